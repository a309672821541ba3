"""InteractionNet block on the K6 kernel (BASELINE north_star "edge-MLP + scatter-add").

BUILD-DEFINED -- the reference's only graph layer is GCNConv (/root/reference/src/gwen/models_gnn.py:
118-130,:147-149); it has no edge features and no edge MLP.  Semantics (restated on the CPU by
oracle/interaction_oracle.py, PARITY UNPINNED):

    m_e   = MLP_e([e, x_src[s(e)], x_dst[d(e)]])          MLP = Linear -> act -> Linear
    agg_d = sum / mean of m_e over the in-edges of d
    x'_d  = x_dst_d + MLP_n([x_dst_d, agg_d]) ,   e' = e + m_e

How it runs (3 launches on a square graph, 4 on a bipartite one; training: _InteractionNetFunction, whose backward
is assembled from atomic-free launches of libgwen_hip.so as well -- csrc/interact_bwd.hip):
  * the node halves of both first layers are projected per NODE, not per edge, by ONE K3 launch with the
    three weights stacked (3xbf16): [Ps | Pd | Q] = x [W1[:, F:2F]; W1[:, 2F:]; W3[:, :F]]^T + [0, b1, b3]
    (sources projected apart when x_src is not x_dst); K6 reads them as strided column blocks;
  * K6 over edges (target-sorted):  e' and agg in one launch -- gathers Ps[s], Pd[d], never forms the
    [E,3F] concatenation, the hidden layer or the message tensor in HBM;
  * K6 over nodes:  x' = x_dst + act(agg W3[:, F:]^T + Q) W4^T + b4.
Edge features live in the graph's STORED order (targets ascending): ``EdgeGraph.sort_edges`` /
``unsort_edges`` convert from / to the order of the ``edge_index`` the graph was built from.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _lib, ops
from .graph import GraphCSR, _ptr, _stream, prepare_bipartite

_ACT = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "silu": _lib.ACT_SILU}


@dataclass
class EdgeGraph:
    """Target-sorted edge list + the row-aligned tilings K6 walks (include/gwen_hip.h, gwen_edge_tiles)."""

    num_src: int
    num_dst: int
    num_edges: int
    rowptr: Tensor        # int32 [num_dst + 1]
    src: Tensor           # int32 [E]  source node of stored edge k
    dst: Tensor           # int32 [E]  target node of stored edge k
    eid: Tensor           # int32 [E]  position of stored edge k in the edge_index it came from
    max_degree: int       # longest target row
    _tiles: Dict[int, Tuple[Tensor, int]] = field(default_factory=dict, repr=False)
    _batched: Dict[int, "EdgeGraph"] = field(default_factory=dict, repr=False)
    _seg: Dict[str, tuple] = field(default_factory=dict, repr=False)

    def segments(self, by: str, mean: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
        """CSR (rowptr, col, val) whose columns are stored-EDGE positions: K2 over it sums, per target
        (``by="dst"``) or per source (``by="src"``) node, rows of an [E, F] array in stored order -- the
        atomic-free form of the backward's scatter-adds.  ``mean``: 1/in-degree weights (targets only).
        Index plumbing, built once per graph."""
        key = by + ("_mean" if mean else "")
        if key not in self._seg:
            dev, e = self.device, self.num_edges
            ones = torch.ones(max(e, 1), dtype=torch.float32, device=dev)[:e]
            if by == "dst":
                col = torch.arange(e, dtype=torch.int32, device=dev)
                val = self.inv_degree().index_select(0, self.dst.long()) if mean else ones
                self._seg[key] = (self.rowptr, col, val.contiguous())
            elif by == "src" and not mean:
                order = torch.sort(self.src.long(), stable=True).indices        # stored order inside a source's run
                counts = torch.bincount(self.src.long(), minlength=self.num_src)
                rowptr = torch.zeros(self.num_src + 1, dtype=torch.int32, device=dev)
                rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
                self._seg[key] = (rowptr, order.to(torch.int32).contiguous(), ones)
            else:
                raise ValueError("segments: by in ('dst', 'src'); mean only by 'dst'")
        return self._seg[key]

    def degree(self) -> Tensor:
        """in-degree per target node, fp32 [num_dst, 1] (cached)."""
        if "deg" not in self._seg:
            self._seg["deg"] = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32).view(-1, 1).contiguous()
        return self._seg["deg"]

    def inv_degree(self) -> Tensor:
        """1 / max(in-degree, 1) per target node, fp32 [num_dst] (mean aggregation)."""
        if "inv_deg" not in self._seg:
            deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32).clamp(min=1.0)
            self._seg["inv_deg"] = (1.0 / deg).contiguous()
        return self._seg["inv_deg"]

    @property
    def device(self) -> torch.device:
        return self.rowptr.device

    def batched(self, members: int) -> "EdgeGraph":
        """The block-diagonal graph of ``members`` independent copies (member m's nodes and edges offset by
        m * num_src / num_dst / num_edges): ONE launch set then serves every local ensemble member -- rows
        of all members go through the same persistent blocks, which load their weight fragments once.
        Index plumbing only (built once per member count, kept with the graph)."""
        if members == 1:
            return self
        if members not in self._batched:
            if members < 1 or members * max(self.num_edges, self.num_src, self.num_dst) >= 2 ** 31 - 1:
                raise ValueError("members out of range for int32 indices")
            dev, e = self.device, self.num_edges
            m = torch.arange(members, dtype=torch.int32, device=dev).view(-1, 1)
            rowptr = torch.cat([(self.rowptr[:-1].view(1, -1) + m * e).reshape(-1),
                                torch.tensor([members * e], dtype=torch.int32, device=dev)])
            g = EdgeGraph(self.num_src * members, self.num_dst * members, e * members, rowptr.contiguous(),
                          (self.src.view(1, -1) + m * self.num_src).reshape(-1).contiguous(),
                          torch.empty(max(e * members, 1), dtype=torch.int32, device=dev)[:e * members],
                          (self.eid.view(1, -1) + m * e).reshape(-1).contiguous(), self.max_degree)
            g.tiles(64)                                   # cuts the tiles and fills dst
            self._batched[members] = g
        return self._batched[members]

    def tiles(self, rows: int) -> Tuple[Tensor, int]:
        """(tile_row int32 [n_tiles + 1], n_tiles) for a kernel that takes ``rows`` rows per pass: the
        tile size rows - (max_degree - 1) makes every tile a single pass on bounded-degree graphs."""
        if rows not in self._tiles:
            t = max(rows - max(self.max_degree - 1, 0), rows // 2) if self.max_degree <= rows else rows
            n_tiles = int(_lib.lib().gwen_edge_tiles_count(self.num_edges, t))
            tile_row = torch.empty(n_tiles + 1, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                rc = _lib.lib().gwen_edge_tiles(_ptr(self.rowptr), self.num_dst, self.num_edges, t,
                                                _ptr(tile_row), _ptr(self.dst), _stream(self.device))
            _lib.check(rc, "gwen_edge_tiles")
            self._tiles[rows] = (tile_row, n_tiles)
        return self._tiles[rows]

    def sort_edges(self, e: Tensor) -> Tensor:
        """Edge features in ``edge_index`` order -> stored order."""
        return e.index_select(0, self.eid.long())

    def unsort_edges(self, e: Tensor) -> Tensor:
        out = torch.empty_like(e)
        out.index_copy_(0, self.eid.long(), e)
        return out


def interaction_graph(edge_index: Tensor, num_src: int, num_dst: int) -> EdgeGraph:
    """Once per graph: sort by target (K1, gwen_gcn_prep_rect); tilings are cut on first use."""
    g: GraphCSR = prepare_bipartite(edge_index, num_src, num_dst, mean=False)
    e, dev = g.num_edges, g.device
    # one read-back per graph: the longest row decides the tile size (a tile = one pass of the kernel)
    max_deg = int((g.rowptr[1:] - g.rowptr[:-1]).max().item()) if num_dst > 0 else 0
    dst = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    graph = EdgeGraph(num_src, num_dst, e, g.rowptr, g.col[:e], dst[:e], g.eid[:e], max_deg)
    graph.tiles(64)              # also fills dst
    return graph


def mlp2_supported(channels: int) -> bool:
    return bool(_lib.lib().gwen_mlp2_supported(channels))


def mlp2(a: Tensor, w1: Tensor, w2: Tensor, b2: Optional[Tensor] = None, *,
         g1: Optional[Tensor] = None, idx1: Optional[Tensor] = None, g2: Optional[Tensor] = None,
         idx2: Optional[Tensor] = None, b1: Optional[Tensor] = None, res: Optional[Tensor] = None,
         act: str = "silu", graph: Optional[EdgeGraph] = None, mean: bool = False,
         want_out: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """K6 (gwen_mlp2_f32): returns (out, agg); ``agg`` only with ``graph`` (rows = its stored edges)."""
    f = a.size(-1)
    if a.dim() != 2 or not mlp2_supported(f):
        raise ValueError(f"K6 needs [rows, F] with F in (32, 64, 128, 256); got {tuple(a.shape)}")
    if tuple(w1.shape) != (f, f) or tuple(w2.shape) != (f, f):
        raise ValueError("W1 and W2 must be [F, F]")
    ts = {"a": a, "w1": w1, "w2": w2, "b1": b1, "b2": b2, "res": res}
    for name, t in ts.items():
        if t is not None:
            ops._require(t, name)
            ts[name] = t.contiguous()
    a, w1, w2, b1, b2, res = (ts[k] for k in ("a", "w1", "w2", "b1", "b2", "res"))
    # a table may be a column block of a wider row-major matrix (several projections from one launch)
    for name, g in (("g1", g1), ("g2", g2)):
        if g is not None:
            ops._require(g, name)
            if g.dim() != 2 or g.size(1) != f or g.stride(1) != 1 or g.stride(0) % 4 or g.stride(0) < f \
                    or g.data_ptr() % 16:
                raise ValueError(f"{name} must be [rows, F] with unit column stride and 16-byte aligned rows")
    rows = a.size(0)
    for name, idx, g in (("idx1", idx1, g1), ("idx2", idx2, g2)):
        if idx is not None and (g is None or idx.dtype != torch.int32 or idx.numel() != rows):
            raise ValueError(f"{name} must be int32 [rows] and come with its table")
        if idx is None and g is not None and g.size(0) != rows:
            raise ValueError("a table without an index must have one row per row of A")
    if res is not None and res.shape != a.shape:
        raise ValueError("res must have the shape of A")
    dev = a.device
    out = torch.empty_like(a) if want_out else None
    agg, tile_row, n_tiles = None, None, 0
    if graph is not None:
        if graph.num_edges != rows:
            raise ValueError("A must hold one row per stored edge of the graph")
        agg = torch.empty(graph.num_dst, f, dtype=torch.float32, device=dev)
        tile_row, n_tiles = graph.tiles(int(_lib.lib().gwen_mlp2_rows(f)))
    nws = int(_lib.lib().gwen_mlp2_workspace_bytes(f))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev) if nws > 0 else None
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_mlp2_f32(
            _ptr(a), _ptr(w1), _ptr(g1), _ptr(idx1), 0 if g1 is None else g1.size(0),
            0 if g1 is None else g1.stride(0), _ptr(g2), _ptr(idx2), 0 if g2 is None else g2.size(0),
            0 if g2 is None else g2.stride(0), _ptr(b1), _ptr(w2), _ptr(b2),
            _ptr(res), _ptr(out), rows, f, _ACT[act],
            _ptr(graph.rowptr) if graph else None, _ptr(tile_row), n_tiles, _ptr(agg),
            graph.num_dst if graph else 0, int(mean),
            _ptr(ws), nws, _stream(dev))
    _lib.check(rc, "gwen_mlp2_f32")
    return out, agg


def _mlp(fin: int, f: int, act: str) -> nn.Sequential:
    a = {"none": nn.Identity, "relu": nn.ReLU, "silu": nn.SiLU}[act]()
    return nn.Sequential(nn.Linear(fin, f), a, nn.Linear(f, f))


class InteractionNet(nn.Module):
    """``forward(x_src, x_dst, e, graph) -> (x_dst', e')``; parameters ``edge_mlp.{0,2}.{weight,bias}``
    ([F,3F] / [F,F]) and ``node_mlp.{0,2}.{weight,bias}`` ([F,2F] / [F,F]).  Forward on K6; when gradients are
    needed the backward runs on libgwen_hip.so as well (``_InteractionNetFunction``: atomic-free, reproducible)."""

    def __init__(self, channels: int, activation: str = "silu", aggr: str = "sum"):
        super().__init__()
        if activation not in _ACT or aggr not in ("sum", "mean"):
            raise ValueError("activation in (none, relu, silu), aggr in (sum, mean)")
        self.channels, self.activation, self.aggr = channels, activation, aggr
        self.edge_mlp = _mlp(3 * channels, channels, activation)
        self.node_mlp = _mlp(2 * channels, channels, activation)
        self._blocks = None          # contiguous [F,F] blocks of the two first layers + their versions

    def _weight_blocks(self):
        """edge_mlp.0.weight = [We | Ws | Wd], node_mlp.0.weight = [Wx | Wa] re-cut for the kernels:
        We, Wa contiguous [F,F]; the three NODE projections stacked as one [3F,F] weight (rows Ws, Wd,
        Wx) with bias [0, b1, b3], so that one K3 launch serves all of them when x_src is x_dst (two
        when not).  Re-cut only when a parameter changed (in-place version counter) or moved."""
        l1, l3, f = self.edge_mlp[0], self.node_mlp[0], self.channels
        key = tuple((p.data_ptr(), p._version) for p in (l1.weight, l1.bias, l3.weight, l3.bias))
        if self._blocks is None or self._blocks[0] != key:
            with torch.no_grad():
                w1, w3 = l1.weight, l3.weight
                we, wa = w1[:, :f].contiguous(), w3[:, f:].contiguous()
                wn = torch.cat([w1[:, f:2 * f], w1[:, 2 * f:], w3[:, :f]], dim=0).contiguous()
                bn = torch.cat([torch.zeros_like(l1.bias), l1.bias, l3.bias]).contiguous()
            self._blocks = (key, (we, wa, wn, bn))
        return self._blocks[1]

    def __getstate__(self):          # the cache is derived data: keep modules picklable and small
        state = self.__dict__.copy()
        state["_blocks"] = None
        return state

    def forward(self, x_src: Tensor, x_dst: Tensor, e: Tensor, graph: EdgeGraph,
                update_edges: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
        f = self.channels
        if x_src.shape != (graph.num_src, f) or x_dst.shape != (graph.num_dst, f) or \
                e.shape != (graph.num_edges, f):
            raise ValueError("x_src / x_dst / e do not match the graph and the channel count")
        if torch.is_grad_enabled() and any(t.requires_grad for t in (x_src, x_dst, e, *self.parameters())):
            return _InteractionNetFunction.apply(self, graph, update_edges, x_src is x_dst, x_src, x_dst, e,
                                                 *self.parameters())
        return self._forward_k6(x_src, x_dst, e, graph, update_edges)

    def _forward_k6(self, x_src: Tensor, x_dst: Tensor, e: Tensor, graph: EdgeGraph,
                    update_edges: bool = True, return_agg: bool = False):
        f = self.channels
        we, wa, wn, bn = self._weight_blocks()
        if x_src is x_dst:                                   # mesh -> mesh: one launch, [N, 3F]
            p = ops.linear(x_dst, wn, bn, exact=False)
            ps, pd, q = p[:, :f], p[:, f:2 * f], p[:, 2 * f:]
        else:                                                # bipartite: sources apart
            ps = ops.linear(x_src, wn[:f], None, exact=False)
            p = ops.linear(x_dst, wn[f:], bn[f:], exact=False)
            pd, q = p[:, :f], p[:, f:]
        e_new, agg = mlp2(e, we, self.edge_mlp[2].weight, self.edge_mlp[2].bias,
                          g1=ps, idx1=graph.src, g2=pd, idx2=graph.dst, res=e, act=self.activation,
                          graph=graph, mean=self.aggr == "mean", want_out=update_edges)
        x_new, _ = mlp2(agg, wa, self.node_mlp[2].weight, self.node_mlp[2].bias, g1=q,
                        res=x_dst, act=self.activation)
        # (training keeps the aggregate and the node projections: node-sized arrays this pass makes anyway)
        return (x_new, e_new, agg, p, None if x_src is x_dst else ps) if return_agg else (x_new, e_new)


# ---- pieces of the backward (csrc/interact_bwd.hip + K2 / K3 / the gradient reductions) -------------------------
_BWD_CONTRACT = "3xbf16"          # the block's forward (K6) contracts on the bf16x3 split; so does its backward


def _act_pair(a: Tensor, act: str, g1: Optional[Tensor] = None, idx1: Optional[Tensor] = None,
              g2: Optional[Tensor] = None, idx2: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(act(pre), act'(pre)) with pre = a + g1[idx1 or row] + g2[idx2 or row]; act(pre) overwrites ``a``."""
    rows, f = a.shape
    d = torch.empty_like(a)
    dev = a.device
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_act_pair_f32(_ptr(a), _ptr(g1), _ptr(idx1), 0 if g1 is None else g1.stride(0),
                                          _ptr(g2), _ptr(idx2), 0 if g2 is None else g2.stride(0), _ptr(a),
                                          _ptr(d), rows, f, _ACT[act], _stream(dev))
    _lib.check(rc, "gwen_act_pair_f32")
    return a, d


def _act_pair_seg(a: Tensor, act: str, g1: Tensor, idx1: Tensor, g2: Tensor, rowptr: Tensor, n_dst: int):
    """gwen_act_pair_seg_f32: (act(pre), act'(pre), per-target sums of act(pre)) for edges stored by target, pre = a +
    g1[idx1] + g2[target]; act(pre) overwrites ``a``."""
    rows, f = a.shape
    dev = a.device
    d = torch.empty_like(a)
    hsum = torch.empty(n_dst, f, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_act_pair_seg_f32(_ptr(a), _ptr(g1), _ptr(idx1), g1.stride(0), _ptr(g2), g2.stride(0),
                                              _ptr(rowptr), _ptr(a), _ptr(d), _ptr(hsum), rows, n_dst, f, _ACT[act],
                                              _stream(dev))
    _lib.check(rc, "gwen_act_pair_seg_f32")
    return a, d, hsum


def _gather_add(a: Optional[Tensor], t: Tensor, idx: Tensor, scale: Optional[Tensor] = None) -> Tensor:
    rows, f = idx.numel(), t.size(-1)
    out = torch.empty(rows, f, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        rc = _lib.lib().gwen_gather_add_f32(_ptr(a), _ptr(t), _ptr(idx), _ptr(scale), _ptr(out), rows, f,
                                            _stream(t.device))
    _lib.check(rc, "gwen_gather_add_f32")
    return out


def _ew(op: int, a: Tensor, b: Tensor) -> Tensor:
    """a * b or a + b, written over ``a``."""
    with torch.cuda.device(a.device):
        rc = _lib.lib().gwen_ew_f32(op, _ptr(a), _ptr(b), _ptr(a), a.numel(), _stream(a.device))
    _lib.check(rc, "gwen_ew_f32")
    return a


def _edge_backward(ge: Tensor, w2t: Tensor, d1: Tensor, t: Tensor, dst: Tensor, wet: Tensor) -> Tuple[Tensor, Tensor]:
    """gwen_mlp2_bwd_f32: (g_pre1, g_e) = ((ge W2 + T[dst]) * d1, ge + g_pre1 We) in one launch of K6's kernel."""
    rows, f = ge.shape
    L = _lib.lib()
    dev = ge.device
    padded = int(L.gwen_mlp2_bwd_rows(rows))
    g_pre1 = torch.empty(padded, f, dtype=torch.float32, device=dev)        # whole passes: the kernel stores every lane
    g_e = torch.empty_like(ge)
    nws = int(L.gwen_mlp2_workspace_bytes(f))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev) if nws > 0 else None
    with torch.cuda.device(dev):
        rc = L.gwen_mlp2_bwd_f32(_ptr(ge), _ptr(w2t), _ptr(d1), _ptr(t), _ptr(dst), t.size(0), t.stride(0), _ptr(wet),
                                 _ptr(g_pre1), _ptr(g_e), rows, f, _ptr(ws), nws, _stream(dev))
    _lib.check(rc, "gwen_mlp2_bwd_f32")
    return g_pre1[:rows], g_e


def _segsum(seg: Tuple[Tensor, Tensor, Tensor], h: Tensor, rows: int) -> Tensor:
    """K2 over an edge-position CSR (``EdgeGraph.segments``): out[i] = sum_s val[s] h[col[s]] in stored order."""
    rowptr, col, val = seg
    f = h.size(-1)
    out = torch.empty(rows, f, dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        rc = _lib.lib().gwen_gcn_propagate_f32(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(h), None, _ptr(out), rows, f,
                                               f, f, 1, h.numel(), rows * f, 0, _stream(h.device))
    _lib.check(rc, "gwen_gcn_propagate_f32")
    return out


class _InteractionNetFunction(torch.autograd.Function):
    """Training through an InteractionNet block, forward AND backward on libgwen_hip.so.  The forward runs on K6 and
    keeps the node-sized intermediates it makes anyway -- the aggregated messages (saves the backward an edge-sized
    projection and a segmented sum) and the node projections [Ps | Pd | Q]; the backward recomputes the two hidden layers
    (K3 + ``gwen_act_pair_f32``, which also yields the activation's derivative) and then walks the block in reverse:
        node MLP:  g_pre3 = (gx W4) * act'(pre3);   g_agg = g_pre3 Wa;   g_x += gx + g_pre3 Wx
        messages:  g_m[e] = ge[e] + g_agg[dst(e)] (/ degree for the mean)          (``gwen_gather_add_f32``)
        edge MLP:  g_pre1 = (g_m W2) * act'(pre1);  g_e = ge + g_pre1 We
        nodes:     G_d = sum of g_pre1 over a target's edges, G_s over a source's   (K2 over ``EdgeGraph.segments``:
                   stored order, no atomics);  g_x_dst += G_d Wd,  g_x_src = G_s Ws
        weights:   every grad_W = (gradient rows)^T (input rows), grad_b = column sums   (fixed-order reductions; the
                   wide ones -- multiples of 64 from 128 x 128 -- on the block's own bf16x3 split: csrc/grad.hip)
    Every launch is atomic-free with a fixed summation order: two backward runs are bitwise equal.  BUILD-DEFINED
    like the block (the reference has no edge MLP); gradients are tested against fp64 autograd of the oracle."""

    @staticmethod
    def forward(ctx, net, graph, update_edges, same, x_src, x_dst, e, *params):
        with torch.no_grad():
            x_new, e_new, agg, proj, ps = net._forward_k6(x_dst if same else x_src, x_dst, e, graph, update_edges,
                                                          return_agg=True)
        ctx.net, ctx.graph, ctx.update_edges, ctx.same = net, graph, update_edges, same
        ctx.save_for_backward(x_src, x_dst, e, agg, *params, proj, proj.new_empty(0) if ps is None else ps)
        if e_new is None:
            e_new = e.new_empty(0)
            ctx.mark_non_differentiable(e_new)
        return x_new, e_new

    @staticmethod
    def backward(ctx, gx, ge):
        x_src, x_dst, e, agg, w1, b1, w2, b2, w3, b3, w4, b4, pall, ps = ctx.saved_tensors
        net, g, same = ctx.net, ctx.graph, ctx.same
        f, act, mean = net.channels, net.activation, net.aggr == "mean"
        n_src, n_dst = g.num_src, g.num_dst
        with torch.no_grad():
            lin = lambda x, w, b=None: ops.linear(x, w, b, contract=_BWD_CONTRACT)          # noqa: E731
            x_src, x_dst, e = x_src.detach().contiguous(), x_dst.detach().contiguous(), e.detach().contiguous()
            gx = gx.contiguous()
            has_ge = bool(ctx.update_edges and ge is not None and ge.numel() > 0)
            ge = ge.contiguous() if has_ge else None
            # every W^T the backward contracts with, from FOUR small copies (the row blocks of a transposed matrix are
            # contiguous): w1 = [We | Ws | Wd], w3 = [Wx | Wa]  (thirteen slice / transpose copies a block before)
            w1t, w3t, w2t, w4t = (w.t().contiguous() for w in (w1, w3, w2, w4))
            wet, wst, wdt, wxt, wat = w1t[:f], w1t[f:2 * f], w1t[2 * f:], w3t[:f], w3t[f:]
            # ---- the two hidden layers again ------------------------------------------------------------------------
            # the node-side projections [Ps | Pd | Q] (stacked when x_src is x_dst) are the forward's own, kept (node-sized:
            # 3 F floats a node -- recomputing them was three launches of the dense kernel a block, 3 % of the step); the
            # kernels below read the column blocks through a row stride
            we, wa, _, _ = net._weight_blocks()
            if same:
                ps, pd, q = pall[:, :f], pall[:, f:2 * f], pall[:, 2 * f:]
            else:
                pd, q = pall[:, :f], pall[:, f:]
            fused_edge = bool(_lib.lib().gwen_mlp2_bwd_supported(f))
            if fused_edge:       # (edges are stored by target: the hidden layer's per-target sums come out of the same pass)
                h1, d1, hagg = _act_pair_seg(lin(e, we), act, ps, g.src, pd, g.rowptr, n_dst)
            else:
                h1, d1 = _act_pair(lin(e, we), act, ps, g.src, pd, g.dst)
            del ps, pd                                   # (agg = sum / mean of the messages: kept by the forward)
            h3, d3 = _act_pair(lin(agg, wa), act, q)
            del pall, q
            # ---- node MLP -----------------------------------------------------------------------------------------
            # (every weight / bias gradient: stage 1 launched where its operands are live, the fixed-order finishes of
            #  all of them in ONE launch at the end -- ops.GradBatch)
            gb = ops.GradBatch()
            gw = lambda a, b_: gb.grad_weight(a, b_, _BWD_CONTRACT)                          # noqa: E731
            gwb = lambda a, b_: gb.grad_weight_bias(a, b_, _BWD_CONTRACT)                    # noqa: E731
            g_w4, g_b4 = gwb(gx, h3)                                   # (weight gradient + column sums of the same rows)
            g_pre3 = _ew(_lib.EW_MUL, lin(gx, w4t), d3)
            del h3, d3
            g_w3a, g_b3 = gwb(g_pre3, x_dst)
            g_w3 = [g_w3a, gw(g_pre3, agg)]
            g_agg = lin(g_pre3, wat)
            g_xd = _ew(_lib.EW_ADD, lin(g_pre3, wxt), gx)
            del g_pre3, agg
            # ---- messages and edge MLP ----------------------------------------------------------------------------
            if fused_edge:
                # ONE launch of K6's kernel for the edge-level half (round 4): by linearity g_m W2 = ge W2 + T[dst] with
                # T = (g_agg / degree) W2 per node, so the message gradient g_m = ge + g_agg[dst] is never formed --
                # its two uses split the same way: g_m^T h1 = ge^T h1 + g_agg_s^T (sum of h1 over a target's edges),
                # column sums of g_m = column sums of ge + sum_d degree_d g_agg_s[d]
                # (a block without an edge output -- the encoder / decoder blocks -- has ge = 0: the same launch on a
                #  zero array still replaces four launches and nine passes, and ge's own gradient terms drop out)
                g_agg_s = g_agg * g.inv_degree().view(-1, 1) if mean else g_agg
                g_pre1, g_e = _edge_backward(ge if has_ge else torch.zeros_like(e), w2t, d1,
                                             lin(g_agg_s, w2t), g.dst, wet)
                del d1
                g_b2 = gb.grad_bias(g_agg_s * g.degree())
                g_w2 = gw(g_agg_s, hagg)
                g_w2e, g_b2e = gwb(ge, h1) if has_ge else (None, None)
                del h1, hagg, g_agg, g_agg_s
                big_d = _segsum(g.segments("dst"), g_pre1, n_dst)
                big_s = _segsum(g.segments("src"), g_pre1, n_src)
                g_w1d, g_b1 = gwb(big_d, x_dst)
                g_w1 = [gw(g_pre1, e), gw(big_s, x_src), g_w1d]
                del g_pre1
                g_xs = lin(big_s, wst)
                g_xd = _ew(_lib.EW_ADD, g_xd, lin(big_d, wdt))
                if same:
                    g_xd = _ew(_lib.EW_ADD, g_xd, g_xs)
                    g_xs = None
                gb.finish()
                g_w1, g_w3 = torch.cat(g_w1, dim=1), torch.cat(g_w3, dim=1)
                if has_ge:
                    g_b2, g_w2 = g_b2 + g_b2e, g_w2 + g_w2e
                need = ctx.needs_input_grad
                pick = lambda k, t: t if need[k] else None                                  # noqa: E731
                return (None, None, None, None, pick(4, g_xs), pick(5, g_xd), pick(6, g_e), pick(7, g_w1),
                        pick(8, g_b1), pick(9, g_w2), pick(10, g_b2), pick(11, g_w3), pick(12, g_b3), pick(13, g_w4),
                        pick(14, g_b4))
            g_m = _gather_add(ge, g_agg, g.dst, g.inv_degree() if mean else None)
            g_b2, g_w2 = gb.grad_bias(g_m), gw(g_m, h1)
            g_pre1 = _ew(_lib.EW_MUL, lin(g_m, w2t), d1)
            del g_m, h1, d1, g_agg
            big_d = _segsum(g.segments("dst"), g_pre1, n_dst)          # per target: sum over its in-edges
            g_b1 = gb.grad_bias(big_d)                                 # = column sums of g_pre1, over N_dst rows instead of E
            big_s = _segsum(g.segments("src"), g_pre1, n_src)          # per source: sum over its out-edges
            g_w1 = [gw(g_pre1, e), gw(big_s, x_src), gw(big_d, x_dst)]
            gb.finish()
            g_w1, g_w3 = torch.cat(g_w1, dim=1), torch.cat(g_w3, dim=1)
            g_e = lin(g_pre1, wet)
            if has_ge:
                g_e = _ew(_lib.EW_ADD, g_e, ge)
            del g_pre1
            g_xs = lin(big_s, wst)
            g_xd = _ew(_lib.EW_ADD, g_xd, lin(big_d, wdt))
            if same:           # x_src is x_dst: one tensor in two argument slots -- its gradient is reported once
                g_xd = _ew(_lib.EW_ADD, g_xd, g_xs)
                g_xs = None
        need = ctx.needs_input_grad
        pick = lambda k, t: t if need[k] else None                                          # noqa: E731
        return (None, None, None, None, pick(4, g_xs), pick(5, g_xd), pick(6, g_e), pick(7, g_w1), pick(8, g_b1),
                pick(9, g_w2), pick(10, g_b2), pick(11, g_w3), pick(12, g_b3), pick(13, g_w4), pick(14, g_b4))
