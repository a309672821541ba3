"""Host side of the whole-stack launcher ``gwen_gnn_forward_f32`` (include/gwen_hip.h).

One C call enqueues every kernel of ``GNNModel.forward`` (/root/reference/src/gwen/models_gnn.py:
292-303) on torch's current HIP stream: no per-layer Python, no allocation inside, hipGraph-capturable.
Used for inference (``torch.no_grad``); training goes through the per-layer autograd Function.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from .graph import GraphCSR, _ptr, _stream

# order string -> (GWEN_ORDER_*, GWEN_CONTRACT_* of an AUTO / FUSED layer).  "auto" is the default precision,
# fp32-class on the kernel's own split ("f16x3": K8 on two scaled fp16 images, every other
# kernel on bf16x6); "auto_x6" / "fused" are bf16x6 in every kernel, "auto_x3" / "fused_x3" the faster bf16x3 split; the
# explicit two-launch orders and "fused_exact" use the fp32-input MFMA whatever the second entry says.
_ORDERS = {"auto": (_lib.ORDER_AUTO, _lib.CONTRACT_F16X3), "auto_x6": (_lib.ORDER_AUTO, _lib.CONTRACT_BF16X6),
           "auto_x3": (_lib.ORDER_AUTO, _lib.CONTRACT_BF16X3),
           "transform_first": (_lib.ORDER_TRANSFORM_FIRST, _lib.CONTRACT_BF16X6),
           "aggregate_first": (_lib.ORDER_AGGREGATE_FIRST, _lib.CONTRACT_BF16X6),
           "fused": (_lib.ORDER_FUSED, _lib.CONTRACT_BF16X6), "fused_x3": (_lib.ORDER_FUSED, _lib.CONTRACT_BF16X3),
           "fused_exact": (_lib.ORDER_FUSED_EXACT, _lib.CONTRACT_BF16X6)}


class KernelEvents:
    """hipEvent pairs recorded by the launcher around each kernel launch of one forward."""

    def __init__(self, max_launches: int):
        self.max_launches = max_launches
        self._ev = (C.c_void_p * (2 * max_launches))()
        self.info = (_lib.LaunchInfo * max_launches)()
        self.n = C.c_int32(0)
        L = _lib.lib()
        for i in range(2 * max_launches):
            h = C.c_void_p()
            _lib.check(L.gwen_event_create(C.byref(h)), "gwen_event_create")
            self._ev[i] = h

    def durations(self) -> List[Tuple[str, int, int, int, float]]:
        """[(kind, layer, fin, fout, seconds)] of the last forward (synchronises on the last event)."""
        L = _lib.lib()
        n = self.n.value
        if n == 0:
            return []
        _lib.check(L.gwen_event_synchronize(self._ev[2 * n - 1]), "gwen_event_synchronize")
        out = []
        ms = C.c_float(0)
        for i in range(n):
            _lib.check(L.gwen_event_elapsed_ms(self._ev[2 * i], self._ev[2 * i + 1], C.byref(ms)),
                       "gwen_event_elapsed_ms")
            it = self.info[i]
            out.append((_lib.KIND_NAMES.get(it.kind, str(it.kind)), it.layer, it.fin, it.fout,
                        ms.value * 1e-3))
        return out

    def close(self) -> None:
        L = _lib.lib()
        for i in range(2 * self.max_launches):
            if self._ev[i]:
                L.gwen_event_destroy(self._ev[i])
                self._ev[i] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def event_bracket_overhead(device, samples: int = 32) -> float:
    """Seconds of stream time one hipEvent record costs, from empty brackets (two records back to back,
    median over ``samples``, halved).  A bracket around a kernel contains one record's worth of it, so
    ``bracket - overhead`` is the kernel-only time (checked against rocprofv3: within 0.4 us)."""
    L = _lib.lib()
    ev = KernelEvents(samples)
    st = _stream(device)
    with torch.cuda.device(device):
        torch.cuda.synchronize(device)
        for i in range(samples):
            _lib.check(L.gwen_event_record(ev._ev[2 * i], st), "gwen_event_record")
            _lib.check(L.gwen_event_record(ev._ev[2 * i + 1], st), "gwen_event_record")
        torch.cuda.synchronize(device)
    ms = C.c_float(0)
    vals = []
    for i in range(samples):
        _lib.check(L.gwen_event_elapsed_ms(ev._ev[2 * i], ev._ev[2 * i + 1], C.byref(ms)),
                   "gwen_event_elapsed_ms")
        vals.append(ms.value * 1e-3)
    ev.close()
    vals.sort()
    return 0.5 * vals[len(vals) // 2]


def pack_weight(weight: Tensor) -> Optional[Tensor]:
    """The 3xbf16 hi / lo images of ``weight`` [Fout, Fin] in MFMA fragment order (gwen_gcn_small_pack_f32)
    for K7, or None when the shape has no packed form.  As many bytes as the weight itself."""
    fout, fin = weight.shape
    nbytes = int(_lib.lib().gwen_gcn_small_pack_bytes(fin, fout))
    if nbytes == 0 or not weight.is_cuda or weight.dtype != torch.float32:
        return None
    w = weight.detach().contiguous()
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.lib().gwen_gcn_small_pack_f32(_ptr(w), fin, fout, _ptr(img), _stream(w.device))
    _lib.check(rc, "gwen_gcn_small_pack_f32")
    return img


class StackForward:
    """A stack of GCN layers bound to one prepared graph: ``run(x)`` is one C call."""

    def __init__(self, layers: Sequence[Tuple[Tensor, Optional[Tensor], bool, str]], graph: GraphCSR,
                 packed: Optional[Sequence[Optional[Tensor]]] = None):
        """``packed``: per layer the ``pack_weight`` image of its weight (or None) -- K7 then streams the
        weights in fragment order (small graphs; the caller owns the cache, see GNNModel)."""
        if not layers:
            raise ValueError("need at least one layer")
        self.graph = graph
        self._keep = []                      # tensors whose pointers sit in the descriptor array
        self.desc = (_lib.LayerDesc * len(layers))()
        for i, (w, b, relu, order) in enumerate(layers):
            if not w.is_cuda or w.dtype != torch.float32:
                raise RuntimeError("layer weights must be float32 on a HIP device (no CPU fallback)")
            w = w.detach().contiguous()
            b = None if b is None else b.detach().contiguous()
            self._keep += [w, b]
            d = self.desc[i]
            d.W, d.bias = w.data_ptr(), (0 if b is None else b.data_ptr())
            d.fout, d.fin = w.size(0), w.size(1)
            d.relu = int(relu)
            d.order, d.contract = _ORDERS[order]
            img = None if packed is None else packed[i]
            d.packed = 0 if img is None else img.data_ptr()
            self._keep.append(img)
        self.fin, self.fout = self.desc[0].fin, self.desc[len(layers) - 1].fout
        # the launcher's own test for the K7 path (small graph, every layer AUTO and of a supported shape):
        # when it holds the grouped layout is never read, so it is not built (a kernel and a read-back
        # per fresh graph -- the reference's loaders hand over a new edge_index per batch)
        self._small = graph.dense() is not None and all(
            d.order == _lib.ORDER_AUTO and _lib.lib().gwen_gcn_small_supported(graph.num_nodes, d.fin, d.fout,
                                                                               _lib.dense_contract(d.contract))
            for d in self.desc)
        # long rows beyond K7's graphs: per-layer K3 + segmented K2 (ops.propagate) instead of the C launcher,
        # whose fused kernels walk a row serially
        self._long = (not self._small) and graph.long_row_levels() is not None
        self._layers = list(layers)
        self._packed_in = packed
        self._inner: dict = {}               # members -> (perm, inv_perm, StackForward on the clustered graph) or None
        self._scratch: Optional[Tensor] = None
        self._scratch_key = None
        self._gd = None
        self._gd_key = None

    def _scratch_for(self, members: int, dev) -> Tensor:
        key = (members, dev)
        if self._scratch_key != key:
            n = int(_lib.lib().gwen_gnn_forward_scratch_floats(self.graph.num_nodes, members, self.desc,
                                                               len(self.desc)))
            if n < 0:
                _lib.check(n, "gwen_gnn_forward_scratch_floats")
            self._scratch = torch.empty(max(n, 4), dtype=torch.float32, device=dev)
            self._scratch_key = key
        return self._scratch

    def _graph_desc(self, members: int) -> "_lib.GraphDesc":
        """The views of the prepared graph the launcher may need (built on first use, kept with the graph):
        grouped layout unless the K7 path is taken, tile layout when some AUTO layer would run as K8."""
        g = self.graph
        want_tiles = (not self._small) and any(
            d.order == _lib.ORDER_AUTO and _lib.lib().gwen_gcn_wide_preferred(g.num_nodes, members, d.fin, d.fout)
            and _lib.lib().gwen_gcn_wide_contract_supported(d.fin, d.fout, _lib.wide_contract(d.fin, d.fout, d.contract))
            for d in self.desc)
        key = (members if want_tiles else 0, want_tiles)
        if self._gd_key != key:
            gr, gc, gv = (None, None, None) if self._small else g.grouped()
            tiles = g.tiles() if want_tiles else None
            gd = _lib.GraphDesc()
            gd.N = g.num_nodes
            gd.rowptr, gd.col, gd.val = g.rowptr.data_ptr(), g.col.data_ptr(), g.val.data_ptr()
            gd.g_rowptr = 0 if gr is None else gr.data_ptr()
            gd.g_col = 0 if gc is None else gc.data_ptr()
            gd.g_val = 0 if gv is None else gv.data_ptr()
            dense = g.dense()
            gd.dense = 0 if dense is None else dense.data_ptr()
            if tiles is not None:
                gd.t_rows, gd.t_lid, gd.t_val = (t.data_ptr() for t in tiles[:3])
                gd.union_max = tiles[3]
            self._gd, self._gd_key = gd, key
        return self._gd

    def _clustered_plan(self, members: int):
        """A caller's arbitrary node order does not tile, so its wide AUTO layers would fall back from K8 to K4
        (0.28 of the HBM peak beyond the caches against 0.4-0.5): when some layer wants K8 and the graph tiles only
        in an order this library grows itself (``GraphCSR.clustered``), the whole stack runs in that order --
        input rows permuted once, output rows permuted back -- with bitwise the same results."""
        if members not in self._inner:
            g, hit = self.graph, None
            wants = (not self._small) and (not self._long) and g.num_src < 0 and any(
                d.order == _lib.ORDER_AUTO and _lib.lib().gwen_gcn_wide_preferred(g.num_nodes, members, d.fin, d.fout)
                and _lib.lib().gwen_gcn_wide_contract_supported(d.fin, d.fout, _lib.wide_contract(d.fin, d.fout, d.contract))
                for d in self.desc)
            if wants and g.tiles() is None:
                cl = g.clustered()
                if cl is not None:
                    hit = (cl[0], cl[1], StackForward(self._layers, cl[2], self._packed_in))
            self._inner[members] = hit
        return self._inner[members]

    def run(self, x: Tensor, out: Optional[Tensor] = None,
            events: Optional[KernelEvents] = None, acts: Optional[List[Tensor]] = None) -> Tensor:
        """``acts``: list of n_layers output tensors ([..., N, fout_l]) -- the TRAINING forward: every
        layer's output is kept (``acts[-1]`` is the result) and no projection is chained across layers."""
        if not x.is_cuda:
            raise RuntimeError("gwen_amd: x must live on a HIP device (no CPU fallback)")
        if x.dtype != torch.float32:
            raise TypeError(f"gwen_amd: x must be float32 (got {x.dtype})")
        x = x.contiguous()
        if x.dim() == 2:
            members = 1
        elif x.dim() == 3:
            members = x.size(0)
        else:
            raise ValueError(f"x must be [N, C] or [members, N, C], got {tuple(x.shape)}")
        n = self.graph.num_nodes
        if x.size(-2) != n or x.size(-1) != self.fin:
            raise ValueError(f"x is {tuple(x.shape)}, expected [..., {n}, {self.fin}]")
        if acts is None:
            inner = self._clustered_plan(members)
            if inner is not None:
                perm, inv, plan = inner
                y = plan.run(x.index_select(-2, perm), events=events)
                if out is None:
                    return y.index_select(-2, inv)
                torch.index_select(y, -2, inv, out=out)
                return out
        if self._long:
            from . import ops
            cur = x
            for i, (w, b, relu, order) in enumerate(self._layers):
                contract = ops.contract_of_order(order)
                if w.size(1) < w.size(0):                      # gather at the narrower width
                    cur = ops.linear(ops.propagate(self.graph, cur), w.detach(), None if b is None else b.detach(),
                                     relu, contract=contract)
                else:
                    cur = ops.propagate(self.graph, ops.linear(cur, w.detach(), contract=contract),
                                        None if b is None else b.detach(), relu)
                if acts is not None:
                    acts[i].copy_(cur)
            if out is not None and acts is None:
                out.copy_(cur)
                return out
            return cur if acts is None else acts[-1]
        dev = x.device
        acts_arr = None
        if acts is not None:
            if len(acts) != len(self.desc):
                raise ValueError("acts must hold one tensor per layer")
            acts_arr = (C.c_void_p * len(acts))(*[a.data_ptr() for a in acts])
            out = acts[-1]
        if out is None:
            out = torch.empty(*x.shape[:-1], self.fout, dtype=torch.float32, device=dev)
        scratch = self._scratch_for(members, dev)
        gd = self._graph_desc(members)
        with torch.cuda.device(dev):
            rc = _lib.lib().gwen_gnn_forward_f32(
                C.byref(gd), self.desc, len(self.desc), _ptr(x),
                _ptr(out), _ptr(scratch), scratch.numel(), members, _stream(dev),
                None if events is None else events._ev,
                None if events is None else events.info,
                0 if events is None else events.max_launches,
                None if events is None else C.byref(events.n), acts_arr)
        _lib.check(rc, "gwen_gnn_forward_f32")
        return out


class GNNStackFunction(torch.autograd.Function):
    """The whole GCN stack as ONE autograd node: forward = ``gwen_gnn_forward_f32`` keeping every layer's
    output, backward = ``gwen_gnn_backward_f32`` -- one host call each (the reference's train step,
    /root/reference/src/gwen/models_gnn.py:365-373, runs ~25 eager launches per layer and direction).
    ``flat`` = weight_0, bias_0, weight_1, bias_1, ...; ``spec`` = ((relu, order), ...) per layer."""

    @staticmethod
    def forward(ctx, x: Tensor, graph: GraphCSR, spec, *flat):
        layers = [(flat[2 * i], flat[2 * i + 1], spec[i][0], spec[i][1]) for i in range(len(spec))]
        plan = StackForward(layers, graph)
        if plan._small:
            raise RuntimeError("GNNStackFunction is for graphs beyond K7's 256 nodes")
        xc = x.contiguous()
        acts = [torch.empty(*xc.shape[:-1], w.size(0), dtype=torch.float32, device=xc.device)
                for w, _, _, _ in layers]
        plan.run(xc, acts=acts)
        ctx.plan, ctx.graph, ctx.n = plan, graph, len(spec)
        ctx.save_for_backward(xc, *acts)
        return acts[-1]

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        xc, *acts = ctx.saved_tensors
        plan, g, n = ctx.plan, ctx.graph, ctx.n
        dev = xc.device
        members = 1 if xc.dim() == 2 else xc.size(0)
        go = grad_out.contiguous()
        t_rowptr, t_col, t_val = g.transposed()
        gr, gc, gv = g.transposed_grouped()
        gd = _lib.GraphDesc()
        gd.N = g.num_nodes
        gd.rowptr, gd.col, gd.val = t_rowptr.data_ptr(), t_col.data_ptr(), t_val.data_ptr()
        gd.g_rowptr = 0 if gr is None else gr.data_ptr()
        gd.g_col, gd.g_val = gc.data_ptr(), gv.data_ptr()
        need = ctx.needs_input_grad
        gx = torch.empty_like(xc) if need[0] else None
        gws = [torch.empty_like(plan._keep[3 * i]) if need[3 + 2 * i] else None for i in range(n)]
        gbs = [torch.empty_like(plan._keep[3 * i + 1]) if (plan._keep[3 * i + 1] is not None and need[4 + 2 * i])
               else None for i in range(n)]
        arr = lambda ts: (C.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])      # noqa: E731
        L = _lib.lib()
        nfl = int(L.gwen_gnn_backward_scratch_floats(g.num_nodes, members, plan.desc, n))
        if nfl < 0:
            _lib.check(nfl, "gwen_gnn_backward_scratch_floats")
        scratch = torch.empty(max(nfl, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.gwen_gnn_backward_f32(C.byref(gd), plan.desc, n, _ptr(xc), arr(acts), _ptr(go), _ptr(gx),
                                         arr(gws), arr(gbs), _ptr(scratch), scratch.numel(), members,
                                         _stream(dev))
        _lib.check(rc, "gwen_gnn_backward_f32")
        flat = []
        for w, b in zip(gws, gbs):
            flat += [w, b]
        return (gx, None, None, *flat)


def stack_apply(x: Tensor, graph: GraphCSR, layers: Sequence[Tuple[Tensor, Optional[Tensor], bool, str]]) -> Tensor:
    """Differentiable forward of a GCN stack through ``GNNStackFunction``."""
    spec = tuple((bool(r), o) for _, _, r, o in layers)
    flat = []
    for w, b, _, _ in layers:
        flat += [w, b]
    return GNNStackFunction.apply(x, graph, spec, *flat)


class GraphedForward:
    """One forward of a StackForward captured into a hipGraph (via ``torch.cuda.CUDAGraph``): a replay is
    one graph launch instead of 6-12 kernel launches, which removes the per-launch gaps on the device.
    The launcher allocates nothing and synchronises nothing, so it is capturable as is.  Input and
    output live in static buffers: ``copy_`` new data into ``self.x`` (or pass ``x`` to ``__call__``)."""

    def __init__(self, stack: StackForward, x: Tensor):
        self.stack = stack
        self.x = x.contiguous().clone()
        self.out = stack.run(self.x)                       # warm-up: scratch, occupancy query, LDS opt-in
        torch.cuda.synchronize(self.x.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            stack.run(self.x, out=self.out)

    def __call__(self, x: Optional[Tensor] = None) -> Tensor:
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.graph.replay()
        return self.out
