"""Tensor-level wrappers over the C ABI (include/gwen_hip.h) and the autograd Function of one layer.

torch is used for device memory, the current HIP stream and autograd bookkeeping only; every
arithmetic step of the layer runs in libgwen_hip.so.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from .graph import GraphCSR, _ptr, _stream


def _require(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"gwen_amd: {name} must live on a HIP device (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"gwen_amd: {name} must be float32 (got {t.dtype})")


def _rows2d(x: Tensor):
    """[N,F] or [M,N,F] -> (members, N, F)."""
    if x.dim() == 2:
        return 1, x.size(0), x.size(1)
    if x.dim() == 3:
        return x.size(0), x.size(1), x.size(2)
    raise ValueError(f"expected [N, F] or [members, N, F], got {tuple(x.shape)}")


AUTO_ORDERS = ("auto", "auto_x6", "auto_x3")


def contract_of_order(order: str) -> str:
    """Contraction of a layer run under ``order``: "auto" -> "f16x3" (the default: fp32-class on the kernel's own
    split -- K8 on two scaled fp16 images, every other kernel on bf16x6), "auto_x6" /
    "fused" -> "bf16x6" in every kernel, "auto_x3" / "fused_x3" -> "3xbf16", every other (explicit) order -> "fp32"."""
    if order == "auto":
        return "f16x3"
    if order in ("auto_x6", "fused"):
        return "bf16x6"
    if order in ("auto_x3", "fused_x3"):
        return "3xbf16"
    return "fp32"


def _contract_code(contract, exact: bool = True) -> int:
    if contract is None:
        return _lib.CONTRACT_F32 if exact else _lib.CONTRACT_BF16X3
    if contract not in _lib.CONTRACT_NAMES:
        raise ValueError(f'contract must be one of {sorted(_lib.CONTRACT_NAMES)}')
    return _lib.CONTRACT_NAMES[contract]


def _dense_code(contract, exact: bool = True) -> int:
    """K3 / K4 / K5 / K7: "f16x3" (fp32-class on the kernel's own split) is bf16x6 there."""
    return _lib.dense_contract(_contract_code(contract, exact))


def propagate(graph: GraphCSR, h: Tensor, bias: Optional[Tensor] = None, relu: bool = False,
              transposed: bool = False) -> Tensor:
    """K2: out[i] = act(sum_s val[s] * h[col[s]] + bias) over the CSR (or its transpose)."""
    _require(h, "h")
    h = h.contiguous()
    m, n_src, f = _rows2d(h)
    n = graph.num_nodes
    if not transposed and n_src != graph.source_nodes:
        raise ValueError(f"h has {n_src} rows but the graph has {graph.source_nodes} source nodes")
    if transposed:                        # A^T h: rows = this graph's sources, h has one row per target
        graph = graph.transposed_graph()
        n = graph.num_nodes
        if n_src != graph.source_nodes:
            raise ValueError(f"h has {n_src} rows but the transposed graph has {graph.source_nodes} sources")
    if bias is not None:
        _require(bias, "bias")
        bias = bias.contiguous()
    dev = h.device
    # long rows (hub nodes, complete graphs beyond K7's 256 nodes): the same kernel over the segment chain --
    # every 32-entry segment summed by its own lane group, then each row's partial sums in segment order
    levels = graph.long_row_levels()
    chain = levels if levels is not None else [(graph.rowptr, graph.col, graph.val, n, n_src)]
    cur = h
    for k, (rowptr, col, val, rows, cols) in enumerate(chain):
        last = k + 1 == len(chain)
        out = torch.empty(*h.shape[:-2], rows, f, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().gwen_gcn_propagate_f32(
                _ptr(rowptr), _ptr(col), _ptr(val), _ptr(cur), _ptr(bias) if last else None, _ptr(out), rows, f,
                f, f, m, cols * f, rows * f, int(relu and last), _stream(dev))
        _lib.check(rc, "gwen_gcn_propagate_f32")
        cur = out
    return cur


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, relu: bool = False,
           exact: bool = True, contract: Optional[str] = None) -> Tensor:
    """K3: act(x @ weight^T + bias); x [..., Fin], weight [Fout, Fin].  ``contract``: "fp32" (fp32-input MFMA, a
    k-ordered fp32 fmaf chain), "bf16x6" (fp32-class split, what AUTO layers use) or "3xbf16" (the faster
    split); when None, ``exact`` picks between "fp32" (True) and "3xbf16" (False)."""
    _require(x, "x")
    _require(weight, "weight")
    x = x.contiguous()
    weight = weight.contiguous()
    fin, fout = weight.size(1), weight.size(0)
    if x.size(-1) != fin:
        raise ValueError(f"x has {x.size(-1)} features but weight expects {fin}")
    rows = math.prod(x.shape[:-1])
    out = torch.empty(*x.shape[:-1], fout, dtype=torch.float32, device=x.device)
    if bias is not None:
        _require(bias, "bias")
        bias = bias.contiguous()
    dev = x.device
    code = _dense_code(contract, exact)
    ws = None
    nws = 0 if code == _lib.CONTRACT_F32 else int(_lib.lib().gwen_gcn_linear_workspace_floats(rows, fin, fout))
    if nws > 0:
        ws = torch.empty(nws, dtype=torch.float32, device=dev)          # split-K partial products
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_gcn_linear_f32(_ptr(x), _ptr(weight), _ptr(bias), _ptr(out), rows, fin,
                                            fout, fin, fout, int(relu), code, _ptr(ws), nws,
                                            _stream(dev))
    _lib.check(rc, "gwen_gcn_linear_f32")
    return out


def linear_nn(x: Tensor, weight_t: Tensor, contract: Optional[str] = None) -> Tensor:
    """K3 with the weight operand as stored by the OTHER product: x [..., K] @ weight_t [K, N] (row-major), no transposing
    copy -- the backward's g_x = g W with W = the layer's [out, in] weight.  Split contractions ("3xbf16", "bf16x6",
    "f16x3"); "fp32" / None transposes and takes the exact kernel."""
    _require(x, "x")
    _require(weight_t, "weight_t")
    code = _dense_code(contract, True)
    # (tall inputs -- mesh-sized rows -- keep the transposed form: their weights are small and K3 runs them on K8's pipeline)
    if code == _lib.CONTRACT_F32 or math.prod(x.shape[:-1]) >= 16384:
        return linear(x, weight_t.t().contiguous(), contract=contract)
    x = x.contiguous()
    weight_t = weight_t.contiguous()
    k, n = weight_t.shape
    if x.size(-1) != k:
        raise ValueError(f"x has {x.size(-1)} features but weight_t has {k} rows")
    rows = math.prod(x.shape[:-1])
    out = torch.empty(*x.shape[:-1], n, dtype=torch.float32, device=x.device)
    dev = x.device
    nws = int(_lib.lib().gwen_gcn_linear_workspace_floats(rows, k, n))
    ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws > 0 else None
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_gcn_linear_nn_f32(_ptr(x), _ptr(weight_t), None, _ptr(out), rows, k, n, k, n, n, 0, code,
                                               _ptr(ws), nws, _stream(dev))
    _lib.check(rc, "gwen_gcn_linear_nn_f32")
    return out


class LinearFunction(torch.autograd.Function):
    """Differentiable K3: y = act(x W^T + b) with the backward on the same library -- g_x = g W (K3), g_W = g^T x and
    g_b = column sums of g (the fixed-order reductions of K4's backward).  ``act``: "none" or "relu"."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], relu: bool, contract: str) -> Tensor:
        y = linear(x, weight, bias, relu, contract=contract)
        ctx.relu, ctx.contract, ctx.has_bias = relu, contract, bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        x, weight, y = ctx.saved_tensors
        g = g.contiguous()
        if ctx.relu:
            g = relu_backward(y, g)
        gx = linear_nn(g, weight, ctx.contract) if ctx.needs_input_grad[0] else None
        gw = grad_weight(g, x, ctx.contract) if ctx.needs_input_grad[1] else None
        gb = grad_bias(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None, None


def linear_autograd(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, relu: bool = False,
                    contract: str = "bf16x6") -> Tensor:
    """K3 with autograd (``LinearFunction``)."""
    return LinearFunction.apply(x, weight, bias, relu, contract)


def layer_supported(fin: int, fout: int) -> bool:
    return bool(_lib.lib().gwen_gcn_layer_supported(fin, fout))


def layer_fused(graph: GraphCSR, x: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
                relu: bool = False, exact: bool = False, contract: Optional[str] = None) -> Tensor:
    """K4: act((A~ x) W^T + b) in one launch (widths in {16,32,64,128,256}).  ``contract``: "bf16x6", "3xbf16" or
    "fp32"; when None, ``exact`` picks between "fp32" (True) and "3xbf16" (False)."""
    _require(x, "x")
    _require(weight, "weight")
    x = x.contiguous()
    weight = weight.contiguous()
    m, n_src, fin = _rows2d(x)
    n = graph.num_nodes
    fout = weight.size(0)
    if n_src != graph.source_nodes or weight.size(1) != fin:
        raise ValueError("shape mismatch between x, weight and the graph")
    if n_src * fin * 4 >= (1 << 32):
        raise ValueError("x is too large for K4's 32-bit row offsets")
    if bias is not None:
        _require(bias, "bias")
        bias = bias.contiguous()
    out = torch.empty(*x.shape[:-2], n, fout, dtype=torch.float32, device=x.device)
    dev = x.device
    g_rowptr, g_col, g_val = graph.grouped()
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_gcn_layer_f32(
            _ptr(g_rowptr), _ptr(g_col), _ptr(g_val), _ptr(x), _ptr(weight), _ptr(bias),
            _ptr(out), n, fin, fout, fin, fout, m, n_src * fin, n * fout, int(relu),
            _dense_code(contract, exact), _stream(dev))
    _lib.check(rc, "gwen_gcn_layer_f32")
    return out


def wide_supported(fin: int, fout: int) -> bool:
    return bool(_lib.lib().gwen_gcn_wide_supported(fin, fout))


def wide_preferred(graph: GraphCSR, x: Tensor, fin: int, fout: int, contract: str = "3xbf16") -> bool:
    """Would the stack launcher run this AUTO layer as K8?  (Same rule, so training and inference agree.)"""
    m, n_src, _ = _rows2d(x)
    code = _lib.wide_contract(fin, fout, _contract_code(contract))
    if not _lib.lib().gwen_gcn_wide_preferred(graph.num_nodes, m, fin, fout) or \
            not _lib.lib().gwen_gcn_wide_contract_supported(fin, fout, code):
        return False
    tiles = graph.tiles()
    return tiles is not None and (code != _lib.CONTRACT_BF16X6 or tiles[3] <= 128)


def wide_layer(graph: GraphCSR, x: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
               relu: bool = False, contract: str = "f16x3") -> Tensor:
    """K8: act((A~ x) W^T + b), tile-staged through LDS (widths in {64,128,256}, graphs that tile:
    ``graph.tiles()``).  ``contract``: "f16x3" (the library default: fp32-class on two scaled fp16 images, ONE launch
    at every width), "bf16x6" (graphs whose tile unions stay
    within 128 rows; 256 -> 256 as two 256 -> 128 launches) or "3xbf16" -- the bf16 splits term for term K4's
    arithmetic."""
    _require(x, "x")
    _require(weight, "weight")
    x = x.contiguous()
    weight = weight.contiguous()
    m, n_src, fin = _rows2d(x)
    n = graph.num_nodes
    fout = weight.size(0)
    if n_src != graph.source_nodes or weight.size(1) != fin:
        raise ValueError("shape mismatch between x, weight and the graph")
    tiles = graph.tiles()
    if tiles is None or not wide_supported(fin, fout):
        raise ValueError("K8 needs a graph that tiles (rows of at most 8 entries, <= 192 distinct sources per "
                         "64 rows) and widths in {64, 128, 256}")
    if bias is not None:
        _require(bias, "bias")
        bias = bias.contiguous()
    out = torch.empty(*x.shape[:-2], n, fout, dtype=torch.float32, device=x.device)
    dev = x.device
    t_rows, t_lid, t_val, umax = tiles
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_gcn_wide_layer_f32(
            _ptr(t_rows), _ptr(t_lid), _ptr(t_val), _ptr(x), _ptr(weight), _ptr(bias), _ptr(out),
            n, n_src, fin, fout, fout, m, n_src * fin, n * fout, int(relu), umax,
            _lib.wide_contract(fin, fout, _contract_code(contract)), _stream(dev))
    _lib.check(rc, "gwen_gcn_wide_layer_f32")
    return out


def small_layer(graph: GraphCSR, x: Tensor, weight: Tensor, bias: Optional[Tensor] = None,
                relu: bool = False, packed: Optional[Tensor] = None, contract: str = "3xbf16") -> Tensor:
    """K7: act(A~ (x W^T) + b) in one launch on a graph of at most 256 nodes (dense adjacency).
    ``packed``: the weight's ``gwen_amd.forward.pack_weight`` image (streamed instead of ``weight``; "3xbf16"
    only).  ``contract``: "3xbf16" or "bf16x6"."""
    _require(x, "x")
    _require(weight, "weight")
    x, weight = x.contiguous(), weight.contiguous()
    m, n_src, fin = _rows2d(x)
    n, fout = graph.num_nodes, weight.size(0)
    dense = graph.dense()
    code = _dense_code(contract)
    if dense is None or not _lib.lib().gwen_gcn_small_supported(n, fin, fout, code):
        raise ValueError("K7 needs a square graph of at most 256 nodes, Fin % 32 == 0, Fout % 16 == 0")
    if n_src != n or weight.size(1) != fin:
        raise ValueError("shape mismatch between x, weight and the graph")
    if bias is not None:
        _require(bias, "bias")
        bias = bias.contiguous()
    dev = x.device
    out = torch.empty(*x.shape[:-2], n, fout, dtype=torch.float32, device=dev)
    nws = int(_lib.lib().gwen_gcn_small_workspace_floats(n, m, fin, fout))
    ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws > 0 else None
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_gcn_small_layer_f32(_ptr(dense), _ptr(x), _ptr(weight), _ptr(packed), _ptr(bias), _ptr(out),
                                                 n, fin, fout, m, n * fin, n * fout, int(relu), _ptr(ws),
                                                 nws, code, _stream(dev))
    _lib.check(rc, "gwen_gcn_small_layer_f32")
    return out


def chain(graph: GraphCSR, x: Tensor, w1: Tensor, w2: Optional[Tensor], bias: Optional[Tensor],
          relu: bool, pre: bool, contract: str = "3xbf16") -> Tensor:
    """K5.  pre=False: act((A~ x) w1^T + bias) w2^T;  pre=True: act(A~ x + bias) w1^T  (inference only).
    ``contract``: "3xbf16" or "bf16x6"."""
    _require(x, "x")
    x = x.contiguous()
    m, n, fin = _rows2d(x)
    f1 = w1.size(0)
    f2 = 0 if w2 is None else w2.size(0)
    fw = f2 if f2 else f1
    out = torch.empty(*x.shape[:-1], fw, dtype=torch.float32, device=x.device)
    g_rowptr, g_col, g_val = graph.grouped()
    dev = x.device
    with torch.cuda.device(dev):
        rc = _lib.lib().gwen_gcn_chain_f32(
            _ptr(g_rowptr), _ptr(g_col), _ptr(g_val), _ptr(x), _ptr(w1.contiguous()),
            None if w2 is None else _ptr(w2.contiguous()), None if bias is None else _ptr(bias.contiguous()),
            _ptr(out), n, fin, f1, f2, int(pre), int(relu), m, n * fin, n * fw, _dense_code(contract),
            _stream(dev))
    _lib.check(rc, "gwen_gcn_chain_f32")
    return out


def _grad_workspace(rows: int, fin: int, fout: int, dev) -> Tensor:
    n = int(_lib.lib().gwen_gcn_grad_workspace_floats(rows, fin, fout))
    return torch.empty(n, dtype=torch.float32, device=dev)


def grad_weight(g: Tensor, x: Tensor, contract: Optional[str] = None) -> Tensor:
    """grad_W [Fout, Fin] = g^T @ x over all leading rows.  ``contract``: the contraction of the layer the gradient
    belongs to -- None / "fp32": exact fp32 products; "bf16x6" / "f16x3" / "3xbf16": the split contractions where the
    widths allow (multiples of 64: 2.7 - 3.7 x faster at 256 channels), the fp32 MFMA elsewhere."""
    g = g.contiguous(); x = x.contiguous()
    fout, fin = g.size(-1), x.size(-1)
    rows = math.prod(g.shape[:-1])
    out = torch.empty(fout, fin, dtype=torch.float32, device=g.device)
    ws = _grad_workspace(rows, fin, fout, g.device)
    with torch.cuda.device(g.device):
        rc = _lib.lib().gwen_gcn_grad_weight_f32(_ptr(g), _ptr(x), _ptr(out), rows, fin, fout, fout,
                                                 fin, _ptr(ws), _contract_code(contract), _stream(g.device))
    _lib.check(rc, "gwen_gcn_grad_weight_f32")
    return out


def grad_bias(g: Tensor) -> Tensor:
    g = g.contiguous()
    f = g.size(-1)
    rows = math.prod(g.shape[:-1])
    out = torch.empty(f, dtype=torch.float32, device=g.device)
    ws = _grad_workspace(rows, f, 1, g.device)
    with torch.cuda.device(g.device):
        rc = _lib.lib().gwen_gcn_grad_bias_f32(_ptr(g), _ptr(out), rows, f, f, _ptr(ws),
                                               _stream(g.device))
    _lib.check(rc, "gwen_gcn_grad_bias_f32")
    return out


class _ReduceTask(ctypes.Structure):          # gwen_reduce_task (include/gwen_hip.h)
    _fields_ = [("partial", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("count", ctypes.c_int64),
                ("nchunks", ctypes.c_int64)]


class GradBatch:
    """Many grad_weight / grad_bias reductions whose fixed-order finishes share ONE launch
    (gwen_reduce_chunks_batched, up to 32 per launch) instead of one each: stage 1 of every reduction is launched
    at once, the returned tensors are complete after ``finish()``.  Same chunking and summation order per reduction
    whatever else is in the batch (deterministic, run to run bitwise)."""

    def __init__(self) -> None:
        self._tasks: list = []
        self._keep: list = []
        self._dev = None

    def _add(self, partial: Tensor, out: Tensor, count: int, nch: int) -> None:
        self._tasks.append((partial.data_ptr(), out.data_ptr(), count, nch))
        self._keep += [partial, out]

    def grad_weight(self, g: Tensor, x: Tensor, contract: Optional[str] = None) -> Tensor:
        g = g.contiguous(); x = x.contiguous()
        fout, fin = g.size(-1), x.size(-1)
        rows = math.prod(g.shape[:-1])
        dev = self._dev = g.device
        out = torch.empty(fout, fin, dtype=torch.float32, device=dev)
        if rows == 0:
            return out.zero_()
        code = _contract_code(contract)
        nch = int(_lib.lib().gwen_gcn_grad_weight_chunks(rows, fin, fout, code))
        dst = out if nch == 1 else torch.empty(nch * fout * fin, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().gwen_gcn_grad_weight_partial_f32(_ptr(g), _ptr(x), _ptr(dst), rows, fin, fout, fout, fin,
                                                             code, _stream(dev))
        _lib.check(rc, "gwen_gcn_grad_weight_partial_f32")
        if nch > 1:
            self._add(dst, out, fout * fin, nch)
        return out

    def grad_weight_bias(self, g: Tensor, x: Tensor, contract: Optional[str] = None):
        """(g^T x, column sums of g): ONE stage-1 launch where the weight gradient runs on the LDS-staged split kernel (the
        g tile is in LDS anyway), the two separate launches elsewhere."""
        g = g.contiguous(); x = x.contiguous()
        fout, fin = g.size(-1), x.size(-1)
        rows = math.prod(g.shape[:-1])
        code = _contract_code(contract)
        L = _lib.lib()
        if rows == 0 or not L.gwen_gcn_grad_weight_bias_supported(fin, fout, code) or g.data_ptr() % 16 \
                or x.data_ptr() % 16:
            return self.grad_weight(g, x, contract), self.grad_bias(g)
        dev = self._dev = g.device
        gw = torch.empty(fout, fin, dtype=torch.float32, device=dev)
        gb = torch.empty(fout, dtype=torch.float32, device=dev)
        nch = int(L.gwen_gcn_grad_weight_chunks(rows, fin, fout, code))
        pw = gw if nch == 1 else torch.empty(nch * fout * fin, dtype=torch.float32, device=dev)
        pb = gb if nch == 1 else torch.empty(nch * fout, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.gwen_gcn_grad_weight_bias_partial_f32(_ptr(g), _ptr(x), _ptr(pw), _ptr(pb), rows, fin, fout, fout, fin,
                                                         code, _stream(dev))
        _lib.check(rc, "gwen_gcn_grad_weight_bias_partial_f32")
        if nch > 1:
            self._add(pw, gw, fout * fin, nch)
            self._add(pb, gb, fout, nch)
        return gw, gb

    def grad_bias(self, g: Tensor) -> Tensor:
        g = g.contiguous()
        f = g.size(-1)
        rows = math.prod(g.shape[:-1])
        dev = self._dev = g.device
        out = torch.empty(f, dtype=torch.float32, device=dev)
        if rows == 0:
            return out.zero_()
        nch = int(_lib.lib().gwen_gcn_grad_chunks(rows))
        dst = out if nch == 1 else torch.empty(nch * f, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().gwen_gcn_grad_bias_partial_f32(_ptr(g), _ptr(dst), rows, f, f, _stream(dev))
        _lib.check(rc, "gwen_gcn_grad_bias_partial_f32")
        if nch > 1:
            self._add(dst, out, f, nch)
        return out

    def finish(self) -> None:
        cap = 32                                             # GWEN_MAX_REDUCE_TASKS
        for i in range(0, len(self._tasks), cap):
            part = self._tasks[i:i + cap]
            arr = (_ReduceTask * len(part))(*[_ReduceTask(*t) for t in part])
            with torch.cuda.device(self._dev):
                rc = _lib.lib().gwen_reduce_chunks_batched(ctypes.cast(arr, ctypes.c_void_p), len(part),
                                                           _stream(self._dev))
            _lib.check(rc, "gwen_reduce_chunks_batched")
        self._tasks, self._keep = [], []


def relu_backward(y: Tensor, g: Tensor) -> Tensor:
    y = y.contiguous(); g = g.contiguous()
    out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        rc = _lib.lib().gwen_relu_backward_f32(_ptr(y), _ptr(g), _ptr(out), g.numel(),
                                               _stream(g.device))
    _lib.check(rc, "gwen_relu_backward_f32")
    return out


def _propagate_transposed(graph: GraphCSR, g: Tensor, contract: Optional[str]) -> Tensor:
    """A~^T g for the backward.  Small square graphs with wide rows (the reference's member graphs: 125 nodes x 1024 ..
    16 384 features) on a split precision: ONE dense product D^T g (K3 with g as the [K, N] operand, as the forward's K7
    contracts D h) instead of K2 walking the 125-entry rows (53.6 -> 10 us at 16 384 features); everything else: K2."""
    if g.dim() == 2 and g.size(-1) >= 256 and contract in ("3xbf16", "bf16x6", "f16x3"):
        d = graph.dense_transposed_square()
        if d is not None:
            return linear_nn(d, g, contract)
    return propagate(graph, g, transposed=True)


class GCNLayerFunction(torch.autograd.Function):
    """act(A~ (x W^T) + b): forward and backward entirely on the HIP kernels.

    ``order``: "auto" picks "small" = K7 on graphs of at most 256 nodes (one launch, dense adjacency),
    else "fused" = K4, one launch, whenever the widths allow; otherwise the
    two linear maps run as two launches, transform-first (x W^T, then aggregate at width Fout -- what
    PyG does) when Fout <= Fin, aggregate-first (A~ x at width Fin, then the projection with the
    bias/ReLU epilogue) when Fin < Fout, so the gather always runs at the narrower width.
    All three equal the reference's result up to fp32 rounding order (A~ is linear).
    """

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], graph: GraphCSR,
                relu: bool, order: str, packed: Optional[Tensor] = None) -> Tensor:
        fout, fin = weight.shape
        # one precision rule for both host paths (this Function and the stack launcher, forward.hip): an AUTO
        # layer contracts with its split ("auto": f16x3 = K8's scaled fp16 split / bf16x6 in every other kernel,
        # "auto_x6": bf16x6, "auto_x3": 3xbf16) whatever kernels it resolves to;
        # explicit orders ("transform_first", "aggregate_first", "fused_exact") use the exact fp32-input MFMA
        contract = contract_of_order(order)
        if order in AUTO_ORDERS:
            if graph.dense() is not None and _lib.lib().gwen_gcn_small_supported(graph.num_nodes, fin, fout,
                                                                                 _dense_code(contract)):
                order = "small"              # K7: the reference's member graphs (<= 256 nodes)
            elif graph.long_row_levels() is not None:
                # rows far beyond 8 entries: the fused kernels walk a row serially, the segment chain does not
                order = "aggregate_first" if fin < fout else "transform_first"
            elif wide_preferred(graph, x, fin, fout, contract):
                order = "wide"               # K8: K4's arithmetic, tile-staged (same backward)
            elif layer_supported(fin, fout):
                order = "fused"
            else:
                order = "aggregate_first" if fin < fout else "transform_first"
        if order == "small":
            out = small_layer(graph, x, weight, bias, relu, packed=packed if contract == "3xbf16" else None,
                              contract=contract)
            saved_in = x
        elif order == "wide":
            out = wide_layer(graph, x, weight, bias, relu, contract=contract)
            saved_in = x
        elif order in ("fused", "fused_x3", "fused_exact"):
            out = layer_fused(graph, x, weight, bias, relu, contract=contract)          # ("f16x3" is bf16x6 in K4)
            saved_in = x
        elif order == "transform_first":
            h = linear(x, weight, contract=contract)
            out = propagate(graph, h, bias, relu)
            saved_in = x
        elif order == "aggregate_first":
            agg = propagate(graph, x)
            out = linear(agg, weight, bias, relu, contract=contract)
            saved_in = agg
        else:
            raise ValueError(f"unknown order {order!r}")
        ctx.graph, ctx.relu, ctx.order, ctx.has_bias = graph, relu, order, bias is not None
        ctx.contract = contract          # the backward contracts on the layer's own precision ("f16x3": bf16x6 in K3)
        ctx.save_for_backward(saved_in, weight, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        saved_in, weight, out = ctx.saved_tensors
        g = grad_out.contiguous()
        if ctx.relu:
            g = relu_backward(out, g)
        gb = grad_bias(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        gx = gw = None
        if ctx.order in ("transform_first", "fused", "fused_x3", "fused_exact", "small", "wide"):   # out = act(A~ x W^T + b) either way
            gh = _propagate_transposed(ctx.graph, g, ctx.contract)  # A~^T g
            if ctx.needs_input_grad[1]:
                gw = grad_weight(gh, saved_in, ctx.contract)       # gh^T x
            if ctx.needs_input_grad[0]:
                gx = linear_nn(gh, weight, ctx.contract)                                 # gh W (split-K, W as stored)
        else:
            if ctx.needs_input_grad[1]:
                gw = grad_weight(g, saved_in, ctx.contract)        # g^T (A~ x)
            if ctx.needs_input_grad[0]:
                gagg = linear_nn(g, weight, ctx.contract)                                # g W
                gx = _propagate_transposed(ctx.graph, gagg, ctx.contract)   # A~^T (g W)
        return gx, gw, gb, None, None, None, None


def gcn_layer(x: Tensor, weight: Tensor, bias: Optional[Tensor], graph: GraphCSR,
              relu: bool = False, order: str = "auto", packed: Optional[Tensor] = None) -> Tensor:
    _require(x, "x")
    _require(weight, "weight")
    if x.device != weight.device or x.device != graph.device:
        raise RuntimeError("x, weight and the graph must be on the same device")
    return GCNLayerFunction.apply(x, weight, bias, graph, relu, order, packed)
