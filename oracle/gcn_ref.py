"""ctypes binding of oracle/libgcn_ref.so (plain-C oracle)  --  TEST INFRASTRUCTURE.

PARITY UNPINNED (see oracle/gcn_ref.c header).  Built by ``make -C oracle`` or
``__graft_entry__.build()``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgcn_ref.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        i64, f32p, f64p, i64p, vp = C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p
        L.gcn_ref_norm.restype = C.c_int64
        L.gcn_ref_norm.argtypes = [vp, vp, i64, i64, C.c_int, C.c_float, vp, vp, vp]
        for name in ("gcn_ref_linear_f32", "gcn_ref_linear_fma_f32"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp, vp, vp, i64, i64, i64]
        L.gcn_ref_propagate_f32.restype = None
        L.gcn_ref_propagate_f32.argtypes = [vp, vp, vp, i64, vp, vp, C.c_int, vp, i64, i64]
        for name in ("gcn_ref_conv_f32", "gcn_ref_conv_f64"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, i64, i64, i64, i64]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def norm(edge_index, edge_weight, n, add_self_loops=True, fill=1.0):
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    e = ei.shape[1]
    ew = _f32(edge_weight)
    s = np.empty(e + n, np.int64); d = np.empty(e + n, np.int64); w = np.empty(e + n, np.float32)
    m = lib().gcn_ref_norm(_p(ei), _p(ew), n, e, int(add_self_loops), fill, _p(s), _p(d), _p(w))
    if m < 0:
        raise IndexError("edge_index out of range")
    return s[:m].copy(), d[:m].copy(), w[:m].copy()


def linear(x, W, fma=False):
    x = _f32(x); W = _f32(W)
    h = np.empty((x.shape[0], W.shape[0]), np.float32)
    fn = lib().gcn_ref_linear_fma_f32 if fma else lib().gcn_ref_linear_f32
    fn(_p(x), _p(W), _p(h), x.shape[0], x.shape[1], W.shape[0])
    return h


def propagate(src, dst, w, h, bias=None, relu=False):
    h = _f32(h); n, f = h.shape
    src = np.ascontiguousarray(src, np.int64); dst = np.ascontiguousarray(dst, np.int64); w = _f32(w)
    b = _f32(bias)
    out = np.empty((n, f), np.float32)
    lib().gcn_ref_propagate_f32(_p(src), _p(dst), _p(w), len(w), _p(h), _p(b), int(relu), _p(out), n, f)
    return out


def conv(x, edge_index, W, bias=None, edge_weight=None, relu=False, add_self_loops=True, fill=1.0, f64=False):
    x = _f32(x); W = _f32(W); b = _f32(bias); ew = _f32(edge_weight)
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    n, fin = x.shape; fout = W.shape[0]
    out = np.empty((n, fout), np.float64 if f64 else np.float32)
    fn = lib().gcn_ref_conv_f64 if f64 else lib().gcn_ref_conv_f32
    rc = fn(_p(x), _p(ei), _p(ew), _p(W), _p(b), int(relu), int(add_self_loops), fill, _p(out), n, ei.shape[1], fin, fout)
    if rc != 0:
        raise IndexError("edge_index out of range")
    return out
