"""ctypes binding of libgwen_hip.so -- the only way gwen_amd reaches the GPU kernels.

There is NO CPU fallback: if the shared library is missing or a launcher fails, a RuntimeError is
raised (the reference's loops catch RuntimeError: /root/reference/src/gwen/models_gnn.py:389).
The library handle lives in this module, never on an nn.Module, so modules stay picklable
(``mp.spawn`` / ``mlflow.pytorch.log_model``: /root/reference/src/gwen/train_gnn.py:144-152,
/root/reference/src/gwen/models_gnn.py:387).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# GWEN_HIP_LIB: an experimental variant build (gwen_amd/build.py::build_variant) instead of the product library --
# for tools/experiments only; tests and bench.py run with it unset.
PRODUCT_LIB = os.path.join(_HERE, "libgwen_hip.so")
LIB_PATH = os.environ.get("GWEN_HIP_LIB") or PRODUCT_LIB


def is_variant() -> bool:
    """True when GWEN_HIP_LIB swapped the product library for an experimental build (ablations included: results may
    be wrong by construction).  tests/conftest.py refuses to run then; bench.py needs --allow-variant and says so."""
    return os.path.abspath(LIB_PATH) != os.path.abspath(PRODUCT_LIB)


def library_stamp() -> dict:
    """What is loaded: path, and whether the product library's source digest (gwen_amd/build.py) matches the sources."""
    info = {"path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE)), "variant": is_variant()}
    try:
        from . import build as _b
        with open(PRODUCT_LIB + ".stamp") as fh:
            stamp = fh.read().strip()
        info["source_digest"] = stamp[:16]
        info["matches_sources"] = stamp == _b._digest()
    except OSError:
        info["source_digest"] = None
    return info

_lib = None
_lock = threading.Lock()

_vp, _i64, _int, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float



class LayerDesc(C.Structure):
    """gwen_layer_desc (include/gwen_hip.h)."""
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("fin", C.c_int32), ("fout", C.c_int32),
                ("relu", C.c_int32), ("order", C.c_int32), ("packed", C.c_void_p),
                ("contract", C.c_int32), ("reserved", C.c_int32)]


class LaunchInfo(C.Structure):
    """gwen_launch_info (include/gwen_hip.h)."""
    _fields_ = [("kind", C.c_int32), ("layer", C.c_int32), ("fin", C.c_int32), ("fout", C.c_int32)]


class GraphDesc(C.Structure):
    """gwen_graph (include/gwen_hip.h)."""
    _fields_ = [("N", C.c_int64), ("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p),
                ("g_rowptr", C.c_void_p), ("g_col", C.c_void_p), ("g_val", C.c_void_p),
                ("dense", C.c_void_p), ("t_rows", C.c_void_p), ("t_lid", C.c_void_p),
                ("t_val", C.c_void_p), ("union_max", C.c_int64)]


ORDER_AUTO, ORDER_TRANSFORM_FIRST, ORDER_AGGREGATE_FIRST, ORDER_FUSED, ORDER_FUSED_EXACT = -1, 0, 1, 2, 3
CONTRACT_BF16X3, CONTRACT_F32, CONTRACT_BF16X6, CONTRACT_F16X3 = 0, 1, 2, 3          # GWEN_CONTRACT_* (include/gwen_hip.h)
CONTRACT_NAMES = {"3xbf16": CONTRACT_BF16X3, "bf16x3": CONTRACT_BF16X3, "fp32": CONTRACT_F32, "bf16x6": CONTRACT_BF16X6,
                  "f16x3": CONTRACT_F16X3}


def dense_contract(code: int) -> int:
    """"f16x3" on a layer means fp32-class on the kernel's own split: K8 has the scaled fp16 split at every
    width, every other kernel -- K3, K4, K5, K7 -- runs bf16x6 (csrc/forward.hip ``dense_contract``)."""
    return CONTRACT_BF16X6 if code == CONTRACT_F16X3 else code


def wide_contract(fin: int, fout: int, code: int) -> int:
    """What K8 runs for a layer contract (csrc/forward.hip ``wide_contract``): f16x3 where it has it (every supported
    width pair today), else bf16x6."""
    if code == CONTRACT_F16X3 and not lib().gwen_gcn_wide_contract_supported(fin, fout, code):
        return CONTRACT_BF16X6
    return code
KIND_PROPAGATE, KIND_LINEAR, KIND_LAYER, KIND_CHAIN, KIND_SMALL, KIND_WIDE = 2, 3, 4, 5, 6, 8
ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2
EW_MUL, EW_ADD = 0, 1
KIND_NAMES = {KIND_PROPAGATE: "propagate", KIND_LINEAR: "linear", KIND_LAYER: "layer",
              KIND_CHAIN: "chain", KIND_SMALL: "small", KIND_WIDE: "wide"}

# name -> (restype, argtypes); mirrors include/gwen_hip.h one to one
SIGNATURES = {
    "gwen_hip_version": (C.c_char_p, []),
    "gwen_hip_error_string": (C.c_char_p, [_int]),
    "gwen_gcn_prep_workspace_bytes": (_int, [_i64, _i64, C.POINTER(C.c_size_t)]),
    "gwen_gcn_prep": (_int, [_vp, _vp, _i64, _i64, _int, _f32, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                             _vp, C.c_size_t, _vp]),
    "gwen_gcn_prep_rect": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                                  C.c_size_t, _vp]),
    "gwen_gcn_transpose": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "gwen_gcn_transpose_rect": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "gwen_gcn_group8_capacity": (_i64, [_i64, _i64]),
    "gwen_gcn_group8": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "gwen_checksum_workspace_bytes": (_i64, []),
    "gwen_checksum128": (_int, [_vp, _i64, _vp, _vp, _i64, _vp]),
    "gwen_gcn_segments_capacity": (_i64, [_i64, _i64, _i64]),
    "gwen_gcn_segments": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "gwen_gcn_propagate_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                      _i64, _i64, _int, _vp]),
    "gwen_gcn_linear_workspace_floats": (_i64, [_i64, _i64, _i64]),
    "gwen_gcn_linear_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp,
                                   _i64, _vp]),
    "gwen_gcn_linear_nn_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _vp, _i64, _vp]),
    "gwen_gcn_layer_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                  _i64, _i64, _i64, _int, _int, _vp]),
    "gwen_gcn_layer_supported": (_int, [_i64, _i64]),
    "gwen_gcn_chain_supported": (_int, [_i64, _i64, _i64, _int, _int]),
    "gwen_gcn_chain_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int,
                                  _int, _i64, _i64, _i64, _int, _vp]),
    "gwen_gcn_tiles64_count": (_i64, [_i64]),
    "gwen_gcn_tiles64": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gwen_cluster_rows64_host": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "gwen_gcn_wide_supported": (_int, [_i64, _i64]),
    "gwen_gcn_wide_preferred": (_int, [_i64, _i64, _i64, _i64]),
    "gwen_gcn_wide_contract_supported": (_int, [_i64, _i64, _int]),
    "gwen_gcn_wide_layer_f32": (_int, [_vp] * 7 + [_i64] * 8 + [_int, _i64, _int, _vp]),
    "gwen_gnn_forward_scratch_floats": (_i64, [_i64, _i64, C.POINTER(LayerDesc), C.c_int32]),
    "gwen_gnn_forward_f32": (_int, [C.POINTER(GraphDesc), C.POINTER(LayerDesc), C.c_int32, _vp, _vp,
                                    _vp, _i64, _i64, _vp, C.POINTER(C.c_void_p),
                                    C.POINTER(LaunchInfo), C.c_int32, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_void_p)]),
    "gwen_gcn_layer_bwd_f32": (_int, [_vp] * 8 + [_i64] * 4 + [_int, _vp]),
    "gwen_gcn_layer_bwd_bias_rows": (_i64, [_i64, _i64]),
    "gwen_gcn_layer_bwd_bias_f32": (_int, [_vp] * 8 + [_i64] * 4 + [_int, _vp, _vp, _vp]),
    "gwen_gnn_backward_scratch_floats": (_i64, [_i64, _i64, C.POINTER(LayerDesc), C.c_int32]),
    "gwen_gnn_backward_f32": (_int, [C.POINTER(GraphDesc), C.POINTER(LayerDesc), C.c_int32, _vp,
                                     C.POINTER(C.c_void_p), _vp, _vp, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), _vp, _i64, _i64, _vp]),
    "gwen_event_create": (_int, [C.POINTER(C.c_void_p)]),
    "gwen_event_destroy": (_int, [_vp]),
    "gwen_event_record": (_int, [_vp, _vp]),
    "gwen_event_synchronize": (_int, [_vp]),
    "gwen_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "gwen_gcn_grad_workspace_floats": (_i64, [_i64, _i64, _i64]),
    "gwen_gcn_grad_weight_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "gwen_gcn_grad_bias_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "gwen_relu_backward_f32": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "gwen_gcn_grad_chunks": (_i64, [_i64]),
    "gwen_gcn_grad_weight_partial_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp]),
    "gwen_gcn_grad_weight_chunks": (_i64, [_i64, _i64, _i64, _int]),
    "gwen_gcn_grad_bias_partial_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "gwen_gcn_grad_weight_bias_supported": (_int, [_i64, _i64, _int]),
    "gwen_gcn_grad_weight_bias_partial_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp]),
    "gwen_reduce_chunks_batched": (_int, [_vp, C.c_int32, _vp]),
    "gwen_transpose_batched": (_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "gwen_gcn_small_pad": (_int, [_i64]),
    "gwen_gcn_small_supported": (_int, [_i64, _i64, _i64, _int]),
    "gwen_gcn_small_workspace_floats": (_i64, [_i64, _i64, _i64, _i64]),
    "gwen_gcn_dense_f32": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "gwen_gcn_small_pack_bytes": (_i64, [_i64, _i64]),
    "gwen_gcn_small_pack_f32": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "gwen_gcn_small_layer_f32": (_int, [_vp] * 6 + [_i64] * 6 + [_int, _vp, _i64, _int, _vp]),
    "gwen_mlp2_supported": (_int, [_i64]),
    "gwen_edge_tiles_count": (_i64, [_i64, _i64]),
    "gwen_edge_tiles": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "gwen_mlp2_f32": (_int, [_vp] * 4 + [_i64, _i64, _vp, _vp, _i64, _i64] + [_vp] * 5 + [_i64, _i64, _int, _vp, _vp, _i64,
                              _vp, _i64, _int, _vp, C.c_size_t, _vp]),
    "gwen_mlp2_workspace_bytes": (_i64, [_i64]),
    "gwen_mlp2_rows": (_int, [_i64]),
    "gwen_act_pair_f32": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _int, _vp]),
    "gwen_act_pair_seg_f32": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "gwen_gather_add_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "gwen_ew_f32": (_int, [_int, _vp, _vp, _vp, _i64, _vp]),
    "gwen_mlp2_bwd_supported": (_int, [_i64]),
    "gwen_mlp2_bwd_rows": (_i64, [_i64]),
    "gwen_mlp2_bwd_f32": (_int, [_vp] * 5 + [_i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, C.c_size_t, _vp]),
    "gwen_masked_l1_workspace_floats": (_i64, []),
    "gwen_masked_l1_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
}


class GwenHipError(RuntimeError):
    """A launcher of libgwen_hip.so returned non-zero."""


def lib() -> C.CDLL:
    """Load libgwen_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: the HIP extension has not been built "
                        "(run `python -m gwen_amd.build`); gwen_amd has no CPU fallback")
                L = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)      # AttributeError if the ABI lost a symbol
                    fn.restype, fn.argtypes = res, args
                _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().gwen_hip_error_string(rc).decode()
        raise GwenHipError(f"{what} failed with code {rc}: {msg}")


def version() -> str:
    return lib().gwen_hip_version().decode()
