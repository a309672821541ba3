// Shared helpers for libgwen_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gwen_hip.h"

#define GWEN_HIP_CHECK(expr)                       \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

// A kernel launch reports configuration errors through hipGetLastError only.
#define GWEN_LAUNCH_CHECK()                        \
  do {                                             \
    hipError_t _e = hipGetLastError();             \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

static inline hipStream_t gwen_stream(gwen_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline size_t gwen_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline bool gwen_aligned(const void *p, size_t a) {
  return (reinterpret_cast<uintptr_t>(p) % a) == 0;
}

// contraction of a layer's dense part in the stack launchers (forward.hip, backward.hip): AUTO / FUSED layers carry it
// (bf16x3, bf16x6 or f16x3), explicit orders are fp32
static inline int gwen_contract_of(const gwen_layer_desc &L) {
  if (L.order == GWEN_ORDER_AUTO || L.order == GWEN_ORDER_FUSED)
    return L.contract == GWEN_CONTRACT_BF16X6 || L.contract == GWEN_CONTRACT_F16X3 ? L.contract : GWEN_CONTRACT_BF16X3;
  return GWEN_CONTRACT_F32;
}
// GWEN_CONTRACT_F16X3 on a layer = "fp32-class, the kernel's own split": K8 has the scaled fp16 split; every
// other kernel (K3, K4, K5, K7, the backward) runs its fp32-class split, bf16x6
static inline int gwen_dense_contract(int c) { return c == GWEN_CONTRACT_F16X3 ? GWEN_CONTRACT_BF16X6 : c; }

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
