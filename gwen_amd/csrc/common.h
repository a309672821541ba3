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

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
