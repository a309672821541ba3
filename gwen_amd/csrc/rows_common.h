// Helpers shared by the row-stationary kernels (interact_rows.hip, linear_rows.hip): compile-time loops, the
// LDS-DMA instruction with its counted waits, the 3xbf16 split of eight values.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int N, typename Fn, int I = 0>
__device__ inline void static_for(Fn &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, Fn, I + 1>(static_cast<Fn &&>(f));
  }
}

__device__ inline const char *uniform_ptr(const void *p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return reinterpret_cast<const char *>(((uint64_t)hi << 32) | lo);
}

// one LDS-DMA wave instruction: lane l copies 16 B from base + voff(l) + IMM to LDS byte address dst + IMM + 16 l
template <int IMM>
__device__ inline void glds16(const void *base, uint32_t voff, uint32_t dst) {
  static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
  uint32_t keep;       // M0 is compiler-reserved: save and restore it inside the statement
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst), "n"(IMM) : "memory");
}

template <int N>
__device__ inline void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ inline void split8(const float4_t a, const float4_t b, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __bf16 h = (__bf16)a[i];
    hi[i] = h; lo[i] = (__bf16)(a[i] - (float)h);
    h = (__bf16)b[i];
    hi[i + 4] = h; lo[i + 4] = (__bf16)(b[i] - (float)h);
  }
}

}  // namespace
