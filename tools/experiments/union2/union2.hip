// EXPERIMENT (not part of libgwen_hip.so): K2 at 64 channels with a block's WHOLE row range (RB rows, two blocks a CU)
// and its halo staged ONCE in LDS by LDS-DMA (own rows: contiguous KiB pieces; halo rows: per-lane row addresses), all
// loads in flight at once, then every gather served from LDS.  Round 1's union experiment staged 64-row tiles through
// registers (168 staged rows for 64 outputs: 18.6 us against K2's 15.4); here 196 rows stage ~271 (1.38 x instead of 8 x
// gathered bytes through L2).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int IMM>
__device__ inline void glds16(const void *base, uint32_t voff, uint32_t dst) {
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst), "n"(IMM) : "memory");
}
__device__ inline const char *uniform_ptr(const void *p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return reinterpret_cast<const char *>(((uint64_t)hi << 32) | lo);
}

#ifndef U2_ABL
#define U2_ABL 0
#endif
constexpr int F = 64;
template <int RB, int HC, int NW>
__global__ __launch_bounds__(NW * 64) void k_union2(const float *__restrict__ x, const float *__restrict__ val,   // [8N]
                                                    const uint16_t *__restrict__ lid,                            // [8N]
                                                    const int32_t *__restrict__ halo,                            // [nb][HC]
                                                    float *__restrict__ out, int N) {
  static_assert(RB % 4 == 0 && HC % 4 == 0, "pieces of four rows");
  __shared__ __attribute__((aligned(1024))) float xs[(RB + HC) * F];
  const uint32_t lds0 = (uint32_t)(uintptr_t)xs;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int q = lane & 15, rr = lane >> 4;
  const int b = blockIdx.x, r0 = b * RB;
  const int nown = N - r0 < RB ? N - r0 : RB;
  const char *xb = uniform_ptr(x);
  // halo ids first (the only dependent chain): piece p covers halo slots 4 p .. 4 p + 3
  constexpr int HP = HC / 4, OP = RB / 4;
  constexpr int HPW = (HP + NW - 1) / NW, OPW = (OP + NW - 1) / NW;
  int32_t hid[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int p = wave + NW * i;
    hid[i] = halo[(int64_t)b * HC + (p < HP ? 4 * p + rr : 0)];
  }
  // own rows: contiguous, no dependency
#pragma unroll
  for (int i = 0; i < OPW; ++i) {
    const int p = wave + NW * i;
    if (p < OP && !(U2_ABL & 1)) {
      int row = 4 * p + rr;
      row = row < nown ? row : nown - 1;
      glds16<0>(xb, (uint32_t)(((int64_t)(r0 + row) * F + 4 * q) * 4), lds0 + p * 1024);
    }
  }
  // this lane's entries (weights + local ids of its rows): rows wave * 4 + rr + 4 NW it
  constexpr int NIT = (RB + 4 * NW - 1) / (4 * NW);
  float4_t w0[NIT], w1[NIT];
  u32x4 ids[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int row = it * 4 * NW + wave * 4 + rr;
    row = row < nown ? row : nown - 1;
    const int64_t e = 8 * (int64_t)(r0 + row);
    w0[it] = *reinterpret_cast<const float4_t *>(val + e);
    w1[it] = *reinterpret_cast<const float4_t *>(val + e + 4);
    ids[it] = *reinterpret_cast<const u32x4 *>(lid + e);
  }
  // halo rows
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int p = wave + NW * i;
    if (p < HP && !(U2_ABL & 1)) glds16<0>(xb, (uint32_t)(((int64_t)hid[i] * F + 4 * q) * 4), lds0 + (OP + p) * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = it * 4 * NW + wave * 4 + rr;
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < ((U2_ABL & 2) ? 1 : 8); ++u) {
      const uint32_t word = ids[it][u >> 1];
      const int l = (u & 1) ? (word >> 16) : (word & 0xffff);
      const float wv = u < 4 ? w0[it][u & 3] : w1[it][u & 3];
      const float4_t v = *reinterpret_cast<const float4_t *>(xs + l * F + 4 * q);
      acc += float4_t{wv, wv, wv, wv} * v;
    }
    if (row < nown) *reinterpret_cast<float4_t *>(out + (int64_t)(r0 + row) * F + 4 * q) = acc;
  }
}

#ifndef U2_RB
#define U2_RB 196
#endif
#ifndef U2_HC
#define U2_HC 104
#endif
#ifndef U2_NW
#define U2_NW 8
#endif
extern "C" int union2_rb() { return U2_RB; }
extern "C" int union2_hc() { return U2_HC; }
extern "C" int union2_launch(const float *x, const float *val, const uint16_t *lid, const int32_t *halo, float *out, int N,
                             void *stream) {
  k_union2<U2_RB, U2_HC, U2_NW><<<(N + U2_RB - 1) / U2_RB, U2_NW * 64, 0, (hipStream_t)stream>>>(x, val, lid, halo, out, N);
  return (int)hipGetLastError();
}
