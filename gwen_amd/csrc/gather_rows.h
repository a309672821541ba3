// Phase 1 of K4 / K5: gather + aggregate destination rows from the GROUPED layout (gwen_gcn_group8).
//
// One wave handles R = 64 / (FIN/4) rows per pass (FIN/4 lanes per row, 16-B loads) and NP passes.
// Every load is unconditional (absent rows read the all-zero null group): a load under a per-lane
// condition makes hipcc branch around it and wait vmcnt(0), which serialises the gathers.
// The passes are software-pipelined by hand -- the compiler keeps them strictly one after another:
// the column indices of pass p+1 are requested before pass p's row gathers are waited for, and
// pass p's weights ride along with its gathers, so a pass costs one memory round trip, not two.
// UNI: uniform layout (every row exactly one group, rowptr == NULL): row r is the group at 8 r.
#pragma once
#include "common.h"

namespace gwen {

typedef int int4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

template <int FIN>
__device__ inline float4_t gather_group(const char *xb, uint32_t lane_off, const int4_u &c0,
                                        const int4_u &c1, const float4_u &w0, const float4_u &w1,
                                        float4_t acc) {
  constexpr uint32_t kRowBytes = FIN * 4;        // x rows are contiguous: base + 32-bit byte offset
  float4_t v[8];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    v[u] = *reinterpret_cast<const float4_t *>(xb + (uint64_t)((uint32_t)c0[u] * kRowBytes + lane_off));
    v[u + 4] = *reinterpret_cast<const float4_t *>(xb + (uint64_t)((uint32_t)c1[u] * kRowBytes + lane_off));
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    acc = __builtin_elementwise_fma(float4_t{w0[u], w0[u], w0[u], w0[u]}, v[u], acc);
#pragma unroll
  for (int u = 0; u < 4; ++u)
    acc = __builtin_elementwise_fma(float4_t{w1[u], w1[u], w1[u], w1[u]}, v[u + 4], acc);
  return acc;
}

// sink(lr, acc): lr = row index inside the block (p * RB + wave * R + gr), acc = aggregated 4 floats
template <int FIN, int NP, int RB, bool UNI, typename Sink>
__device__ inline void gather_passes(const int32_t *__restrict__ rowptr,
                                     const int32_t *__restrict__ col,
                                     const float *__restrict__ val, const char *xb, int32_t N,
                                     int b0, int wave, int gr, uint32_t lane_off, Sink &&sink) {
  constexpr int R = 64 / (FIN / 4);
  // group offset and row end of pass p (absent rows: the null group, which ends at once)
  auto locate = [&](int p, int32_t &s, int32_t &rb) {
    const int r = b0 + p * RB + wave * R + gr;
    const bool ok = r < N;
    if constexpr (UNI) {
      s = ok ? 8 * r : 8 * N;
      rb = 0;
    } else {
      const int32_t ra = rowptr[ok ? r : N];
      rb = rowptr[ok ? r + 1 : N];
      s = rb > ra ? ra : rowptr[N];
    }
  };
  int32_t s, rb;
  locate(0, s, rb);
  int4_u c0 = *reinterpret_cast<const int4_u *>(col + s);
  int4_u c1 = *reinterpret_cast<const int4_u *>(col + s + 4);
  // a rolled loop: fully unrolled, hipcc hoists several passes' gathers at once and the register
  // count (172 VGPRs at Fin = 128) costs more occupancy than the extra overlap returns
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    int4_u n0 = c0, n1 = c1;
    int32_t ns = s, nrb = rb;
    if (p + 1 < NP) {                                    // next pass's source rows, one pass ahead
      locate(p + 1, ns, nrb);
      n0 = *reinterpret_cast<const int4_u *>(col + ns);
      n1 = *reinterpret_cast<const int4_u *>(col + ns + 4);
    }
    const float4_u w0 = *reinterpret_cast<const float4_u *>(val + s);
    const float4_u w1 = *reinterpret_cast<const float4_u *>(val + s + 4);
    float4_t acc = gather_group<FIN>(xb, lane_off, c0, c1, w0, w1, float4_t{0.f, 0.f, 0.f, 0.f});
    if constexpr (!UNI) {
      for (int32_t q = s + 8; q < rb; q += 8) {          // rows longer than one group of 8
        const int4_u d0 = *reinterpret_cast<const int4_u *>(col + q);
        const int4_u d1 = *reinterpret_cast<const int4_u *>(col + q + 4);
        const float4_u x0 = *reinterpret_cast<const float4_u *>(val + q);
        const float4_u x1 = *reinterpret_cast<const float4_u *>(val + q + 4);
        acc = gather_group<FIN>(xb, lane_off, d0, d1, x0, x1, acc);
      }
    }
    sink(p * RB + wave * R + gr, acc);
    c0 = n0; c1 = n1; s = ns; rb = nrb;
  }
}

}  // namespace gwen
