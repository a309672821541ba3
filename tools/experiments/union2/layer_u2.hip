// EXPERIMENT: K4 (one GCN layer, aggregate-first) at 64 -> 64 with the block's whole row range + halo staged ONCE in LDS
// (union2.hip) and K4's own arithmetic behind it (same fma chain per row, same split, same MFMA order: bitwise K4).
//   hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -I../../../gwen_amd/csrc layer_u.hip -o liblayer_u.so
#include "common.h"
#include "split.h"
#include "rows_common.h"
#ifndef U2_AHEAD
#define U2_AHEAD 3
#endif
#ifndef U2_ABL
#define U2_ABL 0
#endif

namespace {
using gwen::bf16x4;
using gwen::bf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int FIN = 64;

template <int FOUT, int NS, int RB, int HC>
__global__ __launch_bounds__(1024) void k_layer_u(const float *__restrict__ x, const float *__restrict__ val,
                                                  const uint16_t *__restrict__ lid, const int32_t *__restrict__ halo,
                                                  const float *__restrict__ W, const float *__restrict__ bias,
                                                  float *__restrict__ out, int N, int relu) {
  constexpr int NW = 16, ST = 64, PB = 72, NJ = FOUT / 16, KS = FIN / 32;
  static_assert(NJ == 4, "one (row tile, column tile) pair per wave");
  constexpr int kUnion = (RB + HC) * FIN * 4, kImg = ST * PB;
  static_assert(RB % 4 == 0 && HC % 4 == 0, "pieces of four rows");
  __shared__ __attribute__((aligned(1024))) char lds[kUnion + NS * kImg * 2];
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const float *xs = reinterpret_cast<const float *>(lds);
  __bf16 *timg = reinterpret_cast<__bf16 *>(lds + kUnion);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int gl = lane & 15, gr = lane >> 4, mi = lane & 15, mh = lane >> 4;
  const int b = blockIdx.x, r0 = b * RB;
  const int nown = N - r0 < RB ? N - r0 : RB;
  const char *xb = uniform_ptr(x);
  constexpr int HP = HC / 4, OP = RB / 4, HPW = (HP + NW - 1) / NW, OPW = (OP + NW - 1) / NW;
  int32_t hid[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int p = wave + NW * i;
    hid[i] = halo[(int64_t)b * HC + (p < HP ? 4 * p + gr : 0)];
  }
#pragma unroll
  for (int i = 0; i < OPW; ++i) {
    const int p = wave + NW * i;
    if (p < OP) {
      int row = 4 * p + gr;
      row = row < nown ? row : nown - 1;
      glds16<0>(xb, (uint32_t)(((int64_t)(r0 + row) * FIN + 4 * gl) * 4), lds0 + p * 1024);
    }
  }
  // this wave's W fragments (its 16 output columns), as K4
  const int j = wave % NJ, tt = wave / NJ;
  const float *wrow = W + (int64_t)(j * 16 + mi) * FIN;
  bf16x8 bw[KS][NS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float wv[8];
    const float *wp = wrow + 8 * (4 * ks + mh);
#pragma unroll
    for (int i = 0; i < 8; i += 4) {
      const float4_t w4 = *reinterpret_cast<const float4_t *>(wp + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[i + e] = w4[e];
    }
    gwen::split_images<8, NS>(wv, bw[ks]);
  }
  float4_t bv4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv4 = *reinterpret_cast<const float4_t *>(bias + j * 16 + 4 * mh);
  // entries of sub-tile 0
  auto entries = [&](int s, float4_t &w0, float4_t &w1, u32x4 &ids) {
    int row = s * ST + wave * 4 + gr;
    row = row < nown ? row : nown - 1;
    const int64_t e = 8 * (int64_t)(r0 + row);
    w0 = *reinterpret_cast<const float4_t *>(val + e);
    w1 = *reinterpret_cast<const float4_t *>(val + e + 4);
    ids = *reinterpret_cast<const u32x4 *>(lid + e);
  };
  // ROLES: waves 0-7 aggregate 32-row sub-tile t into image buffer t & 1 while waves 8-15 contract and store sub-tile
  // t - 1 from the other buffer; one barrier a step
  constexpr int ST2 = 32, NSTEP = (RB + ST2 - 1) / ST2, AHEAD = U2_AHEAD, kImg2 = ST2 * PB;
  const bool producer = wave < 8;
  const int cw = wave - 8, j2 = cw & 3, tt2 = cw >> 2;
  auto entries2 = [&](int t, float4_t &a0, float4_t &a1, u32x4 &ii) {
    int row = t * ST2 + (wave & 7) * 4 + gr;
    row = row < nown ? row : nown - 1;
    const int64_t e = 8 * (int64_t)(r0 + row);
    a0 = *reinterpret_cast<const float4_t *>(val + e);
    a1 = *reinterpret_cast<const float4_t *>(val + e + 4);
    ii = *reinterpret_cast<const u32x4 *>(lid + e);
  };
  float4_t w0[NSTEP], w1[NSTEP];
  u32x4 ids[NSTEP];
  if (producer) {
#pragma unroll
    for (int t = 0; t < AHEAD && t < NSTEP; ++t) entries2(t, w0[t], w1[t], ids[t]);
  }
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int p = wave + NW * i;
    if (p < HP) glds16<0>(xb, (uint32_t)(((int64_t)hid[i] * FIN + 4 * gl) * 4), lds0 + (OP + p) * 1024);
  }
  // the consumers' W fragments (column tile j2)
  const float *wrow2 = W + (int64_t)((producer ? 0 : j2) * 16 + mi) * FIN;
  bf16x8 bw2[KS][NS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float wv[8];
    const float *wp = wrow2 + 8 * (4 * ks + mh);
#pragma unroll
    for (int i = 0; i < 8; i += 4) {
      const float4_t w4 = *reinterpret_cast<const float4_t *>(wp + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[i + e] = w4[e];
    }
    gwen::split_images<8, NS>(wv, bw2[ks]);
  }
  float4_t bv42 = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv42 = *reinterpret_cast<const float4_t *>(bias + (producer ? 0 : j2) * 16 + 4 * mh);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int nstep = (nown + ST2 - 1) / ST2;
  static_for<NSTEP + 1>([&](auto tt_) {
    constexpr int t = decltype(tt_)::value;
    if (t <= nstep) {
      if (producer) {
        if constexpr (t < NSTEP) {
          if (t < nstep) {
            if constexpr (t + AHEAD < NSTEP) entries2(t + AHEAD, w0[t + AHEAD], w1[t + AHEAD], ids[t + AHEAD]);
            float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const uint32_t word = ids[t][u >> 1];
              const int l = (u & 1) ? (word >> 16) : (word & 0xffff);
              const float wv = u < 4 ? w0[t][u & 3] : w1[t][u & 3];
              const float4_t v = *reinterpret_cast<const float4_t *>(xs + l * FIN + 4 * gl);
              acc = __builtin_elementwise_fma(float4_t{wv, wv, wv, wv}, v, acc);
            }
            const float a4[4] = {acc[0], acc[1], acc[2], acc[3]};
            bf16x4 im[NS];
            gwen::split_images<4, NS>(a4, im);
            const int lr = wave * 4 + gr;
            __bf16 *buf = timg + (t & 1) * NS * kImg2;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) *reinterpret_cast<bf16x4 *>(buf + s_ * kImg2 + lr * PB + gl * 4) = im[s_];
          }
        }
      } else {
        if constexpr (t >= 1) {
          const __bf16 *buf = timg + ((t - 1) & 1) * NS * kImg2;
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          const int arow = (tt2 * 16 + mi) * PB;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            bf16x8 a[NS];
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_)
              a[s_] = *reinterpret_cast<const bf16x8 *>(buf + s_ * kImg2 + arow + 8 * (4 * ks + mh));
            d = gwen::mma_split<8, NS>(bw2[ks], a, d);
          }
          const int lr = (t - 1) * ST2 + tt2 * 16 + mi;
          float4_t o = float4_t{d[0], d[1], d[2], d[3]} + bv42;
          if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = o[q] < 0.0f ? 0.0f : o[q];
          }
          if (lr < nown) *reinterpret_cast<float4_t *>(out + (int64_t)(r0 + lr) * FOUT + j2 * 16 + 4 * mh) = o;
        }
      }
      lds_barrier();
    }
  });
}
}  // namespace

#ifndef U2_AHEAD
#define U2_AHEAD 3
#endif
#ifndef U2_AHEAD
#define U2_AHEAD 3
#endif
#ifndef U2_ABL
#define U2_ABL 0
#endif
#ifndef U2_RB
#define U2_RB 392
#endif
#ifndef U2_HC
#define U2_HC 136
#endif
#ifndef U2_NS
#define U2_NS 3
#endif
extern "C" int union2_rb() { return U2_RB; }
extern "C" int union2_hc() { return U2_HC; }
extern "C" int union2_ns() { return U2_NS; }
extern "C" int layer_u_launch(const float *x, const float *val, const uint16_t *lid, const int32_t *halo, const float *W,
                              const float *bias, float *out, int N, int relu, void *stream) {
  k_layer_u<64, U2_NS, U2_RB, U2_HC><<<(N + U2_RB - 1) / U2_RB, 1024, 0, (hipStream_t)stream>>>(x, val, lid, halo, W, bias,
                                                                                                 out, N, relu);
  return (int)hipGetLastError();
}
