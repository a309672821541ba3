// K5 -- K4 with the NEXT layer's projection chained on, and the "activation first" variant.
//
// A~ is linear, so each GCNConv layer may run transform-first (PyG's order: h = x W^T, then
// aggregate at width Fout) or aggregate-first (aggregate at width Fin, then project); the gather --
// the HBM/fabric-bound part -- is cheapest at min(Fin, Fout).  To gather a SHRINKING layer
// (Fout < Fin) at Fout, its projection has to exist before its gather starts: this kernel appends it
// to the kernel that produces the layer's input, while the rows are still in LDS:
//
//   PRE = 0:  t1 = act(A~ x W1^T + b) ;  out = t1 W2^T          (layer l, then layer l+1's `lin`)
//   PRE = 1:  t0 = act(A~ h + b)      ;  out = t0 W1^T          (layer l was pre-projected: bias and
//                                                                 ReLU come BEFORE the contraction,
//                                                                 which is layer l+1's `lin`)
// e.g. GNNModel(C=64,H=64) = 64->64->32->16->32->64->64 runs as
//   [gather 64, W1, b1+ReLU, W2 -> 32] [gather 32, b2+ReLU, W3 -> 16] [gather 16, b3+ReLU]
//   [gather 16, W4, ...] [gather 32, W5, ...] [gather 64, W6, ...]   -- gathered widths 64,32,16,16,32,64
// instead of 64,64,32,16,32,64 (reference call sites: /root/reference/src/gwen/models_gnn.py:147-149,
// :204-206; the re-bracketing only changes fp32 rounding order).
// Structure, layout, contraction (3xbf16 split, fp32 accumulate) and numerics are K4's (layer.hip).
#include "common.h"
#include "gather_rows.h"

namespace {

constexpr int kTile = 16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int K> struct BF;
template <> struct BF<8> { using T = bf16x8; };
template <> struct BF<4> { using T = bf16x4; };

template <int K>
__device__ inline void split_bf16(const float (&x)[K], typename BF<K>::T &hi, typename BF<K>::T &lo) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}

constexpr int pitch_bf16(int f) { return ((f / 2) % 16 == 8 ? f / 2 : f / 2 + 8) * 2; }

// B fragments (hi/lo) of output-column tile j of a [FO, FI] weight: W[16 j + mi][KF (4 ks + mh) .. +KF)
template <int FI>
struct Frag {
  static constexpr int KF = FI >= 32 ? 8 : 4;
  static constexpr int KS = FI / (4 * KF);
  using T = typename BF<KF>::T;
  T hi[KS], lo[KS];
  __device__ inline void load(const float *W, int j, int mi, int mh) {
    const float *wrow = W + (int64_t)(j * 16 + mi) * FI;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float wv[KF];
      const float *wp = wrow + KF * (4 * ks + mh);
#pragma unroll
      for (int i = 0; i < KF; i += 4) {
        const float4_t w4 = *reinterpret_cast<const float4_t *>(wp + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[i + e] = w4[e];
      }
      split_bf16<KF>(wv, hi[ks], lo[ks]);
    }
  }
  // d += W-fragment (A operand) x tile rows arow.. from the hi/lo LDS images (B operand): the product
  // comes out TRANSPOSED -- lane (mi, mh) holds row mi, columns 16 j + 4 mh .. +3
  __device__ inline f32x4 mma(const __bf16 *thi, const __bf16 *tlo, int arow, int mh, f32x4 d) const {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const T ahi = *reinterpret_cast<const T *>(thi + arow + KF * (4 * ks + mh));
      const T alo = *reinterpret_cast<const T *>(tlo + arow + KF * (4 * ks + mh));
      if constexpr (KF == 8) {
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[ks], alo, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo[ks], ahi, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[ks], ahi, d, 0, 0, 0);
      } else {
        d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hi[ks], alo, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(lo[ks], ahi, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hi[ks], ahi, d, 0, 0, 0);
      }
    }
    return d;
  }
};

// FIN: gathered width.  F1: width after the first contraction.  F2: width after the second (0: none).
template <int FIN, int F1, int F2, bool PRE>
struct Cfg {
  static constexpr int G = FIN / 4, R = 64 / G;
  static constexpr int NJ1 = F1 / 16, NJ2 = F2 / 16;
  static constexpr int NWB = NJ1 > 4 ? 8 : 4;
  static constexpr int RB = NWB * R;
  static constexpr int BRMIN = FIN >= 128 ? 128 : 64;
  static constexpr int BR = RB > BRMIN ? RB : BRMIN;
  static constexpr int NP = BR / RB, NT = BR / kTile;
  static constexpr int PB0 = pitch_bf16(FIN), PB1 = pitch_bf16(F1);
  static constexpr int FW = F2 > 0 ? F2 : F1;              // stored width
  static constexpr size_t lds_elems = (size_t)2 * BR * PB0 + (F2 > 0 ? (size_t)2 * BR * PB1 : 0);
  static_assert(!(PRE && F2 > 0), "activation-first has one contraction");
  static_assert(NWB % NJ1 == 0 && (F2 == 0 || NWB % NJ2 == 0), "waves must tile the columns");
};

template <int FIN, int F1, int F2, bool PRE, bool UNI>
__global__ __launch_bounds__((F1 > 64 ? 512 : 256)) void k_chain(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, const float *__restrict__ W1,
    const float *__restrict__ W2, const float *__restrict__ bias, float *__restrict__ out, int32_t N,
    int64_t mstride_x, int64_t mstride_o, int relu) {
  using C = Cfg<FIN, F1, F2, PRE>;
  __shared__ __attribute__((aligned(16))) __bf16 lds[C::lds_elems];
  __bf16 *t0hi = lds, *t0lo = lds + C::BR * C::PB0;                      // aggregated rows [BR][PB0]
  __bf16 *t1hi = lds + 2 * C::BR * C::PB0, *t1lo = t1hi + C::BR * C::PB1; // first product [BR][PB1]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gl = lane % C::G, gr = lane / C::G;
  const int mi = lane & 15, mh = lane >> 4;

  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, r8 = nb & 7;
  const int lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int b0 = lb * C::BR;

  const char *xb = reinterpret_cast<const char *>(x + (int64_t)blockIdx.y * mstride_x);
  float *om = out + (int64_t)blockIdx.y * mstride_o;
  const uint32_t lane_off = gl * 16;

  // weights of both contractions for this wave's column tiles, issued before the gathers
  const int j1 = wave % C::NJ1;
  Frag<FIN> b1;
  b1.load(W1, j1, mi, mh);
  const int j2 = wave % (F2 > 0 ? C::NJ2 : 1);
  Frag<(F2 > 0 ? F1 : 16)> b2;
  if constexpr (F2 > 0) b2.load(W2, j2, mi, mh);
  float4_t bpre = {0.f, 0.f, 0.f, 0.f}, bpost = {0.f, 0.f, 0.f, 0.f};
  if constexpr (PRE) { if (bias) bpre = *reinterpret_cast<const float4_t *>(bias + gl * 4); }
  else               { if (bias) bpost = *reinterpret_cast<const float4_t *>(bias + j1 * 16 + 4 * mh); }

  // ---- phase 1: gather + aggregate (+ bias, ReLU when activation-first) -> LDS hi/lo -------------
  gwen::gather_passes<FIN, C::NP, C::RB, UNI>(
      rowptr, col, val, xb, N, b0, wave, gr, lane_off, [&](int lr, float4_t acc) {
        if constexpr (PRE) {
          acc = acc + bpre;
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] < 0.0f ? 0.0f : acc[e];
          }
        }
        const float a4[4] = {acc[0], acc[1], acc[2], acc[3]};
        bf16x4 h4, l4;
        split_bf16<4>(a4, h4, l4);
        *reinterpret_cast<bf16x4 *>(t0hi + lr * C::PB0 + gl * 4) = h4;
        *reinterpret_cast<bf16x4 *>(t0lo + lr * C::PB0 + gl * 4) = l4;
      });
  __syncthreads();

  // ---- phase 2: first contraction; result to global (F2 == 0) or to the second LDS image ----------
#pragma unroll
  for (int tt = wave / C::NJ1; tt < C::NT; tt += C::NWB / C::NJ1) {
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = b1.mma(t0hi, t0lo, (tt * kTile + mi) * C::PB0, mh, d);
    const int lr = tt * kTile + mi;
    float4_t o = {d[0], d[1], d[2], d[3]};
    if constexpr (!PRE) {
      o = o + bpost;
      if (relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = o[t] < 0.0f ? 0.0f : o[t];
      }
    }
    if constexpr (F2 > 0) {
      const float o4[4] = {o[0], o[1], o[2], o[3]};
      bf16x4 h4, l4;
      split_bf16<4>(o4, h4, l4);
      *reinterpret_cast<bf16x4 *>(t1hi + lr * C::PB1 + j1 * 16 + 4 * mh) = h4;
      *reinterpret_cast<bf16x4 *>(t1lo + lr * C::PB1 + j1 * 16 + 4 * mh) = l4;
    } else {
      if (b0 + lr < N)
        *reinterpret_cast<float4_t *>(om + (int64_t)(b0 + lr) * F1 + j1 * 16 + 4 * mh) = o;
    }
  }
  if constexpr (F2 > 0) {
    __syncthreads();
    // ---- phase 3: second contraction (the next layer's `lin`), stored at width F2 ---------------
#pragma unroll
    for (int tt = wave / C::NJ2; tt < C::NT; tt += C::NWB / C::NJ2) {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      d = b2.mma(t1hi, t1lo, (tt * kTile + mi) * C::PB1, mh, d);
      const int r = b0 + tt * kTile + mi;
      if (r < N)
        *reinterpret_cast<float4_t *>(om + (int64_t)r * F2 + j2 * 16 + 4 * mh) =
            float4_t{d[0], d[1], d[2], d[3]};
    }
  }
}

// Activation-first layer with nothing chained: out = act(A~ h + bias) on the grouped layout (the
// fma form of K2; K2 itself keeps the rounded-product order that is bit-identical to the CPU path).
template <int FIN, bool UNI>
__global__ __launch_bounds__(256) void k_gather(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, const float *__restrict__ bias,
    float *__restrict__ out, int32_t N, int64_t mstride_x, int64_t mstride_o, int relu) {
  constexpr int G = FIN / 4, R = 64 / G, BR = 4 * R;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gl = lane % G, gr = lane / G;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, r8 = nb & 7;
  const int lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const char *xb = reinterpret_cast<const char *>(x + (int64_t)blockIdx.y * mstride_x);
  float *om = out + (int64_t)blockIdx.y * mstride_o;
  const uint32_t lane_off = gl * 16;
  gwen::gather_passes<FIN, 1, BR, UNI>(
      rowptr, col, val, xb, N, lb * BR, wave, gr, lane_off, [&](int lr, float4_t acc) {
        if (bias) acc = acc + *reinterpret_cast<const float4_t *>(bias + gl * 4);
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = acc[e] < 0.0f ? 0.0f : acc[e];
        }
        if (lb * BR + lr < N)
          *reinterpret_cast<float4_t *>(om + (int64_t)(lb * BR + lr) * FIN + gl * 4) = acc;
      });
}

template <int FIN>
int launch_gather(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
                  const float *bias, float *out, int64_t N, int64_t members, int64_t msx,
                  int64_t mso, int relu, hipStream_t st) {
  constexpr int BR = 4 * (64 / (FIN / 4));
  dim3 grid((unsigned)((N + BR - 1) / BR), (unsigned)members);
  if (!rowptr)
    k_gather<FIN, true><<<grid, 256, 0, st>>>(rowptr, col, val, x, bias, out, (int32_t)N, msx, mso, relu);
  else
    k_gather<FIN, false><<<grid, 256, 0, st>>>(rowptr, col, val, x, bias, out, (int32_t)N, msx, mso, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

template <int FIN, int F1, int F2, bool PRE>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
           const float *W1, const float *W2, const float *bias, float *out, int64_t N,
           int64_t members, int64_t msx, int64_t mso, int relu, hipStream_t st) {
  using C = Cfg<FIN, F1, F2, PRE>;
  const int64_t blocks = (N + C::BR - 1) / C::BR;
  dim3 grid((unsigned)blocks, (unsigned)members);
  if (!rowptr)
    k_chain<FIN, F1, F2, PRE, true><<<grid, C::NWB * 64, 0, st>>>(rowptr, col, val, x, W1, W2, bias, out,
                                                                  (int32_t)N, msx, mso, relu);
  else
    k_chain<FIN, F1, F2, PRE, false><<<grid, C::NWB * 64, 0, st>>>(rowptr, col, val, x, W1, W2, bias,
                                                                   out, (int32_t)N, msx, mso, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

constexpr bool width_ok(int64_t f) { return f == 16 || f == 32 || f == 64 || f == 128; }

}  // namespace

// LDS of one block (bytes), as Cfg computes it; the chained form keeps two hi/lo images
inline int64_t chain_lds_bytes(int64_t Fin, int64_t F1, int64_t F2) {
  const int64_t R = 64 / (Fin / 4), nwb = F1 / 16 > 4 ? 8 : 4, rb = nwb * R;
  const int64_t brmin = Fin >= 128 ? 128 : 64, br = rb > brmin ? rb : brmin;
  return 2 * br * pitch_bf16((int)Fin) * 2 + (F2 > 0 ? 2 * br * pitch_bf16((int)F1) * 2 : 0);
}

extern "C" int gwen_gcn_chain_supported(int64_t Fin, int64_t F1, int64_t F2, int pre) {
  if (pre && F1 == 0 && F2 == 0) return width_ok(Fin) ? 1 : 0;      // activation-first, nothing chained
  if (!width_ok(Fin) || !width_ok(F1)) return 0;
  if (pre) return F2 == 0 ? 1 : 0;
  if (!(width_ok(F2) && F2 < F1)) return 0;          // chained projection of a SHRINKING next layer
  // worth it only while two blocks still fit a CU's 160 KiB of LDS (otherwise K4, then K3 + K2)
  return chain_lds_bytes(Fin, F1, F2) <= 80 * 1024 ? 1 : 0;
}

extern "C" int gwen_gcn_chain_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                  const float *x, const float *W1, const float *W2,
                                  const float *bias, float *out, int64_t N, int64_t Fin, int64_t F1,
                                  int64_t F2, int pre, int relu, int64_t members, int64_t mstride_x,
                                  int64_t mstride_o, gwen_stream_t stream_) {
  if (N < 0 || members < 0) return GWEN_EINVAL;
  if (!gwen_gcn_chain_supported(Fin, F1, F2, pre)) return GWEN_EINVAL;
  if (N == 0 || members == 0) return GWEN_OK;
  if (!col || !val || !x || (F1 > 0 && !W1) || !out || x == out || (F2 > 0 && !W2))
    return GWEN_EINVAL;                                    // rowptr NULL = uniform layout
  if (N >= (int64_t(1) << 28) || members > 65535) return GWEN_ERANGE;
  if (!gwen_aligned(x, 16) || !gwen_aligned(out, 16) || (W1 && !gwen_aligned(W1, 16)) ||
      (W2 && !gwen_aligned(W2, 16)) || (bias && !gwen_aligned(bias, 16)) || mstride_x % 4)
    return GWEN_EINVAL;
  if (N * Fin * 4 >= (int64_t(1) << 32)) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  if (pre && F1 == 0) {
#define GWEN_G(FI)                                                                                  \
  if (Fin == FI)                                                                                    \
    return launch_gather<FI>(rowptr, col, val, x, bias, out, N, members, mstride_x, mstride_o, relu, st)
    GWEN_G(16); GWEN_G(32); GWEN_G(64); GWEN_G(128);
#undef GWEN_G
  }
#define GWEN_P(FI, FA)                                                                              \
  if (pre && Fin == FI && F1 == FA)                                                                 \
    return launch<FI, FA, 0, true>(rowptr, col, val, x, W1, W2, bias, out, N, members, mstride_x,   \
                                   mstride_o, relu, st)
#define GWEN_C(FI, FA, FB)                                                                          \
  if (!pre && Fin == FI && F1 == FA && F2 == FB)                                                    \
    return launch<FI, FA, FB, false>(rowptr, col, val, x, W1, W2, bias, out, N, members, mstride_x, \
                                     mstride_o, relu, st)
#define GWEN_ROW(FI)                                                                                \
  GWEN_P(FI, 16); GWEN_P(FI, 32); GWEN_P(FI, 64); GWEN_P(FI, 128);                                  \
  GWEN_C(FI, 32, 16); GWEN_C(FI, 64, 16); GWEN_C(FI, 64, 32);                                       \
  GWEN_C(FI, 128, 16); GWEN_C(FI, 128, 32); GWEN_C(FI, 128, 64)
  GWEN_ROW(16); GWEN_ROW(32); GWEN_ROW(64); GWEN_ROW(128);
#undef GWEN_ROW
#undef GWEN_C
#undef GWEN_P
  return GWEN_EINVAL;
}
