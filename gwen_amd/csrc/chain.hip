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
// Structure, layout, contraction (bf16 split with NS images per operand, fp32 accumulate: split.h) and numerics
// are K4's (layer.hip).
#include "common.h"
#include "gather_rows.h"
#include "split.h"

namespace {

constexpr int kTile = 16;
using gwen::bf16x4;
using gwen::bf16x8;
template <int K> using BF = gwen::BFv<K>;

constexpr int pitch_bf16(int f) { return ((f / 2) % 16 == 8 ? f / 2 : f / 2 + 8) * 2; }

// B fragments (NS images) of output-column tile j of a [FO, FI] weight: W[16 j + mi][KF (4 ks + mh) .. +KF)
template <int FI, int NS>
struct Frag {
  static constexpr int KF = FI >= 32 ? 8 : 4;
  static constexpr int KS = FI / (4 * KF);
  using T = typename BF<KF>::T;
  T im[KS][NS];
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
      gwen::split_images<KF, NS>(wv, im[ks]);
    }
  }
  // d += W-fragment (A operand) x tile rows arow.. from the NS LDS images, `img` elements apart (B operand): the
  // product comes out TRANSPOSED -- lane (mi, mh) holds row mi, columns 16 j + 4 mh .. +3
  __device__ inline f32x4 mma(const __bf16 *t, int img, int arow, int mh, f32x4 d) const {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      T a[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const T *>(t + s * img + arow + KF * (4 * ks + mh));
      d = gwen::mma_split<KF, NS>(im[ks], a, d);
    }
    return d;
  }
};

// FIN: gathered width.  F1: width after the first contraction.  F2: width after the second (0: none).
template <int FIN, int F1, int F2, bool PRE, int NS>
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
  // bf16x6 (three images): the first product's images take the place of the aggregated rows' (one more barrier,
  // the product waits in registers meanwhile) -- with both sets resident 64 -> 64 -> 32 kept two blocks per CU
  // instead of three and ran 30.5 us against 22.9 for bf16x3
  static constexpr bool ALIAS = NS == 3 && F2 > 0;
  static constexpr size_t img0 = (size_t)NS * BR * PB0, img1 = F2 > 0 ? (size_t)NS * BR * PB1 : 0;
  static constexpr size_t lds_elems = ALIAS ? (img0 > img1 ? img0 : img1) : img0 + img1;
  static_assert(!(PRE && F2 > 0), "activation-first has one contraction");
  static_assert(NWB % NJ1 == 0 && (F2 == 0 || NWB % NJ2 == 0), "waves must tile the columns");
};

template <int FIN, int F1, int F2, bool PRE, bool UNI, int NS>
__global__ __launch_bounds__((F1 > 64 ? 512 : 256)) void k_chain(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, const float *__restrict__ W1,
    const float *__restrict__ W2, const float *__restrict__ bias, float *__restrict__ out, int32_t N,
    int64_t mstride_x, int64_t mstride_o, int relu) {
  using C = Cfg<FIN, F1, F2, PRE, NS>;
  __shared__ __attribute__((aligned(16))) __bf16 lds[C::lds_elems];
  constexpr int kImg0 = C::BR * C::PB0, kImg1 = C::BR * C::PB1;
  __bf16 *t0 = lds;                                                      // aggregated rows: NS x [BR][PB0]
  __bf16 *t1 = C::ALIAS ? lds : lds + NS * kImg0;                        // first product:   NS x [BR][PB1]
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
  Frag<FIN, NS> b1;
  b1.load(W1, j1, mi, mh);
  const int j2 = wave % (F2 > 0 ? C::NJ2 : 1);
  Frag<(F2 > 0 ? F1 : 16), NS> b2;
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
        bf16x4 im[NS];
        gwen::split_images<4, NS>(a4, im);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4 *>(t0 + s * kImg0 + lr * C::PB0 + gl * 4) = im[s];
      });
  __syncthreads();

  // ---- phase 2: first contraction; result to global (F2 == 0) or to the second LDS image ----------
  constexpr int NIT = C::NT / (C::NWB / C::NJ1);                          // row tiles of this wave
  float4_t keep[C::ALIAS ? NIT : 1];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int tt = wave / C::NJ1 + it * (C::NWB / C::NJ1);
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = b1.mma(t0, kImg0, (tt * kTile + mi) * C::PB0, mh, d);
    const int lr = tt * kTile + mi;
    float4_t o = {d[0], d[1], d[2], d[3]};
    if constexpr (!PRE) {
      o = o + bpost;
      if (relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = o[t] < 0.0f ? 0.0f : o[t];
      }
    }
    if constexpr (C::ALIAS) {
      keep[it] = o;
    } else if constexpr (F2 > 0) {
      const float o4[4] = {o[0], o[1], o[2], o[3]};
      bf16x4 im[NS];
      gwen::split_images<4, NS>(o4, im);
#pragma unroll
      for (int s = 0; s < NS; ++s)
        *reinterpret_cast<bf16x4 *>(t1 + s * kImg1 + lr * C::PB1 + j1 * 16 + 4 * mh) = im[s];
    } else {
      if (b0 + lr < N)
        *reinterpret_cast<float4_t *>(om + (int64_t)(b0 + lr) * F1 + j1 * 16 + 4 * mh) = o;
    }
  }
  if constexpr (C::ALIAS) {
    __syncthreads();                                   // every wave is done reading the aggregated rows
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int lr = (wave / C::NJ1 + it * (C::NWB / C::NJ1)) * kTile + mi;
      const float o4[4] = {keep[it][0], keep[it][1], keep[it][2], keep[it][3]};
      bf16x4 im[NS];
      gwen::split_images<4, NS>(o4, im);
#pragma unroll
      for (int s = 0; s < NS; ++s)
        *reinterpret_cast<bf16x4 *>(t1 + s * kImg1 + lr * C::PB1 + j1 * 16 + 4 * mh) = im[s];
    }
  }
  if constexpr (F2 > 0) {
    __syncthreads();
    // ---- phase 3: second contraction (the next layer's `lin`), stored at width F2 ---------------
#pragma unroll
    for (int tt = wave / C::NJ2; tt < C::NT; tt += C::NWB / C::NJ2) {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      d = b2.mma(t1, kImg1, (tt * kTile + mi) * C::PB1, mh, d);
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

template <int FIN, int F1, int F2, bool PRE, int NS>
int launch_ns(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
              const float *W1, const float *W2, const float *bias, float *out, int64_t N,
              int64_t members, int64_t msx, int64_t mso, int relu, hipStream_t st) {
  using C = Cfg<FIN, F1, F2, PRE, NS>;
  const int64_t blocks = (N + C::BR - 1) / C::BR;
  dim3 grid((unsigned)blocks, (unsigned)members);
  if (!rowptr)
    k_chain<FIN, F1, F2, PRE, true, NS><<<grid, C::NWB * 64, 0, st>>>(rowptr, col, val, x, W1, W2, bias, out,
                                                                      (int32_t)N, msx, mso, relu);
  else
    k_chain<FIN, F1, F2, PRE, false, NS><<<grid, C::NWB * 64, 0, st>>>(rowptr, col, val, x, W1, W2, bias,
                                                                       out, (int32_t)N, msx, mso, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

template <int FIN, int F1, int F2, bool PRE>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
           const float *W1, const float *W2, const float *bias, float *out, int64_t N,
           int64_t members, int64_t msx, int64_t mso, int relu, int contract, hipStream_t st) {
  if (contract == GWEN_CONTRACT_BF16X6) {
    // width triples whose three images do not fit a CU's LDS are refused by gwen_gcn_chain_supported already
    if constexpr (Cfg<FIN, F1, F2, PRE, 3>::lds_elems * 2 <= 160 * 1024)
      return launch_ns<FIN, F1, F2, PRE, 3>(rowptr, col, val, x, W1, W2, bias, out, N, members, msx, mso, relu, st);
    else
      return GWEN_EINVAL;
  }
  return launch_ns<FIN, F1, F2, PRE, 2>(rowptr, col, val, x, W1, W2, bias, out, N, members, msx, mso, relu, st);
}

constexpr bool width_ok(int64_t f) { return f == 16 || f == 32 || f == 64 || f == 128; }

}  // namespace

// LDS of one block (bytes), as Cfg computes it; the chained form keeps two sets of NS images
inline int64_t chain_lds_bytes(int64_t Fin, int64_t F1, int64_t F2, int ns) {
  const int64_t R = 64 / (Fin / 4), nwb = F1 / 16 > 4 ? 8 : 4, rb = nwb * R;
  const int64_t brmin = Fin >= 128 ? 128 : 64, br = rb > brmin ? rb : brmin;
  const int64_t i0 = ns * br * pitch_bf16((int)Fin) * 2, i1 = F2 > 0 ? ns * br * pitch_bf16((int)F1) * 2 : 0;
  return ns == 3 && F2 > 0 ? (i0 > i1 ? i0 : i1) : i0 + i1;        // bf16x6: the second set replaces the first
}

extern "C" int gwen_gcn_chain_supported(int64_t Fin, int64_t F1, int64_t F2, int pre, int contract) {
  if (contract != GWEN_CONTRACT_BF16X3 && contract != GWEN_CONTRACT_BF16X6) return 0;
  const int ns = gwen::images_of(contract);
  if (pre && F1 == 0 && F2 == 0) return width_ok(Fin) ? 1 : 0;      // activation-first, nothing chained
  if (!width_ok(Fin) || !width_ok(F1)) return 0;
  if (pre) return F2 == 0 && chain_lds_bytes(Fin, F1, 0, ns) <= 150 * 1024 ? 1 : 0;   // one block must fit the CU
  if (!(width_ok(F2) && F2 < F1)) return 0;          // chained projection of a SHRINKING next layer
  // worth it only while two blocks still fit a CU's 160 KiB of LDS (otherwise K4, then K3 + K2)
  return chain_lds_bytes(Fin, F1, F2, ns) <= 80 * 1024 ? 1 : 0;
}

extern "C" int gwen_gcn_chain_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                  const float *x, const float *W1, const float *W2,
                                  const float *bias, float *out, int64_t N, int64_t Fin, int64_t F1,
                                  int64_t F2, int pre, int relu, int64_t members, int64_t mstride_x,
                                  int64_t mstride_o, int contract, gwen_stream_t stream_) {
  if (N < 0 || members < 0) return GWEN_EINVAL;
  if (!gwen_gcn_chain_supported(Fin, F1, F2, pre, contract)) return GWEN_EINVAL;
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
                                   mstride_o, relu, contract, st)
#define GWEN_C(FI, FA, FB)                                                                          \
  if (!pre && Fin == FI && F1 == FA && F2 == FB)                                                    \
    return launch<FI, FA, FB, false>(rowptr, col, val, x, W1, W2, bias, out, N, members, mstride_x, \
                                     mstride_o, relu, contract, st)
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
