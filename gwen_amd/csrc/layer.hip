// K4 -- one whole GCNConv layer (+ReLU) in a single launch, aggregate-first:
//            out = act( (A~ x) W^T + bias )
// Replaces, per layer of the reference, the sequence  lin (SGEMM) -> index_select -> mul ->
// scatter_add_ -> + bias -> relu  (torch-geometric 2.3.1 GCNConv.forward as called from
// /root/reference/src/gwen/models_gnn.py:147-149,:204-206) by ONE kernel: the [N,Fout] intermediate
// `h`, the [E',F] message tensor and four elementwise passes never touch HBM.
//
// Shape (wave64): many SHORT blocks -- K2's shape -- so that the waves sharing a SIMD are in
// different phases (gather / MFMA / store) and the per-wave critical path is one or two gathers:
//   block = NWB waves (one per 16-column output tile, at least 4) = BR consecutive destination rows;
//   phase 1  every wave gathers R = 64/(Fin/4) rows per pass (Fin/4 lanes per row, 16-B loads, the 8
//            entries of a group in flight together) from the GROUPED layout (rows padded to whole
//            groups of 8 with weight-0 entries, null group for absent rows), so every load in the
//            loop is unconditional -- a load under a per-lane condition makes hipcc branch around it
//            and wait vmcnt(0), which serialises the gathers.  The aggregated rows go to the block's
//            LDS tile.  Meanwhile each wave fetches ITS slice of W^T -- the B fragments of its 16
//            output columns -- straight from global memory (16 KB, L1/L2-resident) into registers;
//   barrier;
//   phase 2  wave w owns output-column tile j = w % NJ of every (NWB/NJ)-th 16-row tile: A fragments
//            from LDS, MFMAs with W as the A operand (so the D tile comes out transposed: one lane =
//            4 consecutive output columns of one row), + bias, ReLU, one 16-B store per lane.
// Contraction (default): "3xbf16" -- x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and
//   x.w ~= lo.hi' + hi.lo' + hi.hi' on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the dropped
//   lo.lo' term and the representation residual are < 2^-16 relative each (measured 8e-6 relative on
//   the 6-layer model; tolerance 1e-4).  The split is done once per element when the aggregated row
//   is written to LDS (hi and lo tiles, row pitch Fin/2+8 dwords => conflict-free 16-B reads).
//   It exists because the exact fp32 MFMA (1/16 of the bf16 rate) cost 7 of a 64->64 layer's 28 us.
// Contraction (exact = 2): "bf16x6" -- three images per operand, six MFMAs per k-step: 24 bits per operand, the
//   fp32-class default of the host API (split.h); one more LDS image of the tile.
// Contraction (exact = 1): v_mfma_f32_16x16x4_f32 on an fp32 tile (k-permutation k = 8q+2(lane>>4)+s,
//   row pitch Fin+4 floats => conflict-free 8-B reads): bit-exact fp32 fmaf chains.
// Blocks are remapped so that the blocks sharing an XCD (blockIdx % 8) own neighbouring rows.
#include "common.h"
#include "gather_rows.h"
#include "split.h"

namespace {

constexpr int kTile = 16;
using gwen::bf16x4;
using gwen::bf16x8;
template <int K> using BF = gwen::BFv<K>;

// NS: bf16 images per operand (2: bf16x3, 3: bf16x6, split.h); 0: the fp32-input MFMA
template <int FIN, int FOUT, int NS, int BRMIN = 32>
struct Cfg {
  static constexpr bool SPLIT = NS > 0;
  static constexpr int G = FIN / 4, R = 64 / G;            // lanes per gathered row, rows per wave pass
  static constexpr int NJ = FOUT / 16;                     // 16-column output tiles
  static constexpr int NWB = NJ > 8 ? 16 : (NJ > 4 ? 8 : 4);   // waves per block: one per column tile
  static constexpr int RB = NWB * R;                       // rows gathered per block pass
  static constexpr int BR = RB > BRMIN ? RB : BRMIN;       // rows per block
  static constexpr int NP = BR / RB;                       // gather passes per wave
  static constexpr int NT = BR / kTile;                    // 16-row tiles per block
  static constexpr int TSTEP = NWB / NJ;                   // row tiles are strided over the waves
  static constexpr int NQ = FIN / 8;                       // exact: k-steps of 8
  static constexpr int KF = FIN >= 32 ? 8 : 4;             // split: bf16 per fragment (K = 32 or 16)
  static constexpr int KS = FIN / (4 * KF);                // split: MFMA k-steps
  static constexpr int PF = FIN + 4;                       // exact: tile row pitch (floats)
  static constexpr int PB = ((FIN / 2) % 16 == 8 ? FIN / 2 : FIN / 2 + 8) * 2;   // split: pitch (bf16)
  static constexpr size_t lds_bytes = SPLIT ? (size_t)NS * BR * PB * 2 : (size_t)BR * PF * 4;
  static_assert(NWB % NJ == 0, "waves must tile the output columns");
  static_assert(BR % RB == 0 && BR % kTile == 0, "a block is whole gather passes and whole row tiles");
};

// BWD (the layer's backward, gwen_gcn_layer_bwd_f32): the aggregated rows are also stored (agg_out: the
// operand of grad_W) and the result is masked by mask > 0 (the ReLU of the layer below), so the launch
// returns the gradient the next backward launch starts from.
template <int FIN, int FOUT, int NS, int BRMIN, bool UNI = false, bool BWD = false>
__global__ __launch_bounds__((FOUT > 128 ? 1024 : (FOUT > 64 ? 512 : 256))) void k_layer(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, const float *__restrict__ W,
    const float *__restrict__ bias, float *__restrict__ out, int32_t N, int64_t ldo,
    int64_t mstride_x, int64_t mstride_o, int relu, float *__restrict__ agg_out = nullptr,
    const float *__restrict__ mask = nullptr, float *__restrict__ bsum_out = nullptr, int32_t chunks_per_member = 0) {
  using C = Cfg<FIN, FOUT, NS, BRMIN>;
  constexpr bool SPLIT = NS > 0;
  constexpr int NI = SPLIT ? NS : 1;
  __shared__ __attribute__((aligned(16))) char lds_raw[C::lds_bytes];
  __shared__ float bred[BWD && FIN * FOUT < 128 * 128 ? 16 * 16 : 1];    // BWD, narrow: the waves' column sums of a chunk
  float *tile = reinterpret_cast<float *>(lds_raw);                      // exact: [BR][PF] fp32
  __bf16 *timg = reinterpret_cast<__bf16 *>(lds_raw);                    // split: NS images [BR][PB]
  constexpr int kImg = C::BR * C::PB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gl = lane % C::G, gr = lane / C::G;
  const int mi = lane & 15, mh = lane >> 4;

  // XCD-aware block remap (bijective for any grid size)
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, r8 = nb & 7;
  const int lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  const char *xb = reinterpret_cast<const char *>(x + (int64_t)blockIdx.y * mstride_x);
  float *om = out + (int64_t)blockIdx.y * mstride_o;
  const uint32_t lane_off = gl * 16;                       // x rows are contiguous (ldx == Fin)

  // ---- this wave's B fragments (its 16 output columns of W^T), issued before the gathers ---------
  const int j = wave % C::NJ;
  const float *wrow = W + (int64_t)(j * 16 + mi) * FIN;
  float2_t bfr[SPLIT ? 1 : C::NQ];
  typename BF<C::KF>::T bw[SPLIT ? C::KS : 1][NI];                       // W images per k-step
  if constexpr (SPLIT) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      float wv[C::KF];
      const float *wp = wrow + C::KF * (4 * ks + mh);
#pragma unroll
      for (int i = 0; i < C::KF; i += 4) {
        const float4_t w4 = *reinterpret_cast<const float4_t *>(wp + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[i + e] = w4[e];
      }
      gwen::split_images<C::KF, NI>(wv, bw[ks]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < C::NQ; ++q)
      bfr[q] = *reinterpret_cast<const float2_t *>(wrow + 8 * q + 2 * mh);
  }
  float4_t bv4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv4 = *reinterpret_cast<const float4_t *>(bias + j * 16 + 4 * mh);

  // A block walks chunks lb, lb + grid, ... of BR rows: with one chunk per block (narrow layers) the
  // loop runs once; wide layers launch one resident set of blocks so that W -- fetched and split into
  // this wave's registers once, above -- serves many chunks (at 256 channels W is half as many bytes
  // as a 64-row chunk gathers)
  constexpr bool kPersist = FIN * FOUT >= 128 * 128;
  const int nchunks = kPersist ? (N + C::BR - 1) / C::BR : lb + 1;
  for (int chunk = lb; chunk < nchunks; chunk += nb) {
  const int b0 = chunk * C::BR;
  // ---- phase 1: gather + aggregate into the LDS tile (gather_rows.h) ------------------------------
  gwen::gather_passes<FIN, C::NP, C::RB, UNI>(
      rowptr, col, val, xb, N, b0, wave, gr, lane_off, [&](int lr, float4_t acc) {
        if constexpr (BWD) {
          if (agg_out && b0 + lr < N)
            *reinterpret_cast<float4_t *>(agg_out + (int64_t)blockIdx.y * mstride_x +
                                          (int64_t)(b0 + lr) * FIN + gl * 4) = acc;
        }
        if constexpr (SPLIT) {
          const float a4[4] = {acc[0], acc[1], acc[2], acc[3]};
          bf16x4 im[NI];
          gwen::split_images<4, NI>(a4, im);
#pragma unroll
          for (int s_ = 0; s_ < NI; ++s_)
            *reinterpret_cast<bf16x4 *>(timg + s_ * kImg + lr * C::PB + gl * 4) = im[s_];
        } else {
          *reinterpret_cast<float4_t *>(tile + lr * C::PF + gl * 4) = acc;
        }
      });
  __syncthreads();

  // ---- phase 2: (tile) x (this wave's 16 columns of W^T), bias, ReLU, store ----------------------
  float4_t cs = {0.f, 0.f, 0.f, 0.f};                  // BWD + bsum_out: this lane's column sums over the wave's row tiles
#pragma unroll
  for (int tt = wave / C::NJ; tt < C::NT; tt += C::TSTEP) {
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SPLIT) {
      const int arow = (tt * kTile + mi) * C::PB;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        using FT = typename BF<C::KF>::T;
        FT a[NI];
#pragma unroll
        for (int s_ = 0; s_ < NI; ++s_)
          a[s_] = *reinterpret_cast<const FT *>(timg + s_ * kImg + arow + C::KF * (4 * ks + mh));
        d = gwen::mma_split<C::KF, NI>(bw[ks], a, d);
      }
    } else {
      const float *ap = tile + (tt * kTile + mi) * C::PF + 2 * mh;
#pragma unroll
      for (int q = 0; q < C::NQ; ++q) {
        const float2_t a = *reinterpret_cast<const float2_t *>(ap + 8 * q);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[q][0], a[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[q][1], a[1], d, 0, 0, 0);
      }
    }
    // W is the A operand, so D is the TRANSPOSED tile: lane (mi, mh) holds destination row mi,
    // output columns 16 j + 4 mh .. +3 -- one 16-B store per lane, 16 rows x 64 B per instruction
    const int r = b0 + tt * kTile + mi;
    float4_t o = float4_t{d[0], d[1], d[2], d[3]} + bv4;
    if (relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t] = o[t] < 0.0f ? 0.0f : o[t];
    }
    if constexpr (BWD) {
      if (mask && r < N) {
        const float4_t y = *reinterpret_cast<const float4_t *>(mask + (int64_t)blockIdx.y * mstride_o +
                                                               (int64_t)r * ldo + j * 16 + 4 * mh);
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = y[t] > 0.0f ? o[t] : 0.0f;
      }
    }
    if (r < N) *reinterpret_cast<float4_t *>(om + (int64_t)r * ldo + j * 16 + 4 * mh) = o;
    if constexpr (BWD && !kPersist) {
      if (bsum_out && r < N) cs = cs + o;
    }
  }
  if constexpr (BWD && !kPersist) {
    // the chunk's column sums of the masked result (= grad_b of the layer below, whose incoming gradient this is): rows
    // of a row tile meet through the 16 lanes that share mh, a column tile's waves through LDS in wave order; one
    // partial row per (member, chunk), every chunk written exactly once: fixed order, no atomics.  Narrow layers only
    // (one chunk per block): in the persistent wide kernels the extra registers cost more than the reduction launch
    // they replace (256 -> 256: 206 -> 289 us)
    if (bsum_out) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1)
#pragma unroll
        for (int t = 0; t < 4; ++t) cs[t] = cs[t] + __shfl_xor(cs[t], m);
      if (mi == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bred[wave * 16 + 4 * mh + t] = cs[t];
      }
      __syncthreads();
      if ((int)threadIdx.x < FOUT) {
        const int cj = threadIdx.x >> 4, cc = threadIdx.x & 15;
        float v = 0.0f;
        for (int w = cj; w < C::NWB; w += C::NJ) v = v + bred[w * 16 + cc];
        bsum_out[((int64_t)blockIdx.y * chunks_per_member + chunk) * FOUT + threadIdx.x] = v;
      }
    }
  }
  if constexpr (kPersist) __syncthreads();     // the tile is free for the next chunk
  }
}

template <int FIN, int FOUT, int NS, int BRMIN>
int launch_rows(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
                const float *W, const float *bias, float *out, int64_t N, int64_t ldo, int64_t members,
                int64_t msx, int64_t mso, int relu, hipStream_t st, bool probe, int64_t *resident_out,
                float *agg_out = nullptr, const float *mask = nullptr, bool bwd = false, float *bsum_out = nullptr,
                int64_t *chunks_out = nullptr) {
  using C = Cfg<FIN, FOUT, NS, BRMIN>;
  static int per_cu = 0;
  if (per_cu == 0) {
    int nbk = 0;
    GWEN_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nbk, reinterpret_cast<const void *>(&k_layer<FIN, FOUT, NS, BRMIN, true>), C::NWB * 64, 0));
    per_cu = nbk < 1 ? 1 : nbk;
  }
  const int64_t resident = (int64_t)256 * per_cu;
  if (resident_out) *resident_out = resident;
  if (probe) return GWEN_OK;
  int64_t blocks = (N + C::BR - 1) / C::BR;
  const int64_t chunks_pm = blocks;                        // chunks of BR rows per member (bsum_out: one partial row each)
  constexpr bool kWide = FIN * FOUT >= 128 * 128;          // the persistent form: no fused column sums (see the kernel)
  if (kWide) bsum_out = nullptr;
  if (chunks_out) *chunks_out = kWide ? 0 : chunks_pm * members;
  if (FIN * FOUT >= 128 * 128 && blocks > resident) blocks = resident;   // wide layer: one resident set
  dim3 grid((unsigned)blocks, (unsigned)members);
  if constexpr (NS == 2 || NS == 3) {          // the backward runs on the layer's own split (bf16x3 / bf16x6)
    if (bwd) {
      if (!rowptr)
        k_layer<FIN, FOUT, NS, BRMIN, true, true><<<grid, C::NWB * 64, 0, st>>>(
            rowptr, col, val, x, W, bias, out, (int32_t)N, ldo, msx, mso, relu, agg_out, mask, bsum_out, (int32_t)chunks_pm);
      else
        k_layer<FIN, FOUT, NS, BRMIN, false, true><<<grid, C::NWB * 64, 0, st>>>(
            rowptr, col, val, x, W, bias, out, (int32_t)N, ldo, msx, mso, relu, agg_out, mask, bsum_out, (int32_t)chunks_pm);
      GWEN_LAUNCH_CHECK();
      return GWEN_OK;
    }
  }
  if (!rowptr)      // uniform layout: row r is the group at 8 r
    k_layer<FIN, FOUT, NS, BRMIN, true><<<grid, C::NWB * 64, 0, st>>>(
        rowptr, col, val, x, W, bias, out, (int32_t)N, ldo, msx, mso, relu);
  else
    k_layer<FIN, FOUT, NS, BRMIN, false><<<grid, C::NWB * 64, 0, st>>>(
        rowptr, col, val, x, W, bias, out, (int32_t)N, ldo, msx, mso, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

template <int FIN, int FOUT, int NS>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
           const float *W, const float *bias, float *out, int64_t N, int64_t ldo, int64_t members,
           int64_t msx, int64_t mso, int relu, hipStream_t st, float *agg_out = nullptr,
           const float *mask = nullptr, bool bwd = false, float *bsum_out = nullptr, int64_t *chunks_out = nullptr) {
#define GWEN_ROWS(BRV, PROBE, RES)                                                                  \
  launch_rows<FIN, FOUT, NS, BRV>(rowptr, col, val, x, W, bias, out, N, ldo, members, msx, mso,  \
                                     relu, st, PROBE, RES, agg_out, mask, bwd, bsum_out, chunks_out)
  if constexpr (FIN <= 64 && FOUT <= 64) {
    // Narrow layers run as ONE round of co-resident blocks when a block size makes that possible: with
    // 64-row blocks the c2 mesh needs 1 563 blocks against 1 024 resident ones (4 per CU at 64 -> 64),
    // i.e. a full round plus a half-empty one of ~10 us each; 112-row blocks (893 of them) fit one round.
    // Rows per block: the smallest of 64 / 96 / 112 / 128 (whole gather passes only) whose grid is
    // co-resident, else 64.
    int64_t res = 0;
    const int64_t work = N * members;
    { const int rc_ = GWEN_ROWS(64, true, &res); if (rc_ != GWEN_OK) return rc_; }
    if ((work + 63) / 64 <= res || members > 1) return GWEN_ROWS(64, false, nullptr);
    constexpr int RB = Cfg<FIN, FOUT, NS, 64>::RB;      // a block is whole gather passes of RB rows
    if constexpr (96 % RB == 0) {
      { const int rc_ = GWEN_ROWS(96, true, &res); if (rc_ != GWEN_OK) return rc_; }
      if ((N + 95) / 96 <= res) return GWEN_ROWS(96, false, nullptr);
    }
    if constexpr (112 % RB == 0) {
      { const int rc_ = GWEN_ROWS(112, true, &res); if (rc_ != GWEN_OK) return rc_; }
      if ((N + 111) / 112 <= res) return GWEN_ROWS(112, false, nullptr);
    }
    if constexpr (128 % RB == 0) {
      { const int rc_ = GWEN_ROWS(128, true, &res); if (rc_ != GWEN_OK) return rc_; }
      if ((N + 127) / 128 <= res) return GWEN_ROWS(128, false, nullptr);
    }
    return GWEN_ROWS(64, false, nullptr);
  } else {
    // rows per block: enough that W (read once per block) stays a small fraction of the gathered bytes
#ifndef K4_BR256X6
#define K4_BR256X6 64
#endif
    // 256: 64 rows keep two blocks per CU in LDS (bf16x3); K4_BR256X6: rows per block at 256 channels on bf16x6
    constexpr int BRMIN = FIN == 128 ? 128 : (FIN == 256 && NS == 3 ? K4_BR256X6 : 64);
    return GWEN_ROWS(BRMIN, false, nullptr);
  }
#undef GWEN_ROWS
}

constexpr bool width_ok(int64_t f) { return f == 16 || f == 32 || f == 64 || f == 128 || f == 256; }

}  // namespace

extern "C" int gwen_gcn_layer_supported(int64_t Fin, int64_t Fout) {
  return width_ok(Fin) && width_ok(Fout) ? 1 : 0;
}

extern "C" int gwen_gcn_layer_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                  const float *x, const float *W, const float *bias, float *out,
                                  int64_t N, int64_t Fin, int64_t Fout, int64_t ldx, int64_t ldo,
                                  int64_t members, int64_t mstride_x, int64_t mstride_o, int relu,
                                  int exact, gwen_stream_t stream_) {
  if (N < 0 || members < 0 || ldx < Fin || ldo < Fout || exact < 0 || exact > 2) return GWEN_EINVAL;
  if (!gwen_gcn_layer_supported(Fin, Fout)) return GWEN_EINVAL;
  if (N == 0 || members == 0) return GWEN_OK;
  if (!col || !val || !x || !W || !out || x == out) return GWEN_EINVAL;   // rowptr NULL = uniform
  if (N >= (int64_t(1) << 28) || members > 65535) return GWEN_ERANGE;     // 8 N must fit int32
  if (!gwen_aligned(x, 16) || !gwen_aligned(out, 16) || !gwen_aligned(W, 16) || ldx != Fin ||
      mstride_x % 4 || ldo % 4 || mstride_o % 4 || (bias && !gwen_aligned(bias, 16)))
    return GWEN_EINVAL;                       // x rows must be contiguous (32-bit row offsets)
  if (N * Fin * 4 >= (int64_t(1) << 32)) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
#define GWEN_L(FI, FO)                                                                           \
  if (Fin == FI && Fout == FO)                                                                   \
    return exact == GWEN_CONTRACT_F32                                                            \
               ? launch<FI, FO, 0>(rowptr, col, val, x, W, bias, out, N, ldo, members, mstride_x, \
                                   mstride_o, relu, st)                                          \
               : (exact == GWEN_CONTRACT_BF16X6                                                  \
                      ? launch<FI, FO, 3>(rowptr, col, val, x, W, bias, out, N, ldo, members,    \
                                          mstride_x, mstride_o, relu, st)                        \
                      : launch<FI, FO, 2>(rowptr, col, val, x, W, bias, out, N, ldo, members,    \
                                          mstride_x, mstride_o, relu, st))
  GWEN_L(16, 16); GWEN_L(16, 32); GWEN_L(16, 64); GWEN_L(16, 128);
  GWEN_L(32, 16); GWEN_L(32, 32); GWEN_L(32, 64); GWEN_L(32, 128);
  GWEN_L(64, 16); GWEN_L(64, 32); GWEN_L(64, 64); GWEN_L(64, 128);
  GWEN_L(128, 16); GWEN_L(128, 32); GWEN_L(128, 64); GWEN_L(128, 128);
  GWEN_L(16, 256); GWEN_L(32, 256); GWEN_L(64, 256); GWEN_L(128, 256);
  GWEN_L(256, 16); GWEN_L(256, 32); GWEN_L(256, 64); GWEN_L(256, 128); GWEN_L(256, 256);
#undef GWEN_L
  return GWEN_EINVAL;
}

// The layer's backward as ONE launch of the same kernel on the TRANSPOSED graph (grouped arrays of the
// transposed CSR):  gh = A~^T g  (stored: grad_W = gh^T x is a separate reduction),  gx = gh Wt^T  masked
// by mask > 0.  g [members, N, Fg]; Wt [Fx, Fg] = the layer's weight as stored ([out, in] = [Fg, Fx])
// TRANSPOSED; gh [members, N, Fg]; gx, mask [members, N, Fx] (mask NULL = no ReLU below).
// ... and, with bias_partial, the column sums of gx per (member, chunk of rows) as well: gx is the incoming gradient of
// the layer BELOW, so these are stage 1 of that layer's grad_b -- *bias_chunks partial rows of Fx floats (at most
// gwen_gcn_layer_bwd_bias_rows(N, members)), finished by gwen_reduce_chunks_batched.
extern "C" int64_t gwen_gcn_layer_bwd_bias_rows(int64_t N, int64_t members) { return members * ((N + 31) / 32); }

extern "C" int gwen_gcn_layer_bwd_bias_f32(const int32_t *t_rowptr, const int32_t *t_col, const float *t_val,
                                           const float *g, const float *Wt, const float *mask, float *gh,
                                           float *gx, int64_t N, int64_t Fg, int64_t Fx, int64_t members,
                                           int contract, float *bias_partial, int64_t *bias_chunks,
                                           gwen_stream_t stream_) {
  if (bias_chunks) *bias_chunks = 0;
  if (N < 0 || members < 0 || (contract != GWEN_CONTRACT_BF16X3 && contract != GWEN_CONTRACT_BF16X6)) return GWEN_EINVAL;
  if (!gwen_gcn_layer_supported(Fg, Fx)) return GWEN_EINVAL;
  if (N == 0 || members == 0) return GWEN_OK;
  if (!t_col || !t_val || !g || !Wt || !gx || g == gx) return GWEN_EINVAL;
  if (N >= (int64_t(1) << 28) || members > 65535) return GWEN_ERANGE;
  if (!gwen_aligned(g, 16) || !gwen_aligned(gx, 16) || !gwen_aligned(Wt, 16) ||
      (gh && !gwen_aligned(gh, 16)) || (mask && !gwen_aligned(mask, 16)))
    return GWEN_EINVAL;
  if (N * Fg * 4 >= (int64_t(1) << 32)) return GWEN_ERANGE;
  if ((bias_partial != nullptr) != (bias_chunks != nullptr) || (bias_partial && !gwen_aligned(bias_partial, 16))) return GWEN_EINVAL;
  hipStream_t st = gwen_stream(stream_);
#define GWEN_L(FI, FO)                                                                           \
  if (Fg == FI && Fx == FO)                                                                      \
    return contract == GWEN_CONTRACT_BF16X6                                                      \
               ? launch<FI, FO, 3>(t_rowptr, t_col, t_val, g, Wt, nullptr, gx, N, Fx, members,  \
                                   N * Fg, N * Fx, 0, st, gh, mask, true, bias_partial, bias_chunks) \
               : launch<FI, FO, 2>(t_rowptr, t_col, t_val, g, Wt, nullptr, gx, N, Fx, members,  \
                                   N * Fg, N * Fx, 0, st, gh, mask, true, bias_partial, bias_chunks)
  GWEN_L(16, 16); GWEN_L(16, 32); GWEN_L(16, 64); GWEN_L(16, 128);
  GWEN_L(32, 16); GWEN_L(32, 32); GWEN_L(32, 64); GWEN_L(32, 128);
  GWEN_L(64, 16); GWEN_L(64, 32); GWEN_L(64, 64); GWEN_L(64, 128);
  GWEN_L(128, 16); GWEN_L(128, 32); GWEN_L(128, 64); GWEN_L(128, 128);
  GWEN_L(16, 256); GWEN_L(32, 256); GWEN_L(64, 256); GWEN_L(128, 256);
  GWEN_L(256, 16); GWEN_L(256, 32); GWEN_L(256, 64); GWEN_L(256, 128); GWEN_L(256, 256);
#undef GWEN_L
  return GWEN_EINVAL;
}

extern "C" int gwen_gcn_layer_bwd_f32(const int32_t *t_rowptr, const int32_t *t_col, const float *t_val,
                                      const float *g, const float *Wt, const float *mask, float *gh,
                                      float *gx, int64_t N, int64_t Fg, int64_t Fx, int64_t members,
                                      int contract, gwen_stream_t stream_) {
  return gwen_gcn_layer_bwd_bias_f32(t_rowptr, t_col, t_val, g, Wt, mask, gh, gx, N, Fg, Fx, members, contract, nullptr,
                                     nullptr, stream_);
}
