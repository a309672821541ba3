// K4 -- one whole GCNConv layer (+ReLU) in a single launch, aggregate-first:
//            out = act( (A~ x) W^T + bias )
// Replaces, per layer of the reference, the sequence  lin (SGEMM) -> index_select -> mul ->
// scatter_add_ -> + bias -> relu  (torch-geometric 2.3.1 GCNConv.forward as called from
// /root/reference/src/gwen/models_gnn.py:147-149,:204-206) by ONE kernel: the [N,Fout] intermediate
// `h`, the [E',F] message tensor and four elementwise passes never touch HBM.
//
// Work split (wave64, 256-thread blocks = 4 waves, all waves independent after W is staged):
//   * every wave owns a CONTIGUOUS range of destination rows (N split evenly over all waves of the
//     grid -> no tail quantisation) and walks it in tiles of 16 rows, R = 64/(FIN/4) rows per pass;
//   * gather: FIN/4 lanes per row (16-B loads); each lane-group walks its row's CSR segment 8
//     neighbours per batch (8 independent row gathers in flight), adds the terms in stored order
//     (rounded product, then add -- same order/rounding as K2) and parks the aggregated row in the
//     wave's private LDS tile agg[16][FIN+4].  The three dependent loads of a CSR row
//     (rowptr -> col/val -> x row) are software-pipelined across passes;
//   * contraction: v_mfma_f32_16x16x4_f32 with A = the finished tile's fragments (taken out of LDS
//     into registers once, so the tile buffer is free for the next gather) and B = W^T from the
//     block's LDS copy W[FOUT][FIN+4]; operands use the k-permutation k = 8q + 2*(lane>>4) + s so
//     each is one 8-B LDS read.  A tile's MFMAs are issued in slices BETWEEN the next tile's gather
//     issue and its wait, so the matrix pipe works while the wave's own loads are in flight;
//   * epilogue: + bias, ReLU, D through a 4-row LDS buffer, whole-row 16-B coalesced stores.
// Blocks are remapped so that the blocks sharing an XCD (blockIdx % 8) own neighbouring row ranges:
// gathered rows are then re-used inside one 4 MiB L2 instead of being fetched by all eight.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int kTile = 16;
constexpr int kBatch = 8;
typedef int int4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

__device__ inline void wave_lds_fence() {
  // LDS ops of one wave execute in order; this only stops the compiler from moving them.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int FIN, int FOUT, int NW>
struct Cfg {
  static constexpr int kWaves = NW;
  static constexpr int kThreads = NW * 64;
  static constexpr int G = FIN / 4;                 // lanes per gathered row
  static constexpr int R = 64 / G;                  // rows gathered per pass by one wave
  static constexpr int P = kTile / R;               // passes per 16-row tile
  static constexpr int NQ = FIN / 8;                // k-steps of 8 in the contraction
  static constexpr int QP = NQ / P;                 // k-steps issued per pass (= 2)
  static constexpr int SW = FIN + 4;                // row pitch of W in LDS (floats)
  static constexpr int ST = FIN + 4;                // row pitch of the aggregated tile
  static constexpr int SE = FOUT + 4;               // row pitch of the 4-row store buffer
  static constexpr int NJ = FOUT / 16;              // 16-column output tiles
  static constexpr int GO = FOUT / 4;               // lanes per stored output row
  static constexpr int RO = 64 / GO;                // output rows covered by one store instruction
  static constexpr int WAVE_LDS = kTile * ST + 4 * SE;
  static constexpr size_t lds_bytes = sizeof(float) * (size_t)(FOUT * SW + kWaves * WAVE_LDS);
  static_assert(QP * P == NQ && QP >= 1, "k-steps must split evenly over the passes");
};

template <int FIN, int FOUT, int NW, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void k_layer(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, const float *__restrict__ W,
    const float *__restrict__ bias, float *__restrict__ out, int32_t N, int64_t ldx, int64_t ldo,
    int64_t mstride_x, int64_t mstride_o, int relu) {
  using C = Cfg<FIN, FOUT, NW>;
  constexpr int kThreads = C::kThreads, kWaves = C::kWaves;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Wl = lds;                                                  // [FOUT][SW], whole block
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *tile = lds + FOUT * C::SW + wave * C::WAVE_LDS;            // [16][ST], this wave only
  float *ebuf = tile + kTile * C::ST;                               // [4][SE],  this wave only

  // stage W once per block (coalesced 16-B loads, 16-B LDS stores)
  for (int idx = threadIdx.x; idx < FOUT * (FIN / 4); idx += kThreads) {
    const int r = idx / (FIN / 4), c4 = (idx % (FIN / 4)) * 4;
    *reinterpret_cast<float4_t *>(Wl + r * C::SW + c4) =
        *reinterpret_cast<const float4_t *>(W + (int64_t)r * FIN + c4);
  }
  __syncthreads();

  // XCD-aware block remap (bijective for any grid size): blocks with equal blockIdx % 8 share an XCD
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nb >> 3, r8 = nb & 7;
  const int lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  // contiguous row range of this wave: N split evenly over all waves of the grid (the host launches
  // a whole number of blocks per CU, all resident at once, so every SIMD carries the same load)
  const int nw = nb * kWaves, gw = lb * kWaves + wave;
  const int base = N / nw, extra = N % nw;
  const int r0 = gw * base + (gw < extra ? gw : extra);
  const int r1 = r0 + base + (gw < extra ? 1 : 0);

  const float *xm = x + (int64_t)blockIdx.y * mstride_x;
  float *om = out + (int64_t)blockIdx.y * mstride_o;
  const int gl = lane % C::G, gr = lane / C::G;           // gather: lane within row group, row slot
  const int mi = lane & 15, mh = lane >> 4;               // MFMA: row/col index, k group
  const int ol = lane % C::GO, orow = lane / C::GO;       // store: lane within row, row slot
  const char *xb = reinterpret_cast<const char *>(xm);   // wave-uniform base of this member's rows
  constexpr uint32_t kRowBytes = FIN * 4;                // x rows are contiguous (ldx == FIN)
  const uint32_t lane_off = gl * 16;
  const float *ap = tile + mi * C::ST + 2 * mh;
  const float *bp = Wl + mi * C::SW + 2 * mh;

  const int nrows = r1 - r0;
  const int ntiles = (nrows + kTile - 1) / kTile;
  const int npass = ntiles * C::P;
  // rowptr/col/val are the GROUPED arrays (gwen_gcn_group8): rows are whole groups of 8 entries,
  // padding has weight 0, and an all-zero null group sits at rowptr[N].  A pass that has no row for
  // a lane group (or an empty row) reads the null group, so the hot loop has no per-entry bounds
  // logic.  All loads are unconditional (a load under a per-lane condition would be branched around
  // and followed by vmcnt(0), draining the gathers that are meant to stay in flight).
  const int32_t null_off = rowptr[N];
  auto load_rp = [&](int k, int32_t &first, int32_t &a, int32_t &b) {
    const int r = r0 + k * C::R + gr;
    const bool ok = k < npass && r < r1;
    const int32_t ra = rowptr[ok ? r : N], rb = rowptr[ok ? r + 1 : N];
    a = ra; b = rb;
    first = rb > ra ? ra : null_off;
  };
  auto load_col = [&](int32_t s, int32_t (&c)[kBatch]) {     // one aligned 32-byte group
    const int4_u c0 = *reinterpret_cast<const int4_u *>(col + s);
    const int4_u c1 = *reinterpret_cast<const int4_u *>(col + s + 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) { c[u] = c0[u]; c[u + 4] = c1[u]; }
  };
  auto load_val = [&](int32_t s, float (&w)[kBatch]) {
    const float4_u w0 = *reinterpret_cast<const float4_u *>(val + s);
    const float4_u w1 = *reinterpret_cast<const float4_u *>(val + s + 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) { w[u] = w0[u]; w[u + 4] = w1[u]; }
  };

  // A fragments of the tile whose product is in flight, its accumulators and its first row
  float2_t afr[C::NQ];
  f32x4 d[C::NJ];
  int pend_t0 = -1;

  auto mfma_steps = [&](int q_begin, int q_end) {        // static bounds after unrolling
#pragma unroll
    for (int q = q_begin; q < q_end; ++q) {
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) {
        const float2_t b = *reinterpret_cast<const float2_t *>(bp + j * 16 * C::SW + 8 * q);
        d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[q][0], b[0], d[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) {
        const float2_t b = *reinterpret_cast<const float2_t *>(bp + j * 16 * C::SW + 8 * q);
        d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[q][1], b[1], d[j], 0, 0, 0);
      }
    }
  };
  // bias, ReLU, then D[row = 4*mh + t][col = 16 j + mi] through the 4-row buffer: round t carries
  // rows {t, 4+t, 8+t, 12+t}; each store instruction writes whole rows (16 B per lane)
  float bv[C::NJ];                                       // this lane's bias columns, loaded once
#pragma unroll
  for (int j = 0; j < C::NJ; ++j) bv[j] = bias ? bias[j * 16 + mi] : 0.0f;
  auto epilogue = [&](int t0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) {
        float vv = d[j][t] + bv[j];
        if (relu) vv = vv < 0.0f ? 0.0f : vv;
        ebuf[mh * C::SE + j * 16 + mi] = vv;
      }
      wave_lds_fence();
#pragma unroll
      for (int e = 0; e < 4; e += C::RO) {
        const int er = e + orow, r = t0 + 4 * er + t;
        if (er < 4 && r < r1)
          *reinterpret_cast<float4_t *>(om + (int64_t)r * ldo + ol * 4) =
              *reinterpret_cast<const float4_t *>(ebuf + er * C::SE + ol * 4);
      }
      wave_lds_fence();
    }
  };

  // ---- software pipeline over passes (R rows each) ----------------------------------------------
  // pass k: issue its row gathers; issue the index loads of pass k+1 and the rowptr loads of pass
  // k+2; issue this pass's share of the PREVIOUS tile's MFMAs (they run in the matrix pipe while
  // the gathers are in flight); then wait for the rows, add them up in stored order and park the
  // aggregated rows in the LDS tile.
  int32_t cs, ca, cb, ns, na, nb2;
  int32_t cc[kBatch], nc[kBatch];
  load_rp(0, cs, ca, cb);
  load_rp(1, ns, na, nb2);
  load_col(cs, cc);

#pragma unroll 1
  for (int tl = 0; tl < ntiles; ++tl) {
#pragma unroll
    for (int p = 0; p < C::P; ++p) {
      const int k = tl * C::P + p;
      // row address = wave-uniform base + 32-bit byte offset (one VALU op per gather)
      float4_t v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        v[u] = *reinterpret_cast<const float4_t *>(
            xb + (uint64_t)((uint32_t)cc[u] * kRowBytes + lane_off));
      }
      float cw[kBatch];
      int32_t fs, fa, fb;
      load_val(cs, cw);                     // this pass's weights ride along with its row gathers
      load_col(ns, nc);                     // next pass's source rows
      load_rp(k + 2, fs, fa, fb);           // row bounds of pass k+2
      if (pend_t0 >= 0) mfma_steps(p * C::QP, (p + 1) * C::QP);
      float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        acc = __builtin_elementwise_fma(float4_t{cw[u], cw[u], cw[u], cw[u]}, v[u], acc);
      for (int32_t s = ca + kBatch; s < cb; s += kBatch) {      // rows longer than one group
        int32_t c2[kBatch];
        float w2[kBatch];
        load_col(s, c2);
        load_val(s, w2);
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
          v[u] = *reinterpret_cast<const float4_t *>(
              xb + (uint64_t)((uint32_t)c2[u] * kRowBytes + lane_off));
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
          acc = __builtin_elementwise_fma(float4_t{w2[u], w2[u], w2[u], w2[u]}, v[u], acc);
      }
      *reinterpret_cast<float4_t *>(tile + (p * C::R + gr) * C::ST + gl * 4) = acc;
      cs = ns; ca = na; cb = nb2; ns = fs; na = fa; nb2 = fb;
#pragma unroll
      for (int u = 0; u < kBatch; ++u) cc[u] = nc[u];
    }
    // the tile is complete: retire the previous tile, then take this one's A fragments out of LDS
    if (pend_t0 >= 0) epilogue(pend_t0);
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) afr[q] = *reinterpret_cast<const float2_t *>(ap + 8 * q);
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    wave_lds_fence();
    pend_t0 = r0 + tl * kTile;
  }
  if (pend_t0 >= 0) {
    mfma_steps(0, C::NQ);
    epilogue(pend_t0);
  }
}

template <int FIN, int FOUT, int NW, int MINW>
int launch_v(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
           const float *W, const float *bias, float *out, int64_t N, int64_t ldx, int64_t ldo,
           int64_t members, int64_t msx, int64_t mso, int relu, hipStream_t st) {
  using C = Cfg<FIN, FOUT, NW>;
  constexpr int kThreads = C::kThreads, kWaves = C::kWaves;
  static int per_cu_cached = 0;
  if (per_cu_cached == 0) {   // once per process: LDS opt-in (> 64 KiB) and measured residency
    const void *fn = reinterpret_cast<const void *>(&k_layer<FIN, FOUT, NW, MINW>);
    GWEN_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)C::lds_bytes));
    int nb = 0;
    GWEN_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, kThreads, C::lds_bytes));
    per_cu_cached = nb < 1 ? 1 : nb;
  }
  // a whole number of blocks per CU (<= measured residency), at least ~one tile of rows per wave
  int64_t per_cu = per_cu_cached;
  while (per_cu > 1 && (int64_t)256 * per_cu * kWaves * kTile > N + 256 * kWaves * kTile) --per_cu;
  int64_t blocks = 256 * per_cu;
  const int64_t max_useful = (N + kTile * kWaves - 1) / (kTile * kWaves);
  if (blocks > max_useful) blocks = max_useful;     // small graphs: fewer than one tile per wave
  if (const char *e = getenv("GWEN_K4_RPW")) {       // TUNING ONLY: rows per wave, many short blocks
    const int64_t rpw = atoi(e);
    if (rpw > 0) blocks = (N + rpw * kWaves - 1) / (rpw * kWaves);
  }
  dim3 grid((unsigned)blocks, (unsigned)members);
  k_layer<FIN, FOUT, NW, MINW><<<grid, kThreads, C::lds_bytes, st>>>(
      rowptr, col, val, x, W, bias, out, (int32_t)N, ldx, ldo, msx, mso, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

// Waves per block: 8 from 64 input channels up (one W copy in LDS serves more rows), else 4.
template <int FIN, int FOUT>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
           const float *W, const float *bias, float *out, int64_t N, int64_t ldx, int64_t ldo,
           int64_t members, int64_t msx, int64_t mso, int relu, hipStream_t st) {
  constexpr int NWV = FIN >= 64 ? 8 : 4;
  if (const char *e = getenv("GWEN_K4_WAVES"))
    if (atoi(e) == 4)
      return launch_v<FIN, FOUT, 4, 1>(rowptr, col, val, x, W, bias, out, N, ldx, ldo, members, msx,
                                       mso, relu, st);
  return launch_v<FIN, FOUT, NWV, 1>(rowptr, col, val, x, W, bias, out, N, ldx, ldo, members, msx,
                                     mso, relu, st);
}

constexpr bool width_ok(int64_t f) { return f == 16 || f == 32 || f == 64 || f == 128; }

}  // namespace

extern "C" int gwen_gcn_layer_supported(int64_t Fin, int64_t Fout) {
  if (!width_ok(Fin) || !width_ok(Fout)) return 0;
  // W + 4 wave tiles must fit one CU's 160 KiB LDS
  const int64_t fmax = Fin > Fout ? Fin : Fout;
  const int64_t bytes = 4 * (Fout * (Fin + 4) + 4 * (kTile * (Fin + 4) + 4 * (Fout + 4)));
  (void)fmax;
  return bytes <= 160 * 1024 ? 1 : 0;
}

extern "C" int gwen_gcn_layer_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                  const float *x, const float *W, const float *bias, float *out,
                                  int64_t N, int64_t Fin, int64_t Fout, int64_t ldx, int64_t ldo,
                                  int64_t members, int64_t mstride_x, int64_t mstride_o, int relu,
                                  gwen_stream_t stream_) {
  if (N < 0 || members < 0 || ldx < Fin || ldo < Fout) return GWEN_EINVAL;
  if (!gwen_gcn_layer_supported(Fin, Fout)) return GWEN_EINVAL;
  if (N == 0 || members == 0) return GWEN_OK;
  if (!rowptr || !col || !val || !x || !W || !out || x == out) return GWEN_EINVAL;
  if (N >= (int64_t(1) << 31) - 1 || members > 65535) return GWEN_ERANGE;
  if (!gwen_aligned(x, 16) || !gwen_aligned(out, 16) || !gwen_aligned(W, 16) || ldx != Fin ||
      ldo % 4 || mstride_x % 4 || mstride_o % 4)
    return GWEN_EINVAL;                       // x rows must be contiguous (32-bit row offsets)
  if (N * Fin * 4 >= (int64_t(1) << 32)) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
#define GWEN_L(FI, FO)                                                                          \
  if (Fin == FI && Fout == FO)                                                                  \
    return launch<FI, FO>(rowptr, col, val, x, W, bias, out, N, ldx, ldo, members, mstride_x,   \
                          mstride_o, relu, st)
  GWEN_L(16, 16); GWEN_L(16, 32); GWEN_L(16, 64); GWEN_L(16, 128);
  GWEN_L(32, 16); GWEN_L(32, 32); GWEN_L(32, 64); GWEN_L(32, 128);
  GWEN_L(64, 16); GWEN_L(64, 32); GWEN_L(64, 64); GWEN_L(64, 128);
  GWEN_L(128, 16); GWEN_L(128, 32); GWEN_L(128, 64); GWEN_L(128, 128);
#undef GWEN_L
  return GWEN_EINVAL;
}
