// K4 -- one whole GCNConv layer (+ReLU) in a single launch, aggregate-first:
//            out = act( (A~ x) W^T + bias )
// Replaces, per layer of the reference, the sequence  lin (SGEMM) -> index_select -> mul ->
// scatter_add_ -> + bias -> relu  (torch-geometric 2.3.1 GCNConv.forward as called from
// /root/reference/src/gwen/models_gnn.py:147-149,:204-206) by ONE kernel: the [N,Fout] intermediate
// `h`, the [E',F] message tensor and four elementwise passes never touch HBM.
//
// Work split (wave64, 256-thread blocks = 4 waves, all waves independent after W is staged):
//   * every wave owns a CONTIGUOUS range of destination rows (N split evenly over all waves of the
//     grid -> no tail quantisation) and walks it in tiles of 16 rows;
//   * gather phase: FIN/4 lanes per row (16-B loads), 64/(FIN/4) rows at a time; each lane-group walks
//     its row's CSR segment 8 neighbours per batch (8 independent row gathers in flight), adds the
//     terms in stored order (rounded product, then add -- same order/rounding as K2) and parks the
//     aggregated row in the wave's private LDS tile  agg[16][FIN+4];
//   * MFMA phase: v_mfma_f32_16x16x4_f32 with A = agg tile, B = W^T from the block's LDS copy of
//     W[FOUT][FIN+4]; both operands are read as ds_read_b64 with the k-permutation
//     k = 8q + 2*(lane>>4) + s, which with a row pitch of FIN+4 floats is bank-conflict free;
//   * epilogue: + bias, ReLU, D tile back through the same LDS tile, 16-B coalesced row stores.
// Blocks are remapped so that the blocks sharing an XCD (blockIdx % 8) own neighbouring row ranges:
// gathered rows are then re-used inside one 4 MiB L2 instead of being fetched by all eight.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kTile = 16;
constexpr int kBatch = 8;

__device__ inline void wave_lds_fence() {
  // LDS ops of one wave execute in order; this only stops the compiler from moving them.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int FIN, int FOUT>
struct Cfg {
  static constexpr int G = FIN / 4;                 // lanes per gathered row
  static constexpr int R = 64 / G;                  // rows gathered at a time by one wave
  static constexpr int FMAX = FIN > FOUT ? FIN : FOUT;
  static constexpr int SW = FIN + 4;                // row pitch of W and of the agg tile (floats)
  static constexpr int ST = FMAX + 4;               // row pitch of the wave tile
  static constexpr int NJ = FOUT / 16;              // 16-column output tiles
  static constexpr int GO = FOUT / 4;               // lanes per stored output row
  static constexpr int RO = 64 / GO;                // output rows stored at a time
  static constexpr size_t lds_bytes = sizeof(float) * (size_t)(FOUT * SW + kWaves * kTile * ST);
};

template <int FIN, int FOUT>
__global__ __launch_bounds__(kThreads) void k_layer(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, const float *__restrict__ W,
    const float *__restrict__ bias, float *__restrict__ out, int32_t N, int64_t ldx, int64_t ldo,
    int64_t mstride_x, int64_t mstride_o, int relu) {
  using C = Cfg<FIN, FOUT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Wl = lds;                                        // [FOUT][SW]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *tile = lds + FOUT * C::SW + wave * (kTile * C::ST);   // [16][ST], private to the wave

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

  // contiguous row range of this wave
  const int nw = nb * kWaves, gw = lb * kWaves + wave;
  const int base = N / nw, extra = N % nw;
  const int r0 = gw * base + (gw < extra ? gw : extra);
  const int r1 = r0 + base + (gw < extra ? 1 : 0);

  const float *xm = x + (int64_t)blockIdx.y * mstride_x;
  float *om = out + (int64_t)blockIdx.y * mstride_o;
  const int gl = lane % C::G, gr = lane / C::G;           // gather: lane within row group, row slot
  const int mi = lane & 15, mh = lane >> 4;               // MFMA: row/col index, k group
  const int ol = lane % C::GO, orow = lane / C::GO;       // store: lane within row, row slot

  for (int t0 = r0; t0 < r1; t0 += kTile) {
    // ---- gather + aggregate 16 rows into the LDS tile ------------------------------------------
#pragma unroll 1
    for (int p = 0; p < kTile; p += C::R) {
      const int lr = p + gr, r = t0 + lr;
      float4_t acc = {0.f, 0.f, 0.f, 0.f};
      if (r < r1) {
        const int32_t s0 = rowptr[r], s1 = rowptr[r + 1];
        const float *xg = xm + gl * 4;
        for (int32_t s = s0; s < s1; s += kBatch) {
          int32_t c[kBatch];
          float w[kBatch];
          float4_t v[kBatch];
#pragma unroll
          for (int u = 0; u < kBatch; ++u) {
            const int32_t pp = (s + u < s1) ? s + u : s1 - 1;
            c[u] = col[pp];
            w[u] = val[pp];
          }
#pragma unroll
          for (int u = 0; u < kBatch; ++u)
            v[u] = *reinterpret_cast<const float4_t *>(xg + (int64_t)c[u] * ldx);
#pragma unroll
          for (int u = 0; u < kBatch; ++u)
            if (s + u < s1) acc = acc + w[u] * v[u];
        }
      }
      *reinterpret_cast<float4_t *>(tile + lr * C::ST + gl * 4) = acc;
    }
    wave_lds_fence();

    // ---- (agg tile) x W^T on the fp32 MFMA -------------------------------------------------------
    f32x4 d[C::NJ];
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ap = tile + mi * C::ST + 2 * mh;
    const float *bp = Wl + mi * C::SW + 2 * mh;
#pragma unroll
    for (int q = 0; q < FIN / 8; ++q) {
      const float2_t a = *reinterpret_cast<const float2_t *>(ap + 8 * q);
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) {
        const float2_t b = *reinterpret_cast<const float2_t *>(bp + j * 16 * C::SW + 8 * q);
        d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], d[j], 0, 0, 0);
        d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], d[j], 0, 0, 0);
      }
    }
    wave_lds_fence();

    // ---- epilogue: bias, ReLU, D[row = 4*mh + t][col = 16 j + mi] -> tile -> coalesced rows ------
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
      const float bv = bias ? bias[j * 16 + mi] : 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = d[j][t];
        if (bias) v = v + bv;
        if (relu) v = v < 0.0f ? 0.0f : v;
        tile[(4 * mh + t) * C::ST + j * 16 + mi] = v;
      }
    }
    wave_lds_fence();
#pragma unroll 1
    for (int p = 0; p < kTile; p += C::RO) {
      const int lr = p + orow, r = t0 + lr;
      if (r < r1)
        *reinterpret_cast<float4_t *>(om + (int64_t)r * ldo + ol * 4) =
            *reinterpret_cast<const float4_t *>(tile + lr * C::ST + ol * 4);
    }
    wave_lds_fence();
  }
}

template <int FIN, int FOUT>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
           const float *W, const float *bias, float *out, int64_t N, int64_t ldx, int64_t ldo,
           int64_t members, int64_t msx, int64_t mso, int relu, hipStream_t st) {
  using C = Cfg<FIN, FOUT>;
  static bool attr_set = false;
  if (!attr_set) {   // > 64 KiB of dynamic LDS needs the opt-in once per process
    GWEN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_layer<FIN, FOUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)C::lds_bytes));
    attr_set = true;
  }
  // as many co-resident blocks as LDS admits (<= 4 per CU), never more waves than 16-row tiles
  int per_cu = (int)((160 * 1024) / C::lds_bytes);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int64_t blocks = 256 * per_cu;
  const int64_t max_useful = (N + kTile * kWaves - 1) / (kTile * kWaves);
  if (blocks > max_useful) blocks = max_useful;
  if (blocks < 1) blocks = 1;
  dim3 grid((unsigned)blocks, (unsigned)members);
  k_layer<FIN, FOUT><<<grid, kThreads, C::lds_bytes, st>>>(rowptr, col, val, x, W, bias, out,
                                                           (int32_t)N, ldx, ldo, msx, mso, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

constexpr bool width_ok(int64_t f) { return f == 16 || f == 32 || f == 64 || f == 128; }

}  // namespace

extern "C" int gwen_gcn_layer_supported(int64_t Fin, int64_t Fout) {
  if (!width_ok(Fin) || !width_ok(Fout)) return 0;
  // W + 4 wave tiles must fit one CU's 160 KiB LDS
  const int64_t fmax = Fin > Fout ? Fin : Fout;
  const int64_t bytes = 4 * (Fout * (Fin + 4) + kWaves * kTile * (fmax + 4));
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
  if (!gwen_aligned(x, 16) || !gwen_aligned(out, 16) || !gwen_aligned(W, 16) || ldx % 4 || ldo % 4 ||
      mstride_x % 4 || mstride_o % 4)
    return GWEN_EINVAL;
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
