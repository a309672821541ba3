// K9 -- a CHAIN of narrow GCN layers in ONE launch by overlapped tiling.
//
// GNNModel's hourglass (/root/reference/src/gwen/models_gnn.py:135-157, :189-212: C -> H -> H/2 -> H/4 -> H/2
// -> H -> C) puts its three middle gathers at widths 32 / 16 / 16 (H = 64): as separate kernels each is a
// ~10 us launch that moves 6-13 MB -- launch-to-drain floors, not work.  A layer's output on a tile of 64
// destination rows depends only on its input on the tile's 1-hop neighbourhood, so L chained layers can be
// computed per tile from the input on the tile's L-hop neighbourhood, RE-computing the shrinking halo at every
// layer instead of exchanging it through memory (overlapped tiling): on the nu = 100 mesh in Hilbert order a
// tile's 1 / 2 / 3-hop sets hold 107 / 153 / 204 rows, at widths <= 32 that redundancy is cheap and everything
// lives in LDS.
//
// gwen_gcn_hops64 (part of K1, once per graph): per tile the LEVEL-MAJOR list of rows -- level 0 = the tile's own
//   rows in order, level k = rows first reached at hop k, ascending -- with counts n_0 <= ... <= n_H, and for
//   every row of levels < H its (up to 8) entries as local indices into that list plus weights.
// gwen_gcn_narrow_chain_f32: stages s = 1..S (S <= H) on shrinking row sets: stage s computes rows [0, n_{S-s})
//     g[r]  = sum_e w_e * in[lid_e]                         (gather at the stage's input width, from LDS)
//     out[r] = act( g[r] W^T + b )   or, for a PRE-PROJECTED input (first stage only),  act( g[r] + b )
//   the last stage computes the tile's own rows and stores them.  fp32 FMAs throughout (exact products).
#include "common.h"

namespace {

constexpr int kRows = GWEN_TILE_ROWS;          // 64
constexpr int kLMax = GWEN_HOPS_LMAX;          // 288 list slots per tile (3-hop sets of the Hilbert-ordered mesh: <= 264)
constexpr int kEMax = GWEN_HOPS_EMAX;          // 192 rows with entries (2-hop sets: <= 186)
constexpr int kNone = 0x7fffffff;

// in-LDS bitonic sort of n2 (power of two) int64 keys by 256 threads
__device__ inline void bitonic(long long *key, int n2) {
  for (int k2 = 2; k2 <= n2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const long long a = key[i], b = key[p];
          const bool up = (i & k2) == 0;
          if ((a > b) == up) { key[i] = b; key[p] = a; }
        }
      }
      __syncthreads();
    }
  }
}

// One block per tile.  list[] grows level by level: the candidates of a level are the columns of the newest
// level's rows; (row, is-candidate) keys of the whole list plus the candidates are sorted together and a
// candidate row is new when no list entry precedes it.
__global__ __launch_bounds__(256) void k_hops64(const int32_t *__restrict__ rowptr,
                                                const int32_t *__restrict__ col,
                                                const float *__restrict__ val, int32_t N, int32_t H,
                                                int32_t *__restrict__ h_cnt, int32_t *__restrict__ h_rows,
                                                uint16_t *__restrict__ h_lid, float *__restrict__ h_val,
                                                int32_t *__restrict__ status) {
  __shared__ long long key[2048];
  __shared__ int32_t list[512];
  __shared__ int32_t cnt[6];
  __shared__ int32_t tcount[256];
  __shared__ int32_t n_new, bad;
  const int t = blockIdx.x, tid = threadIdx.x;
  const int b0 = t * kRows;
  const int n0 = N - b0 < kRows ? N - b0 : kRows;
  if (tid < n0) list[tid] = b0 + tid;
  if (tid == 0) { cnt[0] = n0; bad = 0; }
  __syncthreads();
  int lo = 0, hi = n0;                            // newest level = list[lo, hi)
  for (int k = 1; k <= H; ++k) {
    // keys: list entries (flag 0), columns of the newest level's rows (flag 1), padding
    const int ncand = (hi - lo) * 8;
    int total = hi + ncand;
    int n2 = 64;
    while (n2 < total) n2 <<= 1;
    if (n2 > 2048) { if (tid == 0) bad = 1; n2 = 2048; total = total < 2048 ? total : 2048; }
    for (int i = tid; i < n2; i += 256) {
      long long kk = ((long long)kNone << 1);
      if (i < hi) kk = ((long long)list[i] << 1);
      else if (i < total) {
        const int c = i - hi, r = list[lo + (c >> 3)], e = c & 7;
        const int32_t a = rowptr[r], b = rowptr[r + 1];
        if (b - a > 8) bad = 1;
        if (a + e < b) kk = ((long long)col[a + e] << 1) | 1;
      }
      key[i] = kk;
    }
    if (tid == 0) n_new = 0;
    __syncthreads();
    bitonic(key, n2);
    // a candidate is new when it is the first key of its row (list keys sort before candidates of the same row)
    // ranks must be ascending by row: count the new rows before each position with a block scan over flags
    // (n2 <= 2048: 8 per thread, serial prefix over per-thread counts through LDS)
    int mine = 0;
    const int per = n2 / 256 > 0 ? n2 / 256 : 1;
    for (int q = 0; q < per; ++q) {
      const int i = tid * per + q;
      if (i < n2) {
        const long long kk = key[i];
        const bool cand = (kk & 1) && (kk >> 1) != kNone;
        const bool first = i == 0 || (key[i - 1] >> 1) != (kk >> 1);
        if (cand && first) ++mine;
      }
    }
    tcount[tid] = mine;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int i = 0; i < 256; ++i) { const int c = tcount[i]; tcount[i] = run; run += c; }
      n_new = run;
    }
    __syncthreads();
    int at = hi + tcount[tid];
    for (int q = 0; q < per; ++q) {
      const int i = tid * per + q;
      if (i < n2) {
        const long long kk = key[i];
        const bool cand = (kk & 1) && (kk >> 1) != kNone;
        const bool first = i == 0 || (key[i - 1] >> 1) != (kk >> 1);
        if (cand && first) {
          if (at < 512) list[at] = (int32_t)(kk >> 1);
          ++at;
        }
      }
    }
    __syncthreads();
    lo = hi;
    hi = hi + n_new;
    if (hi > 512) { if (tid == 0) bad = 1; hi = 512; }
    if (tid == 0) cnt[k] = hi;
    __syncthreads();
  }
  // ---- outputs ------------------------------------------------------------------------------------------
  const int nl = hi, ne = cnt[H - 1];              // list length, rows with entries (levels < H)
  if (tid <= H) h_cnt[t * 6 + tid] = cnt[tid] < kLMax ? cnt[tid] : kLMax;
  for (int i = tid; i < kLMax; i += 256) h_rows[(int64_t)t * kLMax + i] = i < nl ? list[i] : -1;
  // (row, position) sorted by row for the local-id look-ups
  int n2 = 64;
  while (n2 < nl) n2 <<= 1;
  for (int i = tid; i < n2; i += 256) key[i] = i < nl ? (((long long)list[i] << 10) | i) : ((long long)kNone << 10);
  __syncthreads();
  bitonic(key, n2);
  for (int s = tid; s < kEMax * 8; s += 256) {
    const int r = s >> 3, e = s & 7;
    uint16_t lid = 0;
    float w = 0.0f;
    if (r < ne) {
      const int32_t row = list[r], a = rowptr[row], b = rowptr[row + 1];
      const int32_t c = a + e < b ? col[a + e] : (a < b ? col[a] : row);      // padding repeats the first entry
      if (a + e < b) w = val[a + e];
      int l2 = 0, h2 = nl;
      while (l2 < h2) {
        const int mid = (l2 + h2) >> 1;
        if ((int32_t)(key[mid] >> 10) < c) l2 = mid + 1; else h2 = mid;
      }
      lid = (uint16_t)(key[l2 < nl ? l2 : 0] & 1023);
    }
    h_lid[(int64_t)t * kEMax * 8 + s] = lid;
    h_val[(int64_t)t * kEMax * 8 + s] = w;
  }
  if (tid == 0) {
    if (bad || nl > kLMax || ne > kEMax) atomicOr(&status[0], 1);
    atomicMax(&status[1], nl);
    atomicMax(&status[2], ne);
  }
}

__global__ void k_hops_init(int32_t *status) { status[0] = 0; status[1] = 0; status[2] = 0; }

// ---- the chain kernel ---------------------------------------------------------------------------------------
struct ChainArgs {
  const float *W[GWEN_HOPS_MAX_STAGES];
  const float *bias[GWEN_HOPS_MAX_STAGES];
  int32_t fin[GWEN_HOPS_MAX_STAGES], fout[GWEN_HOPS_MAX_STAGES], relu[GWEN_HOPS_MAX_STAGES];
  int32_t woff[GWEN_HOPS_MAX_STAGES];                        // float offset of the stage's W in the LDS image
  int32_t n_stages;
};

constexpr int kChainThreads = 512;
constexpr int kWMax = GWEN_HOPS_WMAX;                        // floats of weights over all stages
constexpr int kKX = (kLMax * 8 + kChainThreads - 1) / kChainThreads;   // 16-B items of X per thread: 5
// LDS: X (kLMax x 32 floats: the staged input, later a stage buffer) | Y (kEMax x 32 floats) | lid (kEMax x 8
// u16) | val (kEMax x 8 f32) | W of every stage | bias of every stage: 80 KB -> two blocks per CU.  A stage
// gathers from one buffer into the other (g) and writes its result back over its own, by then dead, source.
// The tables and input rows of the NEXT tile travel in registers while the current one is computed (its row ids
// one tile further ahead), so only the first tile of a block waits for memory.
constexpr int kOffX = 0;
constexpr int kOffY = kOffX + kLMax * 32 * 4;               // 36864
constexpr int kOffLid = kOffY + kEMax * 32 * 4;             // + 24576
constexpr int kOffVal = kOffLid + kEMax * 8 * 2;            // + 3072
constexpr int kOffW = kOffVal + kEMax * 8 * 4;              // + 6144
constexpr int kOffB = kOffW + kWMax * 4;                    // + 10240
constexpr int kChainLds = kOffB + GWEN_HOPS_MAX_STAGES * 64 * 4;   // 81920
static_assert(kChainLds <= 81920, "two blocks per CU");

__device__ inline int log2i(int v) { return 31 - __builtin_clz(v); }

#ifdef GWEN_HOPS_STAMPS   // diagnostic build only (tools/experiments/hops_stamps.py): s_memtime per phase and wave
__device__ uint64_t *g_stamps = nullptr;
#define STAMP_DECL uint64_t tacc[12] = {}; uint64_t tprev = __builtin_amdgcn_s_memtime()
#define STAMP(k) do { const uint64_t tn = __builtin_amdgcn_s_memtime(); tacc[k] += tn - tprev; tprev = tn; } while (0)
#define STAMP_FLUSH do { if ((threadIdx.x & 63) == 0 && g_stamps) for (int k = 0; k < 12; ++k) \
    g_stamps[(blockIdx.x * (kChainThreads / 64) + (threadIdx.x >> 6)) * 12 + k] = tacc[k]; } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#endif

__global__ __launch_bounds__(kChainThreads) void k_narrow_chain(
    const int32_t *__restrict__ h_cnt, const int32_t *__restrict__ h_rows,
    const uint16_t *__restrict__ h_lid, const float *__restrict__ h_val, const float *__restrict__ x,
    float *__restrict__ out, int32_t N, int32_t T, int32_t G, int64_t mstride_x, int64_t mstride_o,
    ChainArgs A) {
  __shared__ __attribute__((aligned(16))) char lds[kChainLds];
  float *X = reinterpret_cast<float *>(lds + kOffX);
  float *Y = reinterpret_cast<float *>(lds + kOffY);
  const uint16_t *lid = reinterpret_cast<const uint16_t *>(lds + kOffLid);
  const float *wv = reinterpret_cast<const float *>(lds + kOffVal);
  float *Wl = reinterpret_cast<float *>(lds + kOffW);
  float *bl = reinterpret_cast<float *>(lds + kOffB);
  const int tid = threadIdx.x;
  const int S = A.n_stages;
  const int f0 = A.fin[0], sh0 = log2i(f0 / 4);
  const int n_items = kLMax << sh0;                          // 16-B items of the input list

  for (int s = 0; s < S; ++s) {                              // weights and biases: once per block
    if (A.W[s])
      for (int i = tid; i < A.fout[s] * A.fin[s]; i += kChainThreads) {      // rows at pitch fin + 4: see the epilogue
        const int c = i / A.fin[s], k = i - c * A.fin[s];
        Wl[A.woff[s] + c * (A.fin[s] + 4) + k] = A.W[s][i];
      }
    if (tid < A.fout[s]) bl[s * 64 + tid] = A.bias[s] ? A.bias[s][tid] : 0.0f;
  }

  int32_t rid[kKX];                                          // row ids of this thread's items, one tile ahead of px
  float4_t px[kKX];
  uint64_t plid = 0;
  float4_t pval = {0.f, 0.f, 0.f, 0.f};
  auto load_ids = [&](int tile) {
    const int t = tile < G ? tile % T : -1;
#pragma unroll
    for (int k = 0; k < kKX; ++k) {
      const int i = tid + k * kChainThreads;
      rid[k] = (t >= 0 && i < n_items) ? h_rows[(int64_t)t * kLMax + (i >> sh0)] : -1;
    }
  };
  auto load_rows = [&](int tile) {                           // needs rid of the same tile
    if (tile >= G) return;
    const int m = tile / T, t = tile - m * T;
    const float *xm = x + (int64_t)m * mstride_x;
#pragma unroll
    for (int k = 0; k < kKX; ++k) {
      const int p = (tid + k * kChainThreads) & ((1 << sh0) - 1);
      if (rid[k] >= 0) px[k] = *reinterpret_cast<const float4_t *>(xm + (int64_t)rid[k] * f0 + 4 * p);
    }
    if (tid < kEMax * 2) {
      plid = reinterpret_cast<const uint64_t *>(h_lid + (int64_t)t * kEMax * 8)[tid];
      pval = reinterpret_cast<const float4_t *>(h_val + (int64_t)t * kEMax * 8)[tid];
    }
  };

  STAMP_DECL;
  int tile = blockIdx.x;
  load_ids(tile);
  load_rows(tile);
  load_ids(tile + gridDim.x);
  for (; tile < G; tile += gridDim.x) {
    const int m = tile / T, t = tile - m * T;
    float *om = out + (int64_t)m * mstride_o;
    const int32_t *cn = h_cnt + t * 6;
    // ---- the prefetched rows and tables go to LDS; the next tile's are requested ---------------------------
    {
      const int n_in = cn[S];
#pragma unroll
      for (int k = 0; k < kKX; ++k) {
        const int i = tid + k * kChainThreads, r = i >> sh0;
        if (r < n_in) *reinterpret_cast<float4_t *>(X + 4 * i) = px[k];
      }
      if (tid < kEMax * 2) {
        reinterpret_cast<uint64_t *>(lds + kOffLid)[tid] = plid;
        reinterpret_cast<float4_t *>(lds + kOffVal)[tid] = pval;
      }
    }
    STAMP(0);
    __syncthreads();
    STAMP(1);
    load_rows(tile + gridDim.x);
    load_ids(tile + 2 * gridDim.x);
    float *src = X, *oth = Y;
    for (int s = 0; s < S; ++s) {
      const int fi = A.fin[s], fo = A.fout[s];
      const int nr = cn[S - 1 - s];                           // rows this stage computes: its (S-1-s)-hop set
      const bool proj = A.W[s] != nullptr;
      const bool last = s + 1 == S;
      const float *Ws = Wl + A.woff[s], *bs = bl + s * 64;
      // gather: fi / 4 lanes per row, the 8 entries in stored order (fma chain, as K4's gather)
      {
        const int sh = log2i(fi / 4);
        for (int i = tid; i < (nr << sh); i += kChainThreads) {
          const int r = i >> sh, p = i - (r << sh);
          const uint4_t l4 = *reinterpret_cast<const uint4_t *>(lid + r * 8);
          const float4_t w0 = *reinterpret_cast<const float4_t *>(wv + r * 8);
          const float4_t w1 = *reinterpret_cast<const float4_t *>(wv + r * 8 + 4);
          float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float w = e < 4 ? w0[e & 3] : w1[e & 3];
            const uint32_t li = (l4[e >> 1] >> (16 * (e & 1))) & 0xffffu;
            const float4_t v = *reinterpret_cast<const float4_t *>(src + (int)li * fi + 4 * p);
            acc = __builtin_elementwise_fma(float4_t{w, w, w, w}, v, acc);
          }
          *reinterpret_cast<float4_t *>(oth + r * fi + 4 * p) = acc;
        }
      }
      STAMP(2);
      __syncthreads();                                        // src is dead from here: the result goes there
      STAMP(3);
      if (proj) {
        // one output column per lane, its W row in registers, the gathered row read as broadcasts: a wave's 16-B
        // reads hit one address per 16-lane group (fo >= 16) -- no bank conflicts, no weight re-reads
        const int c = tid & (fo - 1), lf = log2i(fo), rstep = kChainThreads >> lf;
        float w[32];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          float4_t t4 = {0.f, 0.f, 0.f, 0.f};
          if (4 * k4 < fi) t4 = *reinterpret_cast<const float4_t *>(Ws + c * (fi + 4) + 4 * k4);
          w[4 * k4] = t4[0]; w[4 * k4 + 1] = t4[1]; w[4 * k4 + 2] = t4[2]; w[4 * k4 + 3] = t4[3];
        }
        const float bc = bs[c];
        for (int r = tid >> lf; r < nr; r += rstep) {
          const float *gr = oth + r * fi;
          float o = 0.0f;
#pragma unroll
          for (int k4 = 0; k4 < 8; ++k4) {
            if (4 * k4 < fi) {
              const float4_t a = *reinterpret_cast<const float4_t *>(gr + 4 * k4);
              o = __builtin_fmaf(a[0], w[4 * k4], o);
              o = __builtin_fmaf(a[1], w[4 * k4 + 1], o);
              o = __builtin_fmaf(a[2], w[4 * k4 + 2], o);
              o = __builtin_fmaf(a[3], w[4 * k4 + 3], o);
            }
          }
          o += bc;
          if (A.relu[s]) o = o < 0.0f ? 0.0f : o;
          if (last) {
            const int row = t * kRows + r;                    // level 0 = the tile's own rows in order
            if (row < N) om[(int64_t)row * fo + c] = o;
          } else {
            src[r * fo + c] = o;
          }
        }
      } else {
        const int sh = log2i(fo / 4);
        for (int i = tid; i < (nr << sh); i += kChainThreads) {
          const int r = i >> sh, p = i - (r << sh);
          float4_t o = *reinterpret_cast<const float4_t *>(oth + r * fi + 4 * p);
          o += *reinterpret_cast<const float4_t *>(bs + 4 * p);
          if (A.relu[s]) {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = o[c] < 0.0f ? 0.0f : o[c];
          }
          if (last) {
            const int row = t * kRows + r;
            if (row < N) *reinterpret_cast<float4_t *>(om + (int64_t)row * fo + 4 * p) = o;
          } else {
            *reinterpret_cast<float4_t *>(src + r * fo + 4 * p) = o;
          }
        }
      }
      STAMP(4 + 2 * s);
      __syncthreads();
      STAMP(5 + 2 * s);
    }
  }
  STAMP_FLUSH;
}

#ifdef GWEN_HOPS_STAMPS
}
extern "C" int gwen_hops_set_stamps(uint64_t *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }
namespace {
#endif

}  // namespace

extern "C" int gwen_gcn_hops64(const int32_t *rowptr, const int32_t *col, const float *val, int64_t N,
                               int64_t H, int32_t *h_cnt, int32_t *h_rows, uint16_t *h_lid,
                               float *h_val, int32_t *status, gwen_stream_t stream_) {
  if (N < 0 || H < 1 || H > GWEN_HOPS_MAX_STAGES || !status) return GWEN_EINVAL;
  if (N >= (int64_t(1) << 31) - 64) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  k_hops_init<<<1, 1, 0, st>>>(status);
  GWEN_LAUNCH_CHECK();
  if (N == 0) return GWEN_OK;
  if (!rowptr || !col || !val || !h_cnt || !h_rows || !h_lid || !h_val) return GWEN_EINVAL;
  const int64_t T = (N + kRows - 1) / kRows;
  k_hops64<<<(unsigned)T, 256, 0, st>>>(rowptr, col, val, (int32_t)N, (int32_t)H, h_cnt, h_rows, h_lid, h_val,
                                        status);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int gwen_gcn_narrow_chain_supported(int32_t n_stages, const int32_t *fin, const int32_t *fout,
                                               const int32_t *has_w) {
  if (n_stages < 1 || n_stages > GWEN_HOPS_MAX_STAGES || !fin || !fout || !has_w) return 0;
  int64_t wsum = 0;
  for (int s = 0; s < n_stages; ++s) {
    // widths 4, 8, 16, 32 (64 out of the last stage): F / 4 lanes per row, a power of two
    if (fin[s] < 4 || fin[s] > 32 || fout[s] < 4 || fout[s] > 64 || (fin[s] & (fin[s] - 1)) ||
        (fout[s] & (fout[s] - 1)))
      return 0;
    if (!has_w[s] && fin[s] != fout[s]) return 0;
    if (s > 0 && (fin[s] != fout[s - 1] || !has_w[s])) return 0;     // only the first stage may be pre-projected
    if (s + 1 < n_stages && fout[s] > 32) return 0;                   // intermediates stay <= 32 wide in LDS
    if (has_w[s]) wsum += (int64_t)(fin[s] + 4) * fout[s];        // LDS rows at pitch fin + 4
  }
  return wsum <= GWEN_HOPS_WMAX;                                      // every stage's W stays in LDS
}

extern "C" int gwen_gcn_narrow_chain_f32(const int32_t *h_cnt, const int32_t *h_rows, const uint16_t *h_lid,
                                         const float *h_val, int64_t H, const float *x, float *out,
                                         int64_t N, int32_t n_stages, const float *const *W,
                                         const float *const *bias, const int32_t *fin, const int32_t *fout,
                                         const int32_t *relu, int64_t members, int64_t mstride_x,
                                         int64_t mstride_o, gwen_stream_t stream_) {
  if (N < 0 || members < 0 || !W || !bias || !fin || !fout || !relu) return GWEN_EINVAL;
  int32_t has_w[GWEN_HOPS_MAX_STAGES];
  if (n_stages < 1 || n_stages > GWEN_HOPS_MAX_STAGES || n_stages > H) return GWEN_EINVAL;
  for (int s = 0; s < n_stages; ++s) has_w[s] = W[s] != nullptr;
  if (!gwen_gcn_narrow_chain_supported(n_stages, fin, fout, has_w)) return GWEN_EINVAL;
  if (N == 0 || members == 0) return GWEN_OK;
  if (!h_cnt || !h_rows || !h_lid || !h_val || !x || !out || x == out) return GWEN_EINVAL;
  if (!gwen_aligned(x, 16) || !gwen_aligned(out, 16) || mstride_x % 4 || mstride_o % 4) return GWEN_EINVAL;
  ChainArgs A;
  for (int s = 0; s < GWEN_HOPS_MAX_STAGES; ++s) {
    const bool on = s < n_stages;
    A.W[s] = on ? W[s] : nullptr; A.bias[s] = on ? bias[s] : nullptr;
    A.fin[s] = on ? fin[s] : 0; A.fout[s] = on ? fout[s] : 0; A.relu[s] = on ? relu[s] : 0;
  }
  A.n_stages = n_stages;
  for (int s = 0, off = 0; s < GWEN_HOPS_MAX_STAGES; ++s) {
    A.woff[s] = off;
    if (A.W[s]) off += (A.fin[s] + 4) * A.fout[s];
  }
  const int64_t T = (N + kRows - 1) / kRows;
  if (T * members >= (int64_t(1) << 31) || members > 65535) return GWEN_ERANGE;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    GWEN_HIP_CHECK(hipGetDevice(&dev));
    GWEN_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = n < 1 ? 1 : n;
  }
  const int64_t G = T * members;
  const int64_t bx = G < 2 * cus ? G : 2 * cus;               // two resident blocks per CU walk the tiles
  k_narrow_chain<<<(unsigned)bx, kChainThreads, 0, gwen_stream(stream_)>>>(
      h_cnt, h_rows, h_lid, h_val, x, out, (int32_t)N, (int32_t)T, (int32_t)G, mstride_x, mstride_o, A);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
