// Tile layout of a prepared graph for K8 (wide.hip): destination rows in tiles of 64, and per tile the
// UNION of the source rows its entries name, so that K8 can stage every source row of a tile ONCE in
// LDS (by LDS-DMA) and serve the 7-8 gathers per destination row from there instead of through the
// vector L1 path.  On a locality-ordered bounded-degree mesh (the c2-c5 geodesic meshes in Morton
// order) a tile's 448 entries name ~116 distinct rows (64 of them its own).
//
// Part of K1 (once per graph; gcn_norm of torch-geometric 2.3.1 as called from
// /root/reference/src/gwen/models_gnn.py:147-149,:204-206 has no such structure: it is recomputed
// per layer there).  Works on the plain CSR (rowptr/col/val of gwen_gcn_prep / gwen_gcn_prep_rect).
//
// One 256-thread block per tile: the tile's (up to) 512 columns are sorted in LDS (bitonic), the
// distinct ones ranked, and every entry gets the rank of its column as a 16-bit local id.
#include "common.h"

namespace {

constexpr int kTileRows = GWEN_TILE_ROWS;       // 64
constexpr int kTileEnt = GWEN_TILE_ROWS * 8;    // 512 entry slots per tile
constexpr int kUnion = GWEN_TILE_UNION;         // 192 union slots per tile
constexpr int kNone = 0x7fffffff;

__global__ __launch_bounds__(256) void k_tiles64(const int32_t *__restrict__ rowptr,
                                                 const int32_t *__restrict__ col,
                                                 const float *__restrict__ val, int32_t N,
                                                 int32_t *__restrict__ t_rows,
                                                 uint16_t *__restrict__ t_lid,
                                                 float *__restrict__ t_val,
                                                 int32_t *__restrict__ status) {
  __shared__ int32_t key[kTileEnt];      // sorted columns
  __shared__ int32_t ent[kTileEnt];      // column per entry slot (stored order), kNone = padding
  __shared__ int32_t uniq[kTileEnt];     // distinct columns, ascending
  __shared__ int32_t rank[kTileEnt];
  __shared__ int32_t nuniq, toolong;
  const int t = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { nuniq = 0; toolong = 0; }
  __syncthreads();
  // ---- entry slots: row lr = slot / 8, position k = slot % 8 ---------------------------------------
  for (int s = tid; s < kTileEnt; s += 256) {
    const int r = t * kTileRows + (s >> 3), k = s & 7;
    int32_t c = kNone;
    float w = 0.0f;
    if (r < N) {
      const int32_t a = rowptr[r], b = rowptr[r + 1];
      if (b - a > 8 && k == 0) toolong = 1;
      if (a + k < b) { c = col[a + k]; w = val[a + k]; }
    }
    ent[s] = c;
    key[s] = c;
    t_val[(int64_t)t * kTileEnt + s] = w;
  }
  __syncthreads();
  // ---- bitonic sort of the 512 keys ------------------------------------------------------------------
  for (int k2 = 2; k2 <= kTileEnt; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < kTileEnt; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const int32_t a = key[i], b = key[p];
          const bool up = (i & k2) == 0;
          if ((a > b) == up) { key[i] = b; key[p] = a; }
        }
      }
      __syncthreads();
    }
  }
  // ---- distinct columns: rank by a serial-free scan (512 flags, 256 threads, 2 per thread) -----------
  for (int i = tid; i < kTileEnt; i += 256)
    rank[i] = (key[i] != kNone && (i == 0 || key[i] != key[i - 1])) ? 1 : 0;
  __syncthreads();
  for (int off = 1; off < kTileEnt; off <<= 1) {          // Hillis-Steele inclusive scan
    int32_t v[2];
    for (int q = 0; q < 2; ++q) {
      const int i = tid + 256 * q;
      v[q] = rank[i] + (i >= off ? rank[i - off] : 0);
    }
    __syncthreads();
    for (int q = 0; q < 2; ++q) rank[tid + 256 * q] = v[q];
    __syncthreads();
  }
  for (int i = tid; i < kTileEnt; i += 256) {
    const bool head = key[i] != kNone && (i == 0 || key[i] != key[i - 1]);
    if (head) uniq[rank[i] - 1] = key[i];
  }
  if (tid == 0) nuniq = rank[kTileEnt - 1];
  __syncthreads();
  const int nu = nuniq;
  // ---- local ids: rank of the entry's column among the distinct ones (binary search) ------------------
  for (int s = tid; s < kTileEnt; s += 256) {
    int32_t c = ent[s];
    if (c == kNone) c = ent[s & ~7];           // padding reads the row's first entry (weight 0)
    int lid = 0;
    if (c != kNone) {
      int lo = 0, hi = nu;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (uniq[mid] < c) lo = mid + 1; else hi = mid;
      }
      lid = lo;
    }
    t_lid[(int64_t)t * kTileEnt + s] = (uint16_t)(lid < kUnion ? lid : 0);
  }
  // ---- union list: slot k holds the k-th distinct row; K8 fetches slots in groups of 4 (one LDS-DMA wave
  // instruction = 4 rows x 256 B): a group that starts past the union is marked -1 (skipped), slots past
  // the union inside a started group repeat the first row ------------------------------------------------
  for (int k = tid; k < kUnion; k += 256)
    t_rows[(int64_t)t * kUnion + k] = k < nu ? uniq[k] : ((k & ~3) < nu ? uniq[0] : -1);
  if (tid == 0) {
    if (nu > kUnion || toolong) atomicOr(&status[0], 1);
    atomicMax(&status[1], nu);
  }
}

__global__ void k_tiles_init(int32_t *status) { status[0] = 0; status[1] = 0; }

}  // namespace

extern "C" int64_t gwen_gcn_tiles64_count(int64_t N) {
  return N <= 0 ? 0 : (N + kTileRows - 1) / kTileRows;
}

extern "C" int gwen_gcn_tiles64(const int32_t *rowptr, const int32_t *col, const float *val,
                                int64_t N, int32_t *t_rows, uint16_t *t_lid, float *t_val,
                                int32_t *status, gwen_stream_t stream_) {
  if (N < 0 || !status) return GWEN_EINVAL;
  if (N >= (int64_t(1) << 31) - 64) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  k_tiles_init<<<1, 1, 0, st>>>(status);
  GWEN_LAUNCH_CHECK();
  if (N == 0) return GWEN_OK;
  if (!rowptr || !col || !val || !t_rows || !t_lid || !t_val) return GWEN_EINVAL;
  const int64_t T = gwen_gcn_tiles64_count(N);
  k_tiles64<<<(unsigned)T, 256, 0, st>>>(rowptr, col, val, (int32_t)N, t_rows, t_lid, t_val, status);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
