// K1 -- graph preparation on the device: gcn_norm + add_remaining_self_loops, once per graph.
//
// Replaces what torch-geometric 2.3.1 does inside EVERY GCNConv.forward of the reference
// (/root/reference/src/gwen/models_gnn.py:147-149,:204-206; constructors :118-184 leave
// cached=False): mask/cat/ones/scatter_add/pow/masked_fill/2x index_select/2x mul, ~9 launches per
// layer, 6 layers per forward, on a graph that never changes.
//
// Output is CSR by TARGET with the self-loop stored LAST in each row and the kept edges in
// ORIGINAL order, i.e. exactly the per-destination visiting order of the reference's CPU
// scatter-add, so K2's sequential sums are bit-identical to that path.
//
// Ordering uses one device radix sort of UNIQUE 64-bit keys (target << ebits | edge id), so no
// stability assumption is needed and no atomics decide an order: the CSR is bitwise reproducible.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr int kThreads = 256;

__host__ __device__ inline int bits_for(uint64_t n) {  // smallest b with (1<<b) > n-1, >= 1
  int b = 1;
  while ((uint64_t(1) << b) < n) ++b;
  return b;
}

__global__ void k_init(int32_t *loop_last, int64_t N, int32_t *status) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < N) loop_last[i] = -1;
  if (i == 0) { status[0] = 0; status[1] = 0; }
}

// key = (target << ebits) | e for kept edges; dropped edges (explicit loops when loops are being
// completed, out-of-range indices) get target = N so they sort behind every row.
__global__ void k_mark(const int64_t *__restrict__ ei, int64_t Nsrc, int64_t N, int64_t E,
                       int add_loops, int ebits, uint64_t *__restrict__ keys,
                       int32_t *__restrict__ loop_last, int32_t *__restrict__ status) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  uint64_t tgt;
  if (s < 0 || s >= Nsrc || d < 0 || d >= N) {
    atomicOr(status, 1);
    tgt = (uint64_t)N;
  } else if (add_loops && s == d) {
    atomicMax(&loop_last[s], (int32_t)e);   // an existing loop keeps its weight, last one wins
    tgt = (uint64_t)N;
  } else {
    tgt = (uint64_t)d;
  }
  keys[e] = (tgt << ebits) | (uint64_t)e;
}

__device__ inline int64_t lower_bound_u64(const uint64_t *a, int64_t n, uint64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// rowptr[r] = (#kept edges with target < r) + (r completed loops before row r)
__global__ void k_rowptr(const uint64_t *__restrict__ ks, int64_t E, int64_t N, int ebits,
                         int add_loops, int32_t *__restrict__ rowptr, int32_t *__restrict__ status) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > N) return;
  int64_t lb = lower_bound_u64(ks, E, (uint64_t)r << ebits);
  int32_t v = (int32_t)(lb + (add_loops ? r : 0));
  rowptr[r] = v;
  if (r == N) status[1] = v;
}

__global__ void k_fill(const uint64_t *__restrict__ ks, const int64_t *__restrict__ ei,
                       const float *__restrict__ ew, int64_t E, int64_t N, int ebits, int add_loops,
                       int32_t *__restrict__ col, float *__restrict__ val, int32_t *__restrict__ eid) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= E) return;
  uint64_t k = ks[i];
  int64_t d = (int64_t)(k >> ebits);
  if (d >= N) return;
  int64_t e = (int64_t)(k & ((uint64_t(1) << ebits) - 1));
  int64_t slot = i + (add_loops ? d : 0);
  col[slot] = (int32_t)ei[e];
  val[slot] = ew ? ew[e] : 1.0f;
  eid[slot] = (int32_t)e;
}

__global__ void k_loops(const int32_t *__restrict__ rowptr, int64_t N, const float *__restrict__ ew,
                        const int32_t *__restrict__ loop_last, float fill, int32_t *__restrict__ col,
                        float *__restrict__ val, int32_t *__restrict__ eid) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= N) return;
  int32_t slot = rowptr[r + 1] - 1;
  int32_t le = loop_last[r];
  col[slot] = (int32_t)r;
  val[slot] = (ew && le >= 0) ? ew[le] : fill;
  eid[slot] = -1;
}

// deg[i] = sum of raw weights of row i, added in stored order (== CPU scatter-add order);
// dis = deg^-1/2 as 1/sqrt (what torch's pow(-0.5) evaluates on CPU), +inf -> 0.
__global__ void k_deg(const int32_t *__restrict__ rowptr, const float *__restrict__ val, int64_t N,
                      float *__restrict__ dis) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= N) return;
  float s = 0.0f;
  for (int32_t p = rowptr[r], e = rowptr[r + 1]; p < e; ++p) s = s + val[p];
  float d = 1.0f / sqrtf(s);
  if (isinf(d) && d > 0.0f) d = 0.0f;
  dis[r] = d;
}

// val = (dis[src] * w) * dis[dst]   -- same association as deg_inv_sqrt[row] * w * deg_inv_sqrt[col]
__global__ void k_norm(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                       const float *__restrict__ dis, int64_t N, float *__restrict__ val) {
  // one wave-sized group of 8 lanes per row keeps long rows (complete graphs) parallel
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t r = t >> 3;
  int lane = (int)(t & 7);
  if (r >= N) return;
  float dd = dis[r];
  for (int32_t p = rowptr[r] + lane, e = rowptr[r + 1]; p < e; p += 8)
    val[p] = (dis[col[p]] * val[p]) * dd;
}

// mean aggregation of a rectangular (bipartite) graph: val /= sum of the row's raw weights
__global__ void k_rowmean(const int32_t *__restrict__ rowptr, int64_t N, float *__restrict__ val) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= N) return;
  const int32_t p0 = rowptr[r], p1 = rowptr[r + 1];
  float s = 0.0f;
  for (int32_t p = p0; p < p1; ++p) s = s + val[p];
  for (int32_t p = p0; p < p1; ++p) val[p] = val[p] / s;
}

// ---- transpose ------------------------------------------------------------------------------
__device__ inline int32_t row_of_slot(const int32_t *rowptr, int64_t N, int32_t s) {
  // largest r with rowptr[r] <= s
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    int64_t mid = (lo + hi + 1) >> 1;
    if (rowptr[mid] <= s) lo = mid; else hi = mid - 1;
  }
  return (int32_t)lo;
}

// N = rows of the CSR being transposed, NT = rows of its transpose (== N on a square graph)
__global__ void k_tmark(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                        int64_t N, int64_t NT, int64_t cap, int sbits, uint64_t *__restrict__ keys) {
  int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s >= cap) return;
  int32_t nnz = rowptr[N];
  uint64_t src = (s < nnz) ? (uint64_t)col[s] : (uint64_t)NT;
  keys[s] = (src << sbits) | (uint64_t)s;
}

__global__ void k_trowptr(const uint64_t *__restrict__ ks, int64_t cap, int64_t N, int sbits,
                          int32_t *__restrict__ t_rowptr) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > N) return;
  t_rowptr[r] = (int32_t)lower_bound_u64(ks, cap, (uint64_t)r << sbits);
}

__global__ void k_tfill(const uint64_t *__restrict__ ks, const int32_t *__restrict__ rowptr,
                        const float *__restrict__ val, int64_t cap, int64_t N, int64_t NT, int sbits,
                        int32_t *__restrict__ t_col, float *__restrict__ t_val) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= cap) return;
  uint64_t k = ks[i];
  if ((int64_t)(k >> sbits) >= NT) return;
  int32_t s = (int32_t)(k & ((uint64_t(1) << sbits) - 1));
  t_col[i] = row_of_slot(rowptr, N, s);
  t_val[i] = val[s];
}

// ---- grouped layout (rows padded to whole groups of 8 entries) ------------------------------
__global__ void k_glen(const int32_t *__restrict__ rowptr, int64_t N, int32_t *__restrict__ glen,
                       int32_t *__restrict__ uniform) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > N) return;
  const int32_t g = r < N ? ((rowptr[r + 1] - rowptr[r] + 7) & ~7) : 0;
  glen[r] = g;
  if (uniform && r < N && g != 8) atomicAnd(uniform, 0);     // some row is not exactly one group
}

__global__ void k_set1(int32_t *p) { *p = 1; }

// ---- row segmentation (long rows): row r of length len is cut into max(1, ceil(len / S)) segments -----
__global__ void k_segcount(const int32_t *__restrict__ rowptr, int64_t N, int32_t S,
                           int32_t *__restrict__ cnt, int32_t *__restrict__ status) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > N) return;
  int32_t c = 0;
  if (r < N) {
    const int32_t len = rowptr[r + 1] - rowptr[r];
    c = len <= S ? 1 : (len + S - 1) / S;
    atomicMax(&status[0], c);
    atomicMax(&status[1], len);
  }
  cnt[r] = c;
}

__global__ void k_segfill(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ rowptr2,
                          int64_t N, int32_t S, int32_t *__restrict__ seg_rowptr,
                          int32_t *__restrict__ col2, float *__restrict__ val2) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > N) return;
  if (r == N) { seg_rowptr[rowptr2[N]] = rowptr[N]; return; }
  const int32_t a = rowptr[r], k0 = rowptr2[r], k1 = rowptr2[r + 1];
  for (int32_t k = k0; k < k1; ++k) {
    seg_rowptr[k] = a + (k - k0) * S;
    col2[k] = k;
    val2[k] = 1.0f;
  }
}

// 8 lanes per row copy its entries and append the padding (weight 0, column of the row's first
// entry: a row that is already part of the sum); lane group N writes the all-zero null group.
__global__ void k_gfill(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                        const float *__restrict__ val, const int32_t *__restrict__ g_rowptr,
                        int64_t N, int32_t *__restrict__ g_col, float *__restrict__ g_val) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t r = t >> 3;
  int lane = (int)(t & 7);
  if (r > N) return;
  if (r == N) {                      // null group right behind the last row
    g_col[g_rowptr[N] + lane] = 0;
    g_val[g_rowptr[N] + lane] = 0.0f;
    return;
  }
  const int32_t s0 = rowptr[r], len = rowptr[r + 1] - s0;
  const int32_t g0 = g_rowptr[r], glen = g_rowptr[r + 1] - g0;
  const int32_t first = len > 0 ? col[s0] : 0;
  for (int32_t p = lane; p < glen; p += 8) {
    g_col[g0 + p] = p < len ? col[s0 + p] : first;
    g_val[g0 + p] = p < len ? val[s0 + p] : 0.0f;
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

struct WsLayout {
  size_t keys_in, keys_out, loop_last, temp, temp_bytes, total;
};

int ws_layout(int64_t N, int64_t E, WsLayout *L) {
  int64_t cap = E + N;                       // the transpose sorts up to cap keys
  size_t off = 0;
  L->keys_in = off;  off = gwen_align_up(off + sizeof(uint64_t) * (size_t)(cap > 0 ? cap : 1), 256);
  L->keys_out = off; off = gwen_align_up(off + sizeof(uint64_t) * (size_t)(cap > 0 ? cap : 1), 256);
  L->loop_last = off; off = gwen_align_up(off + sizeof(int32_t) * (size_t)(N > 0 ? N : 1), 256);
  size_t tb = 0;
  hipError_t e = rocprim::radix_sort_keys(nullptr, tb, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                          (size_t)(cap > 0 ? cap : 1), 0u, 64u);
  if (e != hipSuccess) return (int)e;
  L->temp = off; L->temp_bytes = tb; off = gwen_align_up(off + tb, 256);
  L->total = off;
  return GWEN_OK;
}

}  // namespace

extern "C" int gwen_gcn_prep_workspace_bytes(int64_t N, int64_t E, size_t *bytes) {
  if (!bytes || N < 0 || E < 0) return GWEN_EINVAL;
  if (N + E >= (int64_t(1) << 31) - 1) return GWEN_ERANGE;
  WsLayout L;
  int rc = ws_layout(N, E, &L);
  if (rc != GWEN_OK) return rc;
  *bytes = L.total;
  return GWEN_OK;
}

extern "C" int gwen_gcn_prep(const int64_t *edge_index, const float *edge_weight, int64_t N,
                             int64_t E, int add_self_loops, float fill_value, int normalize,
                             int32_t *rowptr, int32_t *col, float *val, int32_t *eid, float *dis,
                             int32_t *status, void *workspace, size_t workspace_bytes,
                             gwen_stream_t stream_) {
  if (N < 0 || E < 0 || !rowptr || !status || (E > 0 && !edge_index)) return GWEN_EINVAL;
  if (N + E >= (int64_t(1) << 31) - 1) return GWEN_ERANGE;
  if ((N + E > 0) && (!col || !val || !eid)) return GWEN_EINVAL;
  if (N > 0 && !dis) return GWEN_EINVAL;
  WsLayout L;
  int rc = ws_layout(N, E, &L);
  if (rc != GWEN_OK) return rc;
  if (!workspace || workspace_bytes < L.total) return GWEN_ENOSPACE;
  hipStream_t stream = gwen_stream(stream_);
  char *ws = static_cast<char *>(workspace);
  uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws + L.keys_in);
  uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws + L.keys_out);
  int32_t *loop_last = reinterpret_cast<int32_t *>(ws + L.loop_last);
  const int ebits = bits_for((uint64_t)(E > 1 ? E : 2));
  const int nbits = bits_for((uint64_t)N + 1);

  k_init<<<blocks_for(N > 0 ? N : 1), kThreads, 0, stream>>>(loop_last, N, status);
  GWEN_LAUNCH_CHECK();
  const uint64_t *ks = keys_in;
  if (E > 0) {
    k_mark<<<blocks_for(E), kThreads, 0, stream>>>(edge_index, N, N, E, add_self_loops, ebits,
                                                   keys_in, loop_last, status);
    GWEN_LAUNCH_CHECK();
    size_t tb = L.temp_bytes;
    GWEN_HIP_CHECK(rocprim::radix_sort_keys(ws + L.temp, tb, keys_in, keys_out, (size_t)E, 0u,
                                            (unsigned)(ebits + nbits), stream));
    ks = keys_out;
  }
  k_rowptr<<<blocks_for(N + 1), kThreads, 0, stream>>>(ks, E, N, ebits, add_self_loops, rowptr,
                                                       status);
  GWEN_LAUNCH_CHECK();
  if (E > 0) {
    k_fill<<<blocks_for(E), kThreads, 0, stream>>>(ks, edge_index, edge_weight, E, N, ebits,
                                                   add_self_loops, col, val, eid);
    GWEN_LAUNCH_CHECK();
  }
  if (N > 0) {
    if (add_self_loops) {
      k_loops<<<blocks_for(N), kThreads, 0, stream>>>(rowptr, N, edge_weight, loop_last,
                                                      fill_value, col, val, eid);
      GWEN_LAUNCH_CHECK();
    }
    if (normalize) {
      k_deg<<<blocks_for(N), kThreads, 0, stream>>>(rowptr, val, N, dis);
      GWEN_LAUNCH_CHECK();
      k_norm<<<blocks_for(N * 8), kThreads, 0, stream>>>(rowptr, col, dis, N, val);
      GWEN_LAUNCH_CHECK();
    }
  }
  return GWEN_OK;
}

extern "C" int gwen_gcn_prep_rect(const int64_t *edge_index, const float *edge_weight,
                                  int64_t N_src, int64_t N_dst, int64_t E, int mean,
                                  int32_t *rowptr, int32_t *col, float *val, int32_t *eid,
                                  int32_t *status, void *workspace, size_t workspace_bytes,
                                  gwen_stream_t stream_) {
  if (N_src < 0 || N_dst < 0 || E < 0 || !rowptr || !status || (E > 0 && !edge_index))
    return GWEN_EINVAL;
  if (N_src + E >= (int64_t(1) << 31) - 1 || N_dst + E >= (int64_t(1) << 31) - 1) return GWEN_ERANGE;
  if (E > 0 && (!col || !val || !eid)) return GWEN_EINVAL;
  WsLayout L;
  int rc = ws_layout(N_dst, E, &L);
  if (rc != GWEN_OK) return rc;
  if (!workspace || workspace_bytes < L.total) return GWEN_ENOSPACE;
  hipStream_t stream = gwen_stream(stream_);
  char *ws = static_cast<char *>(workspace);
  uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws + L.keys_in);
  uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws + L.keys_out);
  int32_t *loop_last = reinterpret_cast<int32_t *>(ws + L.loop_last);
  const int ebits = bits_for((uint64_t)(E > 1 ? E : 2));
  const int nbits = bits_for((uint64_t)N_dst + 1);
  k_init<<<blocks_for(N_dst > 0 ? N_dst : 1), kThreads, 0, stream>>>(loop_last, N_dst, status);
  GWEN_LAUNCH_CHECK();
  const uint64_t *ks = keys_in;
  if (E > 0) {
    k_mark<<<blocks_for(E), kThreads, 0, stream>>>(edge_index, N_src, N_dst, E, 0, ebits, keys_in,
                                                   loop_last, status);
    GWEN_LAUNCH_CHECK();
    size_t tb = L.temp_bytes;
    GWEN_HIP_CHECK(rocprim::radix_sort_keys(ws + L.temp, tb, keys_in, keys_out, (size_t)E, 0u,
                                            (unsigned)(ebits + nbits), stream));
    ks = keys_out;
  }
  k_rowptr<<<blocks_for(N_dst + 1), kThreads, 0, stream>>>(ks, E, N_dst, ebits, 0, rowptr, status);
  GWEN_LAUNCH_CHECK();
  if (E > 0) {
    k_fill<<<blocks_for(E), kThreads, 0, stream>>>(ks, edge_index, edge_weight, E, N_dst, ebits, 0,
                                                   col, val, eid);
    GWEN_LAUNCH_CHECK();
    if (mean && N_dst > 0) {
      k_rowmean<<<blocks_for(N_dst), kThreads, 0, stream>>>(rowptr, N_dst, val);
      GWEN_LAUNCH_CHECK();
    }
  }
  return GWEN_OK;
}

extern "C" int gwen_gcn_transpose_rect(const int32_t *rowptr, const int32_t *col, const float *val,
                                       int64_t N, int64_t N_t, int64_t cap, int32_t *t_rowptr,
                                       int32_t *t_col, float *t_val, void *workspace,
                                       size_t workspace_bytes, gwen_stream_t stream_) {
  if (N < 0 || N_t < 0 || cap < 0 || !rowptr || !t_rowptr) return GWEN_EINVAL;
  if (cap >= (int64_t(1) << 31) - 1 || N_t >= (int64_t(1) << 31) - 1) return GWEN_ERANGE;
  if (cap > 0 && (!col || !val || !t_col || !t_val)) return GWEN_EINVAL;
  WsLayout L;
  const int64_t nmax = N > N_t ? N : N_t;
  int rc = ws_layout(nmax, cap, &L);
  if (rc != GWEN_OK) return rc;
  if (!workspace || workspace_bytes < L.total) return GWEN_ENOSPACE;
  hipStream_t stream = gwen_stream(stream_);
  char *ws = static_cast<char *>(workspace);
  uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws + L.keys_in);
  uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws + L.keys_out);
  const int sbits = bits_for((uint64_t)(cap > 1 ? cap : 2));
  const int nbits = bits_for((uint64_t)N_t + 1);
  const uint64_t *ks = keys_in;
  if (cap > 0) {
    k_tmark<<<blocks_for(cap), kThreads, 0, stream>>>(rowptr, col, N, N_t, cap, sbits, keys_in);
    GWEN_LAUNCH_CHECK();
    size_t tb = L.temp_bytes;
    GWEN_HIP_CHECK(rocprim::radix_sort_keys(ws + L.temp, tb, keys_in, keys_out, (size_t)cap, 0u,
                                            (unsigned)(sbits + nbits), stream));
    ks = keys_out;
  }
  k_trowptr<<<blocks_for(N_t + 1), kThreads, 0, stream>>>(ks, cap, N_t, sbits, t_rowptr);
  GWEN_LAUNCH_CHECK();
  if (cap > 0) {
    k_tfill<<<blocks_for(cap), kThreads, 0, stream>>>(ks, rowptr, val, cap, N, N_t, sbits, t_col, t_val);
    GWEN_LAUNCH_CHECK();
  }
  return GWEN_OK;
}

extern "C" int gwen_gcn_transpose(const int32_t *rowptr, const int32_t *col, const float *val,
                                  int64_t N, int64_t cap, int32_t *t_rowptr, int32_t *t_col,
                                  float *t_val, void *workspace, size_t workspace_bytes,
                                  gwen_stream_t stream_) {
  return gwen_gcn_transpose_rect(rowptr, col, val, N, N, cap, t_rowptr, t_col, t_val, workspace,
                                 workspace_bytes, stream_);
}

extern "C" int64_t gwen_gcn_group8_capacity(int64_t N, int64_t cap) {
  if (N < 0 || cap < 0) return GWEN_EINVAL;
  return cap + 7 * N + 8;            // every row padded by at most 7 entries + the null group
}

extern "C" int gwen_gcn_group8(const int32_t *rowptr, const int32_t *col, const float *val,
                               int64_t N, int64_t cap, int32_t *g_rowptr, int32_t *g_col,
                               float *g_val, int32_t *uniform, void *workspace,
                               size_t workspace_bytes, gwen_stream_t stream_) {
  if (N < 0 || cap < 0 || !rowptr || !g_rowptr || !g_col || !g_val) return GWEN_EINVAL;
  if (cap > 0 && (!col || !val)) return GWEN_EINVAL;
  if (cap + 7 * N + 8 >= (int64_t(1) << 31) - 1) return GWEN_ERANGE;
  hipStream_t stream = gwen_stream(stream_);
  const size_t glen_bytes = gwen_align_up(sizeof(int32_t) * (size_t)(N + 1), 256);
  size_t tb = 0;
  GWEN_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, (int32_t *)nullptr, (int32_t *)nullptr, 0,
                                         (size_t)(N + 1), rocprim::plus<int32_t>(), stream));
  if (!workspace || workspace_bytes < glen_bytes + tb) return GWEN_ENOSPACE;
  int32_t *glen = static_cast<int32_t *>(workspace);
  void *temp = static_cast<char *>(workspace) + glen_bytes;
  if (uniform) {
    k_set1<<<1, 1, 0, stream>>>(uniform);
    GWEN_LAUNCH_CHECK();
  }
  k_glen<<<blocks_for(N + 1), kThreads, 0, stream>>>(rowptr, N, glen, uniform);
  GWEN_LAUNCH_CHECK();
  GWEN_HIP_CHECK(rocprim::exclusive_scan(temp, tb, glen, g_rowptr, 0, (size_t)(N + 1),
                                         rocprim::plus<int32_t>(), stream));
  k_gfill<<<blocks_for((N + 1) * 8), kThreads, 0, stream>>>(rowptr, col, val, g_rowptr, N, g_col,
                                                            g_val);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int64_t gwen_gcn_segments_capacity(int64_t N, int64_t nnz, int64_t S) {
  if (N < 0 || nnz < 0 || S < 1) return GWEN_EINVAL;
  return N + nnz / S + 1;                        // every row at least one segment, a long row len / S + 1
}

extern "C" int gwen_gcn_segments(const int32_t *rowptr, int64_t N, int64_t S, int32_t *rowptr2,
                                 int32_t *seg_rowptr, int32_t *col2, float *val2, int32_t *status,
                                 void *workspace, size_t workspace_bytes, gwen_stream_t stream_) {
  if (N < 0 || S < 1 || S >= (int64_t(1) << 30) || !rowptr || !rowptr2 || !seg_rowptr || !col2 || !val2 ||
      !status)
    return GWEN_EINVAL;
  if (N >= (int64_t(1) << 31) - 2) return GWEN_ERANGE;
  hipStream_t stream = gwen_stream(stream_);
  const size_t cnt_bytes = gwen_align_up(sizeof(int32_t) * (size_t)(N + 1), 256);
  size_t tb = 0;
  GWEN_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, (int32_t *)nullptr, (int32_t *)nullptr, 0,
                                         (size_t)(N + 1), rocprim::plus<int32_t>(), stream));
  if (!workspace || workspace_bytes < cnt_bytes + tb) return GWEN_ENOSPACE;
  int32_t *cnt = static_cast<int32_t *>(workspace);
  void *temp = static_cast<char *>(workspace) + cnt_bytes;
  GWEN_HIP_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int32_t), stream));
  k_segcount<<<blocks_for(N + 1), kThreads, 0, stream>>>(rowptr, N, (int32_t)S, cnt, status);
  GWEN_LAUNCH_CHECK();
  GWEN_HIP_CHECK(rocprim::exclusive_scan(temp, tb, cnt, rowptr2, 0, (size_t)(N + 1),
                                         rocprim::plus<int32_t>(), stream));
  k_segfill<<<blocks_for(N + 1), kThreads, 0, stream>>>(rowptr, rowptr2, N, (int32_t)S, seg_rowptr, col2,
                                                        val2);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
