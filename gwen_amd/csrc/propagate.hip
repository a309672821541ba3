// K2 -- fused gather / scale / segmented sum / bias / ReLU over a CSR-by-target graph.
//
// Replaces, per GCNConv layer of the reference, the eager chain  index_select (materialises
// [E',F]) -> mul -> scatter_add_ (float atomics) -> add bias -> relu
// (torch-geometric 2.3.1 MessagePassing.propagate as called from
//  /root/reference/src/gwen/models_gnn.py:147-149,:204-206; ReLU at :147-149,:204-205)
// by ONE pass with no [E',F] intermediate, no atomics, one store per output element.
//
// Mapping (wave64): a row (destination node) x feature-chunk item is owned by G lanes, each lane
// holding V consecutive floats (V = 4 -> 16-B loads); 256/G items per 256-thread block.
//   F = 64 -> G = 16, four destination rows per wave; F = 256 -> one wave per row;
//   F = 16 -> G = 4.  Wider F loops over chunks of G*V features (item = row * nchunks + chunk).
// The neighbour list is walked U = 8 at a time: indices/weights first, then U independent row
// gathers in flight, then the adds in stored order, each product rounded before its add (no FMA;
// this TU is compiled with -ffp-contract=off), so out is bit-identical to a sequential CPU
// scatter-add in edge order and identical run to run.
#include "common.h"

namespace {

template <int V> struct Vec;
template <> struct Vec<4> { using T = float4_t; };
template <> struct Vec<2> { using T = float2_t; };
template <> struct Vec<1> { using T = float; };

template <int V>
__device__ inline typename Vec<V>::T vload(const float *p) {
  return *reinterpret_cast<const typename Vec<V>::T *>(p);
}
template <int V>
__device__ inline void vstore(float *p, typename Vec<V>::T v) {
  *reinterpret_cast<typename Vec<V>::T *>(p) = v;
}
template <int V>
__device__ inline typename Vec<V>::T vzero() {
  typename Vec<V>::T z = {};
  return z;
}
__device__ inline float relu_f(float v) { return v < 0.0f ? 0.0f : v; }   // NaN stays NaN
template <int V>
__device__ inline typename Vec<V>::T vrelu(typename Vec<V>::T v) {
  if constexpr (V == 1) return relu_f(v);
  else {
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = relu_f(v[i]);
    return v;
  }
}

constexpr int kBlock = 256;
constexpr int kUnroll = 8;

template <int G, int V, bool REMAP>
__global__ __launch_bounds__(kBlock) void k_propagate(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ h, const float *__restrict__ bias,
    float *__restrict__ out, int64_t n_items, int32_t nchunks, int32_t F, int64_t ldh, int64_t ldo,
    int64_t mstride_h, int64_t mstride_o, int relu) {
  using VT = typename Vec<V>::T;
  int64_t lb = blockIdx.x;
  if constexpr (REMAP) {   // blocks sharing an XCD (blockIdx % 8) take neighbouring rows
    const int64_t nb = gridDim.x, xcd = lb & 7, q8 = nb >> 3, r8 = nb & 7;
    lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lb >> 3);
  }
  const int64_t item = lb * (kBlock / G) + threadIdx.x / G;
  const int lane = threadIdx.x % G;
  if (item >= n_items) return;
  const int64_t r = nchunks == 1 ? item : item / nchunks;
  const int32_t f = (int32_t)(item - r * nchunks) * (G * V) + lane * V;
  if (f >= F) return;
  const float *hm = h + (int64_t)blockIdx.y * mstride_h + f;
  float *om = out + (int64_t)blockIdx.y * mstride_o + r * ldo + f;

  const int32_t s0 = rowptr[r], s1 = rowptr[r + 1];
  VT acc = vzero<V>();
  for (int32_t s = s0; s < s1; s += kUnroll) {
    int32_t c[kUnroll];
    float w[kUnroll];
    VT v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int32_t p = (s + u < s1) ? s + u : s1 - 1;
      c[u] = col[p];
      w[u] = val[p];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = vload<V>(hm + (int64_t)c[u] * ldh);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
      if (s + u < s1) acc = acc + w[u] * v[u];
  }
  if (bias) acc = acc + vload<V>(bias + f);
  if (relu) acc = vrelu<V>(acc);
  vstore<V>(om, acc);
}

template <int G, int V>
int launch(const int32_t *rowptr, const int32_t *col, const float *val, const float *h,
           const float *bias, float *out, int64_t N, int64_t F, int64_t ldh, int64_t ldo,
           int64_t members, int64_t msh, int64_t mso, int relu, hipStream_t stream) {
  const int64_t nchunks = (F + G * V - 1) / (G * V);
  const int64_t n_items = N * nchunks;
  const int64_t per_block = kBlock / G;
  const int64_t blocks = (n_items + per_block - 1) / per_block;
  if (blocks > 0x7fffffffLL || members > 65535) return GWEN_ERANGE;
  dim3 grid((unsigned)blocks, (unsigned)members);
  // XCD remap pays when rows of one XCD are neighbours (one chunk per row); with several feature
  // chunks per row the items of a row already sit side by side
  if (nchunks == 1)
    k_propagate<G, V, true><<<grid, kBlock, 0, stream>>>(rowptr, col, val, h, bias, out, n_items,
                                                         (int32_t)nchunks, (int32_t)F, ldh, ldo, msh,
                                                         mso, relu);
  else
    k_propagate<G, V, false><<<grid, kBlock, 0, stream>>>(rowptr, col, val, h, bias, out, n_items,
                                                          (int32_t)nchunks, (int32_t)F, ldh, ldo,
                                                          msh, mso, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

template <int V>
int dispatch_g(int64_t lanes_needed, const int32_t *rowptr, const int32_t *col, const float *val,
               const float *h, const float *bias, float *out, int64_t N, int64_t F, int64_t ldh,
               int64_t ldo, int64_t members, int64_t msh, int64_t mso, int relu, hipStream_t st) {
#define GWEN_CASE(G)                                                                          \
  if (lanes_needed <= G)                                                                      \
    return launch<G, V>(rowptr, col, val, h, bias, out, N, F, ldh, ldo, members, msh, mso, relu, st)
  GWEN_CASE(1);
  GWEN_CASE(2);
  GWEN_CASE(4);
  GWEN_CASE(8);
  GWEN_CASE(16);
  GWEN_CASE(32);
#undef GWEN_CASE
  return launch<64, V>(rowptr, col, val, h, bias, out, N, F, ldh, ldo, members, msh, mso, relu, st);
}

}  // namespace

extern "C" int gwen_gcn_propagate_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                      const float *h, const float *bias, float *out, int64_t N,
                                      int64_t F, int64_t ldh, int64_t ldo, int64_t members,
                                      int64_t mstride_h, int64_t mstride_o, int relu,
                                      gwen_stream_t stream_) {
  if (N < 0 || F < 0 || members < 0 || ldh < F || ldo < F) return GWEN_EINVAL;
  if (N == 0 || F == 0 || members == 0) return GWEN_OK;
  if (!rowptr || !h || !out || h == out) return GWEN_EINVAL;
  if (N >= (int64_t(1) << 31) - 1 || F >= (int64_t(1) << 31) - 1) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  // widest vector the shapes and pointers allow
  const bool a16 = gwen_aligned(h, 16) && gwen_aligned(out, 16) && (!bias || gwen_aligned(bias, 16));
  const bool a8 = gwen_aligned(h, 8) && gwen_aligned(out, 8) && (!bias || gwen_aligned(bias, 8));
  auto all4 = [&](int64_t m) { return F % m == 0 && ldh % m == 0 && ldo % m == 0 &&
                                      mstride_h % m == 0 && mstride_o % m == 0; };
  if (a16 && all4(4))
    return dispatch_g<4>((F + 3) / 4, rowptr, col, val, h, bias, out, N, F, ldh, ldo, members,
                         mstride_h, mstride_o, relu, st);
  if (a8 && all4(2))
    return dispatch_g<2>((F + 1) / 2, rowptr, col, val, h, bias, out, N, F, ldh, ldo, members,
                         mstride_h, mstride_o, relu, st);
  return dispatch_g<1>(F, rowptr, col, val, h, bias, out, N, F, ldh, ldo, members, mstride_h,
                       mstride_o, relu, st);
}
