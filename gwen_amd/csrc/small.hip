// K7 -- one whole GCNConv layer on a SMALL graph (N <= 256 nodes) with WIDE features, the shape of the
// reference's own workload: its graph is the complete graph over ~125-150 ensemble members
// (/root/reference/src/gwen/utils.py:175-176) and its features are flattened fields, C = height x
// ncells -> hidden 1024 (/root/reference/src/gwen/config.json:9,12; layers
// /root/reference/src/gwen/models_gnn.py:118-130,:172-184).  At that shape K3 + K2 are latency-bound:
// a 125-row projection is 2-8 blocks of the 128 x 128 tile kernel (18-30 us each) and every propagate
// walks 125-entry rows 8 entries per memory round trip (14-17 us each): 272 us per forward.
//
// Here the normalised adjacency is a dense NP x NP matrix D (NP = 128 or 256, zero padded, built once
// per graph by gwen_gcn_dense_f32) and a layer is two chained contractions per block of output columns,
//        h[:, cols] = x W[cols, :]^T          (K = Fin, streamed from global memory, no LDS)
//        out[:, cols] = act( D h[:, cols] + b[cols] )      (K = NP, h through LDS, transposed)
// both as 3xbf16 split MFMAs with fp32 accumulation (see layer.hip).  Only the summation ORDER differs
// from the sequential edge-order sum of K2 (fp32 rounding; parity tolerance 1e-4 as everywhere).
//   block = 4 waves = all NP (padded) rows x BN = 16 NC output columns; wave w owns NP/64 row tiles;
//   W is the MFMA A operand, x the B operand, both read straight from global memory in fragment layout
//   (8 consecutive floats per lane = 32 B, 128 B contiguous per row per k-step), 2-4 k-steps in flight;
//   weights that are re-used come pre-split in fragment order (gwen_gcn_small_pack_f32): one contiguous
//   KB per wave load;
//   a long K (the C -> 1024 projection) is cut over blockIdx.z: partial h tiles go to a workspace and
//   k_small_finish adds them in split order before the second contraction.
#include "common.h"
#include "split.h"

namespace {

using gwen::bf16x8;

// NP = padded node count: 128 (2 row tiles per wave) or 256 (4); pitch of the transposed h tile
// NP + 8 bf16 => conflict-free 16-B reads
constexpr int kMaxNodes = 256;
inline int pad_nodes(int64_t N) { return N <= 128 ? 128 : 256; }

template <int NS>
__device__ inline void split8n(const float4_t a, const float4_t b, bf16x8 (&im)[NS]) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  gwen::split_images<8, NS>(v, im);
}

__device__ inline void split8(const float4_t a, const float4_t b, bf16x8 &hi, bf16x8 &lo) {
  bf16x8 im[2];
  split8n<2>(a, b, im);
  hi = im[0];
  lo = im[1];
}

// d += sum of image products a[i] . b[t - i], smallest terms first; NS = 2: (a0,b1) (a1,b0) (a0,b0)
template <int NS>
__device__ inline f32x4 mma_n(const bf16x8 (&a)[NS], const bf16x8 (&b)[NS], f32x4 d) {
  return gwen::mma_split<8, NS>(a, b, d);
}

// second contraction + bias + ReLU + store, from the h tile this lane holds in D layout
// (d[rt][n]: row (2 wave + rt) 16 + mi, columns c0 + 16 n + 4 mh .. +3)
template <int NP, int NC, int NS>
__device__ inline void aggregate_store(f32x4 (&d)[NP / 64][NC], const float *__restrict__ dense,
                                       const float *__restrict__ bias, float *__restrict__ om,
                                       int N, int Fout, int c0, int relu, __bf16 *ht) {
  constexpr int RT = NP / 64, PJ = NP + 8, KS = NP / 32, kImg = 16 * NC * PJ;   // NS images of h^T, kImg apart
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, mi = lane & 15, mh = lane >> 4;
  // this lane's fragments of D (rows of its row tiles): the first k-step is requested before anything
  // else, every later one a step ahead of the MFMAs that use it
  float4_t dr[RT][2], dn[RT][2];
  auto fetch_d = [&](int ks, float4_t (&r)[RT][2]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float *dp = dense + ((RT * wave + rt) * 16 + mi) * NP + 32 * ks + 8 * mh;
      r[rt][0] = *reinterpret_cast<const float4_t *>(dp);
      r[rt][1] = *reinterpret_cast<const float4_t *>(dp + 4);
    }
  };
  fetch_d(0, dr);
  // h^T into LDS (column-major: ht[c][j]), split once per element
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int j = (RT * wave + rt) * 16 + mi;
#pragma unroll
    for (int n = 0; n < NC; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float r = d[rt][n][i];
        const int c = 16 * n + 4 * mh + i;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
          const __bf16 h = (__bf16)r;
          ht[s_ * kImg + c * PJ + j] = h;
          r = r - (float)h;
        }
      }
  }
  __syncthreads();
  f32x4 o[RT][NC];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int n = 0; n < NC; ++n) o[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) fetch_d(ks + 1, dn);
    bf16x8 a[NC][NS];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int off = (16 * n + mi) * PJ + 32 * ks + 8 * mh;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) a[n][s_] = *reinterpret_cast<const bf16x8 *>(ht + s_ * kImg + off);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      bf16x8 b[NS];
      split8n<NS>(dr[rt][0], dr[rt][1], b);
#pragma unroll
      for (int n = 0; n < NC; ++n) o[rt][n] = mma_n<NS>(a[n], b, o[rt][n]);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      dr[rt][0] = dn[rt][0];
      dr[rt][1] = dn[rt][1];
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = (RT * wave + rt) * 16 + mi;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = c0 + 16 * n + 4 * mh;
      float4_t v = {o[rt][n][0], o[rt][n][1], o[rt][n][2], o[rt][n][3]};
      if (bias) v += *reinterpret_cast<const float4_t *>(bias + c);
      if (relu) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] < 0.0f ? 0.0f : v[i];
      }
      if (row < N) *reinterpret_cast<float4_t *>(om + (int64_t)row * Fout + c) = v;
    }
  }
}

template <int NP, int NC, bool SPLIT, bool PACKED, int NS>
__global__ __launch_bounds__(256) void k_small(const float *__restrict__ dense,
                                               const float *__restrict__ x,
                                               const float *__restrict__ W,
                                               const float *__restrict__ bias, float *__restrict__ out,
                                               float *__restrict__ part, int N, int Fin, int Fout,
                                               int kchunk, int relu, int64_t mstride_x,
                                               int64_t mstride_o) {
  constexpr int RT = NP / 64, PJ = NP + 8;
  static_assert(!PACKED || NS == 2, "the packed weight images are the two bf16x3 images");
  __shared__ __attribute__((aligned(16))) __bf16 ht[SPLIT ? 8 : NS * 16 * NC * PJ];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, mi = lane & 15, mh = lane >> 4;
  int cb = blockIdx.x, sp = blockIdx.z;
  if (SPLIT && gridDim.y == 1 && gridDim.z % 8 == 0) {
    // the column blocks of one K range share one chunk of x: deal them to ONE XCD (linear block id
    // % 8), whose L2 then serves that chunk after the first read (x does not fit an L2 as a whole)
    const int lin = blockIdx.x + gridDim.x * blockIdx.z, q = lin >> 3;
    cb = q % gridDim.x;
    sp = (q / gridDim.x) * 8 + (lin & 7);
  }
  const int c0 = cb * 16 * NC, member = blockIdx.y;
  const int k0 = sp * kchunk, k1 = k0 + kchunk < Fin ? k0 + kchunk : Fin;
  const float *xp[RT], *wp[NC];
  // PACKED: W points at the bf16 images of gwen_gcn_small_pack_f32 -- fragment (jt, ks) of lane l at
  // ((jt KS + ks) 64 + l) x 16 bytes (hi image, then lo image): one contiguous KB per wave load, a
  // contiguous stream per column tile, no split arithmetic
  const bf16x8 *ip[NC];
  const int64_t lo_off = (int64_t)Fout * Fin / 8;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = (RT * wave + rt) * 16 + mi;
    row = row < N ? row : N - 1;                  // padded rows repeat the last one (finite values)
    xp[rt] = x + member * mstride_x + (int64_t)row * Fin + 8 * mh;
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    wp[n] = W + (int64_t)(c0 + 16 * n + mi) * Fin + 8 * mh;
    ip[n] = reinterpret_cast<const bf16x8 *>(W) + (int64_t)(c0 / 16 + n) * (Fin / 32) * 64 + lane;
  }

  f32x4 d[RT][NC];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int n = 0; n < NC; ++n) d[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // KU k-steps are in flight at any time: slot u is re-requested (KU steps ahead) as soon as its
  // values have been split -- with one step ahead the 32-deep loop ran at one memory round trip per
  // step (27 us for 1024 -> 512 on 125 rows)
  constexpr int KU = (NC == 1 && RT == 2) ? 4 : 2;
  float4_t xr[KU][RT][2], wr[KU][NC][2];
  auto fetch = [&](int u, int k) {
    k = k < k1 ? k : k1 - 32;                       // unconditional load, clamped into the range
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      xr[u][rt][0] = *reinterpret_cast<const float4_t *>(xp[rt] + k);
      xr[u][rt][1] = *reinterpret_cast<const float4_t *>(xp[rt] + k + 4);
    }
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      if constexpr (PACKED) {
        wr[u][n][0] = *reinterpret_cast<const float4_t *>(ip[n] + (k >> 5) * 64);
        wr[u][n][1] = *reinterpret_cast<const float4_t *>(ip[n] + (k >> 5) * 64 + lo_off);
      } else {
        wr[u][n][0] = *reinterpret_cast<const float4_t *>(wp[n] + k);
        wr[u][n][1] = *reinterpret_cast<const float4_t *>(wp[n] + k + 4);
      }
    }
  };
#pragma unroll
  for (int u = 0; u < KU; ++u) fetch(u, k0 + 32 * u);
#pragma unroll 1
  for (int k = k0; k < k1; k += 32 * KU) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      bf16x8 xi[RT][NS], wi[NC][NS];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) split8n<NS>(xr[u][rt][0], xr[u][rt][1], xi[rt]);
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        if constexpr (PACKED) {
          wi[n][0] = __builtin_bit_cast(bf16x8, wr[u][n][0]);
          wi[n][1] = __builtin_bit_cast(bf16x8, wr[u][n][1]);
        } else {
          split8n<NS>(wr[u][n][0], wr[u][n][1], wi[n]);
        }
      }
      fetch(u, k + 32 * (KU + u));
      if (k + 32 * u < k1) {                        // uniform: the range need not be a multiple of KU
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) d[rt][n] = mma_n<NS>(wi[n], xi[rt], d[rt][n]);
      }
    }
  }
  if constexpr (SPLIT) {
    float *pm = part + ((int64_t)sp * gridDim.y + member) * N * Fout;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = (RT * wave + rt) * 16 + mi;
#pragma unroll
      for (int n = 0; n < NC; ++n)
        if (row < N)
          *reinterpret_cast<float4_t *>(pm + (int64_t)row * Fout + c0 + 16 * n + 4 * mh) =
              float4_t{d[rt][n][0], d[rt][n][1], d[rt][n][2], d[rt][n][3]};
    }
  } else {
    aggregate_store<NP, NC, NS>(d, dense, bias, out + member * mstride_o, N, Fout, c0, relu, ht);
  }
}

// adds the partial h tiles in split order, then the second contraction
template <int NP, int NC, int NS>
__global__ __launch_bounds__(256) void k_small_finish(const float *__restrict__ dense,
                                                      const float *__restrict__ part,
                                                      const float *__restrict__ bias,
                                                      float *__restrict__ out, int N, int Fout,
                                                      int nsplit, int relu, int64_t mstride_o) {
  constexpr int RT = NP / 64, PJ = NP + 8;
  __shared__ __attribute__((aligned(16))) __bf16 ht[NS * 16 * NC * PJ];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, mi = lane & 15, mh = lane >> 4;
  const int c0 = blockIdx.x * 16 * NC, member = blockIdx.y;
  f32x4 d[RT][NC];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int row = (RT * wave + rt) * 16 + mi;
    row = row < N ? row : N - 1;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      float4_t acc = {0.f, 0.f, 0.f, 0.f};
      const float *pp = part + (int64_t)member * N * Fout + (int64_t)row * Fout + c0 + 16 * n + 4 * mh;
      const int64_t ss = (int64_t)gridDim.y * N * Fout;
      for (int s = 0; s < nsplit; s += 8) {          // 8 partials in flight, added in split order
        float4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = *reinterpret_cast<const float4_t *>(pp + (s + u < nsplit ? s + u : nsplit - 1) * ss);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s + u < nsplit) acc += v[u];
      }
      d[rt][n] = f32x4{acc[0], acc[1], acc[2], acc[3]};
    }
  }
  aggregate_store<NP, NC, NS>(d, dense, bias, out + member * mstride_o, N, Fout, c0, relu, ht);
}

// dense[i][j] = sum of the stored weights of entries (i <- j), zero elsewhere (NP x NP); one thread per
// row adds its entries in stored order (multi-edges accumulate deterministically)
__global__ void k_dense(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                        const float *__restrict__ val, int N, int NP, float *__restrict__ dense) {
  const int i = blockIdx.x, j = threadIdx.x;            // NP blocks x NP threads
  __shared__ float row[kMaxNodes];
  row[j] = 0.0f;
  __syncthreads();
  if (j == 0 && i < N)
    for (int32_t s = rowptr[i]; s < rowptr[i + 1]; ++s) row[col[s]] += val[s];
  __syncthreads();
  dense[i * NP + j] = row[j];
}

// W [Fout, Fin] fp32 -> hi / lo bf16 images in fragment order; one wave per (column tile, k-step)
__global__ __launch_bounds__(64) void k_pack_w(const float *__restrict__ W, int Fin, int64_t lo_off,
                                               bf16x8 *__restrict__ img) {
  const int lane = threadIdx.x, mi = lane & 15, mh = lane >> 4;
  const int64_t frag = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;       // jt * KS + ks
  const float *wp = W + ((int64_t)blockIdx.y * 16 + mi) * Fin + 32 * blockIdx.x + 8 * mh;
  bf16x8 hi, lo;
  split8(*reinterpret_cast<const float4_t *>(wp), *reinterpret_cast<const float4_t *>(wp + 4), hi, lo);
  img[frag * 64 + lane] = hi;
  img[lo_off + frag * 64 + lane] = lo;
}

struct Shape {
  int nc, nsplit, kchunk;
};

inline Shape shape_for(int64_t N, int64_t Fin, int64_t Fout) {
  Shape s;
  // wide blocks (64 columns) re-read x a quarter as often: worth it when x is large (long K) or when
  // there are plenty of column blocks anyway
  // (with 4 row tiles per wave -- more than 128 nodes -- the registers allow 32 columns, not 64)
  const int wide = pad_nodes(N) == 128 ? 4 : 2;
  s.nc = (Fout % (16 * wide) == 0 && (Fin >= 4096 || Fout >= 4096)) ? wide : 1;
  const int64_t blocks = Fout / (16 * s.nc);
  // K is cut over blocks whenever the column blocks alone leave most CUs idle: a block then walks at
  // least 128 of K (4 k-steps) and the partial tiles are added by k_small_finish
  int64_t n = 1;
  if (Fin >= 512 && blocks < 256) {
    n = (256 + blocks - 1) / blocks;                     // aim at ~256 blocks ...
    if (Fin >= 4096) n *= 2;                             // ... ~512 when K is very long
    if (n > Fin / 128) n = Fin / 128;
    if (n < 1) n = 1;
  }
  s.kchunk = (int)(((Fin + n - 1) / n + 31) / 32 * 32);
  s.nsplit = (int)((Fin + s.kchunk - 1) / s.kchunk);
  return s;
}

}  // namespace

extern "C" int gwen_gcn_small_pad(int64_t N) { return N >= 1 && N <= kMaxNodes ? pad_nodes(N) : GWEN_EINVAL; }

static int small_shape_ok(int64_t N, int64_t Fin, int64_t Fout) {
  return N >= 1 && N <= kMaxNodes && Fin >= 32 && Fin % 32 == 0 && Fout >= 16 && Fout % 16 == 0 ? 1 : 0;
}

extern "C" int gwen_gcn_small_supported(int64_t N, int64_t Fin, int64_t Fout, int contract) {
  if (contract != GWEN_CONTRACT_BF16X3 && contract != GWEN_CONTRACT_BF16X6) return 0;
  return small_shape_ok(N, Fin, Fout);
}

extern "C" int64_t gwen_gcn_small_workspace_floats(int64_t N, int64_t members, int64_t Fin,
                                                   int64_t Fout) {
  if (!small_shape_ok(N, Fin, Fout) || members < 0) return 0;
  const Shape s = shape_for(N, Fin, Fout);
  return s.nsplit > 1 ? (int64_t)s.nsplit * members * N * Fout : 0;
}

extern "C" int gwen_gcn_dense_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                  int64_t N, float *dense, gwen_stream_t stream_) {
  if (N < 1 || N > kMaxNodes || !rowptr || !col || !val || !dense) return GWEN_EINVAL;
  const int np = pad_nodes(N);
  k_dense<<<np, np, 0, gwen_stream(stream_)>>>(rowptr, col, val, (int)N, np, dense);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int64_t gwen_gcn_small_pack_bytes(int64_t Fin, int64_t Fout) {
  return Fin >= 32 && Fin % 32 == 0 && Fout >= 16 && Fout % 16 == 0 ? Fin * Fout * 4 : 0;
}

extern "C" int gwen_gcn_small_pack_f32(const float *W, int64_t Fin, int64_t Fout, void *packed,
                                       gwen_stream_t stream_) {
  if (!gwen_gcn_small_pack_bytes(Fin, Fout) || !W || !packed) return GWEN_EINVAL;
  if (!gwen_aligned(W, 16) || !gwen_aligned(packed, 16) || Fout / 16 > 65535) return GWEN_EINVAL;
  k_pack_w<<<dim3((unsigned)(Fin / 32), (unsigned)(Fout / 16)), 64, 0, gwen_stream(stream_)>>>(
      W, (int)Fin, Fin * Fout / 8, reinterpret_cast<bf16x8 *>(packed));
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int gwen_gcn_small_layer_f32(const float *dense, const float *x, const float *W,
                                        const void *packed,
                                        const float *bias, float *out, int64_t N, int64_t Fin,
                                        int64_t Fout, int64_t members, int64_t mstride_x,
                                        int64_t mstride_o, int relu, float *workspace,
                                        int64_t workspace_floats, int contract, gwen_stream_t stream_) {
  if (members < 0 || !gwen_gcn_small_supported(N, Fin, Fout, contract)) return GWEN_EINVAL;
  if (members == 0) return GWEN_OK;
  const bool x6 = contract == GWEN_CONTRACT_BF16X6;
  if (x6 && W) packed = nullptr;          // the packed images are the two of bf16x3: bf16x6 splits W itself
  if (!dense || !x || (!W && !packed) || (x6 && !W) || !out || x == out || members > 65535) return GWEN_EINVAL;
  const void *al[] = {dense, x, W, packed, bias, out, workspace};
  for (const void *p : al)
    if (p && !gwen_aligned(p, 16)) return GWEN_EINVAL;
  if (mstride_x % 4 || mstride_o % 4) return GWEN_EINVAL;
  const Shape s = shape_for(N, Fin, Fout);
  if (s.nsplit > 1 && (!workspace || workspace_floats < (int64_t)s.nsplit * members * N * Fout))
    return GWEN_ENOSPACE;
  hipStream_t st = gwen_stream(stream_);
  const dim3 grid((unsigned)(Fout / (16 * s.nc)), (unsigned)members, (unsigned)s.nsplit);
  const dim3 fgrid((unsigned)(Fout / 16), (unsigned)members);   // the finish always 16 columns a block
  const float *wsrc = packed ? reinterpret_cast<const float *>(packed) : W;
#define GWEN_K(NPV, NCV, SP, PK, NSV)                                                                \
  k_small<NPV, NCV, SP, PK, NSV><<<grid, 256, 0, st>>>(dense, x, wsrc, bias, out, workspace, (int)N,  \
                                                       (int)Fin, (int)Fout, s.kchunk, relu, mstride_x, \
                                                       mstride_o)
#define GWEN_F(NPV, NSV)                                                                             \
  k_small_finish<NPV, 1, NSV><<<fgrid, 256, 0, st>>>(dense, workspace, bias, out, (int)N, (int)Fout,  \
                                                     s.nsplit, relu, mstride_o)
#define GWEN_S(NPV, NCV)                                                                             \
  if (pad_nodes(N) == NPV && s.nc == NCV) {                                                          \
    if (s.nsplit > 1) {                                                                              \
      if (x6) GWEN_K(NPV, NCV, true, false, 3);                                                      \
      else if (packed) GWEN_K(NPV, NCV, true, true, 2);                                              \
      else GWEN_K(NPV, NCV, true, false, 2);                                                         \
      if (x6) GWEN_F(NPV, 3); else GWEN_F(NPV, 2);                                                   \
    } else {                                                                                         \
      if (x6) GWEN_K(NPV, NCV, false, false, 3);                                                     \
      else if (packed) GWEN_K(NPV, NCV, false, true, 2);                                             \
      else GWEN_K(NPV, NCV, false, false, 2);                                                        \
    }                                                                                                \
  }
  GWEN_S(128, 1) GWEN_S(128, 4) GWEN_S(256, 1) GWEN_S(256, 2)
#undef GWEN_S
#undef GWEN_F
#undef GWEN_K
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
