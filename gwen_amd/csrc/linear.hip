// K3 -- dense projection h = x @ W^T (+bias, ReLU) on the fp32-input MFMA of gfx950.
//
// Replaces GCNConv.lin (PyG Linear(Fin,Fout,bias=False) -> cuBLAS/rocBLAS SGEMM in the reference:
// /root/reference/src/gwen/models_gnn.py:118-130,:172-184 constructors, called inside every
// conv(x, edge_index) at :147-149,:204-206).
//
// v_mfma_f32_32x32x2_f32: A lane l holds A[i = l&31][k = l>>5], B lane l holds B[k = l>>5][j = l&31],
// D register t of lane l is D[row = (t&3) + 8*(t>>2) + 4*(l>>5)][col = l&31]; the accumulation is an
// exact k-ordered fp32 fmaf chain (no reduced-precision path is used: 1e-4 parity is fp32).
//
// Tile: 256 threads = 4 waves; block = 128 rows x 64 output columns; wave = 32 rows x 64 columns
// (two 32x32 accumulators); K is walked in slabs of 32 staged through LDS with rows padded to 33
// floats so the per-k-step ds_read_b32 of 32 different rows is bank-conflict free.
#include "common.h"
#include "split.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 32, LDT = BK + 1, kThreads = 256;

template <bool VEC>
__device__ inline void stage(const float *__restrict__ g, int64_t ld, int64_t row0, int64_t nrows,
                             int k0, int K, float *__restrict__ tile, int tile_rows) {
  // tile[tile_rows][LDT] <- g[row0 .. row0+tile_rows)[k0 .. k0+BK), zero-filled outside
  for (int idx = threadIdx.x; idx < tile_rows * (BK / 4); idx += kThreads) {
    const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    const int64_t gr = row0 + r;
    if (gr < nrows) {
      const float *p = g + gr * ld + k0 + kq;
      if (VEC && k0 + kq + 3 < K) {
        v = *reinterpret_cast<const float4_t *>(p);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (k0 + kq + i < K) v[i] = p[i];
      }
    }
    float *t = tile + r * LDT + kq;
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
  }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_linear(const float *__restrict__ x,
                                                     const float *__restrict__ W,
                                                     const float *__restrict__ bias,
                                                     float *__restrict__ h, int64_t rows, int Fin,
                                                     int Fout, int64_t ldx, int64_t ldh, int relu) {
  __shared__ float As[BM * LDT];
  __shared__ float Bs[BN * LDT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;

  f32x16 acc0 = {}, acc1 = {};
  for (int k0 = 0; k0 < Fin; k0 += BK) {
    stage<VEC>(x, ldx, row0, rows, k0, Fin, As, BM);
    stage<VEC>(W, Fin, col0, Fout, k0, Fin, Bs, BN);
    __syncthreads();
    const float *a = As + (wave * 32 + li) * LDT + lh;
    const float *b0 = Bs + li * LDT + lh;
    const float *b1 = Bs + (32 + li) * LDT + lh;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float av = a[kk];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[kk], acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: register t -> row (t&3) + 8*(t>>2) + 4*lh of the wave's 32 rows, column li
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int c = col0 + half * 32 + li;
    if (c >= Fout) continue;
    const float bv = bias ? bias[c] : 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int64_t r = row0 + wave * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
      if (r >= rows) continue;
      float v = half == 0 ? acc0[t] : acc1[t];
      if (bias) v = v + bv;
      if (relu) v = v < 0.0f ? 0.0f : v;
      h[r * ldh + c] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3xbf16 variant (default): x = hi + lo, W = hi' + lo' in bf16, x.w ~= lo.hi' + hi.lo' + hi.hi' on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (same scheme and error as K4, layer.hip): the exact
// fp32 MFMA runs at 1/16 of the bf16 rate and made this kernel MFMA-bound from Fin = 128 up.
// Tile 128 x 128, K slabs of 32; operands are split once when they are staged into LDS (hi and lo
// images, row pitch 48 bf16 => conflict-free 16-B fragment reads); wave = 32 rows x 128 columns.
// ---------------------------------------------------------------------------------------------
using gwen::bf16x4;
using gwen::bf16x8;
constexpr int SBM = 128, SBN = 128, SBK = 32, SPB = 48;

// One thread's share of a 128-row x 32-deep fp32 slab: 4 x 16 B, rows idx/8, k (idx%8)*4.
template <bool VEC>
__device__ inline void slab_load(const float *__restrict__ g, int64_t ld, int64_t row0,
                                 int64_t nrows, int k0, int K, float4_t (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int r = idx >> 3, kq = (idx & 7) * 4;
    const int64_t gr = row0 + r;
    if constexpr (VEC) {
      // unconditional 16-B load from a clamped (always valid) address, zeroed by select: a load under
      // a per-lane condition would be branched around and followed by vmcnt(0) (K % 4 == 0 here, so a
      // quad is inside or outside K as a whole)
      const bool ok = gr < nrows && k0 + kq < K;
      const int64_t cr = gr < nrows ? gr : nrows - 1;
      const int ck = k0 + kq < K ? k0 + kq : 0;
      const float4_t t = *reinterpret_cast<const float4_t *>(g + cr * ld + ck);
      v[i] = ok ? t : float4_t{0.f, 0.f, 0.f, 0.f};
    } else {
      v[i] = float4_t{0.f, 0.f, 0.f, 0.f};
      if (gr < nrows) {
        const float *p = g + gr * ld + k0 + kq;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k0 + kq + e < K) v[i][e] = p[e];
      }
    }
  }
}

// split the slab share into NS bf16 images and park them in the LDS images NS x [128][SPB]
template <int NS>
__device__ inline void slab_store(const float4_t (&v)[4], __bf16 *__restrict__ img) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int r = idx >> 3, kq = (idx & 7) * 4;
    const float f4[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
    bf16x4 im[NS];
    gwen::split_images<4, NS>(f4, im);
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4 *>(img + s * 128 * SPB + r * SPB + kq) = im[s];
  }
}

// WT ("NN" products: the weight operand given as Wt [K, N] row-major -- the backward's gx = gh W with W as stored, no
// transposing copy): a 32-deep x 128-column slab is read along N (consecutive lanes = consecutive columns, one dword per
// k: coalesced) and parked as bf16x4 K-QUADS, image layout [8 quads][128 columns] -- writes and the two 8-byte fragment
// reads of a lane (quads 2 mh, 2 mh + 1 of its column) both run over consecutive columns: conflict-free without padding.
// One thread's share: 4 (quad, column) items, 4 dwords each.
__device__ inline void slab_load_t(const float *__restrict__ wt, int64_t ld, int k0, int K, int col0, int ncols,
                                   float4_t (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int n = idx & 127, kq = idx >> 7;
    const int c = col0 + n;
    const int cc = c < ncols ? c : ncols - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + 4 * kq + e;
      const float t = wt[(int64_t)(k < K ? k : K - 1) * ld + cc];            // unconditional, clamped, zeroed by select
      v[i][e] = (k < K && c < ncols) ? t : 0.0f;
    }
  }
}
template <int NS>
__device__ inline void slab_store_t(const float4_t (&v)[4], __bf16 *__restrict__ img) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int n = idx & 127, kq = idx >> 7;
    const float f4[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
    bf16x4 im[NS];
    gwen::split_images<4, NS>(f4, im);
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4 *>(img + s * 128 * SPB + (kq * 128 + n) * 4) = im[s];
  }
}

// ... and with 16-byte loads (Fout % 4 == 0, Wt 16-byte aligned): a thread owns ONE k-quad of four consecutive columns --
// four float4 loads (32 lanes = 512 contiguous bytes of a row of Wt: the dword form above is one 256-byte instruction per
// k and was instruction-bound, 2.7 x slower than the transposed product it replaces), turned 4 x 4 in registers, parked as
// 32 contiguous bytes of the quad's row.
__device__ inline void slab_load_tv(const float *__restrict__ wt, int64_t ld, int k0, int K, int col0, int ncols,
                                    float4_t (&v)[4]) {
  const int kq = threadIdx.x >> 5, n4 = (threadIdx.x & 31) * 4;
  const int c = col0 + n4;
  const int cc = c < ncols ? c : 0;                     // (ncols % 4 == 0: a quad of columns is inside or outside as a whole)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k0 + 4 * kq + e;
    const float4_t t = *reinterpret_cast<const float4_t *>(wt + (int64_t)(k < K ? k : K - 1) * ld + cc);
    v[e] = (k < K && c < ncols) ? t : float4_t{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NS>
__device__ inline void slab_store_tv(const float4_t (&v)[4], __bf16 *__restrict__ img) {
  const int kq = threadIdx.x >> 5, n4 = (threadIdx.x & 31) * 4;
  bf16x4 im[4][NS];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float f4[4] = {v[0][j], v[1][j], v[2][j], v[3][j]};
    gwen::split_images<4, NS>(f4, im[j]);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    __bf16 *dst = img + s * 128 * SPB + (kq * 128 + n4) * 4;
    *reinterpret_cast<bf16x8 *>(dst) = bf16x8{im[0][s][0], im[0][s][1], im[0][s][2], im[0][s][3],
                                              im[1][s][0], im[1][s][1], im[1][s][2], im[1][s][3]};
    *reinterpret_cast<bf16x8 *>(dst + 8) = bf16x8{im[2][s][0], im[2][s][1], im[2][s][2], im[2][s][3],
                                                  im[3][s][0], im[3][s][1], im[3][s][2], im[3][s][3]};
  }
}

// NCT = 16-column tiles per block (8: 128 columns; 4: 64 columns for narrow outputs).
// The next slab's global loads are issued before the current slab's MFMAs and parked in LDS after
// them, so their latency hides under the matrix work instead of standing in front of it.
template <bool VEC, int NCT, int NS, int WT = 0>      // WT: 0 W [Fout, Fin]; 1 Wt [Fin, Fout], dword loads; 2 ..., 16-byte loads
__global__ __launch_bounds__(kThreads) void k_linear_split(const float *__restrict__ x,
                                                           const float *__restrict__ W,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ h, int64_t rows,
                                                           int Fin, int Fout, int64_t ldx,
                                                           int64_t ldh, int relu, int kchunk) {
  // split-K: blockIdx.z contracts k in [z * kchunk, min(Fin, (z+1) * kchunk)) and writes its RAW partial
  // product to slab z of h (h is then the workspace, ldh = Fout; bias/ReLU happen in the reduction)
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * NS * 128 * SPB];
  constexpr int kImg = 128 * SPB;
  __bf16 *aimg = lds, *bimg = lds + NS * kImg;            // NS images of the x slab, NS of the W slab
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mi = lane & 15, mh = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * SBM;
  const int col0 = blockIdx.y * (NCT * 16);

  f32x4 acc[2][NCT];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int k_lo = blockIdx.z * kchunk;
  const int k_hi = k_lo + kchunk < Fin ? k_lo + kchunk : Fin;
  const bool partial = gridDim.z > 1;
  if (partial) h += (int64_t)blockIdx.z * rows * Fout;
  float4_t pa[4], pb[4];
  const int ncol_end = col0 + NCT * 16 < Fout ? col0 + NCT * 16 : Fout;
  slab_load<VEC>(x, ldx, row0, rows, k_lo, k_hi, pa);
  if constexpr (WT == 2) slab_load_tv(W, Fout, k_lo, k_hi, col0, ncol_end, pb);
  else if constexpr (WT == 1) slab_load_t(W, Fout, k_lo, k_hi, col0, ncol_end, pb);
  else slab_load<VEC>(W, Fin, col0, (int64_t)ncol_end, k_lo, k_hi, pb);
  slab_store<NS>(pa, aimg);
  if constexpr (WT == 2) slab_store_tv<NS>(pb, bimg);
  else if constexpr (WT == 1) slab_store_t<NS>(pb, bimg);
  else slab_store<NS>(pb, bimg);
  __syncthreads();
  for (int k0 = k_lo; k0 < k_hi; k0 += SBK) {
    const bool more = k0 + SBK < k_hi;
    // issued unconditionally (past the last slab the loads are clamped and their result unused)
    slab_load<VEC>(x, ldx, row0, rows, k0 + SBK, k_hi, pa);
    if constexpr (WT == 2) slab_load_tv(W, Fout, k0 + SBK, k_hi, col0, ncol_end, pb);
    else if constexpr (WT == 1) slab_load_t(W, Fout, k0 + SBK, k_hi, col0, ncol_end, pb);
    else slab_load<VEC>(W, Fin, col0, (int64_t)ncol_end, k0 + SBK, k_hi, pb);
    bf16x8 fa[2][NS];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int off = (wave * 32 + rt * 16 + mi) * SPB + 8 * mh;
#pragma unroll
      for (int s = 0; s < NS; ++s) fa[rt][s] = *reinterpret_cast<const bf16x8 *>(aimg + s * kImg + off);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int off = (ct * 16 + mi) * SPB + 8 * mh;
      bf16x8 fb[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if constexpr (WT != 0) {
          const bf16x4 q0 = *reinterpret_cast<const bf16x4 *>(bimg + s * kImg + ((2 * mh) * 128 + ct * 16 + mi) * 4);
          const bf16x4 q1 = *reinterpret_cast<const bf16x4 *>(bimg + s * kImg + ((2 * mh + 1) * 128 + ct * 16 + mi) * 4);
          fb[s] = bf16x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
        } else {
          fb[s] = *reinterpret_cast<const bf16x8 *>(bimg + s * kImg + off);
        }
      }
      // terms (x image t - i) . (W image i), from the smallest to hi . hi (split.h's order with x as the A operand)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = NS - 1; t >= 0; --t)
#pragma unroll
          for (int i = 0; i <= t; ++i)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[rt][t - i], fb[i], acc[rt][ct], 0, 0, 0);
    }
    __syncthreads();                       // every wave is done reading this slab
    if (more) {
      slab_store<NS>(pa, aimg);
      if constexpr (WT == 2) slab_store_tv<NS>(pb, bimg);
      else if constexpr (WT == 1) slab_store_t<NS>(pb, bimg);
      else slab_store<NS>(pb, bimg);
      __syncthreads();
    }
  }

  // D[row = 4*mh + t][col = mi] of each 16x16 tile
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int c = col0 + ct * 16 + mi;
    if (c >= Fout) continue;
    const float bv = (bias && !partial) ? bias[c] : 0.0f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t r = row0 + wave * 32 + rt * 16 + 4 * mh + t;
        if (r >= rows) continue;
        float v = acc[rt][ct][t] + bv;
        if (relu && !partial) v = v < 0.0f ? 0.0f : v;
        h[r * ldh + c] = v;
      }
  }
}

// out[r][c] = act(sum_z partial[z][r][c] + bias[c]), slabs added in ascending z (deterministic)
__global__ void k_splitk_reduce(const float *__restrict__ partial, const float *__restrict__ bias,
                                float *__restrict__ out, int64_t rows, int Fout, int64_t ldh,
                                int nsplit, int relu) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows * Fout) return;
  const int64_t r = i / Fout;
  const int c = (int)(i - r * Fout);
  float s = 0.0f;
  for (int z = 0; z < nsplit; ++z) s = s + partial[(int64_t)z * rows * Fout + i];
  if (bias) s = s + bias[c];
  if (relu) s = s < 0.0f ? 0.0f : s;
  out[r * ldh + c] = s;
}

// split-K pays when the output tiles alone cannot fill the chip and K is long: few rows x wide input
// (the reference's own C -> 1024 projection on ~125 member-nodes)
inline int splitk_factor(int64_t rows, int64_t Fin, int64_t Fout) {
  const int64_t bn = Fout <= 64 ? 64 : SBN;
  const int64_t blocks = ((rows + SBM - 1) / SBM) * ((Fout + bn - 1) / bn);
  if (blocks >= 256 || Fin < 1024) return 1;
  int64_t n = (512 + blocks - 1) / blocks;           // aim at ~512 blocks
  const int64_t maxn = Fin / 256;                      // at least 256 of K per block
  if (n > maxn) n = maxn;
  return n < 2 ? 1 : (int)n;
}

}  // namespace

// wide.hip: the K8 pipeline without a graph (tall inputs, Fin and Fout in {64, 128, 256}; bf16x6 up to Fin = 128)
int gwen_wide_dense_f32(const float *x, const float *W, const float *bias, float *h, int64_t rows, int64_t Fin,
                        int64_t Fout, int64_t ldx, int64_t ldh, int relu, int contract, hipStream_t st);

extern "C" int64_t gwen_gcn_linear_workspace_floats(int64_t rows, int64_t Fin, int64_t Fout) {
  if (rows <= 0 || Fin <= 0 || Fout <= 0) return 0;
  const int n = splitk_factor(rows, Fin, Fout);
  return n > 1 ? (int64_t)n * rows * Fout : 0;
}

extern "C" int gwen_gcn_linear_f32(const float *x, const float *W, const float *bias, float *h,
                                   int64_t rows, int64_t Fin, int64_t Fout, int64_t ldx,
                                   int64_t ldh, int relu, int exact, float *workspace,
                                   int64_t workspace_floats, gwen_stream_t stream_) {
  if (rows < 0 || Fin < 0 || Fout < 0 || ldx < Fin || ldh < Fout || exact < 0 || exact > 2) return GWEN_EINVAL;
  if (rows == 0 || Fout == 0) return GWEN_OK;
  if (!h || (Fin > 0 && (!x || !W))) return GWEN_EINVAL;
  const bool x6 = exact == GWEN_CONTRACT_BF16X6;
  exact = exact == GWEN_CONTRACT_F32;
  if (Fin >= (1 << 30) || Fout >= (1 << 30)) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  const bool vec = Fin % 4 == 0 && ldx % 4 == 0 && gwen_aligned(x, 16) && gwen_aligned(W, 16);
  // Tall inputs at the widths K8 is built for run on its pipeline (rows DMA-staged through LDS, W resident in
  // registers, persistent blocks): output columns in groups of 256 / 128 / 64, each group one pass over x.
  // (measured, tools/experiments/k3_time.py: 256 -> 256 on 100 002 rows 76 -> 58 us, on 400 008 rows 338 -> 231 us;
  // equal from ~16 000 rows down, where the 128 x 128 tiles of k_linear_split fill the chip as well)
  const bool tall = rows >= 16384;
  if (!exact && tall && vec && (Fin == 64 || Fin == 128 || (Fin == 256 && !x6)) && Fout % 64 == 0 && Fout <= 2048 &&
      ldh % 4 == 0 && gwen_aligned(h, 16) && (!bias || gwen_aligned(bias, 16)) &&
      rows * ldx * 4 < (int64_t(1) << 32) && x != h) {
    for (int64_t c0 = 0; c0 < Fout;) {
      const int64_t left = Fout - c0, g = left >= 256 ? 256 : left >= 128 ? 128 : 64;
      const int rc = gwen_wide_dense_f32(x, W + c0 * Fin, bias ? bias + c0 : nullptr, h + c0, rows, Fin, g, ldx,
                                         ldh, relu, x6 ? GWEN_CONTRACT_BF16X6 : GWEN_CONTRACT_BF16X3, st);
      if (rc != GWEN_OK) return rc;
      c0 += g;
    }
    return GWEN_OK;
  }
  if (!exact && Fin > 0) {      // Fin == 0: nothing to contract, the exact kernel writes act(bias)
    const int bn = Fout <= 64 ? 64 : SBN;           // narrow outputs: 64-column blocks, no empty tiles
    const int64_t sx = (rows + SBM - 1) / SBM, sy = (Fout + bn - 1) / bn;
    if (sx > 0x7fffffffLL || sy > 65535) return GWEN_ERANGE;
    int nsplit = splitk_factor(rows, Fin, Fout);
    if (nsplit > 1 && (!workspace || workspace_floats < (int64_t)nsplit * rows * Fout)) nsplit = 1;
    const int kchunk = nsplit > 1 ? (int)(((Fin + nsplit - 1) / nsplit + SBK - 1) / SBK * SBK) : (int)Fin;
    const int nz = (int)((Fin + kchunk - 1) / kchunk);
    dim3 sgrid((unsigned)sx, (unsigned)sy, (unsigned)nz);
    float *dst = nz > 1 ? workspace : h;
    const int64_t dld = nz > 1 ? Fout : ldh;
#define GWEN_LS(V, T)                                                                               \
  do {                                                                                              \
    if (x6)                                                                                         \
      k_linear_split<V, T, 3><<<sgrid, kThreads, 0, st>>>(x, W, bias, dst, rows, (int)Fin, (int)Fout, ldx, \
                                                          dld, relu, kchunk);                       \
    else                                                                                            \
      k_linear_split<V, T, 2><<<sgrid, kThreads, 0, st>>>(x, W, bias, dst, rows, (int)Fin, (int)Fout, ldx, \
                                                          dld, relu, kchunk);                       \
  } while (0)
    if (vec) { if (bn == 64) GWEN_LS(true, 4); else GWEN_LS(true, 8); }
    else     { if (bn == 64) GWEN_LS(false, 4); else GWEN_LS(false, 8); }
#undef GWEN_LS
    GWEN_LAUNCH_CHECK();
    if (nz > 1) {
      const int64_t count = rows * Fout;
      k_splitk_reduce<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(workspace, bias, h, rows,
                                                                       (int)Fout, ldh, nz, relu);
      GWEN_LAUNCH_CHECK();
    }
    return GWEN_OK;
  }
  const int64_t gx = (rows + BM - 1) / BM, gy = (Fout + BN - 1) / BN;
  if (gx > 0x7fffffffLL || gy > 65535) return GWEN_ERANGE;
  dim3 grid((unsigned)gx, (unsigned)gy);
  if (vec)
    k_linear<true><<<grid, kThreads, 0, st>>>(x, W, bias, h, rows, (int)Fin, (int)Fout, ldx, ldh, relu);
  else
    k_linear<false><<<grid, kThreads, 0, st>>>(x, W, bias, h, rows, (int)Fin, (int)Fout, ldx, ldh, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

// h = act(x Wt + bias) with the weight operand given as Wt [Fin, Fout] row-major ("NN"): the backward's gx = gh W with W
// as the layer stores it ([out, in] = [K, N] of this product) -- no transposing copy (at the reference's own shape,
// 16 384 x 1024 weights, torch's transposes were 88 us of a 1.2 ms training step) -- and out = D g for a dense
// adjacency.  Split contractions only (bf16x3 / bf16x6; f16x3 = bf16x6 here); split-K as gwen_gcn_linear_f32.
extern "C" int gwen_gcn_linear_nn_f32(const float *x, const float *Wt, const float *bias, float *h, int64_t rows,
                                      int64_t Fin, int64_t Fout, int64_t ldx, int64_t ldw, int64_t ldh, int relu,
                                      int contract, float *workspace, int64_t workspace_floats, gwen_stream_t stream_) {
  if (rows < 0 || Fin <= 0 || Fout < 0 || ldx < Fin || ldw < Fout || ldh < Fout) return GWEN_EINVAL;
  if (contract != GWEN_CONTRACT_BF16X3 && contract != GWEN_CONTRACT_BF16X6 && contract != GWEN_CONTRACT_F16X3) return GWEN_EINVAL;
  if (rows == 0 || Fout == 0) return GWEN_OK;
  if (!h || !x || !Wt || x == h) return GWEN_EINVAL;
  if (Fin >= (1 << 30) || Fout >= (1 << 30)) return GWEN_ERANGE;
  const bool x6 = contract != GWEN_CONTRACT_BF16X3;
  hipStream_t st = gwen_stream(stream_);
  const bool vec = Fin % 4 == 0 && ldx % 4 == 0 && gwen_aligned(x, 16);
  const int bn = Fout <= 64 ? 64 : SBN;
  const int64_t sx = (rows + SBM - 1) / SBM, sy = (Fout + bn - 1) / bn;
  if (sx > 0x7fffffffLL || sy > 65535) return GWEN_ERANGE;
  int nsplit = splitk_factor(rows, Fin, Fout);
  if (nsplit > 1 && (!workspace || workspace_floats < (int64_t)nsplit * rows * Fout)) nsplit = 1;
  const int kchunk = nsplit > 1 ? (int)(((Fin + nsplit - 1) / nsplit + SBK - 1) / SBK * SBK) : (int)Fin;
  const int nz = (int)((Fin + kchunk - 1) / kchunk);
  dim3 sgrid((unsigned)sx, (unsigned)sy, (unsigned)nz);
  float *dst = nz > 1 ? workspace : h;
  const int64_t dld = nz > 1 ? Fout : ldh;
  // (the kernel reads Wt with the row pitch it is given as "Fout": pass ldw through that argument when they differ)
  if (ldw != Fout) return GWEN_EINVAL;
  const bool wvec = Fout % 4 == 0 && gwen_aligned(Wt, 16);
#define GWEN_LT(V, T, WV)                                                                                      \
  do {                                                                                                         \
    if (x6)                                                                                                    \
      k_linear_split<V, T, 3, WV><<<sgrid, kThreads, 0, st>>>(x, Wt, bias, dst, rows, (int)Fin, (int)Fout, ldx, dld, \
                                                              relu, kchunk);                                   \
    else                                                                                                       \
      k_linear_split<V, T, 2, WV><<<sgrid, kThreads, 0, st>>>(x, Wt, bias, dst, rows, (int)Fin, (int)Fout, ldx, dld, \
                                                              relu, kchunk);                                   \
  } while (0)
#define GWEN_LT2(V, T) do { if (wvec) GWEN_LT(V, T, 2); else GWEN_LT(V, T, 1); } while (0)
  if (vec) { if (bn == 64) GWEN_LT2(true, 4); else GWEN_LT2(true, 8); }
  else     { if (bn == 64) GWEN_LT2(false, 4); else GWEN_LT2(false, 8); }
#undef GWEN_LT2
#undef GWEN_LT
  GWEN_LAUNCH_CHECK();
  if (nz > 1) {
    const int64_t count = rows * Fout;
    k_splitk_reduce<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(workspace, bias, h, rows, (int)Fout, ldh, nz, relu);
    GWEN_LAUNCH_CHECK();
  }
  return GWEN_OK;
}
