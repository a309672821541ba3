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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int SBM = 128, SBN = 128, SBK = 32, SPB = 48;

template <bool VEC>
__device__ inline void stage_split(const float *__restrict__ g, int64_t ld, int64_t row0,
                                   int64_t nrows, int k0, int K, __bf16 *__restrict__ hi,
                                   __bf16 *__restrict__ lo) {
  // [128][SPB] hi/lo <- g[row0 .. +128)[k0 .. k0+32), zero-filled outside
  for (int idx = threadIdx.x; idx < 128 * (SBK / 4); idx += kThreads) {
    const int r = idx / (SBK / 4), kq = (idx % (SBK / 4)) * 4;
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
    bf16x4 h4, l4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __bf16 h = (__bf16)v[i];
      h4[i] = h;
      l4[i] = (__bf16)(v[i] - (float)h);
    }
    *reinterpret_cast<bf16x4 *>(hi + r * SPB + kq) = h4;
    *reinterpret_cast<bf16x4 *>(lo + r * SPB + kq) = l4;
  }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_linear_split(const float *__restrict__ x,
                                                           const float *__restrict__ W,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ h, int64_t rows,
                                                           int Fin, int Fout, int64_t ldx,
                                                           int64_t ldh, int relu) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[4 * 128 * SPB];
  __bf16 *ahi = lds, *alo = lds + 128 * SPB, *bhi = lds + 2 * 128 * SPB, *blo = lds + 3 * 128 * SPB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mi = lane & 15, mh = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * SBM;
  const int col0 = blockIdx.y * SBN;

  f32x4 acc[2][8];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < Fin; k0 += SBK) {
    stage_split<VEC>(x, ldx, row0, rows, k0, Fin, ahi, alo);
    stage_split<VEC>(W, Fin, col0, Fout, k0, Fin, bhi, blo);
    __syncthreads();
    bf16x8 fah[2], fal[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int off = (wave * 32 + rt * 16 + mi) * SPB + 8 * mh;
      fah[rt] = *reinterpret_cast<const bf16x8 *>(ahi + off);
      fal[rt] = *reinterpret_cast<const bf16x8 *>(alo + off);
    }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const int off = (ct * 16 + mi) * SPB + 8 * mh;
      const bf16x8 fbh = *reinterpret_cast<const bf16x8 *>(bhi + off);
      const bf16x8 fbl = *reinterpret_cast<const bf16x8 *>(blo + off);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[rt], fbh, acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[rt], fbl, acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[rt], fbh, acc[rt][ct], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // D[row = 4*mh + t][col = mi] of each 16x16 tile
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) {
    const int c = col0 + ct * 16 + mi;
    if (c >= Fout) continue;
    const float bv = bias ? bias[c] : 0.0f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t r = row0 + wave * 32 + rt * 16 + 4 * mh + t;
        if (r >= rows) continue;
        float v = acc[rt][ct][t] + bv;
        if (relu) v = v < 0.0f ? 0.0f : v;
        h[r * ldh + c] = v;
      }
  }
}

}  // namespace

extern "C" int gwen_gcn_linear_f32(const float *x, const float *W, const float *bias, float *h,
                                   int64_t rows, int64_t Fin, int64_t Fout, int64_t ldx,
                                   int64_t ldh, int relu, int exact, gwen_stream_t stream_) {
  if (rows < 0 || Fin < 0 || Fout < 0 || ldx < Fin || ldh < Fout) return GWEN_EINVAL;
  if (rows == 0 || Fout == 0) return GWEN_OK;
  if (!h || (Fin > 0 && (!x || !W))) return GWEN_EINVAL;
  if (Fin >= (1 << 30) || Fout >= (1 << 30)) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  const bool vec = Fin % 4 == 0 && ldx % 4 == 0 && gwen_aligned(x, 16) && gwen_aligned(W, 16);
  if (!exact) {
    const int64_t sx = (rows + SBM - 1) / SBM, sy = (Fout + SBN - 1) / SBN;
    if (sx > 0x7fffffffLL || sy > 65535) return GWEN_ERANGE;
    dim3 sgrid((unsigned)sx, (unsigned)sy);
    if (vec)
      k_linear_split<true><<<sgrid, kThreads, 0, st>>>(x, W, bias, h, rows, (int)Fin, (int)Fout, ldx, ldh, relu);
    else
      k_linear_split<false><<<sgrid, kThreads, 0, st>>>(x, W, bias, h, rows, (int)Fin, (int)Fout, ldx, ldh, relu);
    GWEN_LAUNCH_CHECK();
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
