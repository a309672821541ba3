// Split contractions on the bf16 matrix cores: an fp32 operand is cut into NS bf16 images
//     x = im[0] + im[1] (+ im[2]),   im[s] = bf16(x - im[0] - .. - im[s-1])      (each difference is exact in fp32)
// and a product x.w is the sum of the image products im_x[i] . im_w[j] with i + j < NS, accumulated in fp32 from the
// smallest terms to the largest:
//   NS = 2  "bf16x3": 3 MFMAs per k-step, ~17 significant bits per product (dropped: im[1].im[1] and the
//           residuals, each < 2^-17 relative) -- 7e-6 relative on the 6-layer c2 model;
//   NS = 3  "bf16x6": 6 MFMAs per k-step, 24 significant bits per operand (dropped terms < 2^-24 relative): the
//           fp32-class contraction -- the arithmetic of the reference's fp32 `lin`
//           (/root/reference/src/gwen/models_gnn.py:118-130 -> PyG Linear) at 6/16 of the fp32-input MFMA's cost.
// bf16 keeps fp32's exponent range, so the split needs no scaling and has no overflow / underflow cases of its own
// (Inf - Inf in the residual gives NaN where fp32 would give Inf; finite inputs are unaffected).
#pragma once
#include "common.h"

namespace gwen {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int K> struct BFv;
template <> struct BFv<8> { using T = bf16x8; };
template <> struct BFv<4> { using T = bf16x4; };

template <int K, int NS>
__device__ inline void split_images(const float (&x)[K], typename BFv<K>::T (&im)[NS]) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    float r = x[i];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const __bf16 h = (__bf16)r;
      im[s][i] = h;
      if (s + 1 < NS) r = r - (float)h;
    }
  }
}

// d += W-images (the MFMA A operand) x row-images (the B operand): KF = 8 -> 16x16x32, KF = 4 -> 16x16x16.
// Term order for NS = 2: (w0,a1) (w1,a0) (w0,a0) -- the order K4 has always used, bit for bit.
template <int KF, int NS>
__device__ inline f32x4 mma_split(const typename BFv<KF>::T (&w)[NS], const typename BFv<KF>::T (&a)[NS], f32x4 d) {
#pragma unroll
  for (int t = NS - 1; t >= 0; --t)
#pragma unroll
    for (int i = 0; i <= t; ++i) {
      if constexpr (KF == 8) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i], a[t - i], d, 0, 0, 0);
      else d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w[i], a[t - i], d, 0, 0, 0);
    }
  return d;
}

// GWEN_CONTRACT_* (include/gwen_hip.h) -> number of images; 0 = the fp32-input MFMA
constexpr int images_of(int contract) { return contract == 1 ? 0 : (contract == 2 ? 3 : 2); }

// ---- f16x3: the fp32-class contraction on TWO fp16 images per operand (K8, GWEN_CONTRACT_F16X3) ----------------------
// fp16 carries 11 significant bits: hi = f16(x) (round to nearest even), lo = f16(x - hi) represent x to within 2^-24
// relative (x - hi is exact in fp32 and has at most 13 significant bits, of which lo keeps 11), and the one dropped
// product lo.lo is < 2^-24 of hi.hi -- the accuracy of bf16x6 at bf16x3's three MFMAs and two images.  What fp16 lacks is
// exponent range, so both operands are brought to [2^14, 2^15) by POWER-OF-TWO scales (exact): W per output column, once
// per block; the rows per (row, 64-feature chunk), with the accumulators re-expressed in the new chunk's scale before its
// first MFMA (a multiplication by 2^delta: exact) and un-scaled once in the epilogue.  Elements more than 2^17 below
// their row chunk's maximum fall into fp16's subnormal range and keep an ABSOLUTE accuracy of 2^-40 of that maximum.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int kF16Top = 141;       // biased fp32 exponent e of a maximum  ->  scale 2^(141 - e): maximum in [2^14, 2^15)
constexpr int kF16Floor = 20;      // exponents below 2^-107 are treated as 2^-107 (the scale stays a normal fp32 number)
constexpr int kF16Back = 16;       // a chunk's scale is at most 2^16 finer than the chunk before (bounded accumulator growth)
constexpr int kF16Guard = 2;       // a freshly chosen scale puts the maximum in [2^12, 2^13): the next chunks may be 4x larger
constexpr int kF16Keep = 10;       // and up to 2^10 smaller (maximum >= 2^4: six binades of full 22-bit elements) under the same scale

// x * 2^k cut into two fp16 images (K values; the images' bits travel in the bf16 vector types of the kernels)
template <int K>
__device__ inline void split_f16(const float (&x)[K], int k, typename BFv<K>::T &hi, typename BFv<K>::T &lo) {
  typedef _Float16 hv __attribute__((ext_vector_type(K)));
  typedef float fv __attribute__((ext_vector_type(K)));
  fv xs;
#pragma unroll
  for (int i = 0; i < K; ++i) xs[i] = __builtin_ldexpf(x[i], k);
  const hv h = __builtin_convertvector(xs, hv);           // v_cvt_pk_f16_f32: round to nearest even
  // r = xs - hi, exact; one v_fma_mix_f32 per element (the fp16 half is widened inside the fma) instead of a
  // conversion back and a subtraction
  typedef uint32_t uv __attribute__((ext_vector_type(K / 2)));
  const uv hp = __builtin_bit_cast(uv, h);
  fv r;
#pragma unroll
  for (int i = 0; i < K; i += 2) {
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp[i / 2]), "v"(xs[i]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp[i / 2]), "v"(xs[i + 1]));
    r[i] = r0; r[i + 1] = r1;
  }
  const hv l = __builtin_convertvector(r, hv);
  hi = __builtin_bit_cast(typename BFv<K>::T, h);
  lo = __builtin_bit_cast(typename BFv<K>::T, l);
}

// biased exponent of |m| (0 for zero / subnormal), for a non-negative m
__device__ inline int f16_exp_of(float m) { return __builtin_bit_cast(int, m) >> 23; }

// d += W-images x row-images on v_mfma_f32_16x16x32_f16, smallest terms first: (w0,a1) (w1,a0) (w0,a0)
__device__ inline f32x4 mma_split_f16(const bf16x8 (&w)[2], const bf16x8 (&a)[2], f32x4 d) {
  const f16x8 w0 = __builtin_bit_cast(f16x8, w[0]), w1 = __builtin_bit_cast(f16x8, w[1]);
  const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, a1, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, a0, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, a0, d, 0, 0, 0);
  return d;
}

}  // namespace gwen
