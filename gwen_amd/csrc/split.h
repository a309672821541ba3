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

}  // namespace gwen
