// EXPERIMENT: dense linear at 64 input channels as a ROW-STREAMING kernel -- no LDS staging of x, no barriers in the loop:
// a wave owns 16 rows at a time, reads them straight from global memory in the MFMA operand layout (two 32-byte pieces a
// lane), cuts them into bf16 images in registers, and contracts with W's images from LDS (W: Fout x 64, staged once per
// block) one 16-column tile after the other; persistent blocks.  Against K3 on K8's pipeline (19.8 us for 100 002 x 64 ->
// 64, 2 launches = 43 us for -> 192).
//   hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -I../../../gwen_amd/csrc rows64.hip -o librows64.so
#include "common.h"
#include "split.h"
#include "rows_common.h"

namespace {
using gwen::bf16x4;
using gwen::bf16x8;
constexpr int FIN = 64, KS = 2, PW = 72;          // W image row pitch in bf16 (144 B: 16-byte aligned fragment reads)

template <int NS, int FOUT>
__global__ __launch_bounds__(256) void k_rows64(const float *__restrict__ x, const float *__restrict__ W,
                                                const float *__restrict__ bias, float *__restrict__ h, int64_t rows, int relu) {
  constexpr int NJ = FOUT / 16;
  __shared__ __attribute__((aligned(16))) __bf16 wimg[NS * FOUT * PW];
  __shared__ float bl[FOUT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, mi = lane & 15, mh = lane >> 4;
  // W -> images in LDS: thread t: 4 consecutive k of one output column at a time
  for (int i = threadIdx.x; i < FOUT * (FIN / 4); i += 256) {
    const int c = i / (FIN / 4), kq = (i % (FIN / 4)) * 4;
    const float4_t w4 = *reinterpret_cast<const float4_t *>(W + (int64_t)c * FIN + kq);
    const float f4[4] = {w4[0], w4[1], w4[2], w4[3]};
    bf16x4 im[NS];
    gwen::split_images<4, NS>(f4, im);
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4 *>(wimg + (s * FOUT + c) * PW + kq) = im[s];
  }
  for (int i = threadIdx.x; i < FOUT; i += 256) bl[i] = bias ? bias[i] : 0.0f;
  __syncthreads();
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t r = t * 16 + mi, rc = r < rows ? r : rows - 1;
    bf16x8 a[KS][NS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float *p = x + rc * FIN + 8 * (4 * ks + mh);
      const float4_t v0 = *reinterpret_cast<const float4_t *>(p), v1 = *reinterpret_cast<const float4_t *>(p + 4);
      const float f8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      gwen::split_images<8, NS>(f8, a[ks]);
    }
#pragma unroll 2
    for (int ct = 0; ct < NJ; ++ct) {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 w[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
          w[s] = *reinterpret_cast<const bf16x8 *>(wimg + (s * FOUT + ct * 16 + mi) * PW + 8 * (4 * ks + mh));
        d = gwen::mma_split<8, NS>(w, a[ks], d);
      }
      // W is the A operand: lane (mi, mh) holds row mi, output columns 16 ct + 4 mh .. + 3
      float4_t o = float4_t{d[0], d[1], d[2], d[3]} + *reinterpret_cast<const float4_t *>(bl + ct * 16 + 4 * mh);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.0f ? 0.0f : o[e];
      }
      if (r < rows) *reinterpret_cast<float4_t *>(h + r * FOUT + ct * 16 + 4 * mh) = o;
    }
  }
}
}  // namespace

extern "C" int rows64_launch(const float *x, const float *W, const float *bias, float *h, int64_t rows, int fout, int ns,
                             int relu, int blocks, void *stream) {
  hipStream_t st = (hipStream_t)stream;
#define R64(NS_, FO_) if (ns == NS_ && fout == FO_) { k_rows64<NS_, FO_><<<blocks, 256, 0, st>>>(x, W, bias, h, rows, relu); return (int)hipGetLastError(); }
  R64(2, 64) R64(2, 128) R64(2, 192) R64(3, 64) R64(3, 128) R64(3, 192)
#undef R64
  return -1;
}
