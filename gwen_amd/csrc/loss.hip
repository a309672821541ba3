// Masked L1 loss of the reference's training loop, forward value and gradient in one pass:
//     loss = mean over the rows r with mask[r] (and all members, channels) of |out - target|
// (/root/reference/src/gwen/models_gnn.py:261-265: F.l1_loss(output[mask], target[mask]); backward :372).
// Written with tensor ops -- subtract, abs, mask, two reductions, and their autograd mirror -- it is 14 small
// launches (~190 us of the 0.7 ms c2 training step, most of it launch gaps); here three: the row count, the
// fused |diff| partial sums + gradient, and a fixed-order final reduction (no atomics: bitwise reproducible).
#include "common.h"

namespace {

constexpr int kThreads = 256;

// rows picked by the mask: one block, 16 mask bytes per load (a byte per load and 98 dependent trips cost 22.7 us on
// 100 002 nodes -- 4 % of the c2 training step's kernels), unaligned heads / tails byte by byte; an integer count, exact
// in any order
__device__ inline int nonzero_bytes(uint32_t w) {
  const uint32_t m = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
  return __builtin_popcount(m);
}
__global__ __launch_bounds__(1024) void k_mask_count(const uint8_t *__restrict__ mask, int64_t N,
                                                     float *__restrict__ ws) {
  __shared__ int32_t part[1024];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int64_t head = N < 16 ? N : (int64_t)((16 - (reinterpret_cast<uintptr_t>(mask) & 15)) & 15);   // bytes before alignment
  const int64_t nvec = (N - head) / 16;
  const u32x4 *v = reinterpret_cast<const u32x4 *>(mask + head);
  int32_t c = 0;
  for (int64_t i = threadIdx.x; i < nvec; i += 1024) {
    const u32x4 w = v[i];
    c += nonzero_bytes(w[0]) + nonzero_bytes(w[1]) + nonzero_bytes(w[2]) + nonzero_bytes(w[3]);
  }
  for (int64_t i = threadIdx.x; i < head; i += 1024) c += mask[i] != 0;
  for (int64_t i = head + nvec * 16 + threadIdx.x; i < N; i += 1024) c += mask[i] != 0;
  part[threadIdx.x] = c;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[0] = (float)part[0];
}

// one thread per 4 consecutive channels of a row; rows = members * N; C % 4 == 0
__global__ __launch_bounds__(kThreads) void k_masked_l1(const float *__restrict__ out,
                                                        const float *__restrict__ target,
                                                        const uint8_t *__restrict__ mask, int64_t N,
                                                        int64_t rows, int32_t C4, float scale_c,
                                                        float *__restrict__ grad, float *__restrict__ ws) {
  __shared__ float part[kThreads];
  const float inv = 1.0f / (ws[0] * scale_c);                 // 1 / (picked rows * C * members)
  const int64_t total = rows * C4;
  float acc = 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t r = i / C4;
    const float m = mask[r % N] ? 1.0f : 0.0f;
    const float4_t o = reinterpret_cast<const float4_t *>(out)[i];
    const float4_t t = reinterpret_cast<const float4_t *>(target)[i];
    float4_t g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = o[k] - t[k];
      acc += m * __builtin_fabsf(d);
      g[k] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * m * inv;     // 0 * inf = NaN with an empty mask, as torch
    }
    if (grad) reinterpret_cast<float4_t *>(grad)[i] = g;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[1 + blockIdx.x] = part[0];
}

__global__ __launch_bounds__(kThreads) void k_l1_final(const float *__restrict__ ws, int32_t nblocks, float scale_c,
                                                       float *__restrict__ loss) {
  __shared__ float part[kThreads];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < nblocks; i += kThreads) acc += ws[1 + i];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = part[0] / (ws[0] * scale_c);
}

constexpr int kMaxBlocks = 2048;

}  // namespace

extern "C" int64_t gwen_masked_l1_workspace_floats(void) { return 1 + kMaxBlocks; }

extern "C" int gwen_masked_l1_f32(const float *out, const float *target, const uint8_t *mask, int64_t members,
                                  int64_t N, int64_t C, float *grad, float *loss, float *workspace,
                                  int64_t workspace_floats, gwen_stream_t stream_) {
  if (members < 0 || N < 0 || C < 0 || C % 4 || !loss) return GWEN_EINVAL;
  if (!workspace || workspace_floats < gwen_masked_l1_workspace_floats()) return GWEN_ENOSPACE;
  const int64_t rows = members * N, total4 = rows * (C / 4);
  if ((total4 > 0 && (!out || !target || !mask)) || rows >= (int64_t(1) << 40)) return GWEN_EINVAL;
  if (!gwen_aligned(out, 16) || !gwen_aligned(target, 16) || (grad && !gwen_aligned(grad, 16))) return GWEN_EINVAL;
  hipStream_t st = gwen_stream(stream_);
  k_mask_count<<<1, 1024, 0, st>>>(mask, N, workspace);
  GWEN_LAUNCH_CHECK();
  int64_t blocks = (total4 + kThreads - 1) / kThreads;
  blocks = blocks < 1 ? 1 : (blocks > kMaxBlocks ? kMaxBlocks : blocks);
  const float scale_c = (float)C * (float)members;
  k_masked_l1<<<(unsigned)blocks, kThreads, 0, st>>>(out, target, mask, N, rows, (int32_t)(C / 4), scale_c, grad,
                                                     workspace);
  GWEN_LAUNCH_CHECK();
  k_l1_final<<<1, kThreads, 0, st>>>(workspace, (int32_t)blocks, scale_c, loss);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
