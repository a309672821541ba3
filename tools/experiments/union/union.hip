// EXPERIMENT (not part of libgwen_hip.so): propagate at 64 channels on a uniform grouped layout with the
// block's DISTINCT source rows staged once in LDS (64 own rows + a halo list), gathers served from LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float float4_t __attribute__((ext_vector_type(4)));
constexpr int F = 64, HC = 104, NR = 64 + HC, PX = F + 4;

extern "C" __global__ __launch_bounds__(256) void k_union(const float *__restrict__ x,
                                                          const float *__restrict__ val,     // [8N]
                                                          const uint8_t *__restrict__ lid,   // [8N]
                                                          const int32_t *__restrict__ halo,  // [nb][HC]
                                                          float *__restrict__ out, int N) {
  __shared__ __attribute__((aligned(16))) float xs[NR * PX];
  const int t = threadIdx.x, b = blockIdx.x, r0 = b * 64;
  const int q = t & 15, rr = t >> 4;                      // 16 lanes per row, 16 rows per pass
  // first trip: halo ids, weights and local ids of this thread's 4 rows
  int32_t hid[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int h = rr + 16 * i;                            // halo slot 0..111 (HC = 104)
    hid[i] = halo[(int64_t)b * HC + (h < HC ? h : HC - 1)];
  }
  float4_t w[4][2];
  uint2 ids[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int r = r0 + 16 * p + rr;
    r = r < N ? r : N - 1;
    w[p][0] = *reinterpret_cast<const float4_t *>(val + 8 * (int64_t)r);
    w[p][1] = *reinterpret_cast<const float4_t *>(val + 8 * (int64_t)r + 4);
    ids[p] = *reinterpret_cast<const uint2 *>(lid + 8 * (int64_t)r);
  }
  // own rows
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int r = r0 + 16 * p + rr;
    r = r < N ? r : N - 1;
    *reinterpret_cast<float4_t *>(xs + (16 * p + rr) * PX + 4 * q) =
        *reinterpret_cast<const float4_t *>(x + (int64_t)r * F + 4 * q);
  }
  // halo rows (second trip)
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int h = rr + 16 * i;
    if (h < HC)
      *reinterpret_cast<float4_t *>(xs + (64 + h) * PX + 4 * q) =
          *reinterpret_cast<const float4_t *>(x + (int64_t)hid[i] * F + 4 * q);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + 16 * p + rr;
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t word = u < 4 ? ids[p].x : ids[p].y;
      const int l = (word >> (8 * (u & 3))) & 255;
      const float wv = u < 4 ? w[p][0][u & 3] : w[p][1][u & 3];
      const float4_t v = *reinterpret_cast<const float4_t *>(xs + l * PX + 4 * q);
      acc += float4_t{wv, wv, wv, wv} * v;
    }
    if (r < N) *reinterpret_cast<float4_t *>(out + (int64_t)r * F + 4 * q) = acc;
  }
}

extern "C" int union_launch(const float *x, const float *val, const uint8_t *lid, const int32_t *halo,
                            float *out, int N, void *stream) {
  k_union<<<(N + 63) / 64, 256, 0, (hipStream_t)stream>>>(x, val, lid, halo, out, N);
  return (int)hipGetLastError();
}
