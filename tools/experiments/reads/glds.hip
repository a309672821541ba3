// LDS-DMA issue microbenchmark: 256 persistent 8-wave blocks stream a [rows, 256] fp32 buffer through LDS by
// global_load_lds_dwordx4 (1 KiB per wave instruction).  Per step every wave issues K DMAs (K = 2, 4, 8), then
// waits vmcnt(0) + barrier (depth 1) or leaves one step in flight (depth 2).  Reports TB/s and the cycles a
// wave spends ISSUING its K DMAs (s_memtime around the issue only).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline void glds16(const void *base, uint32_t voff, uint32_t dst) {
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}
template <int K, int DEPTH>
__global__ __launch_bounds__(512) void k(const float *x, unsigned long long *issue_cycles, int steps_total) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * 8 * 8 * 1024];
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int per_block = steps_total / gridDim.x;
  unsigned long long acc = 0;
  const char *base = reinterpret_cast<const char *>(x);
  for (int s = 0; s < per_block; ++s) {
    const size_t step_off = ((size_t)blockIdx.x * per_block + s) * (size_t)(8 * K * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const char *b = base + step_off + (size_t)(wave * K + q) * 1024;
      const uint64_t a = reinterpret_cast<uint64_t>(b);
      const char *bu = reinterpret_cast<const char *>(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                                                      __builtin_amdgcn_readfirstlane((uint32_t)a));
      glds16(bu, lane * 16, lds0 + ((s & 1) * 8 * K + wave * K + q) * 1024);
    }
    acc += __builtin_amdgcn_s_memtime() - t0;
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) issue_cycles[blockIdx.x * 8 + wave] = acc;
}
template <int K, int D> void run(const float *d, unsigned long long *ic, size_t bytes) {
  const int steps_total = (int)(bytes / (8 * K * 1024)) / 256 * 256;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) k<K, D><<<256, 512>>>(d, ic, steps_total);
  (void)hipEventRecord(a);
  for (int r = 0; r < 5; ++r) k<K, D><<<256, 512>>>(d, ic, steps_total);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2048]; (void)hipMemcpy(h, ic, sizeof(h), hipMemcpyDeviceToHost);
  double tot = 0; for (int i = 0; i < 2048; ++i) tot += (double)h[i];
  const double us = ms / 5 * 1e3, mb = (double)steps_total * 8 * K * 1024 / 1e6;
  printf("K=%d DMAs per wave per step (%3d KB per CU step), depth %d: %7.1f us  %.2f TB/s   issue %.0f cycles per step per wave (%.0f per DMA)\n",
         K, 8 * K, D, us, mb / us, tot / 2048 / (steps_total / 256), tot / 2048 / (steps_total / 256) / K);
}
int main() {
  const size_t bytes = (size_t)410 << 20;
  float *d; unsigned long long *ic; (void)hipMalloc(&d, bytes); (void)hipMalloc(&ic, 2048 * 8); (void)hipMemset(d, 0, bytes);
  run<2, 1>(d, ic, bytes); run<4, 1>(d, ic, bytes); run<8, 1>(d, ic, bytes);
  run<2, 2>(d, ic, bytes); run<4, 2>(d, ic, bytes); run<8, 2>(d, ic, bytes);
  return 0;
}
