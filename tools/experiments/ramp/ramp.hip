// EXPERIMENT: how long does the dispatcher take to start and retire N trivial 256-thread workgroups?
#include <hip/hip_runtime.h>
#include <stdio.h>
extern "C" __global__ __launch_bounds__(256) void k_empty(float *p, int lds_bytes) {
  extern __shared__ float s[];
  if (threadIdx.x == 0 && lds_bytes < 0) p[blockIdx.x] = s[0];
}
int main() {
  float *p; hipMalloc(&p, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grids[] = {256, 782, 1563, 3126, 6252, 25000};
  const int ldss[] = {0, 20480, 65536};
  for (int lds : ldss) {
    hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int g : grids) {
      for (int i = 0; i < 20; ++i) k_empty<<<g, 256, lds>>>(p, lds);
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int i = 0; i < 200; ++i) k_empty<<<g, 256, lds>>>(p, lds);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("lds %6d B  grid %6d x 256 threads: %7.2f us per launch (back to back)\n", lds, g, ms / 200 * 1e3);
    }
  }
  return 0;
}
