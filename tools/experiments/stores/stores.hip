// Store-pattern microbenchmark: 256 persistent blocks write [M*N, 256] fp32 rows tile by tile (64 rows), in
// the shapes K8's epilogue could use.  Pattern 0: wave instr = 16 rows x 64 B (K8/K4 today);
// 1: 8 rows x 128 B; 2: 4 rows x 256 B; 3: 1 row x 1 KB (K2).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void k(float *out, int tiles, int burst) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4_t v = {1.f, 2.f, 3.f, (float)lane};
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    float *base = out + (size_t)t * 64 * 256;
    // 64 rows x 1 KB = 64 wave instructions per tile, 64 / NW per wave
    for (int i = 0; i < 64 / NW; ++i) {
      const int q = i * NW + wave;          // which of the 64 wave instructions
      size_t off;
      if (P == 0) { const int ct = q & 15, rt = q >> 4; off = (size_t)(rt * 16 + (lane & 15)) * 256 + ct * 16 + (lane >> 4) * 4; }
      else if (P == 1) { const int ct = q & 7, rt = q >> 3; off = (size_t)(rt * 8 + (lane & 7)) * 256 + ct * 32 + (lane >> 3) * 4; }
      else if (P == 2) { const int ct = q & 3, rt = q >> 2; off = (size_t)(rt * 4 + (lane & 3)) * 256 + ct * 64 + (lane >> 2) * 4; }
      else { off = (size_t)q * 256 + lane * 4; }
      *reinterpret_cast<float4_t *>(base + off) = v;
    }
    if (burst) __syncthreads();
  }
}
template <int P, int NW> float run(float *d, int tiles) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) k<P, NW><<<256, NW * 64>>>(d, tiles, 1);
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) k<P, NW><<<256, NW * 64>>>(d, tiles, 1);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10 * 1e3f;
}
int main() {
  const int tiles = 4 * 1563;
  float *d; hipMalloc(&d, (size_t)tiles * 64 * 1024);
  const double mb = tiles * 64.0 * 1024 / 1e6;
  printf("writing %.0f MB per launch, 256 persistent blocks\n", mb);
  float t;
  t = run<0, 8>(d, tiles);  printf("P0 16 rows x 64 B  NW=8 : %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<1, 8>(d, tiles);  printf("P1  8 rows x 128 B NW=8 : %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<2, 8>(d, tiles);  printf("P2  4 rows x 256 B NW=8 : %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<3, 8>(d, tiles);  printf("P3  1 row  x 1 KB  NW=8 : %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<0, 16>(d, tiles); printf("P0 16 rows x 64 B  NW=16: %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<3, 16>(d, tiles); printf("P3  1 row  x 1 KB  NW=16: %7.1f us  %.2f TB/s\n", t, mb / t);
  return 0;
}
