// Read-pattern microbenchmark for K8's staging: 256 persistent 8-wave blocks read [tiles*64, 256] fp32
// rows tile by tile in 4 steps of 16 KB, each step waited for before the next (as a barrier-paced DMA):
//   P0: step c reads bytes [256c, 256c+256) of each of the tile's 64 rows (K8's 64-feature chunks)
//   P1: step c reads rows 16c .. 16c+15 whole (16 KB contiguous)
//   DEPTH: how many steps are in flight before the first wait (1 or 2)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int P, int DEPTH>
__global__ __launch_bounds__(512) void k(const float *x, float *sink, int tiles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4_t acc = {0, 0, 0, 0};
  float4_t v[DEPTH][2];
  const int nt = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int steps = nt * 4;
  auto addr = [&](int s, int h) {
    const int t = blockIdx.x + (s >> 2) * gridDim.x, c = s & 3;
    const float *base = x + (size_t)t * 64 * 256;
    // 16 KB per step = 16 wave instructions of 1 KB; wave w issues 2 (h = 0, 1)
    const int q = 2 * wave + h;       // 0..15
    if (P == 0) return base + (size_t)(4 * q + (lane >> 4)) * 256 + c * 64 + (lane & 15) * 4;
    return base + (size_t)(16 * c + q) * 256 + lane * 4;
  };
  for (int s = 0; s < DEPTH && s < steps; ++s) { v[s % DEPTH][0] = *(const float4_t *)addr(s, 0); v[s % DEPTH][1] = *(const float4_t *)addr(s, 1); }
  for (int s = 0; s < steps; ++s) {
    float4_t a = v[s % DEPTH][0], b = v[s % DEPTH][1];
    acc += a + b;
    if (s + DEPTH < steps) { v[s % DEPTH][0] = *(const float4_t *)addr(s + DEPTH, 0); v[s % DEPTH][1] = *(const float4_t *)addr(s + DEPTH, 1); }
    __syncthreads();
  }
  if (acc[0] == 12345.f) sink[threadIdx.x] = acc[1] + acc[2] + acc[3];
}
template <int P, int D> float run(const float *d, float *sink, int tiles) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) k<P, D><<<256, 512>>>(d, sink, tiles);
  (void)hipEventRecord(a);
  for (int r = 0; r < 10; ++r) k<P, D><<<256, 512>>>(d, sink, tiles);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 10 * 1e3f;
}
int main() {
  const int tiles = 4 * 1563;
  float *d, *sink; (void)hipMalloc(&d, (size_t)tiles * 64 * 1024); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(d, 0, (size_t)tiles * 64 * 1024);
  const double mb = tiles * 64.0 * 1024 / 1e6;
  printf("reading %.0f MB per launch, 256 persistent 8-wave blocks, 16 KB per step\n", mb);
  float t;
  t = run<0, 1>(d, sink, tiles); printf("P0 256-B pieces of 64 rows, depth 1: %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<1, 1>(d, sink, tiles); printf("P1 16 whole rows,           depth 1: %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<0, 2>(d, sink, tiles); printf("P0 256-B pieces of 64 rows, depth 2: %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<1, 2>(d, sink, tiles); printf("P1 16 whole rows,           depth 2: %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<0, 4>(d, sink, tiles); printf("P0 256-B pieces of 64 rows, depth 4: %7.1f us  %.2f TB/s\n", t, mb / t);
  t = run<1, 4>(d, sink, tiles); printf("P1 16 whole rows,           depth 4: %7.1f us  %.2f TB/s\n", t, mb / t);
  return 0;
}
