// How long does ONE wave take per global store instruction, and what does it scale with?  (K8's stamps: ~400 cycles per
// dwordx4 store per wave, whatever else runs.)  Blocks of NW waves, one per CU; every wave issues back-to-back stores
// (inline asm: nothing for hipcc to merge or hoist) to its own rows of a per-block window (cache-resident) or to a stream.
//   hipcc -O3 -w --offload-arch=gfx950 stwave.hip -o stwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float float4_t __attribute__((ext_vector_type(4)));
// MODE 0: dwordx4, all lanes; 1: dwordx4, lanes 0-31; 2: dword, all lanes; 3: dwordx4 streaming (never rewrites a line);
// 4: dwordx4 with s_waitcnt vmcnt(0) after each (round trip); 5: dwordx2; 6: dwordx4, the 64 lanes on ONE contiguous KiB
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(float *out, int reps, size_t stream_stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4_t v = {1.f, 2.f, 3.f, (float)lane};
  float *base = out + (size_t)blockIdx.x * 64 * 256;
  if (MODE == 1 && lane >= 32) return;
  for (int t = 0; t < reps; ++t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = (i * NW + wave) & 63;
      const int ct = q & 15, rt = q >> 4;
      float *p = base + (size_t)(rt * 16 + (lane & 15)) * 256 + ct * 16 + (lane >> 4) * 4;
      if (MODE == 6) p = base + (size_t)q * 256 + lane * 4;
      if (MODE == 3) p += (size_t)t * stream_stride;
      if (MODE == 2) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v[0]) : "memory");
      else if (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(__builtin_shufflevector(v, v, 0, 1)) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
      if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}
template <int MODE, int NW> void run(const char *name, float *d, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const size_t stride = (size_t)256 * 64 * 256;     // floats: one window per block per repetition
  k<MODE, NW><<<256, NW * 64>>>(d, 4, stride);
  hipEventRecord(a);
  k<MODE, NW><<<256, NW * 64>>>(d, reps, stride);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ns = ms * 1e6 / (reps * 8.0);
  printf("%-44s %2d waves: %7.1f ns per store instruction per wave (%5.0f clk at 2.1 GHz), %6.1f ns per instr per CU\n", name, NW, ns, ns * 2.1, ns / NW);
}
int main() {
  const int reps = 400;
  float *d; hipMalloc(&d, (size_t)256 * 64 * 1024 * (reps + 1));      // 6.7 GB for the streaming mode
  run<0, 8>("dwordx4, 64 lanes, rewriting a 64 KiB window", d, reps);
  run<0, 4>("dwordx4, 64 lanes, rewriting a 64 KiB window", d, reps);
  run<0, 2>("dwordx4, 64 lanes, rewriting a 64 KiB window", d, reps);
  run<0, 1>("dwordx4, 64 lanes, rewriting a 64 KiB window", d, reps);
  run<0, 16>("dwordx4, 64 lanes, rewriting a 64 KiB window", d, reps);
  run<1, 8>("dwordx4, 32 lanes", d, reps);
  run<5, 8>("dwordx2, 64 lanes", d, reps);
  run<2, 8>("dword,   64 lanes", d, reps);
  run<6, 8>("dwordx4, 64 lanes on one contiguous KiB", d, reps);
  run<6, 2>("dwordx4, 64 lanes on one contiguous KiB", d, reps);
  run<3, 8>("dwordx4, 64 lanes, streaming (new lines)", d, reps);
  run<3, 16>("dwordx4, 64 lanes, streaming (new lines)", d, reps);
  run<4, 8>("dwordx4 + vmcnt(0) after each (round trip)", d, reps);
  return 0;
}
