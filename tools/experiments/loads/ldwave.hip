// Vector LOAD issue cost per CU (the store side: stores/stwave.hip -- ~58 clk per store instruction per CU whatever the
// width): global_load_dwordx4 / dword by shape on an L2-resident window, and the LDS-DMA form, 8 waves per CU.
//   hipcc -O3 -w --offload-arch=gfx950 ldwave.hip -o ldwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float float4_t __attribute__((ext_vector_type(4)));
// MODE 0: dwordx4, 16 rows x 64 B; 1: dwordx4, 1 KiB contiguous; 2: dword, 16 rows x 16 B... (64 lanes x 4 B, rows of 64 B);
// 3: LDS-DMA dwordx4, 16 rows x 64 B; 4: LDS-DMA dwordx4 contiguous; 5: dwordx4, 4 rows x 256 B
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(const float *in, float *out, int reps) {
  __shared__ __attribute__((aligned(1024))) char lds[NW * 8 * 1024];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float *base = in + (size_t)blockIdx.x * 64 * 256;          // this block's 64 rows x 1 KiB window
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds + wave * 8 * 1024;
  float4_t v[8];
  for (int t = 0; t < reps; ++t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = (i * NW + wave) & 63;
      const float *p;
      if (MODE == 0 || MODE == 2 || MODE == 3) { const int ct = q & 15, rt = q >> 4; p = base + (size_t)(rt * 16 + (lane & 15)) * 256 + ct * 16 + (lane >> 4) * 4; }
      else if (MODE == 5) { const int ct = q & 3, rt = q >> 2; p = base + (size_t)(rt * 4 + (lane & 3)) * 256 + ct * 64 + (lane >> 2) * 4; }
      else p = base + (size_t)q * 256 + lane * 4;
      if (MODE == 2) asm volatile("global_load_dword %0, %1, off" : "=v"(v[i][0]) : "v"(p) : "memory");
      else if (MODE == 3 || MODE == 4) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + i * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory");
      } else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (reps < 0) out[threadIdx.x] = v[0][0] + v[7][1];
}
template <int MODE, int NW> void run(const char *name, float *d, float *o, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, NW><<<256, NW * 64>>>(d, o, 4);
  hipEventRecord(a);
  k<MODE, NW><<<256, NW * 64>>>(d, o, reps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ns = ms * 1e6 / (reps * 8.0);
  printf("%-44s %2d waves: %6.1f ns per instr per CU (%4.0f clk at 2.1 GHz)\n", name, NW, ns / NW, ns / NW * 2.1);
}
int main() {
  const int reps = 400;
  float *d, *o; hipMalloc(&d, (size_t)256 * 64 * 1024); hipMalloc(&o, 4096 * 4);
  hipMemset(d, 0, (size_t)256 * 64 * 1024);
  run<0, 8>("load dwordx4, 16 rows x 64 B", d, o, reps);
  run<5, 8>("load dwordx4,  4 rows x 256 B", d, o, reps);
  run<1, 8>("load dwordx4, 1 KiB contiguous", d, o, reps);
  run<2, 8>("load dword,   16 rows x 16 B", d, o, reps);
  run<3, 8>("LDS-DMA dwordx4, 16 rows x 64 B", d, o, reps);
  run<4, 8>("LDS-DMA dwordx4, 1 KiB contiguous", d, o, reps);
  run<0, 16>("load dwordx4, 16 rows x 64 B", d, o, reps);
  return 0;
}
