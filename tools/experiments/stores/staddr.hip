// Store issue cost by ADDRESS MODE (stwave.hip: ~58 clk per store instruction per CU whatever its width, i.e. ~1 lane per
// clock with 64-bit per-lane addresses): the same 16 rows x 64 B dwordx4 store as  (0) global, 64-bit VGPR address;
// (1) global, SGPR base + 32-bit VGPR offset;  (2) raw buffer store (resource + 32-bit offset).
//   hipcc -O3 -w --offload-arch=gfx950 staddr.hip -o staddr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(float *out, int reps) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float4_t v = {1.f, 2.f, 3.f, (float)lane};
  float *base = out + (size_t)blockIdx.x * 64 * 256;
  const uint64_t b64 = (uint64_t)base;
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b64), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32));
  const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 64 * 1024, 0x00020000);
  for (int t = 0; t < reps; ++t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = (i * NW + wave) & 63;
      const int ct = q & 15, rt = q >> 4;
      const uint32_t off = (uint32_t)(((rt * 16 + (lane & 15)) * 256 + ct * 16 + (lane >> 4) * 4) * 4);
      if (MODE == 0) {
        float *p = reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off);
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
      } else if (MODE == 1) {
        asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(v), "s"(sbase) : "memory");
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 0);
      }
    }
  }
}
template <int MODE, int NW> void run(const char *name, float *d, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, NW><<<256, NW * 64>>>(d, 4);
  hipEventRecord(a);
  k<MODE, NW><<<256, NW * 64>>>(d, reps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ns = ms * 1e6 / (reps * 8.0);
  printf("%-44s %2d waves: %6.1f ns per instr per CU (%4.0f clk at 2.1 GHz)\n", name, NW, ns / NW, ns / NW * 2.1);
}
int main() {
  const int reps = 400;
  float *d; hipMalloc(&d, (size_t)256 * 64 * 1024);
  run<0, 8>("global, 64-bit VGPR address", d, reps);
  run<1, 8>("global, SGPR base + 32-bit VGPR offset", d, reps);
  run<2, 8>("raw buffer store, 32-bit offset", d, reps);
  run<1, 2>("global, SGPR base + 32-bit VGPR offset", d, reps);
  run<2, 2>("raw buffer store, 32-bit offset", d, reps);
  return 0;
}
