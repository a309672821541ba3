// Can the matrix pipe and the vector ALU of one SIMD run at the same time?  8 waves per CU (2 per SIMD), per iteration and SIMD
// 96 MFMA 16x16x32 bf16 (1 536 cycles of the matrix pipe) and 432 independent-chain v_fma_f32 (1 728 cycles of the VALU):
//   A  both waves: 48 MFMA, then 216 FMA                      (serial inside a wave)
//   B  both waves: (1 MFMA, 4-5 FMA) x 48                      (woven inside a wave)
//   C  wave 0: 96 MFMA only; wave 1: 432 FMA only              (split between the waves)
//   M  both waves: 48 MFMA only        V  both waves: 216 FMA only
//   hipcc -O3 -w --offload-arch=gfx950 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
  const float w = 1.0001f, z = 0.5f;
  auto mfma = [&](int n) {
#pragma unroll
    for (int i = 0; i < n; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
  };
  auto fma = [&](int n) {
#pragma unroll
    for (int i = 0; i < n; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(w), "v"(z));
  };
  f32x16 big[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
  auto mfma32 = [&](int n) {                     // 32x32x16: twice the flops of a 16x16x32 per instruction
#pragma unroll
    for (int i = 0; i < n; ++i) big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i & 3], 0, 0, 0);
  };
  const bool second = wave >= 4;                 // waves w and w + 4 share a SIMD
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { mfma(48); fma(216); }
    else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
        fma(i & 1 ? 5 : 4);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 2) { if (second) fma(432); else mfma(96); }
    else if (MODE == 3) mfma(48);
    else if (MODE == 4) fma(216);
    else if (MODE == 5) mfma32(24);
    else if (MODE == 6) { mfma32(24); fma(216); }
    else if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i & 3], 0, 0, 0);
        fma(9);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 8) { if (second) fma(432); else mfma32(48); }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
  for (int i = 0; i < 4; ++i) s += big[i][0];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MODE> void run(const char *name, float *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  k<MODE><<<256, 512>>>(d, 10);
  hipEventRecord(e0);
  k<MODE><<<256, 512>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %7.0f ns per iteration = %5.0f cycles at 2.1 GHz\n", name, ms * 1e6 / iters, ms * 1e6 / iters * 2.1);
}
int main() {
  float *d; hipMalloc(&d, 4096 * 4);
  run<3>("M  both waves 48 MFMA", d);
  run<4>("V  both waves 216 FMA", d);
  run<0>("A  both waves 48 MFMA then 216 FMA", d);
  run<1>("B  both waves woven (1 MFMA, 4-5 FMA) x 48", d);
  run<2>("C  one wave 96 MFMA, the other 432 FMA", d);
  run<5>("M32 both waves 24 MFMA 32x32x16", d);
  run<6>("A32 both waves 24 MFMA 32x32x16 then 216 FMA", d);
  run<7>("B32 both waves woven (1 MFMA 32x32x16, 9 FMA) x 24", d);
  run<8>("C32 one wave 48 MFMA 32x32x16, the other 432 FMA", d);
  return 0;
}
