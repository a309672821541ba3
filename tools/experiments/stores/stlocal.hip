// Store-instruction throughput of ONE CU on cache-resident lines: every block rewrites its own 64-row x 1 KiB window
// (64 KiB, stays in L2), 8 waves, by shape.  ns per wave instruction per CU, and B/clk at 2.1 GHz.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-result stlocal.hip -o stlocal
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void k(float *out, int reps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4_t v = {1.f, 2.f, 3.f, (float)lane};
  float *base = out + (size_t)blockIdx.x * 64 * 256;
  for (int t = 0; t < reps; ++t) {
    for (int i = 0; i < 64 / NW; ++i) {
      const int q = i * NW + wave;          // which of the 64 wave instructions of the window
      size_t off;
      if (P == 0 || P == 4) { const int ct = q & 15, rt = q >> 4; off = (size_t)(rt * 16 + (lane & 15)) * 256 + ct * 16 + (lane >> 4) * 4; }
      else if (P == 1) { const int ct = q & 7, rt = q >> 3; off = (size_t)(rt * 8 + (lane & 7)) * 256 + ct * 32 + (lane >> 3) * 4; }
      else if (P == 2) { const int ct = q & 3, rt = q >> 2; off = (size_t)(rt * 4 + (lane & 3)) * 256 + ct * 64 + (lane >> 2) * 4; }
      else { off = (size_t)q * 256 + lane * 4; }
      if (P == 4) {   // the same 16 x 64 B footprint as two dwordx2 instructions
        *reinterpret_cast<float2_t *>(base + off) = float2_t{v[0], v[1]};
        *reinterpret_cast<float2_t *>(base + off + 2) = float2_t{v[2], v[3]};
      } else {
        *reinterpret_cast<float4_t *>(base + off) = v;
      }
    }
  }
}
template <int P, int NW> void run(const char *name, float *d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 400;
  k<P, NW><<<256, NW * 64>>>(d, 10);
  hipEventRecord(a);
  k<P, NW><<<256, NW * 64>>>(d, reps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ns = ms * 1e6 / (reps * 64.0);
  printf("%-34s %2d waves: %6.1f ns per 1 KiB of stores per CU  (%.1f B/clk at 2.1 GHz)\n", name, NW, ns, 1024.0 / (ns * 2.1));
}
int main() {
  float *d; hipMalloc(&d, (size_t)256 * 64 * 1024);
  run<0, 8>("16 rows x 64 B   (dwordx4)", d);
  run<4, 8>("16 rows x 64 B   (2 x dwordx2)", d);
  run<1, 8>(" 8 rows x 128 B  (dwordx4)", d);
  run<2, 8>(" 4 rows x 256 B  (dwordx4)", d);
  run<3, 8>(" 1 row  x 1 KiB  (dwordx4)", d);
  run<0, 4>("16 rows x 64 B   (dwordx4)", d);
  run<3, 4>(" 1 row  x 1 KiB  (dwordx4)", d);
  run<0, 16>("16 rows x 64 B   (dwordx4)", d);
  return 0;
}
