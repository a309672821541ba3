// LDS read bandwidth per CU for K8's access patterns (8 waves, one block per CU): bytes / clock / CU
//   hipcc -O3 --offload-arch=gfx950 ldsbw.hip -o ldsbw && ./ldsbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int PAT, int NW>
__global__ __launch_bounds__(NW * 64) void k(float *out, int iters, const uint32_t *rows) {
  __shared__ __attribute__((aligned(1024))) char lds[128 * 1024];
  for (int i = threadIdx.x; i < 32 * 1024; i += NW * 64) reinterpret_cast<float *>(lds)[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mi = lane & 15, mh = lane >> 4;
  uint32_t r8[8], a[8];
  for (int u = 0; u < 8; ++u) r8[u] = rows[(wave * 4 + mh) * 8 + u];      // 0..127
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (PAT == 0) a[u] = r8[u] * 256 + mi * 16;                                          // gather: 4 rows x 256 B
    else if (PAT == 1) a[u] = ((u & 3) * 16 + mi) * 160 + mh * 16 + (u >> 2) * 64;      // A fragment, 160-B pitch
    else if (PAT == 2) a[u] = lane * 16 + u * 1024;                                      // linear
    else if (PAT == 3) a[u] = r8[u] * 272 + mi * 16;                                     // gather, 272-B row pitch
    else if (PAT == 4) a[u] = r8[u] * 256 + ((mi * 16 + mh * 64) & 255);                 // gather, row groups rotated by 64 B
    else if (PAT == 5) a[u] = r8[u] * 256 + ((mi * 16 + mh * 128) & 255);                // gather, rotated by 128 B
    else if (PAT == 6) a[u] = (r8[u] & ~3u) * 256 + mh * 256 + mi * 16;                  // 4 consecutive rows (1 KiB linear)
    else a[u] = ((u & 3) * 16 + mi) * 144 + mh * 16 + (u >> 2) * 64;                     // A fragment, 144-B pitch
    a[u] += (uint32_t)(uintptr_t)lds;
  }
  float4_t v;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a[u]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (iters < 0) out[threadIdx.x] = v[0];
}

template <int PAT, int NW>
void run(const char *name, float *out, uint32_t *rows) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256;
  k<PAT, NW><<<blocks, NW * 64>>>(out, 100, rows);
  hipEventRecord(e0);
  k<PAT, NW><<<blocks, NW * 64>>>(out, iters, rows);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 8 * NW * 64 * 16;     // per CU
  printf("%-28s %2d waves: %.1f B/clk/CU at 2.4 GHz (%.3f ms)\n", name, NW, bytes / (ms * 1e-3 * 2.4e9), ms);
}

int main() {
  float *out; uint32_t *rows, h[16 * 4 * 8];
  hipMalloc(&out, 4096 * 4); hipMalloc(&rows, sizeof(h));
  uint32_t s = 12345;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) & 127; }
  hipMemcpy(rows, h, sizeof(h), hipMemcpyHostToDevice);
  run<2, 4>("linear 1 KiB", out, rows); run<2, 8>("linear 1 KiB", out, rows); run<2, 16>("linear 1 KiB", out, rows);
  run<0, 8>("gather 4 rows x 256 B", out, rows);  run<0, 16>("gather 4 rows x 256 B", out, rows);
  run<6, 8>("4 consecutive rows", out, rows);
  run<3, 8>("gather, 272-B row pitch", out, rows);
  run<4, 8>("gather, groups rot 64 B", out, rows);
  run<5, 8>("gather, groups rot 128 B", out, rows);
  run<1, 8>("A fragment (160-B pitch)", out, rows); run<1, 16>("A fragment (160-B pitch)", out, rows);
  run<7, 8>("A fragment (144-B pitch)", out, rows);
  return 0;
}
