// K3 for tall inputs at Fin = 256, ROW-STATIONARY:  h = act(x W^T + b),  Fout = G x 256 (G <= 4).
//
// Same 3xbf16 contraction, term for term and in the same k order, as k_linear_split (linear.hip) and K8's dense
// mode (wide.hip) -- the tests compare them bitwise -- with the operand roles of interact_rows.hip:
//   * a wave owns 16 rows: their 256 input channels sit in registers in the MFMA B-operand layout (lane =
//     (row lane % 16, group lane / 16) holds the 8 channels 32 ks + 8 g .. + 7 of every k-step ks: the four groups
//     of a row read 128 contiguous bytes per load), the 256 accumulators of the current column group in the MFMA
//     output layout (W is the A operand);
//   * the weights are the shared operand: pre-split into fragment-ordered bf16 hi / lo images (k_split_wl: one
//     32 KB chunk per (column group, k-step)) and brought by LDS-DMA into a ring of three slots two k-steps
//     ahead, read by all 8 waves; one barrier per k-step;
//   * x is read from memory ONCE however many column groups there are (the stacked node projections of the
//     InteractionNet block are 256 -> 768: the W-stationary kernels re-read x per group of 256 columns); the
//     next pass's rows move into the same registers k-step by k-step as the LAST column group releases them;
//   * a finished column group (+ bias, ReLU) leaves through a 64-column LDS tile in the ROW layout: 16 lanes
//     store the 256 contiguous bytes of a row piece (the MFMA layout's own store is 64-B pieces of 16 rows).
// Blocks are persistent (one per CU, 131 KB of LDS) and walk 128-row passes interleaved.
#include "common.h"
#include "rows_common.h"
#ifndef LR_ABL
#define LR_ABL 0      // timing ablations: 1 no stores, 2 no row loads, 3 no W reads from LDS
#endif

namespace {

constexpr int kF = 256;                       // Fin
constexpr int kNJ = kF / 16, kKS = kF / 32;   // column tiles per group, k-steps
constexpr int kNW = 8, kRowsL = kNW * 16;     // waves, rows per pass
constexpr int kStep = kNJ * 2 * 1024;         // one (group, k-step) of W fragments: 32 KB
constexpr int kSlots = 3, kDPW = kStep / 1024 / kNW;
constexpr int kYC = 64, kPY = kYC + 4;        // store tile: columns, pitch (floats)
constexpr int kOffYL = kSlots * kStep;
constexpr int kOffBL = kOffYL + kRowsL * kPY * 4;
constexpr int kMaxG = 4;
constexpr int kLdsL = kOffBL + kMaxG * kF * 4;

// W [G * 256, 256] fp32 -> chunk (g, ks): column tile jo, (hi, lo), lane l = (i = l % 16, q = l / 16): the 8 values
// W[256 g + 16 jo + i][32 ks + 8 q .. + 7]   (the natural k order of k_linear_split / K8)
__global__ __launch_bounds__(64) void k_split_wl(const float *__restrict__ W, bf16x8 *__restrict__ img) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  const int jo = blockIdx.x / kKS, ks = blockIdx.x % kKS, g = blockIdx.y;
  const float *wp = W + (int64_t)(g * kF + jo * 16 + i) * kF + 32 * ks + 8 * q;
  bf16x8 hi, lo;
  split8(*reinterpret_cast<const float4_t *>(wp), *reinterpret_cast<const float4_t *>(wp + 4), hi, lo);
  bf16x8 *dst = img + (int64_t)(g * kKS + ks) * (kStep / 16) + (jo * 2) * 64 + lane;
  dst[0] = hi;
  dst[64] = lo;
}

__device__ inline void wait_vm_rt(int n) {            // s_waitcnt vmcnt(n) for a wave-uniform run-time n
  switch (n) {
#define GWEN_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    GWEN_VM(4) GWEN_VM(6) GWEN_VM(8) GWEN_VM(20) GWEN_VM(24)
#undef GWEN_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

__global__ __launch_bounds__(kNW * 64) void k_linear_rows(const float *__restrict__ x, const char *__restrict__ img,
                                                          const float *__restrict__ bias, float *__restrict__ h,
                                                          int32_t rows, int32_t G, int64_t ldx, int64_t ldh,
                                                          int relu) {
  __shared__ __attribute__((aligned(1024))) char lds[kLdsL];
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  float *ytile = reinterpret_cast<float *>(lds + kOffYL);
  float *bl = reinterpret_cast<float *>(lds + kOffBL);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mi = lane & 15, g4 = lane >> 4;
  const int n_pass = (rows + kRowsL - 1) / kRowsL;
  if ((int)blockIdx.x >= n_pass) return;

  for (int f = t; f < G * kF; f += kNW * 64) bl[f] = bias ? bias[f] : 0.0f;

  const char *img_w = uniform_ptr(img + (int64_t)wave * kDPW * 1024);
  const uint32_t lane16 = (uint32_t)lane * 16;
  const int n_chunks = G * kKS;
  int slot = 0;
  auto dma = [&](int chunk, int into) {
    uint32_t off = (uint32_t)chunk * kStep;
    asm volatile("" : "+s"(off));
    const char *src = img_w + off;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + into * kStep + wave * kDPW * 1024);
    static_for<kDPW>([&](auto qq) { glds16<decltype(qq)::value * 1024>(src, lane16, dst); });
  };
  // this lane's row of pass p (clamped: rows past the end read the last row and are never stored)
  auto row_of = [&](int p) {
    const int r = p * kRowsL + wave * 16 + mi;
    return r < rows ? r : rows - 1;
  };
  auto load_rows = [&](float4_t (&X)[2 * kKS], int p, int ks) {       // k-step ks of pass p's rows: 32 B per lane
    int gs = g4;
    asm volatile("" : "+v"(gs));
    const float *xp = x + (int64_t)row_of(p) * ldx + 32 * ks + 8 * gs;
    X[2 * ks] = *reinterpret_cast<const float4_t *>(xp);
    X[2 * ks + 1] = *reinterpret_cast<const float4_t *>(xp + 4);
  };

  float4_t X[2 * kKS], acc[kNJ];
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) load_rows(X, blockIdx.x, ks);
  dma(0, 0);
  dma(1 % n_chunks, 1);

  int carry = 0;                     // what may stay in flight at the next step 0 (0: drain)
  for (int pass = blockIdx.x; pass < n_pass; pass += (int)gridDim.x) {
    const int next = pass + (int)gridDim.x;
    const int pn = next < n_pass ? next : pass;                        // last pass: load it again, unused
    for (int g = 0; g < G; ++g) {
      const bool last_g = g + 1 == G;
#pragma unroll
      for (int j = 0; j < kNJ; ++j) acc[j] = float4_t{0.f, 0.f, 0.f, 0.f};
      static_for<kKS>([&](auto ss) {
        constexpr int ks = decltype(ss)::value;
        // chunk (g, ks) has landed (its DMA is two steps old); the next chunk's DMA (kDPW instructions) and the
        // loads of the steps since may stay in flight: the last column group's steps carry 2 row loads each.
        // Step 0 follows the stores of a finished group: 16 per wave when all its 16 rows exist (then they may
        // stay in flight too -- draining them would cost an HBM write round trip per group), else a full drain.
        if constexpr (ks == 0) wait_vm_rt(carry);
        else if constexpr (ks == 1) wait_vm_rt(kDPW + (last_g ? 2 : 0));
        else wait_vm_rt(kDPW + (last_g ? 4 : 0));
        __syncthreads();
        {
          const int into = slot >= 1 ? slot - 1 : kSlots - 1;          // (slot + 2) % 3
          dma((g * kKS + ks + 2) % n_chunks, into);
        }
        bf16x8 bh, bo;
        split8(X[2 * ks], X[2 * ks + 1], bh, bo);
#if LR_ABL != 2
        if (last_g) load_rows(X, pn, ks);                              // released: the next pass's rows move in
#endif
        const char *wb = lds + slot * kStep + lane * 16;
#pragma unroll
        for (int jo = 0; jo < kNJ; jo += 2) {
#if LR_ABL == 3
          const int jr = 0;
#else
          const int jr = jo;
#endif
          const bf16x8 wh0 = *reinterpret_cast<const bf16x8 *>(wb + (jr * 2) * 1024);
          const bf16x8 wl0 = *reinterpret_cast<const bf16x8 *>(wb + (jr * 2 + 1) * 1024);
          const bf16x8 wh1 = *reinterpret_cast<const bf16x8 *>(wb + (jr * 2 + 2) * 1024);
          const bf16x8 wl1 = *reinterpret_cast<const bf16x8 *>(wb + (jr * 2 + 3) * 1024);
          acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0, bo, acc[jo], 0, 0, 0);
          acc[jo + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1, bo, acc[jo + 1], 0, 0, 0);
          acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl0, bh, acc[jo], 0, 0, 0);
          acc[jo + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl1, bh, acc[jo + 1], 0, 0, 0);
          acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0, bh, acc[jo], 0, 0, 0);
          acc[jo + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1, bh, acc[jo + 1], 0, 0, 0);
        }
        slot = slot + 1 == kSlots ? 0 : slot + 1;
      });
      // ---- the finished group leaves in the row layout, 64 columns at a time, through the wave's own 16 rows of
      // the store tile (no other wave touches them: no barrier) -----------------------------------------------------
      {
        int gs = g4, ms = mi;
        asm volatile("" : "+v"(gs), "+v"(ms));
        float *yown = ytile + wave * 16 * kPY;
        const int r0 = pass * kRowsL + wave * 16;
        float *po = h + (int64_t)(r0 + gs) * ldh + g * kF + 4 * ms;
#pragma unroll
        for (int c = 0; c < kF / kYC; ++c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // bias AFTER the contraction, as the other K3 kernels: bitwise the same result
            float4_t o = acc[4 * c + j] + *reinterpret_cast<const float4_t *>(bl + g * kF + 64 * c + 16 * j + 4 * gs);
            if (relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = o[e] < 0.0f ? 0.0f : o[e];
            }
            *reinterpret_cast<float4_t *>(yown + ms * kPY + 16 * j + 4 * gs) = o;
          }
          float4_t v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4_t *>(yown + (4 * k + gs) * kPY + 4 * ms);
#if LR_ABL == 1
          if (rows < 0)
#endif
          if (r0 + 16 <= rows) {                                       // every lane stores: an exact count
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<float4_t *>(po + (int64_t)(4 * k) * ldh + kYC * c) = v[k];
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (r0 + 4 * k + gs < rows) *reinterpret_cast<float4_t *>(po + (int64_t)(4 * k) * ldh + kYC * c) = v[k];
          }
        }
        // younger than the chunk the next step 0 waits for: the loads of steps 6 and 7, step 7's DMA, these stores
        carry = r0 + 16 <= rows ? kDPW + 16 + (last_g ? 4 : 0) : 0;
#if LR_ABL != 0
        carry = 0;
#endif
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the DMAs issued past the last chunk used
}

}  // namespace

// workspace (floats) the row-stationary path needs for W's image: Fout * 256 (hi + lo bf16 = 4 B per weight)
int64_t gwen_linear_rows_workspace_floats(int64_t Fin, int64_t Fout) {
  return (Fin == kF && Fout % kF == 0 && Fout >= kF && Fout <= kMaxG * kF) ? Fout * kF : 0;
}

// linear.hip's dispatch for tall inputs at Fin = 256, Fout = G x 256 (pointers and alignment validated there)
int gwen_linear_rows_f32(const float *x, const float *W, const float *bias, float *h, int64_t rows, int64_t Fout,
                         int64_t ldx, int64_t ldh, int relu, float *workspace, hipStream_t st) {
  const int G = (int)(Fout / kF);
  if (G < 1 || G > kMaxG || rows < 1 || rows >= (int64_t(1) << 31) - kRowsL) return GWEN_ERANGE;
  bf16x8 *img = reinterpret_cast<bf16x8 *>(workspace);
  k_split_wl<<<dim3(kNJ * kKS, G), 64, 0, st>>>(W, img);
  GWEN_LAUNCH_CHECK();
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    GWEN_HIP_CHECK(hipGetDevice(&dev));
    GWEN_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = n < 8 ? 8 : n;
  }
  const int64_t n_pass = (rows + kRowsL - 1) / kRowsL;
  const int64_t blocks = n_pass < cus ? n_pass : cus;
  k_linear_rows<<<(unsigned)blocks, kNW * 64, 0, st>>>(x, reinterpret_cast<const char *>(img), bias, h, (int32_t)rows,
                                                       G, ldx, ldh, relu);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
