// Backward pieces of the GCNConv layer (the reference trains through it:
// /root/reference/src/gwen/models_gnn.py:372 loss.backward(), :373 optimizer.step()).
//   grad_W = g^T @ x   reduction over the node axis, fp32 MFMA, two deterministic stages
//   grad_b = column sums of g
//   ReLU backward mask
// (grad wrt the aggregated features is K2 on the transposed CSR; grad wrt x is K3 with W^T.)
#include "common.h"
#include "split.h"
#include "rows_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunkRows = 256;    // rows reduced by one block before the cross-chunk stage
constexpr int kWideChunkRows = 4096;   // ... at most, by one 128 x 128 block of the split kernels
constexpr int kWideBlocksPerChip = 768;    // 128 x 128 blocks resident on 256 CUs (3 per CU: 132 registers a lane;
                                           // forced to 4 per CU the kernel is no faster: it is bound by its VALU + MFMA work)
constexpr int kNarrowChunkRows = 512;  // ... at 64 x 64 .. 64 x 128 (one or two waves per block: more blocks to fill the chip)
// rows per block of the split kernels.  128 x 128 blocks (4 waves, 32 KiB of LDS, 3 waves per SIMD by registers: the 256
// CUs hold 768 of them at once) are launched in whole ROUNDS of the chip, because a launch of 2.04 rounds takes three: at
// 256 x 256 on 100 000 rows the ~2 048-block rule below gave 1 564 blocks on 768 places (87 us; 190 chunks in ONE round: 65,
// and half the partial sums to write and to reduce -- 119 -> 82 us with the finish; 600 000 rows 509 -> 385).  The fewest
// rounds that keep a chunk under the cap.  The narrower blocks (1 - 2 waves) keep ~2 048 blocks per launch (a node-sized
// gradient -- 100 000 rows -- would otherwise be 49 blocks): measured better there than whole rounds (600 000 x 64 x 64:
// 72 vs 84 us).  Whole 16-row steps; never under kChunkRows rows (the workspaces are sized by
// gwen_gcn_grad_chunks(rows) = rows / kChunkRows slots).
inline int split_chunk_rows(int64_t rows, int64_t Fin, int64_t Fout) {
  int64_t cr;
  if (Fin % 128 == 0 && Fout % 128 == 0) {
    const int64_t tiles = (Fin / 128) * (Fout / 128);
    const int64_t per_round = kWideBlocksPerChip / tiles > 0 ? kWideBlocksPerChip / tiles : 1;       // chunks in one round
    const int64_t rounds = (rows + kWideChunkRows * per_round - 1) / (kWideChunkRows * per_round);
    const int64_t chunks = (rounds > 0 ? rounds : 1) * per_round;
    cr = ((rows + chunks - 1) / chunks + 15) / 16 * 16;
  } else {
    const int64_t tiles = ((Fin + 127) / 128) * ((Fout + 127) / 128);
    cr = (rows * tiles / 2048 + 15) / 16 * 16;
    cr = cr > kNarrowChunkRows ? kNarrowChunkRows : cr;
  }
  return (int)(cr < kChunkRows ? kChunkRows : cr);
}
constexpr int kU = 8;              // independent loads in flight per thread in every reduction loop

// One wave = one 32x32 tile of grad_W (rows = output channel, cols = input channel) over one chunk
// of node rows.  A[i = l&31][k = l>>5] = g[row k][o0 + i], B[k][j = l&31] = x[row k][i0 + j]:
// both operands are 128-B contiguous global reads per half-wave, so no LDS staging is needed.
// Every load is unconditional (clamped row / column, value masked afterwards) and kU row pairs are
// requested before the first MFMA: with one dependent load pair per MFMA and 1024-row chunks this
// kernel ran at one memory round trip per two rows (206 us for 64 x 64 on 100 002 rows).
__global__ __launch_bounds__(kThreads) void k_grad_w(const float *__restrict__ g,
                                                     const float *__restrict__ x,
                                                     float *__restrict__ dst, int64_t rows, int Fin,
                                                     int Fout, int64_t ldg, int64_t ldx,
                                                     int tiles_i, int ntiles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.y * 4 + wave;
  if (tile >= ntiles) return;
  const int o0 = (tile / tiles_i) * 32, i0 = (tile % tiles_i) * 32;
  const int64_t r0 = (int64_t)blockIdx.x * kChunkRows;
  const int64_t r1 = (r0 + kChunkRows < rows) ? r0 + kChunkRows : rows;
  const bool ao = o0 + li < Fout, ai = i0 + li < Fin;
  const float *gp = g + (ao ? o0 + li : Fout - 1), *xp = x + (ai ? i0 + li : Fin - 1);
  f32x16 acc = {};
  for (int64_t r = r0 + lh; r < r1 + lh; r += 2 * kU) {     // every lane runs the same trip count
    float a[kU], b[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t rr = r + 2 * u, rc = rr < r1 ? rr : r1 - 1;
      a[u] = gp[rc * ldg];
      b[u] = xp[rc * ldx];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool in = r + 2 * u < r1;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32((in && ao) ? a[u] : 0.0f, (in && ai) ? b[u] : 0.0f,
                                                 acc, 0, 0, 0);
    }
  }
  float *d = dst + (int64_t)blockIdx.x * Fout * Fin;
  if (!ai) return;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int o = o0 + (t & 3) + 8 * (t >> 2) + 4 * lh;
    if (o < Fout) d[(int64_t)o * Fin + i0 + li] = acc[t];
  }
}

// The same for gradients of ONE or TWO 32 x 32 tiles (the narrow layers of the c2 model: 32 x 16, 16 x 16, 64 x 32 ...),
// where three or two of the block's four waves had nothing to do and a launch was 391 single waves (16.5 us each on
// 100 002 rows, latency-bound): a tile's 16-row steps are dealt over P = 4 / tiles waves, whose accumulators are added in
// phase order through LDS at the end (fixed order: bitwise reproducible).
template <int P>
__global__ __launch_bounds__(kThreads) void k_grad_w_few(const float *__restrict__ g, const float *__restrict__ x,
                                                         float *__restrict__ dst, int64_t rows, int Fin, int Fout,
                                                         int64_t ldg, int64_t ldx, int tiles_i, int ntiles) {
  __shared__ float red[4 * 16 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = wave / P, ph = wave % P;
  const bool live = tile < ntiles;
  const int o0 = live ? (tile / tiles_i) * 32 : 0, i0 = live ? (tile % tiles_i) * 32 : 0;
  const int64_t r0 = (int64_t)blockIdx.x * kChunkRows;
  const int64_t r1 = (r0 + kChunkRows < rows) ? r0 + kChunkRows : rows;
  const bool ao = o0 + li < Fout, ai = i0 + li < Fin;
  const float *gp = g + (ao ? o0 + li : Fout - 1), *xp = x + (ai ? i0 + li : Fin - 1);
  f32x16 acc = {};
  if (live) {
    for (int64_t r = r0 + lh + 2 * kU * ph; r < r1 + lh; r += 2 * kU * P) {     // every lane of a wave: the same trip count
      float a[kU], b[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t rr = r + 2 * u, rc = rr < r1 ? rr : r1 - 1;
        a[u] = gp[rc * ldg];
        b[u] = xp[rc * ldx];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool in = r + 2 * u < r1;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((in && ao) ? a[u] : 0.0f, (in && ai) ? b[u] : 0.0f, acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) red[(wave * 16 + t) * 64 + lane] = acc[t];
  __syncthreads();
  if (!live || ph != 0 || !ai) return;
  float *d = dst + (int64_t)blockIdx.x * Fout * Fin;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    float v = acc[t];
#pragma unroll
    for (int q = 1; q < P; ++q) v = v + red[((wave + q) * 16 + t) * 64 + lane];
    const int o = o0 + (t & 3) + 8 * (t >> 2) + 4 * lh;
    if (o < Fout) d[(int64_t)o * Fin + i0 + li] = v;
  }
}

// grad_W for WIDE layers on the split contractions (round 4).  At 256 channels the fp32-input MFMA above is the whole
// cost of a training step's weight gradients (InteractionNet forecaster, 600 000 edge rows: 54 launches, 27 of the step's
// 59 ms): 1/16 of the bf16 matrix rate, every operand value read by 8 waves.  Here one wave owns a 64 x 64 tile of
// grad_W over a chunk of 2 048 rows: per 16 rows a lane reads its 8 rows of two 32-column pieces of g and of x (one
// dword each, 128 contiguous bytes per half-wave: the same no-LDS access as above, four tiles' worth of products per
// value instead of one), cuts them into NS bf16 images (split.h) and issues 4 x (3 | 6) v_mfma_f32_32x32x16_bf16.
// NS = 2 (bf16x3) for layers on the "3xbf16" tier and the InteractionNet block, NS = 3 (bf16x6, operands kept to 2^-24)
// for fp32-class layers; rows are summed in stored order by one wave, the chunks in a fixed order by k_reduce_*:
// bitwise reproducible.  Widths: Fin, Fout multiples of 64 (no column guards); everything else stays on k_grad_w.
template <int NS>
__device__ inline f32x16 mma32_split(const gwen::bf16x8 (&a)[NS], const gwen::bf16x8 (&b)[NS], f32x16 d) {
#pragma unroll
  for (int t = NS - 1; t >= 0; --t)
#pragma unroll
    for (int i = 0; i <= t; ++i) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[t - i], d, 0, 0, 0);
  return d;
}

template <int NS>
__global__ __launch_bounds__(kThreads) void k_grad_w_split(const float *__restrict__ g, const float *__restrict__ x,
                                                           float *__restrict__ dst, int64_t rows, int Fin, int Fout,
                                                           int64_t ldg, int64_t ldx, int tiles_i, int ntiles,
                                                           int chunk_rows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.y * 4 + wave;
  if (tile >= ntiles) return;
  const int o0 = (tile / tiles_i) * 64, i0 = (tile % tiles_i) * 64;
  const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
  const int64_t r1 = (r0 + chunk_rows < rows) ? r0 + chunk_rows : rows;
  const float *gp = g + o0 + li, *xp = x + i0 + li;
  f32x16 acc[2][2] = {};
  // this lane's 8 rows of a 16-row step: r + 8 lh .. + 7 (the k index of the MFMA's 8 values per lane), clamped to the
  // chunk's last row and zeroed on the g side past it.  The next step's 32 values are requested before this step's
  // products are issued.
  float a[2][8], b[2][8];
  auto load = [&](int64_t r) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int64_t rr = r + 8 * lh + t, rc = rr < r1 ? rr : r1 - 1;
      a[0][t] = gp[rc * ldg];
      a[1][t] = gp[rc * ldg + 32];
      b[0][t] = xp[rc * ldx];
      b[1][t] = xp[rc * ldx + 32];
    }
  };
  load(r0);
  for (int64_t r = r0; r < r1; r += 16) {
    gwen::bf16x8 ai[2][NS], bi[2][NS];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float av[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) av[t] = r + 8 * lh + t < r1 ? a[s][t] : 0.0f;
      gwen::split_images<8, NS>(av, ai[s]);
      gwen::split_images<8, NS>(b[s], bi[s]);
    }
    if (r + 16 < r1) load(r + 16);
#pragma unroll
    for (int so = 0; so < 2; ++so)
#pragma unroll
      for (int si = 0; si < 2; ++si) acc[so][si] = mma32_split<NS>(ai[so], bi[si], acc[so][si]);
  }
  float *d = dst + (int64_t)blockIdx.x * Fout * Fin;
#pragma unroll
  for (int so = 0; so < 2; ++so)
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int o = o0 + 32 * so + (t & 3) + 8 * (t >> 2) + 4 * lh;
        d[(int64_t)o * Fin + i0 + 32 * si + li] = acc[so][si][t];
      }
}

// The same tiles with the operands STAGED THROUGH LDS: k_grad_w_split reads every value with its own dword load -- 32
// vector-memory instructions per wave and 16-row step, which is what bounds it (627 us for 600 000 x 256 x 256 on bf16x3
// where the matrix pipe needs ~160).  Here a block owns BO x BI outputs (one wave per 64 x 64 piece); per 16-row step the
// 16 x (BO + BI) fp32 values arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, (BO + BI) / 16 of
// them per BLOCK) into one of two buffers while the step before is being contracted, and a lane reads its 8 rows of a
// column from the row-major tile (consecutive lanes = consecutive dwords: conflict-free).  One barrier per step; the
// DMAs are inline asm (hipcc neither counts nor drains them) and the only visible vector-memory instructions are the
// tile's stores after the loop.
// BIAS: the chunk's column sums of g as well (grad_b of the same layer: the g tile is in LDS anyway) -- the waves that
// own the block's first 64 x-columns, in the blocks of the first x-column tile, add their g values per column as they
// cut them; bdst [chunks, Fout].
template <int NS, int BO, int BI, bool BIAS = false>
__global__ __launch_bounds__((BO / 64) * (BI / 64) * 64) void k_grad_w_lds(
    const float *__restrict__ g, const float *__restrict__ x, float *__restrict__ dst, int64_t rows, int Fin, int Fout,
    int64_t ldg, int64_t ldx, int tiles_i, int chunk_rows, float *__restrict__ bdst = nullptr) {
  constexpr int NWB = (BO / 64) * (BI / 64);                 // waves per block
  constexpr int kStepBytes = 16 * (BO + BI) * 4;             // one step's tiles: g [16][BO] | x [16][BI]
  constexpr int PG = BO / 16, NPC = (BO + BI) / 16, PPW = NPC / NWB;     // 1-KiB pieces: of g, in all, per wave
  static_assert(NPC % NWB == 0, "the pieces must deal out over the waves");
  __shared__ __attribute__((aligned(1024))) char lds[2 * kStepBytes];
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wo = wave / (BI / 64), wi = wave % (BI / 64);
  const int o0 = ((int)blockIdx.y / tiles_i) * BO, i0 = ((int)blockIdx.y % tiles_i) * BI;
  const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
  const int64_t r1 = (r0 + chunk_rows < rows) ? r0 + chunk_rows : rows;
  const char *gb = uniform_ptr(g + r0 * ldg + o0), *xb = uniform_ptr(x + r0 * ldx + i0);
  const int nsteps = (int)((r1 - r0 + 15) / 16), nrows = (int)(r1 - r0);
  // step st's tiles -> buffer buf: piece p of g holds 1024 / (4 BO) rows of BO floats (lane l: 16 B at column 4 (l % (BO / 4))
  // of row l / (BO / 4)); rows past the chunk are clamped to its last row (and zeroed when they are read)
  auto issue = [&](int st, int buf) {
    static_for<PPW>([&](auto qq) {
      const int piece = wave * PPW + decltype(qq)::value;              // wave-uniform
      const bool is_g = piece < PG;
      const int pp = is_g ? piece : piece - PG;
      const int lpr = (is_g ? BO : BI) / 4;                          // lanes per row
      int row = st * 16 + pp * (64 / lpr) + lane / lpr;
      row = row < nrows ? row : nrows - 1;
      const uint32_t voff = (uint32_t)(((int64_t)row * (is_g ? ldg : ldx) + 4 * (lane % lpr)) * 4);
      glds16<0>(is_g ? gb : xb, voff, lds0 + buf * kStepBytes + (is_g ? 0 : 16 * BO * 4) + pp * 1024);
    });
  };
  f32x16 acc[2][2] = {};
  float bsum[2] = {0.0f, 0.0f};
  const bool bias_wave = BIAS && wi == 0 && (int)blockIdx.y % tiles_i == 0;      // wave-uniform
  issue(0, 0);
  // one 16-row step; TAIL = the chunk's last step, the only one that can hold rows past the chunk (zeroed as they are
  // read: the 16 compare + select pairs are not paid in the other steps)
  auto step = [&](int st, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    wait_vm<0>();                                   // this wave's pieces of step st (issued a step ago) have landed
    __syncthreads();                                // ... everyone's; and everyone is done reading the other buffer
    if constexpr (!TAIL) issue(st + 1, (st + 1) & 1);
    const float *lg = reinterpret_cast<const float *>(lds + (st & 1) * kStepBytes), *lx = lg + 16 * BO;
    gwen::bf16x8 ai[2][NS], bi[2][NS];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float av[8], bv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float a = lg[(8 * lh + t) * BO + 64 * wo + 32 * s + li];
        av[t] = (!TAIL || st * 16 + 8 * lh + t < nrows) ? a : 0.0f;
        bv[t] = lx[(8 * lh + t) * BI + 64 * wi + 32 * s + li];
      }
      if constexpr (BIAS) {
        if (bias_wave) bsum[s] = bsum[s] + (((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7])));
      }
      gwen::split_images<8, NS>(av, ai[s]);
      gwen::split_images<8, NS>(bv, bi[s]);
    }
#pragma unroll
    for (int so = 0; so < 2; ++so)
#pragma unroll
      for (int si = 0; si < 2; ++si) acc[so][si] = mma32_split<NS>(ai[so], bi[si], acc[so][si]);
  };
  for (int st = 0; st + 1 < nsteps; ++st) step(st, std::false_type{});
  step(nsteps - 1, std::true_type{});
  if constexpr (BIAS) {
    if (bias_wave) {                                  // rows 8 lh + t of every step: the two halves of a column meet here
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float v = bsum[s] + __shfl_xor(bsum[s], 32);
        if (lh == 0) bdst[(int64_t)blockIdx.x * Fout + o0 + 64 * wo + 32 * s + li] = v;
      }
    }
  }
  float *d = dst + (int64_t)blockIdx.x * Fout * Fin;
#pragma unroll
  for (int so = 0; so < 2; ++so)
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int o = o0 + 64 * wo + 32 * so + (t & 3) + 8 * (t >> 2) + 4 * lh;
        d[(int64_t)o * Fin + i0 + 64 * wi + 32 * si + li] = acc[so][si][t];
      }
}

// the split kernel takes a weight gradient when its widths allow and the contraction is a bf16 split
inline bool wide_grad(int64_t Fin, int64_t Fout, int contract) {
  return (contract == GWEN_CONTRACT_BF16X3 || contract == GWEN_CONTRACT_BF16X6) && Fin % 64 == 0 && Fout % 64 == 0;
}

// dst[j] = sum over chunks of partial[c][j], in a fixed order: 16 phases (phase p adds chunks p, p+16,
// ... ascending, kU loads in flight), folded through LDS in phase order.  Block = 16 elements x 16 phases.
__global__ __launch_bounds__(kThreads) void k_reduce_chunks(const float *__restrict__ partial,
                                                            float *__restrict__ dst, int64_t count,
                                                            int nchunks) {
  __shared__ float red[kThreads];
  const int e = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int64_t j = (int64_t)blockIdx.x * 16 + e, jc = j < count ? j : count - 1;
  float s = 0.0f;
  for (int c = ph; c < nchunks; c += 16 * kU) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int cc = c + 16 * u < nchunks ? c + 16 * u : nchunks - 1;
      v[u] = partial[(int64_t)cc * count + jc];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (c + 16 * u < nchunks) s = s + v[u];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (ph == 0 && j < count) {
    float tot = red[e];
#pragma unroll
    for (int p = 1; p < 16; ++p) tot = tot + red[16 * p + e];
    dst[j] = tot;
  }
}

// column sums of one chunk of rows: thread = column, 4 row phases per block folded through LDS,
// kU loads in flight per thread
__global__ __launch_bounds__(kThreads) void k_grad_b(const float *__restrict__ g,
                                                     float *__restrict__ dst, int64_t rows, int F,
                                                     int64_t ldg) {
  __shared__ float red[kThreads];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), phase = threadIdx.x >> 6;
  const int cc = c < F ? c : F - 1;
  const int64_t r0 = (int64_t)blockIdx.x * kChunkRows;
  const int64_t r1 = (r0 + kChunkRows < rows) ? r0 + kChunkRows : rows;
  float s = 0.0f;
  for (int64_t r = r0 + phase; r < r1; r += 4 * kU) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t rr = r + 4 * u, rc = rr < r1 ? rr : r1 - 1;
      v[u] = g[rc * ldg + cc];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (r + 4 * u < r1) s = s + v[u];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (phase == 0 && c < F)
    dst[(int64_t)blockIdx.x * F + c] = ((red[threadIdx.x] + red[threadIdx.x + 64]) +
                                        red[threadIdx.x + 128]) + red[threadIdx.x + 192];
}

__global__ void k_relu_bwd(const float *__restrict__ y, const float *__restrict__ g,
                           float *__restrict__ gin, int64_t count) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < count) gin[i] = y[i] > 0.0f ? g[i] : 0.0f;
}

inline int64_t nchunks_for(int64_t rows) { return rows > 0 ? (rows + kChunkRows - 1) / kChunkRows : 1; }
inline int64_t nchunks_w(int64_t rows, int64_t Fin, int64_t Fout, int contract) {     // partial slots of a weight gradient
  if (!wide_grad(Fin, Fout, contract)) return nchunks_for(rows);
  const int64_t cr = split_chunk_rows(rows, Fin, Fout);
  return rows > 0 ? (rows + cr - 1) / cr : 1;
}

// stage 1 of grad_W into `dst` ([slots, Fout * Fin]; slots = nchunks_w)
int launch_grad_w(const float *g, const float *x, float *dst, int64_t rows, int64_t Fin, int64_t Fout, int64_t ldg,
                  int64_t ldx, int contract, hipStream_t st, float *bdst = nullptr) {
  if (wide_grad(Fin, Fout, contract)) {
    const int64_t nc = nchunks_w(rows, Fin, Fout, contract);
    const int cr = split_chunk_rows(rows, Fin, Fout);
    if (nc > 0x7fffffffLL || ldg * 4 * cr >= (int64_t(1) << 31) || ldx * 4 * cr >= (int64_t(1) << 31)) return GWEN_ERANGE;
    if (!gwen_aligned(g, 16) || !gwen_aligned(x, 16) || ldg % 4 || ldx % 4) {     // (the DMA moves 16-byte pieces)
      if (bdst) return GWEN_EINVAL;                 // the fused column sums exist on the LDS-staged kernel only
      const int tiles_i = (int)(Fin / 64), ntiles = (int)(Fin / 64 * (Fout / 64));
      if ((ntiles + 3) / 4 > 65535) return GWEN_ERANGE;
      dim3 grid((unsigned)nc, (unsigned)((ntiles + 3) / 4));
      if (contract == GWEN_CONTRACT_BF16X6)
        k_grad_w_split<3><<<grid, kThreads, 0, st>>>(g, x, dst, rows, (int)Fin, (int)Fout, ldg, ldx, tiles_i, ntiles, cr);
      else
        k_grad_w_split<2><<<grid, kThreads, 0, st>>>(g, x, dst, rows, (int)Fin, (int)Fout, ldg, ldx, tiles_i, ntiles, cr);
      GWEN_LAUNCH_CHECK();
      return GWEN_OK;
    }
#define GWEN_GW(NS_, BO_, BI_)                                                                                     \
  do {                                                                                                             \
    const int ti = (int)(Fin / BI_);                                                                               \
    const int64_t nt = (int64_t)ti * (Fout / BO_);                                                                 \
    if (nt > 65535) return GWEN_ERANGE;                                                                            \
    if (bdst)                                                                                                      \
      k_grad_w_lds<NS_, BO_, BI_, true><<<dim3((unsigned)nc, (unsigned)nt), (BO_ / 64) * (BI_ / 64) * 64, 0, st>>>(\
          g, x, dst, rows, (int)Fin, (int)Fout, ldg, ldx, ti, cr, bdst);                                           \
    else                                                                                                           \
    k_grad_w_lds<NS_, BO_, BI_><<<dim3((unsigned)nc, (unsigned)nt), (BO_ / 64) * (BI_ / 64) * 64, 0, st>>>(        \
        g, x, dst, rows, (int)Fin, (int)Fout, ldg, ldx, ti, cr);                                                   \
  } while (0)
    const bool o128 = Fout % 128 == 0, i128 = Fin % 128 == 0, x6 = contract == GWEN_CONTRACT_BF16X6;
    if (o128 && i128) { if (x6) GWEN_GW(3, 128, 128); else GWEN_GW(2, 128, 128); }
    else if (o128) { if (x6) GWEN_GW(3, 128, 64); else GWEN_GW(2, 128, 64); }
    else if (i128) { if (x6) GWEN_GW(3, 64, 128); else GWEN_GW(2, 64, 128); }
    else { if (x6) GWEN_GW(3, 64, 64); else GWEN_GW(2, 64, 64); }
#undef GWEN_GW
    GWEN_LAUNCH_CHECK();
    return GWEN_OK;
  }
  if (bdst) return GWEN_EINVAL;
  const int64_t nc = nchunks_for(rows);
  const int tiles_i = (int)((Fin + 31) / 32), tiles_o = (int)((Fout + 31) / 32);
  const int64_t ntiles = (int64_t)tiles_i * tiles_o;
  if (ntiles > 4 * 65535LL || nc > 0x7fffffffLL) return GWEN_ERANGE;
  dim3 grid((unsigned)nc, (unsigned)((ntiles + 3) / 4));
  if (ntiles == 1) k_grad_w_few<4><<<grid, kThreads, 0, st>>>(g, x, dst, rows, (int)Fin, (int)Fout, ldg, ldx, tiles_i, (int)ntiles);
  else if (ntiles == 2) k_grad_w_few<2><<<grid, kThreads, 0, st>>>(g, x, dst, rows, (int)Fin, (int)Fout, ldg, ldx, tiles_i, (int)ntiles);
  else k_grad_w<<<grid, kThreads, 0, st>>>(g, x, dst, rows, (int)Fin, (int)Fout, ldg, ldx, tiles_i, (int)ntiles);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

// k_reduce_chunks for up to GWEN_MAX_REDUCE_TASKS (partial, dst, count, nchunks) tasks in ONE launch: the
// finish stages of every grad_W / grad_b of a backward pass (blockIdx.y = task).
struct TaskTable { gwen_reduce_task t[GWEN_MAX_REDUCE_TASKS]; };
__global__ __launch_bounds__(kThreads) void k_reduce_tasks(TaskTable tab) {
  __shared__ float red[kThreads];
  const gwen_reduce_task tk = tab.t[blockIdx.y];
  const int64_t count = tk.count;
  if ((int64_t)blockIdx.x * 16 >= count) return;
  const int nchunks = (int)tk.nchunks;
  const int e = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int64_t j = (int64_t)blockIdx.x * 16 + e, jc = j < count ? j : count - 1;
  float s = 0.0f;
  for (int c = ph; c < nchunks; c += 16 * kU) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int cc = c + 16 * u < nchunks ? c + 16 * u : nchunks - 1;
      v[u] = tk.partial[(int64_t)cc * count + jc];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (c + 16 * u < nchunks) s = s + v[u];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (ph == 0 && j < count) {
    float tot = red[e];
#pragma unroll
    for (int p = 1; p < 16; ++p) tot = tot + red[16 * p + e];
    tk.dst[j] = tot;
  }
}

struct TransposeTable { const float *w[GWEN_MAX_REDUCE_TASKS]; float *wt[GWEN_MAX_REDUCE_TASKS];
                        int rows[GWEN_MAX_REDUCE_TASKS], cols[GWEN_MAX_REDUCE_TASKS]; };
__global__ void k_transpose_tasks(TransposeTable tab) {            // wt [cols, rows] = w [rows, cols]^T
  const int k = blockIdx.y, rows = tab.rows[k], cols = tab.cols[k];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) tab.wt[k][(i % cols) * rows + i / cols] = tab.w[k][i];
}

}  // namespace

extern "C" int64_t gwen_gcn_grad_workspace_floats(int64_t rows, int64_t Fin, int64_t Fout) {
  const int64_t nc = nchunks_for(rows);
  const int64_t w = nc > 1 ? nc * Fin * Fout : 0;
  const int64_t b = nc > 1 ? nc * (Fout > Fin ? Fout : Fin) : 0;
  return (w > b ? w : b) + 1;
}

extern "C" int gwen_gcn_grad_weight_f32(const float *g, const float *x, float *grad_W,
                                        int64_t rows, int64_t Fin, int64_t Fout, int64_t ldg,
                                        int64_t ldx, float *partial, int contract, gwen_stream_t stream_) {
  if (rows < 0 || Fin < 0 || Fout < 0 || ldg < Fout || ldx < Fin || contract < 0 || contract > GWEN_CONTRACT_F16X3)
    return GWEN_EINVAL;
  if (Fin == 0 || Fout == 0) return GWEN_OK;
  if (!grad_W || (rows > 0 && (!g || !x))) return GWEN_EINVAL;
  if (Fin >= (1 << 30) || Fout >= (1 << 30) || (Fin * Fout + 15) / 16 > 0x7fffffffLL) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  contract = gwen_dense_contract(contract);
  if (rows == 0) return (int)hipMemsetAsync(grad_W, 0, sizeof(float) * Fin * Fout, st);
  const int64_t nc = nchunks_w(rows, Fin, Fout, contract);
  if (nc > 1 && !partial) return GWEN_EINVAL;
  float *dst = nc > 1 ? partial : grad_W;
  const int rc = launch_grad_w(g, x, dst, rows, Fin, Fout, ldg, ldx, contract, st);
  if (rc != GWEN_OK) return rc;
  if (nc > 1) {
    const int64_t count = Fin * Fout;
    k_reduce_chunks<<<(unsigned)((count + 15) / 16), kThreads, 0, st>>>(partial, grad_W, count, (int)nc);
    GWEN_LAUNCH_CHECK();
  }
  return GWEN_OK;
}

extern "C" int gwen_gcn_grad_bias_f32(const float *g, float *grad_b, int64_t rows, int64_t F,
                                      int64_t ldg, float *partial, gwen_stream_t stream_) {
  if (rows < 0 || F < 0 || ldg < F) return GWEN_EINVAL;
  if (F == 0) return GWEN_OK;
  if (!grad_b || (rows > 0 && !g)) return GWEN_EINVAL;
  hipStream_t st = gwen_stream(stream_);
  const int64_t nc = nchunks_for(rows);
  if (nc > 1 && !partial) return GWEN_EINVAL;
  if ((F + 63) / 64 > 65535 || nc > 0x7fffffffLL) return GWEN_ERANGE;
  float *dst = nc > 1 ? partial : grad_b;
  dim3 grid((unsigned)nc, (unsigned)((F + 63) / 64));
  k_grad_b<<<grid, kThreads, 0, st>>>(g, dst, rows, (int)F, ldg);
  GWEN_LAUNCH_CHECK();
  if (nc > 1) {
    k_reduce_chunks<<<(unsigned)((F + 15) / 16), kThreads, 0, st>>>(partial, grad_b, F, (int)nc);
    GWEN_LAUNCH_CHECK();
  }
  return GWEN_OK;
}

extern "C" int gwen_relu_backward_f32(const float *y, const float *g, float *gin, int64_t count,
                                      gwen_stream_t stream_) {
  if (count < 0) return GWEN_EINVAL;
  if (count == 0) return GWEN_OK;
  if (!y || !g || !gin) return GWEN_EINVAL;
  const int64_t blocks = (count + 255) / 256;
  if (blocks > 0x7fffffffLL) return GWEN_ERANGE;
  k_relu_bwd<<<(unsigned)blocks, 256, 0, gwen_stream(stream_)>>>(y, g, gin, count);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

// ---- building blocks of the whole-stack backward (gwen_gnn_backward_f32) --------------------------------------
// Stage 1 only: per-chunk partial sums [nchunks, Fout * Fin] / [nchunks, F] (nchunks = gwen_gcn_grad_chunks(rows));
// the fixed-order finish of MANY such reductions is one gwen_reduce_chunks_batched launch.
extern "C" int64_t gwen_gcn_grad_chunks(int64_t rows) { return nchunks_for(rows); }
extern "C" int64_t gwen_gcn_grad_weight_chunks(int64_t rows, int64_t Fin, int64_t Fout, int contract) {
  return nchunks_w(rows, Fin, Fout, gwen_dense_contract(contract));
}

extern "C" int gwen_gcn_grad_weight_partial_f32(const float *g, const float *x, float *partial, int64_t rows,
                                                int64_t Fin, int64_t Fout, int64_t ldg, int64_t ldx, int contract,
                                                gwen_stream_t stream_) {
  if (rows <= 0 || Fin <= 0 || Fout <= 0 || ldg < Fout || ldx < Fin || !g || !x || !partial || contract < 0 ||
      contract > GWEN_CONTRACT_F16X3)
    return GWEN_EINVAL;
  if (Fin >= (1 << 30) || Fout >= (1 << 30)) return GWEN_ERANGE;
  return launch_grad_w(g, x, partial, rows, Fin, Fout, ldg, ldx, gwen_dense_contract(contract), gwen_stream(stream_));
}

extern "C" int gwen_gcn_grad_weight_bias_supported(int64_t Fin, int64_t Fout, int contract) {
  return contract >= 0 && contract <= GWEN_CONTRACT_F16X3 && wide_grad(Fin, Fout, gwen_dense_contract(contract)) ? 1 : 0;
}

extern "C" int gwen_gcn_grad_weight_bias_partial_f32(const float *g, const float *x, float *partial_w, float *partial_b,
                                                     int64_t rows, int64_t Fin, int64_t Fout, int64_t ldg, int64_t ldx,
                                                     int contract, gwen_stream_t stream_) {
  if (rows <= 0 || Fin <= 0 || Fout <= 0 || ldg < Fout || ldx < Fin || !g || !x || !partial_w || !partial_b ||
      contract < 0 || contract > GWEN_CONTRACT_F16X3)
    return GWEN_EINVAL;
  if (Fin >= (1 << 30) || Fout >= (1 << 30)) return GWEN_ERANGE;
  if (!gwen_gcn_grad_weight_bias_supported(Fin, Fout, contract)) return GWEN_EINVAL;
  return launch_grad_w(g, x, partial_w, rows, Fin, Fout, ldg, ldx, gwen_dense_contract(contract), gwen_stream(stream_),
                       partial_b);
}

extern "C" int gwen_gcn_grad_bias_partial_f32(const float *g, float *partial, int64_t rows, int64_t F,
                                              int64_t ldg, gwen_stream_t stream_) {
  if (rows <= 0 || F <= 0 || ldg < F || !g || !partial) return GWEN_EINVAL;
  const int64_t nc = nchunks_for(rows);
  if ((F + 63) / 64 > 65535 || nc > 0x7fffffffLL) return GWEN_ERANGE;
  dim3 grid((unsigned)nc, (unsigned)((F + 63) / 64));
  k_grad_b<<<grid, kThreads, 0, gwen_stream(stream_)>>>(g, partial, rows, (int)F, ldg);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int gwen_reduce_chunks_batched(const gwen_reduce_task *tasks, int32_t n_tasks, gwen_stream_t stream_) {
  if (n_tasks < 0 || n_tasks > GWEN_MAX_REDUCE_TASKS || (n_tasks > 0 && !tasks)) return GWEN_EINVAL;
  if (n_tasks == 0) return GWEN_OK;
  TaskTable tab;
  int64_t cmax = 0;
  for (int i = 0; i < n_tasks; ++i) {
    if (!tasks[i].partial || !tasks[i].dst || tasks[i].count <= 0 || tasks[i].nchunks <= 0 ||
        tasks[i].nchunks > 0x7fffffffLL)
      return GWEN_EINVAL;
    tab.t[i] = tasks[i];
    if (tasks[i].count > cmax) cmax = tasks[i].count;
  }
  if ((cmax + 15) / 16 > 0x7fffffffLL) return GWEN_ERANGE;
  dim3 grid((unsigned)((cmax + 15) / 16), (unsigned)n_tasks);
  k_reduce_tasks<<<grid, kThreads, 0, gwen_stream(stream_)>>>(tab);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int gwen_transpose_batched(const float *const *w, float *const *wt, const int32_t *rows,
                                      const int32_t *cols, int32_t n, gwen_stream_t stream_) {
  if (n < 0 || n > GWEN_MAX_REDUCE_TASKS || (n > 0 && (!w || !wt || !rows || !cols))) return GWEN_EINVAL;
  if (n == 0) return GWEN_OK;
  TransposeTable tab;
  int cmax = 0;
  for (int i = 0; i < n; ++i) {
    if (!w[i] || !wt[i] || rows[i] <= 0 || cols[i] <= 0 || (int64_t)rows[i] * cols[i] > 0x7fffffffLL)
      return GWEN_EINVAL;
    tab.w[i] = w[i]; tab.wt[i] = wt[i]; tab.rows[i] = rows[i]; tab.cols[i] = cols[i];
    if (rows[i] * cols[i] > cmax) cmax = rows[i] * cols[i];
  }
  dim3 grid((unsigned)((cmax + 255) / 256), (unsigned)n);
  k_transpose_tasks<<<grid, 256, 0, gwen_stream(stream_)>>>(tab);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
