// K8 -- a whole WIDE GCNConv layer (+ReLU) per launch, tile-staged:   out = act( (A~ x) W^T + bias )
//
// Same contract as K4 (layer.hip; replaces lin -> index_select -> mul -> scatter_add_ -> + bias -> relu of
// torch-geometric 2.3.1 GCNConv.forward as called from /root/reference/src/gwen/models_gnn.py:147-149,
// :204-206), built for the regime K4 is weakest in: Fin >= 64 on a locality-ordered bounded-degree mesh,
// where K4 pulls every gathered source row (~7 per destination row) through the vector L1 path
// (measured 50-70 GB/s per CU: 49 us per 256-wide pass in cache, 2.9 TB/s of compulsory traffic once the
// working set leaves the Infinity Cache).  Here a block owns TILES of 64 destination rows
// (gwen_gcn_tiles64: per tile the union of the source rows its entries name, ~116 rows on the nu = 100
// mesh in Morton order, and a 16-bit local id per entry) and
//   * stages the union ONCE in LDS, 64 features (256 B per row) at a time, by LDS-DMA
//     (global_load_lds_dwordx4: no VGPR round trip, 1 KiB per wave instruction) -- 1.8x the compulsory
//     bytes through the L1 path instead of 7x; the entries' weights and local ids ride the same DMA;
//   * aggregates the 64 destination rows from LDS (8 ds_read_b128 per lane, fma chain in stored order:
//     the same order and rounding as K4's gather), splits the sums into bf16 hi/lo images;
//   * contracts the 64 x 64 chunk with W on the matrix cores (3xbf16, fp32 accumulate, as K4) while the
//     DMA of the chunk after next is in flight and the next chunk is being aggregated;
//   * adds bias / ReLU and stores after the last chunk.
// One block per CU (16 waves; 8 at Fin = 256, where W alone takes 128 registers per wave) walks its tiles
// persistently; ONE barrier per 64-feature chunk ("step"):
//     [scalar loads of the row ids of chunk s+D+1 -- EARLYR: in the step before]
//     s_waitcnt vmcnt(n): DMA of chunk s+1 landed;  barrier
//     2 x (row tiles per wave) regions, each: a share of the step's memory instructions -- the DMAs of chunk
//     s+D+1 -> stage[s % (D+1)] and the stores of the tile finished in step s-1 --, the LDS reads of a share of
//     aggregate(s+1), one (row tile, k-step) unit of mfma(s), that share's VALU work woven behind the MFMAs.
// One chunk of DMA is in flight behind the chunk being aggregated; 128 union slots are staged per chunk when
// the graph's unions allow (the cube-sphere Hilbert node order keeps the nu = 100 mesh at <= 121), 192 otherwise.
// The DMAs are issued by inline asm, so hipcc neither counts nor drains them (cdna_hip_programming.md
// section 5 "Pipelining across barriers"); every other global read of the loop is a scalar load (the union
// rows of the tile) and the only vector-memory instructions hipcc sees are the output stores.
// XCD-aware: the tile list (members x tiles) is cut into 8 contiguous ranges, one per blockIdx % 8, and
// the blocks of an XCD walk their range interleaved, so the halos of concurrently staged tiles meet in
// that XCD's L2.
// Round 3 (256 channels, bf16x3, unions <= 128 rows; DESIGN.md section 4 carries the measurements):
//   SHIFT  -- the vector work of an aggregate stage runs one region behind the stage's LDS reads (two row buffers);
//   SKEW   -- row tile t of a tile runs t steps behind row tile 0 (its A slices wait in a ring), the step's units start
//             with the row tile on its last chunk, and that ONE row tile is stored per step behind the step's DMAs:
//             a CU takes ~58 cycles per store instruction and its memory pipe is in order, so 8 stores per wave in one
//             step stood ~3 500 cycles in front of that step's DMAs;
//   window / EARLYR (every width) -- (member, tile) pairs of tiles i-1 .. i+HI split once per tile and carried in scalar
//             registers; a tile's row ids loaded once, behind the last DMA of the step before, instead of in front of every
//             barrier's fence;
//   VOFFQ / PERM -- vector instructions taken out of the step (the matrix pipe and the vector ALU of a SIMD take turns:
//             their times add): DMA row offsets once per tile, a gathered row's LDS address by one v_perm_b32.
// Numerics: identical to K4's split path term for term (same aggregation order, same split, same MFMA
// sequence per output element) -- tests compare the two bitwise.
#include "common.h"
#include "split.h"
#include <type_traits>

namespace {

// ---- choices of the shipped kernel, each the winner of an A/B on one box (DESIGN.md section 4 "K8" has the numbers; the
// lab version of this file -- timing ablations, in-kernel s_memtime stamps, the dropped alternatives as -D knobs -- is
// tools/experiments/wide_lab.patch, applied on top of this file by tools/experiments/k8_variants.sh) ------------------
constexpr int kWeaveValu = 2;        // vector instructions woven behind every MFMA of a region
constexpr int kEarlyRegion = 6;      // the region that loads the next tile's row ids (205.5-207 us; region 4: 208.8, 7: 207.5-208)
constexpr int kPrefRows = 300000;    // rows x members from which the planner takes K8 at 64 channels (below: K4)
// chunks of DMA in flight behind the one being aggregated: 2 at 64 channels (a step is one whole tile and short), 1 above
constexpr int dma_depth(int fin) { return fin == 64 ? 2 : 1; }
// bf16x6 at 128 input channels: 8 waves of 256 registers for 128 -> 256 (16 waves spill 9-14 there: 464 us against 376 on
// 8 members), 16 waves of 128 for 128 -> 64 / 128
constexpr int x6_waves_128(int fout) { return fout == 256 ? 8 : 16; }

using gwen::bf16x4;
using gwen::bf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = GWEN_TILE_ROWS;        // 64 destination rows per tile
constexpr int kUCap = GWEN_TILE_UNION;       // 192 union slots per tile in t_rows
constexpr int kFC = 64;                      // features per chunk
// A-chunk row pitch in bf16: 80 (160 B: conflict-free 16-B fragment reads); 72 (144 B: two-way conflicts on a third
// of the fragment reads) where three images AND a third stage buffer must share the CU's 160 KiB -- bf16x6 with two
// chunks of DMA in flight, i.e. 64 channels, where a wave reads 6 fragments per tile and the conflicts do not show
constexpr int pitch_a(int ns, int d) { return ns == 3 && d == 2 ? 72 : 80; }
constexpr int kEntBytes = kRows * 8 * 4 + kRows * 8 * 2;   // weights fp32 + local ids u16 = 3072
// LDS of one block: (D + 1) stage buffers of KU slots x 256 B | 2 A chunks of NS images | 3 entry sets | bias
constexpr int lds_bytes(int nstg, int ku, int ns) {
  return nstg * ku * kFC * 4 + 2 * ns * (kRows * pitch_a(ns, nstg - 1) * 2) + 3 * kEntBytes + 1024;
}

// aggregated rows -> NS bf16 images (split.h)
template <int NS>
__device__ inline void split4n(const float4_t &a, bf16x4 (&im)[NS]) {
  const float f4[4] = {a[0], a[1], a[2], a[3]};
  gwen::split_images<4, NS>(f4, im);
}

// ReLU of four values.  `x < 0 ? 0 : x` keeps NaN (as torch.relu does) but is a compare + select per element;
// max(bits(x), 0) on the bits as SIGNED integers is one v_max_i32: a float with the sign bit clear is a
// non-negative integer and stays, one with the sign bit set is a negative integer and becomes +0 -- identical for
// every finite value, -0 and +-Inf, and a NaN with a clear sign bit (the one arithmetic produces here: Inf - Inf,
// 0 * Inf give 0x7fc00000 on gfx950) stays a NaN; only a NaN carrying a SET sign bit would read as 0.
// `floor_bits` = 0 with the ReLU, INT_MIN without (max with INT_MIN changes nothing): no branch, no select.
__device__ inline float4_t relu4(float4_t v, int relu, int floor_bits) {
  typedef int int4_t __attribute__((ext_vector_type(4)));
  int4_t b = __builtin_bit_cast(int4_t, v);
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = b[e] > floor_bits ? b[e] : floor_bits;
  return __builtin_bit_cast(float4_t, b);
}

// acc + w * v on four lanes of a row, as four v_fma_f32 with the weight as an operand (the vector form makes hipcc emit
// v_pk_fma_f32 + a move per weight: 1.3-1.7 % slower at 256 channels, A/B on one box; the same bits either way)
__device__ inline float4_t fma4(float w, const float4_t &v, float4_t acc) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r = acc[e];
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(w), "v"(v[e]));      // opaque to the vectoriser
    acc[e] = r;
  }
  return acc;
}

template <int N, typename F, int I = 0>
__device__ inline void gwen_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    gwen_static_for<N, F, I + 1>(static_cast<F &&>(f));
  }
}

// a wave-uniform pointer, pinned to SGPRs (an "s" asm operand is not moved there by hipcc when instruction
// selection happened to compute it on the vector ALU)
__device__ inline const char *uniform_ptr(const void *p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return reinterpret_cast<const char *>(((uint64_t)hi << 32) | lo);
}

// one LDS-DMA wave instruction: lane l copies 16 B from base + voff(l) to LDS byte address dst + 16 l.
// (non-temporal staging loads measured slower where it matters -- 256 -> 256 x 4 members 242.0 -> 249.2 us, 64 -> 64 x 16
//  bf16x3 198.0 -> 203.0: the halo rows neighbouring tiles share are re-read from L2 shortly after, and nt lines leave first)
__device__ inline void glds16(const void *base, uint32_t voff, uint32_t dst) {
  uint32_t keep;       // M0 is compiler-reserved: save and restore it inside the statement
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ inline void wait_vmcnt(int n) {
  switch (n) {
#define GWEN_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    GWEN_VM(1) GWEN_VM(2) GWEN_VM(3) GWEN_VM(4) GWEN_VM(5) GWEN_VM(6) GWEN_VM(7) GWEN_VM(8) GWEN_VM(9)
    GWEN_VM(10) GWEN_VM(11) GWEN_VM(12) GWEN_VM(13) GWEN_VM(14) GWEN_VM(15) GWEN_VM(16) GWEN_VM(17)
    GWEN_VM(18) GWEN_VM(19) GWEN_VM(20) GWEN_VM(21) GWEN_VM(22) GWEN_VM(23) GWEN_VM(24)
#undef GWEN_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// D = chunks of DMA in flight behind the one being aggregated: D = 2 needs unions of at most 128 rows
// (3 stage buffers of 128 slots), D = 1 takes unions up to 192 (2 buffers of 192 slots).
// DENSE: no graph -- the "union" of a tile is its own 64 rows and the aggregate is the row itself: the kernel is
// then K3's 3xbf16 projection out = act(x W^T + b) for tall inputs (gwen_gcn_linear_f32 sends them here), with
// x rows at pitch ldx.
// F16: the fp32-class contraction on two SCALED fp16 images per operand (split.h "f16x3"): NS = 2's registers, LDS and MFMA
// count.  The aggregator scales every (row, 64-feature chunk) by a power of two before cutting it and leaves the exponent
// and the step from the row's previous chunk in the 32 pad bytes of the row's first image ({e, delta} at byte 128); the
// matrix side multiplies a row tile's accumulators by 2^delta before the first MFMA of a chunk (lane = destination row in
// the D layout, so the factor is per lane) and the epilogue un-scales by the row's last exponent and the column's.
template <int FIN, int FOUT, int NW, int D, int KU, bool EARLY, bool DENSE, int NS, bool F16>
__global__ __launch_bounds__(NW * 64) void k_wide(
    const int32_t *__restrict__ t_rows, const uint16_t *__restrict__ t_lid,
    const float *__restrict__ t_val, const float *__restrict__ x, const float *__restrict__ W,
    const float *__restrict__ bias, float *__restrict__ out, int32_t N, int32_t T, int32_t G,
    int64_t ldo, int64_t mstride_x, int64_t mstride_o, int relu, int32_t ldx, int nt) {
  // ROLES (below): the two waves of a SIMD take the step's two halves in opposite order
  constexpr bool ROLES = FIN == 64 && D == 1 && !DENSE && !F16;      // (measured faster there only; 256 channels: 267.7 vs 265.5 us)
  constexpr int NSTG = D + 1;                           // stage buffers
  constexpr int kStageBytes = KU * kFC * 4;
  constexpr int NC = FIN / kFC;                         // chunks per tile
  constexpr int NJ = FOUT / 16;                         // 16-column output tiles
  constexpr int CT = NJ >= NW ? NJ / NW : 1;            // column tiles per wave
  constexpr int TSTEP = NJ >= NW ? 1 : NW / NJ;         // waves sharing a column tile split the row tiles
  constexpr int NTT = 4 / TSTEP;                        // row tiles per wave
  constexpr int KS = FIN / 32;                          // MFMA k-steps over the whole Fin
  constexpr int NQ = KU / (4 * NW);                     // DMA wave instructions per wave and chunk
  constexpr int NP = kRows / (4 * NW);                  // aggregate passes per wave and chunk
  static_assert(NC >= 1 && 4 % TSTEP == 0 && (NJ % NW == 0 || NW % NJ == 0), "unsupported widths");
  static_assert(NQ * 4 * NW == KU && NP * 4 * NW == kRows, "waves must tile the union and the rows");
  constexpr int kPB = pitch_a(NS, D), kAImg = kRows * kPB * 2;     // one bf16 image of a chunk: 10240 B (9216)
  constexpr int kABytes = NS * kAImg;                   // one A chunk: NS images
  // SKEW (Fin = 256 -> 128 / 256, bf16x3, unions <= 128 rows): the matrix work of row tile t on chunk s' runs in step
  // s' + t, so that row tile t of a tile is complete -- and stored, two instructions per wave -- in step (t + 3) % 4 + 1
  // of the stream instead of all four row tiles (8 instructions per wave, 64 KiB per CU) in the tile's last step.
  // Measured before: the step that carried a tile's stores ran 4 600 cycles of regions against 1 400 for the others
  // (in-kernel stamps), and a build that stored nothing ran 167 us against 234 (256 -> 256 x 4 members): a CU drains
  // stores at ~32 GB/s and the in-order waves stand behind them.  The aggregate of a chunk is kept per ROW TILE in a
  // ring of t + 2 slices (written in step s' - 1, read in step s' + t): 14 slices of 5 KiB instead of 2 chunks of 20.
  // (the dense form -- K3's tall case, two chunks of DMA in flight over 64-slot buffers -- takes the same schedule)
  constexpr bool SKEW = !ROLES && NC == 4 && NTT == 4 && TSTEP == 1 && NS == 2 &&
                        (DENSE ? (D == 2 && KU == kRows) : (D == 1 && KU == 128));
  constexpr int kSlImg = 16 * kPB * 2, kSl = NS * kSlImg;        // one image / all images of a row tile's slice of a chunk
  constexpr int kImgStride = SKEW ? kSlImg : kAImg;
  constexpr int kOffStage = 0, kOffA = NSTG * kStageBytes, kOffEnt = kOffA + (SKEW ? 14 * kSl : 2 * kABytes);
  constexpr int kOffBias = kOffEnt + 3 * kEntBytes, kOffCs = kOffBias + 1024, kLds = kOffCs + (F16 ? 1024 : 0);
  static_assert(SKEW || kLds == lds_bytes(NSTG, KU, NS) + (F16 ? 1024 : 0), "LDS layout");
  static_assert(!F16 || (NS == 2 && !DENSE && !ROLES && kPB * 2 >= kFC * 2 + 8), "f16x3: two images, graph form, pad bytes for {e, delta}");
  constexpr int kSide = kFC * 2;                        // F16: byte offset of a row's {e, delta} in its first image's row
  auto ring_base = [](int ti) { return (ti * (ti + 3) / 2) * kSl; };     // rings of 2, 3, 4, 5 slices: first slice 0, 2, 5, 9
  static_assert(kLds <= 160 * 1024, "the block's LDS exceeds a CU's");
  static_assert(!DENSE || KU == kRows, "a dense tile stages its own rows");
  const uint32_t row_pitch = DENSE ? (uint32_t)ldx * 4u : (uint32_t)(FIN * 4);
  const int floor_bits = relu ? 0 : (int)0x80000000;     // relu4
  __shared__ __attribute__((aligned(1024))) char lds[kLds];
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int mi = lane & 15, mh = lane >> 4;

  // ---- this block's tiles: the tile list is cut into one contiguous range per XCD (blockIdx % 8; with
  // fewer than 8 blocks, per block) and the blocks of an XCD walk their range interleaved -----------------
  const int nx = gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = blockIdx.x % nx, idx = blockIdx.x / nx;
  const int nbx = (int)gridDim.x / nx + (xcd < (int)gridDim.x % nx ? 1 : 0);
  const int g0 = (int)((int64_t)xcd * G / nx), g1 = (int)((int64_t)(xcd + 1) * G / nx);
  if (g0 + idx >= g1) return;                                  // whole block: no barrier reached yet
  const int ntl = (g1 - g0 - idx + nbx - 1) / nbx;             // >= 1
  auto tile_of = [&](int i) { return g0 + idx + (i < ntl ? i : ntl - 1) * nbx; };   // clamped

  // (member, tile in member) of global tile g without an integer division (G < 2^31, members < 2^23)
  const float inv_t = 1.0f / (float)T;
  auto split_tile = [&](int g, int &m, int &t) {
    m = __builtin_amdgcn_readfirstlane((int)((float)g * inv_t));     // g is wave-uniform
    t = g - m * T;
    if (t < 0) { --m; t += T; }
    if (t >= T) { ++m; t -= T; }
    m = __builtin_amdgcn_readfirstlane(m);
    t = __builtin_amdgcn_readfirstlane(t);
  };

  // ---- W fragments (this wave's CT x 16 output columns, all of Fin) -- as K4; bias -> LDS ---------------
  const int jw = NJ >= NW ? wave : wave % NJ;                  // this wave's column tiles: CT jw .. + CT - 1
  const int tt0 = NJ >= NW ? 0 : wave / NJ;                    // its first row tile
  bf16x8 bw[CT][KS][NS];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int j = CT * jw + ct;                                // adjacent tiles: full 128-B lines per row
    const float *wrow = W + (int64_t)(j * 16 + mi) * FIN;
    int kw = 0;                                                // F16: this output column's scale is 2^kw
    if constexpr (F16) {
      // the column's largest |w| (its Fin values sit in the four lanes mi, mi + 16, mi + 32, mi + 48); W is read twice
      // (cache-hot) rather than kept raw in registers beside its images
      float m = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float *wp = wrow + 8 * (4 * ks + mh);
        const float4_t w0 = *reinterpret_cast<const float4_t *>(wp);
        const float4_t w1 = *reinterpret_cast<const float4_t *>(wp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(w0[e]), __builtin_fabsf(w1[e])));
      }
      m = __builtin_fmaxf(m, __shfl_xor(m, 16));
      m = __builtin_fmaxf(m, __shfl_xor(m, 32));
      int ew = gwen::f16_exp_of(m);
      ew = ew < gwen::kF16Floor ? gwen::kF16Floor : ew;
      kw = gwen::kF16Top - ew;
      // the epilogue's column factor 2^-kw (a normal number: ew >= 20)
      if (mh == 0) reinterpret_cast<float *>(lds + kOffCs)[j * 16 + mi] = __builtin_bit_cast(float, (ew - (gwen::kF16Top - 127)) << 23);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float *wp = wrow + 8 * (4 * ks + mh);
      const float4_t w0 = *reinterpret_cast<const float4_t *>(wp);
      const float4_t w1 = *reinterpret_cast<const float4_t *>(wp + 4);
      const float w8[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
      if constexpr (F16) gwen::split_f16<8>(w8, kw, bw[ct][ks][0], bw[ct][ks][1]);
      else gwen::split_images<8, NS>(w8, bw[ct][ks]);
    }
  }
  {
    float *bl = reinterpret_cast<float *>(lds + kOffBias);
    for (int f = threadIdx.x; f < FOUT; f += NW * 64) bl[f] = bias ? bias[f] : 0.0f;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)       // every global load above is waited for here, not inside the pipeline
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) asm volatile("" : "+v"(bw[ct][ks][s_]));

  // ---- F16: scale + cut of an aggregated row piece; the epilogue's un-scale -----------------------------------
  // eprev[p]: the biased exponent this lane's row of aggregate pass p was scaled by in the chunk before (a lane
  // aggregates the same rows of every chunk of a tile)
  int eprev[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) eprev[p] = gwen::kF16Floor;
  // the images of this lane's 4 features of a row chunk -> `a` (first image; the others `stride` bytes apart).  F16: the
  // row chunk's largest |value| (16 lanes of a DPP row hold its 64 features) picks the scale 2^(141 - e); a chunk is never
  // scaled more than 2^16 finer than the chunk before it (the accumulators grow by at most that factor when they are
  // re-expressed), `first` = a tile's first chunk (nothing accumulated yet)
  auto put_images = [&](const float4_t &acc, char *a, int stride, int p, bool first) {
    if constexpr (F16) {
      float m3, m;
      asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m3) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
      asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(m) : "v"(m3), "v"(acc[3]));
      int e = gwen::f16_exp_of(m);
      e = max(e, __builtin_amdgcn_update_dpp(0, e, 0xb1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
      e = max(e, __builtin_amdgcn_update_dpp(0, e, 0x4e, 0xf, 0xf, true));       // quad_perm [2,3,0,1]
      e = max(e, __builtin_amdgcn_update_dpp(0, e, 0x141, 0xf, 0xf, true));      // row_half_mirror
      e = max(e, __builtin_amdgcn_update_dpp(0, e, 0x140, 0xf, 0xf, true));      // row_mirror: all 16 lanes hold the row's
      // a fresh scale leaves kF16Guard binades of headroom, and a chunk whose maximum fits under the scale of the chunk
      // before it -- and is not more than kF16Keep binades below it -- KEEPS that scale (delta = 0): the matrix side then
      // has nothing to re-express, which it checks per wave (one compare + branch instead of 4 CT multiplications)
      int ecur = max(e + gwen::kF16Guard, gwen::kF16Floor);
      int delta = 0;
      if (!first) {
        const int ep = eprev[p];
        const bool keep = e <= ep && e + gwen::kF16Keep >= ep;
        ecur = keep ? ep : max(ecur, ep - gwen::kF16Back);
        delta = ep - ecur;
      }
      eprev[p] = ecur;
      const float f4[4] = {acc[0], acc[1], acc[2], acc[3]};
      bf16x4 h, l;
      gwen::split_f16<4>(f4, gwen::kF16Top - ecur, h, l);
      *reinterpret_cast<bf16x4 *>(a) = h;
      *reinterpret_cast<bf16x4 *>(a + stride) = l;
      typedef int int2_t __attribute__((ext_vector_type(2)));
      if (mi == 0) *reinterpret_cast<int2_t *>(a + kSide) = int2_t{ecur, delta};     // (mi == 0: `a` is the row's first byte)
    } else {
      bf16x4 im[NS];
      split4n<NS>(acc, im);
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) *reinterpret_cast<bf16x4 *>(a + s_ * stride) = im[s_];
    }
  };
  // accumulators of 4 output columns (from col0) of a destination row -> the layer's output before the ReLU.  F16: the
  // accumulators are in the scale of the row's last chunk (exponent er) and of the columns (factors in LDS): two exact
  // multiplications by powers of two, the second fused with the bias add (one rounding, as d + bias)
  auto finish4 = [&](const f32x4 &dv, int er, int col0, const float4_t &b4) -> float4_t {
    if constexpr (F16) {
      const float4_t cs = *reinterpret_cast<const float4_t *>(lds + kOffCs + col0 * 4);
      float4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(__builtin_ldexpf(dv[e], er - gwen::kF16Top), cs[e], b4[e]);
      return o;
    } else {
      return float4_t{dv[0], dv[1], dv[2], dv[3]} + b4;
    }
  };

  // ---- pipeline pieces ------------------------------------------------------------------------------------
  // DMA of (tile g, chunk c) into stage[sb]; with c == 0 also the tile's entry weights / local ids into
  // ent[eb].  Union slot k = 4 (NW q + wave) + (lane >> 4); a group of 4 slots past the union starts with
  // -1 and is skipped.  Returns the number of DMA instructions this wave issued.
  auto issue = [&](int g, int c, int sb, int eb) -> int {
    int m, t;
    split_tile(g, m, t);
    const int32_t *rp = t_rows + (int64_t)t * kUCap + 4 * wave;   // wave-uniform: scalar loads
    const char *xm = uniform_ptr(x + (int64_t)m * mstride_x);
    int n = 0;
    if constexpr (DENSE) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        int32_t row = t * kRows + 4 * (NW * q + wave) + mh;
        row = row < N ? row : N - 1;                             // rows past the end: read a valid row, never stored
        const uint32_t voff = (uint32_t)row * row_pitch + (uint32_t)(c * kFC * 4 + mi * 16);
        glds16(xm, voff, lds0 + kOffStage + sb * kStageBytes + 4 * (NW * q + wave) * (kFC * 4));
        ++n;
      }
      return n;
    }
    int32_t r[4 * NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) r[4 * q + k] = rp[4 * NW * q + k];
    // every row id in an SGPR before the first DMA (one batch of scalar loads, selects instead of branches)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      asm volatile("" : "+s"(r[4 * q]), "+s"(r[4 * q + 1]), "+s"(r[4 * q + 2]), "+s"(r[4 * q + 3]));
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (r[4 * q] >= 0) {
        int32_t row = r[4 * q];
        row = mh == 1 ? r[4 * q + 1] : row;
        row = mh == 2 ? r[4 * q + 2] : row;
        row = mh == 3 ? r[4 * q + 3] : row;
        const uint32_t voff = (uint32_t)row * (uint32_t)(FIN * 4) + (uint32_t)(c * kFC * 4 + mi * 16);
        glds16(xm, voff, lds0 + kOffStage + sb * kStageBytes + 4 * (NW * q + wave) * (kFC * 4));
        ++n;
      }
    }
    if (!DENSE && c == 0 && wave < 3) {
      const char *src = wave < 2 ? reinterpret_cast<const char *>(t_val) + (int64_t)t * (kRows * 32) + wave * 1024
                                 : reinterpret_cast<const char *>(t_lid) + (int64_t)t * (kRows * 16);
      src = uniform_ptr(src);
      glds16(src, (uint32_t)lane * 16, lds0 + kOffEnt + eb * kEntBytes + wave * 1024);
      ++n;
    }
    return n;
  };
  // aggregate one chunk: 64 rows x 16 lanes (4 features each) = 1024 lanes of work = NP passes per wave
  auto aggregate = [&](int sb, int ab, int eb) {
    const char *ent = lds + kOffEnt + eb * kEntBytes;
    const char *stg = lds + kOffStage + sb * kStageBytes + mi * 16;
    if constexpr (DENSE) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int lr = 4 * NW * p + 4 * wave + mh;
        const float4_t a4 = *reinterpret_cast<const float4_t *>(stg + lr * (kFC * 4));
        bf16x4 im[NS];
        split4n<NS>(a4, im);
        char *a = lds + kOffA + ab * kABytes + (lr * kPB + mi * 4) * 2;
        if constexpr (SKEW)                               // (the prologue's chunk 0: slot 0 of the row tile's ring)
          a = lds + kOffA + ring_base(2 * p + (wave >> 2)) + ((lr & 15) * kPB + mi * 4) * 2;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) *reinterpret_cast<bf16x4 *>(a + s_ * kImgStride) = im[s_];
      }
      return;
    }
    u32x4 lid4[NP];
    float4_t wgt[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p) {                      // entry weights / local ids of every pass first
      const int lr = 4 * NW * p + 4 * wave + mh;
      lid4[p] = *reinterpret_cast<const u32x4 *>(ent + kRows * 32 + lr * 16);
      wgt[p][0] = *reinterpret_cast<const float4_t *>(ent + lr * 32);
      wgt[p][1] = *reinterpret_cast<const float4_t *>(ent + lr * 32 + 16);
    }
    float4_t acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {                    // 4 NP row reads in flight, then their fma chains
      float4_t v[NP][4];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t pair = lid4[p][2 * hf + (u >> 1)];
          const uint32_t lid = (u & 1) ? (pair >> 16) : (pair & 0xffffu);
          v[p][u] = *reinterpret_cast<const float4_t *>(stg + lid * (kFC * 4));
        }
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float w = wgt[p][hf][u];
          acc[p] = fma4(w, v[p][u], acc[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int lr = 4 * NW * p + 4 * wave + mh;
      char *a = lds + kOffA + ab * kABytes + (lr * kPB + mi * 4) * 2;
      if constexpr (SKEW)                               // (the prologue's chunk 0: slot 0 of the row tile's ring)
        a = lds + kOffA + ring_base(2 * p + (wave >> 2)) + ((lr & 15) * kPB + mi * 4) * 2;
      put_images(acc[p], a, kImgStride, p, true);        // (only called for a tile's chunk 0)
    }
  };

  // the same, one pass after the other with half a row's entries in flight (what the register budget at
  // Fin = 256 leaves when nothing else of the step is interleaved: ROLES)
  auto aggregate_lean = [&](int sb, int ab, int eb) {
    const char *ent = lds + kOffEnt + eb * kEntBytes;
    const char *stg = lds + kOffStage + sb * kStageBytes + mi * 16;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int lr = 4 * NW * p + 4 * wave + mh;
      const u32x4 l4 = *reinterpret_cast<const u32x4 *>(ent + kRows * 32 + lr * 16);
      float4_t a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const float4_t w4 = *reinterpret_cast<const float4_t *>(ent + lr * 32 + 16 * hf);
        float4_t vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t pair = l4[2 * hf + (u >> 1)];
          const uint32_t lid = (u & 1) ? (pair >> 16) : (pair & 0xffffu);
          vv[u] = *reinterpret_cast<const float4_t *>(stg + lid * (kFC * 4));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          a4 = fma4(w4[u], vv[u], a4);
      }
      bf16x4 im[NS];
      split4n<NS>(a4, im);
      char *a = lds + kOffA + ab * kABytes + (lr * kPB + mi * 4) * 2;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) *reinterpret_cast<bf16x4 *>(a + s_ * kAImg) = im[s_];
    }
  };

  f32x4 d[CT][NTT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < NTT; ++i) d[ct][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int erow[NTT];                 // F16: exponent of the scale row tile i's accumulators are in once its last chunk ran
#pragma unroll
  for (int i = 0; i < NTT; ++i) erow[i] = gwen::kF16Top;

  // Stores of a finished tile, issued AFTER the step's DMAs so that a later top-of-step wait can leave them
  // in flight.  The waits count instructions, so the count must be exact: a tile whose 64 rows all exist is
  // stored with every lane active (CT * NTT store instructions); the last tile of a member is stored under
  // its row guard and drained at once.  Returns the number of stores left in flight.
  auto store_tile = [&](int g) -> int {
    int m, t;
    split_tile(g, m, t);
    float *om = out + (int64_t)m * mstride_o;
    const bool whole = (t + 1) * kRows <= N;
    const float *bl = reinterpret_cast<const float *>(lds + kOffBias);
#pragma unroll
    for (int i = 0; i < NTT; ++i) {
      const int r = t * kRows + (tt0 + i * TSTEP) * 16 + mi;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int j = CT * jw + ct;
        float4_t o = finish4(d[ct][i], erow[i], j * 16 + 4 * mh, *reinterpret_cast<const float4_t *>(bl + j * 16 + 4 * mh));
        o = relu4(o, relu, floor_bits);
        float *dst = om + (int64_t)r * ldo + j * 16 + 4 * mh;
        if (whole || r < N) *reinterpret_cast<float4_t *>(dst) = o;
        d[ct][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    (void)whole;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // everything this wave issued so far has completed
    return -1;
  };

  // ---- prologue: chunks 0 .. D in flight, aggregate(0) -------------------------------------------------
  // chunk s = (tile s / NC, chunk s % NC) lives in stage[s % NSTG], its aggregate in A[s % 2], its tile's
  // entries in ent[(s / NC) % 3]
#pragma unroll
  for (int s0 = 0; s0 <= D; ++s0) issue(tile_of(s0 / NC), s0 % NC, s0 % NSTG, (s0 / NC) % 3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (ROLES) aggregate_lean(0, 0, 0);
  else aggregate(0, 0, 0);

  // ---- steady state: interval s = i NC + c ---------------------------------------------------------------
  //   [scalar loads of the row ids of chunk s+D+1]      (their latency sits behind the wait and the barrier)
  //   wait: DMA(s+1) landed;  barrier
  //   NU = 2 NTT regions, one per (row tile, k-step) unit of mfma(s).  Region u:
  //       its share of the step's memory instructions -- the DMAs of chunk s+D+1 into stage[s % NSTG]
  //       (chunk s was aggregated before the barrier) and, with c == 0, the stores of tile i-1 (row tile t's
  //       stores before region 2t overwrites its accumulators) -- spread out because the CU's one vector
  //       memory pipe takes ~16 cycles per 1 KiB instruction: issued back to back by every wave after the
  //       barrier they cost each wave ~1 000 cycles of issue stall per step (in-kernel stamps);
  //       the LDS reads of its share of aggregate(s+1), the unit's MFMAs, that share's VALU work, woven
  //       (a few VALU instructions behind every MFMA) so that the matrix pipe paces the region.
  constexpr int NU = 2 * NTT;                          // regions per step
  constexpr int NSTAGE = 4 * NP;                       // aggregate stages: per pass 0 entries, 1 / 2 halves, 3 split
  int sb = 0;                 // s % NSTG
  int eb = 0;                 // i % 3
  int young = 0;              // operations that may stay in flight at the next wait
  int prev_ops = 0;           // what the interval before this one issued (D = 3)
  // (member, tile) of the tiles a step names -- i-1 (its stores) .. i+HI (the chunk the NEXT step issues) -- are split ONCE per
  // tile and carried in a window of scalar registers (the float reciprocal + readfirstlanes of split_tile twice per step sat
  // between the barrier and the step's first MFMA): w[k] = tile i - 1 + k
  constexpr int HI = (NC + D + 1) / NC, WN = HI + 2;
  int wm[WN], wt[WN];
#pragma unroll
  for (int k = 1; k < WN; ++k) split_tile(tile_of(k - 1), wm[k], wt[k]);
  wm[0] = wm[1]; wt[0] = wt[1];
  // EARLYR: the row ids of the chunk a step issues are loaded (scalar loads) in the step BEFORE, behind its last DMA,
  // instead of in front of the step's own barrier -- the barrier's fence waits for every outstanding scalar load
  // (lgkmcnt), i.e. their whole latency stood in front of every barrier
  constexpr bool EARLYR = !DENSE && !ROLES;
  constexpr int kLastDma = NQ - 1 < NU - 1 ? NQ - 1 : NU - 1;                   // region of a step's last DMA
  constexpr int kEarlyU = NU == 8 && kEarlyRegion > kLastDma ? kEarlyRegion : (kLastDma + 1 < NU ? kLastDma + 1 : NU - 1);
  int32_t r[4 * NQ];
  // VOFFQ (4 chunks per tile): this lane's byte offset of its row in each DMA group is computed ONCE per tile (the row
  // select by lane quarter + multiply: ~5 vector instructions per DMA) and kept; the chunk's 256 bytes go into the scalar
  // base.  The matrix pipe and the vector ALU of a SIMD do not run side by side (tools/experiments/pipes/overlap.hip:
  // their times add), so every vector instruction taken out of the step is time
  constexpr bool VOFFQ = EARLYR && NC == 4 && NQ == 4;     // (unions <= 128 rows: four DMA groups per wave)
  u32x4 voffq = {0, 0, 0, 0};                          // (a vector, not an array: hipcc put the array in scratch)
  auto make_voffq = [&]() {
    gwen_static_for<(VOFFQ ? 4 : 0)>([&](auto qq) {
      constexpr int q = decltype(qq)::value;
      int32_t r0 = r[4 * q], r1 = r[4 * q + 1], r2 = r[4 * q + 2], r3 = r[4 * q + 3];
      asm volatile("" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3));      // in SGPRs: selects (not a lookup in a scratch array)
      int32_t row = r0;
      row = mh == 1 ? r1 : row;
      row = mh == 2 ? r2 : row;
      row = mh == 3 ? r3 : row;
      voffq[q] = (uint32_t)row * (uint32_t)(FIN * 4) + (uint32_t)(mi * 16);
    });
  };
  if constexpr (EARLYR) {
    const int32_t *rp0 = t_rows + (int64_t)wt[(D + 1) / NC + 1] * kUCap + 4 * wave;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) r[4 * q + k] = rp0[4 * NW * q + k];
    if constexpr (VOFFQ) make_voffq();
  }
  constexpr int sreg0 = CT == 1 ? 6 : 5, sreg1 = 7;       // SKEW: the regions of a step's store(s), behind its DMAs (regions 0 .. 3)
  int rd0 = 0, rd1 = 2, rd2 = 2, rd3 = 2;   // SKEW: ring slot row tile t READS in this step = (s - t) mod (t + 2); it writes the slot before
  for (int i = 0; i < ntl; ++i) {
    gwen_static_for<NC>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const int s = i * NC + c;
      // ---- chunk to issue: (tile i2, chunk c2) -> stage[sb], entries -> ent[e2] --------------------------
      constexpr int a2 = c + D + 1, c2 = a2 % NC;
      const int i2 = i + a2 / NC, e2 = (eb + a2 / NC) % 3;
      int m2, t2;
      m2 = wm[a2 / NC + 1];                               // (the window: tile i + a2 / NC)
      t2 = wt[a2 / NC + 1];
      if constexpr (VOFFQ && c2 == 0) {                     // a new tile's rows (loaded in the step before)
        __builtin_amdgcn_sched_barrier(0);                // (hoisted into the step before, the four registers spill there)
        make_voffq();
      }
      const int32_t *rp = t_rows + (int64_t)t2 * kUCap + 4 * wave;          // wave-uniform: scalar loads
      const char *xm = uniform_ptr(x + (int64_t)m2 * mstride_x);
      if constexpr (!DENSE && !EARLYR) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int k = 0; k < 4; ++k) r[4 * q + k] = rp[4 * NW * q + k];
      }
      const float *bl = reinterpret_cast<const float *>(lds + kOffBias);
      wait_vmcnt(young);
      __syncthreads();
      int n_ops = 0;
      // ---- stores of tile i-1: a tile whose 64 rows all exist is stored region by region with every lane
      // active (exact instruction counts for the waits); the last tile of a member at once, guarded, drained
      bool spread_stores = false;
      float *obase = nullptr;                             // this lane's first output element of the tile
      // SKEW: the step's units run row tile (c + 1) % 4 FIRST -- the one on its tile's last chunk: complete after region 1,
      // stored in this step's LATE regions (its next tile starts in the LAST regions of the next step).  Tile i's row tile 0
      // completes in step 3, row tiles 1 .. 3 of tile i - 1 in steps 0 .. 2.
      constexpr int tfin = (c + 1) & 3;
      const char *abase = lds + kOffA + (s & 1) * kABytes;
      // F16: {e, delta} of destination row mi in the chunk row tile ti is about to contract (the pad bytes of the row)
      auto side_src = [&](int ti) -> const char * {
        if constexpr (SKEW) return lds + kOffA + ring_base(ti) + (ti == 0 ? rd0 : ti == 1 ? rd1 : ti == 2 ? rd2 : rd3) * kSl + mi * (kPB * 2) + kSide;
        else return abase + ((tt0 + ti * TSTEP) * 16 + mi) * (kPB * 2) + kSide;
      };
      float *srow = nullptr;
      bool s_any = false, s_whole = false, s_ok = false;
      auto store_target = [&]() {                         // (called inside region 2: off the path from the barrier to the first MFMA)
        if (SKEW && (c == 3 || i > 0)) {
          const int ms = c == 3 ? wm[1] : wm[0], ts = c == 3 ? wt[1] : wt[0];
          const int r = ts * kRows + tfin * 16 + mi;
          s_any = true;
          s_whole = (ts + 1) * kRows <= N;
          s_ok = r < N;
          srow = out + (int64_t)ms * mstride_o + (int64_t)r * ldo + (CT * jw * 16 + 4 * mh);
          if constexpr (F16) erow[tfin] = *reinterpret_cast<const int *>(side_src(tfin));   // (its last chunk's slot: read this step)
        }
      };
      float4_t sbias[CT];
      int late_ops = 0;                                   // SKEW: stores issued behind the step's last DMA (they may stay in flight)
      int aw0 = 0, aw1 = 0;                               // SKEW: where this step's two aggregate passes write (byte offset in A)
      if constexpr (SKEW) {
        const int w0 = rd0 == 0 ? 1 : rd0 - 1, w1 = rd1 == 0 ? 2 : rd1 - 1, w2 = rd2 == 0 ? 3 : rd2 - 1, w3 = rd3 == 0 ? 4 : rd3 - 1;
        aw0 = (wave >> 2) ? ring_base(1) + w1 * kSl : ring_base(0) + w0 * kSl;
        aw1 = (wave >> 2) ? ring_base(3) + w3 * kSl : ring_base(2) + w2 * kSl;
      }
      if (!SKEW && c == 0 && i > 0) {
        const int ms = wm[0], ts = wt[0];
        if ((ts + 1) * kRows <= N) {
          spread_stores = true;
          obase = out + (int64_t)ms * mstride_o + (int64_t)(ts * kRows + tt0 * 16 + mi) * ldo +
                  (CT * jw * 16 + 4 * mh);
        } else {
          store_tile(tile_of(i - 1));                   // drains
        }
      }
      // ---- aggregate(s+1) state -------------------------------------------------------------------------------
      constexpr int a1 = c + 1;
      const int sb1 = sb + 1 == NSTG ? 0 : sb + 1, ab1 = (s + 1) & 1, eb1 = (eb + a1 / NC) % 3;
      const char *ent = lds + kOffEnt + eb1 * kEntBytes;
      const char *stg = lds + kOffStage + sb1 * kStageBytes + mi * 16;
      u32x4 lid4 = {0, 0, 0, 0};
      float4_t wa = {0.f, 0.f, 0.f, 0.f}, wb = wa, acc = wa;
      // (the matrix pipe and the vector ALU of a SIMD take turns -- pipes/overlap.hip -- so address arithmetic is step time)
      constexpr bool PERM = !DENSE && NSTG == 2 && NC % 2 == 0 && kUCap <= 256 && NW == 8 && KU == 128;   // (elsewhere the extra register spills)
      const uint32_t lane16b = (uint32_t)(mi * 16);
      // SHIFT: stage k's vector work is issued in the region AFTER its LDS reads (the last stage's in the last region),
      // the two halves of a row's entries landing in two register buffers; without it (one buffer) a region's fma
      // chains wait for the reads issued at its own start: ~150-250 cycles of LDS latency in front of the region's MFMAs
      constexpr bool SHIFT = !DENSE && NSTAGE == NU && NU >= 4 && !(NS == 3 && NW == 16) &&   // (128-register waves: bf16x6 would spill)
                             !(F16 && FIN == 256 && FOUT == 256);   // (f16x3 at 256 -> 256: the second row buffer does not fit the
                                                                    //  registers -- 10 spilled, reloaded from scratch in every step: 327 us
                                                                    //  against 268 without it; three-row buffers and reads issued at the
                                                                    //  end of the region before measured 277 / 269)
      constexpr int NVB = SHIFT ? 2 : 1;
      float4_t vbuf[NVB][4];
      auto stage_loads = [&](int k) {                   // k = 4 pass + stage
        const int p = k >> 2, j = k & 3;
        const int lr = 4 * NW * p + 4 * wave + mh;
        float4_t (&v)[4] = vbuf[SHIFT && j == 2 ? NVB - 1 : 0];
        // entry q (0 .. 7) of this lane's row -> dst
        auto row_load = [&](int q, float4_t &dst) {
          const uint32_t pair = lid4[q >> 1];
          if constexpr (PERM) {
            // the row's byte address in ONE instruction: byte 0 = this lane's 16 mi, byte 1 = the local id (< 256: one
            // byte of its 16-bit slot), i.e. 256 lid + 16 mi; the stage buffer's base rides in the read's immediate
            // offset (two buffers, four steps per tile: which one is known per step at compile time)
            const uint32_t a = __builtin_amdgcn_perm(pair, lane16b, (q & 1) ? 0x0c0c0600u : 0x0c0c0400u);
            dst = *reinterpret_cast<const float4_t *>(lds + (kOffStage + ((c + 1) & 1) * kStageBytes) + a);
          } else {
            const uint32_t lid = (q & 1) ? (pair >> 16) : (pair & 0xffffu);
            dst = *reinterpret_cast<const float4_t *>(stg + lid * (kFC * 4));
          }
        };
        if constexpr (DENSE) {
          if (j == 1) v[0] = *reinterpret_cast<const float4_t *>(stg + lr * (kFC * 4));
        } else if (j == 0) {
          lid4 = *reinterpret_cast<const u32x4 *>(ent + kRows * 32 + lr * 16);
          wa = *reinterpret_cast<const float4_t *>(ent + lr * 32);
          wb = *reinterpret_cast<const float4_t *>(ent + lr * 32 + 16);
        } else if (j == 1 || j == 2) {
#pragma unroll
          for (int u = 0; u < 4; ++u) row_load(4 * (j - 1) + u, v[u]);
        }
      };
      auto stage_valu = [&](int k) {
        const int p = k >> 2, j = k & 3;
        const int lr = 4 * NW * p + 4 * wave + mh;
        float4_t (&v)[4] = vbuf[SHIFT && j == 2 ? NVB - 1 : 0];
        if constexpr (DENSE) {
          if (j == 1) acc = v[0];
        }
        if (!DENSE && (j == 1 || j == 2)) {
          const float4_t w4 = j == 1 ? wa : wb;
          if (j == 1) acc = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < 4; ++u)
            acc = fma4(w4[u], v[u], acc);
        }
        if (j == 3) {
          char *a = lds + kOffA + ab1 * kABytes + (lr * kPB + mi * 4) * 2;
          if constexpr (SKEW) a = lds + kOffA + (p == 0 ? aw0 : aw1) + ((lr & 15) * kPB + mi * 4) * 2;
          put_images(acc, a, kImgStride, p, (c + 1) % NC == 0);
        }
      };
      // first image of the A fragment of unit (row tile ti, k-step k2 of its chunk)
      auto a_src = [&](int ti, int k2) -> const char * {
        if constexpr (SKEW) return lds + kOffA + ring_base(ti) + (ti == 0 ? rd0 : ti == 1 ? rd1 : ti == 2 ? rd2 : rd3) * kSl + (mi * kPB + 8 * mh) * 2 + k2 * 64;
        else return abase + (((tt0 + ti * TSTEP) * 16 + mi) * kPB + 8 * mh) * 2 + k2 * 64;
      };
      bf16x8 afrag[NS];                                   // A fragments (NS images) of the unit about to run
      {
        const char *ap = a_src(SKEW ? ((c + 1) & 3) : 0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) afrag[s_] = *reinterpret_cast<const bf16x8 *>(ap + s_ * kImgStride);
      }
      if constexpr (F16) {
        // the row tiles that continue a tile in this step (not on their first chunk): accumulators into the new chunk's
        // scale, x 2^delta (exact).  The aggregator keeps a row's scale whenever it can, so ONE test per wave and step
        // -- the deltas of the step's row tiles or-ed, read with the first A fragments -- skips this almost always.
        int dl[NTT], any = 0;
        gwen_static_for<NTT>([&](auto tt) {
          constexpr int ti = decltype(tt)::value;
          constexpr int cu = SKEW ? ((c - ti) & 3) : c;
          dl[ti] = 0;
          if constexpr (cu != 0) {
            dl[ti] = *reinterpret_cast<const int *>(side_src(ti) + 4);
            any |= dl[ti];
          }
        });
        if (__builtin_amdgcn_ballot_w64(any != 0) != 0) {
          gwen_static_for<NTT>([&](auto tt) {
            constexpr int ti = decltype(tt)::value;
            constexpr int cu = SKEW ? ((c - ti) & 3) : c;
            if constexpr (cu != 0) {
#pragma unroll
              for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) d[ct][ti][e] = __builtin_ldexpf(d[ct][ti][e], dl[ti]);
            }
          });
        }
      }
      if constexpr (ROLES) {
        // ---- ROLES: every DMA of chunk s+2 first, then the step's two halves -- X = aggregate(s+1) (LDS reads,
        // VALU) and Y = mfma(s) (matrix pipe) -- in OPPOSITE order on the two waves of a SIMD (waves w, w + NW/2),
        // so that one wave's matrix work runs beside the other's vector work instead of both weaving the same
        // mixture in lockstep between the same two barriers.
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          int32_t r0 = r[4 * q], r1 = r[4 * q + 1], r2 = r[4 * q + 2], r3 = r[4 * q + 3];
          asm volatile("" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3));
          if (r0 >= 0) {
            int32_t row = r0;
            row = mh == 1 ? r1 : row;
            row = mh == 2 ? r2 : row;
            row = mh == 3 ? r3 : row;
            const uint32_t voff = (uint32_t)row * (uint32_t)(FIN * 4) + (uint32_t)(c2 * kFC * 4 + mi * 16);
            glds16(xm, voff, lds0 + kOffStage + sb * kStageBytes + 4 * (NW * q + wave) * (kFC * 4));
            ++n_ops;
          }
        }
        if (c2 == 0 && wave < 3) {
          const char *src = wave < 2 ? reinterpret_cast<const char *>(t_val) + (int64_t)t2 * (kRows * 32) + wave * 1024
                                     : reinterpret_cast<const char *>(t_lid) + (int64_t)t2 * (kRows * 16);
          glds16(uniform_ptr(src), (uint32_t)lane * 16, lds0 + kOffEnt + e2 * kEntBytes + wave * 1024);
          ++n_ops;
        }
        auto half_y = [&]() {
          if (c == 0 && spread_stores) {
#pragma unroll
            for (int ti = 0; ti < NTT; ++ti) {
              float *orow = obase + (int64_t)(ti * TSTEP * 16) * ldo;
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                float4_t o = float4_t{d[ct][ti][0], d[ct][ti][1], d[ct][ti][2], d[ct][ti][3]} +
                             *reinterpret_cast<const float4_t *>(bl + (CT * jw + ct) * 16 + 4 * mh);
                o = relu4(o, relu, floor_bits);
                if (nt) __builtin_nontemporal_store(o, reinterpret_cast<float4_t *>(orow + ct * 16));
                else *reinterpret_cast<float4_t *>(orow + ct * 16) = o;
              }
            }
          }
          gwen_static_for<NU>([&](auto uu) {
            constexpr int u = decltype(uu)::value;
            constexpr int ti = u >> 1, k2 = u & 1, ks = 2 * c + k2;
            bf16x8 acur[NS];
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) acur[s_] = afrag[s_];
            if constexpr (u + 1 < NU) {
              constexpr int tn = (u + 1) >> 1, kn = (u + 1) & 1;
              const char *ap = abase + (((tt0 + tn * TSTEP) * 16 + mi) * kPB + 8 * mh) * 2 + kn * 64;
#pragma unroll
              for (int s_ = 0; s_ < NS; ++s_) afrag[s_] = *reinterpret_cast<const bf16x8 *>(ap + s_ * kAImg);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              // the first unit of a tile starts from a ZERO C operand (an inline constant of the MFMA) instead of
              // re-zeroed accumulator registers: 32 moves per tile and wave less on the vector ALU
              const f32x4 cin = (c == 0 && k2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : d[ct][ti];
              d[ct][ti] = gwen::mma_split<8, NS>(bw[ct][ks], acur, cin);
            }
          });
        };
        // (with every row read of a half in flight instead of 4 at a time -- aggregate() -- the 256-channel kernel spills
        //  52 registers: 389 us against 252; the lean form spills 20 there: 292 us.  ROLES is not for 256 channels.)
        auto half_x = [&]() { aggregate_lean(sb1, ab1, eb1); };
        if (wave < NW / 2) {
          half_x();
          __builtin_amdgcn_sched_barrier(0);
          half_y();
        } else {
          half_y();
          __builtin_amdgcn_sched_barrier(0);
          half_x();
        }
        // D = 2: the chunk the next wait is for is OLDER than everything this interval issued (its DMAs and the
        // stores of half Y), so all of that may stay in flight; D = 1: that chunk is among them
        if (c == 0 && spread_stores) n_ops += CT * NTT;
        young = D == 2 ? n_ops : 0;
        prev_ops = n_ops;
        sb = sb + 1 == NSTG ? 0 : sb + 1;
        return;
      }
      // ---- the regions --------------------------------------------------------------------------------------
      gwen_static_for<NU>([&](auto uu) {
        constexpr int u = decltype(uu)::value;
        // memory instructions of this region
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if ((EARLY ? (q < NU ? q : NU - 1) : q * NU / NQ) != u) continue;
          if constexpr (DENSE) {
            int32_t row = t2 * kRows + 4 * (NW * q + wave) + mh;
            row = row < N ? row : N - 1;
            const uint32_t voff = (uint32_t)row * row_pitch + (uint32_t)(c2 * kFC * 4 + mi * 16);
            glds16(xm, voff, lds0 + kOffStage + sb * kStageBytes + 4 * (NW * q + wave) * (kFC * 4));
            ++n_ops;
            continue;
          }
          int32_t r0 = r[4 * q], r1 = r[4 * q + 1], r2 = r[4 * q + 2], r3 = r[4 * q + 3];
          if constexpr (VOFFQ) {
            if (r0 >= 0) {
              glds16(xm + c2 * kFC * 4, q == 0 ? voffq[0] : q == 1 ? voffq[1] : q == 2 ? voffq[2] : voffq[3], lds0 + kOffStage + sb * kStageBytes + 4 * (NW * q + wave) * (kFC * 4));
              ++n_ops;
            }
            continue;
          }
          asm volatile("" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3));      // in SGPRs: selects, no branches
          if (r0 >= 0) {
            int32_t row = r0;
            row = mh == 1 ? r1 : row;
            row = mh == 2 ? r2 : row;
            row = mh == 3 ? r3 : row;
            const uint32_t voff = (uint32_t)row * (uint32_t)(FIN * 4) + (uint32_t)(c2 * kFC * 4 + mi * 16);
            glds16(xm, voff, lds0 + kOffStage + sb * kStageBytes + 4 * (NW * q + wave) * (kFC * 4));
            ++n_ops;
          }
        }
        if (!DENSE && u == (SKEW ? NQ - 1 : NU - 1) && c2 == 0 && wave < 3) {     // (SKEW: with the last DMA: the stores behind are counted)
          const char *src = wave < 2 ? reinterpret_cast<const char *>(t_val) + (int64_t)t2 * (kRows * 32) + wave * 1024
                                     : reinterpret_cast<const char *>(t_lid) + (int64_t)t2 * (kRows * 16);
          glds16(uniform_ptr(src), (uint32_t)lane * 16, lds0 + kOffEnt + e2 * kEntBytes + wave * 1024);
          ++n_ops;
        }
        // (only when the next step starts on another tile: the chunks of one tile name the same rows, r[] stays)
        if constexpr (EARLYR && u == kEarlyU && (c + D + 2) % NC == 0) {
          constexpr int an = c + D + 2;                   // the chunk the NEXT step issues, counted from this tile's chunk 0
          const int32_t *rpn = t_rows + (int64_t)wt[an / NC + 1] * kUCap + 4 * wave;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) r[4 * q + k] = rpn[4 * NW * q + k];
        }
        if constexpr (SKEW && u == 2) store_target();
        if constexpr (SKEW && !F16) {           // a store's bias fragment is read one region ahead of it
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            if ((ct == 0 ? sreg0 : sreg1) - 1 == u) sbias[ct] = *reinterpret_cast<const float4_t *>(bl + (CT * jw + ct) * 16 + 4 * mh);
        }
        if constexpr (SKEW && u >= 2) {
          // the stores of the row tile that completed in region 1, BEHIND the step's DMAs (regions 0 .. 3).  A CU takes ~58
          // cycles per store INSTRUCTION whatever its width (tools/experiments/stores/stwave.hip), one at a time, and a
          // wave whose next memory instruction meets stores in the queue stands behind them.  (Dealing the 8 x CT stores
          // over regions 2 .. 7 by wave, among the DMAs, measured slower: 225-227 against 222-223 us on one box.)
          if (s_any) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              if ((ct == 0 ? sreg0 : sreg1) != u) continue;          // (wave-uniform: a scalar branch)
              float4_t o = finish4(d[ct][tfin], erow[tfin], (CT * jw + ct) * 16 + 4 * mh,
                                   !F16 ? sbias[ct] : *reinterpret_cast<const float4_t *>(bl + (CT * jw + ct) * 16 + 4 * mh));
              o = relu4(o, relu, floor_bits);
              float4_t *dst = reinterpret_cast<float4_t *>(srow + ct * 16);
              if (s_whole) {                                 // every lane stores: exactly one instruction (the waits count)
                if (nt) __builtin_nontemporal_store(o, dst);
                else *dst = o;
                ++late_ops;                                  // behind the wave's last DMA (region 3)
              } else if (s_ok) {
                *dst = o;
              }
            }
          }
        }
        if (c == 0) {
#pragma unroll
          for (int ti = 0; ti < NTT; ++ti) {
            if ((ti == 0 ? 0 : 2 * ti - 1) != u) continue;
            if (spread_stores) {
              float *orow = obase + (int64_t)(ti * TSTEP * 16) * ldo;
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                float4_t o = finish4(d[ct][ti], erow[ti], (CT * jw + ct) * 16 + 4 * mh,
                                     *reinterpret_cast<const float4_t *>(bl + (CT * jw + ct) * 16 + 4 * mh));
                o = relu4(o, relu, floor_bits);
                  if (nt) __builtin_nontemporal_store(o, reinterpret_cast<float4_t *>(orow + ct * 16));
                else *reinterpret_cast<float4_t *>(orow + ct * 16) = o;
              }
              n_ops += CT;
            }
          }
        }
        // aggregate stages of this region: LDS reads, then the unit's MFMAs, then the VALU work
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k)
          if (k * NU / NSTAGE == u) stage_loads(k);
        {
          // the unit's A fragments were read one region earlier (their LDS latency sits behind that region's
          // MFMAs); the next unit's are requested now
          constexpr int k2 = u & 1;
          constexpr int ti = SKEW ? (((u >> 1) + c + 1) & 3) : (u >> 1);     // SKEW: the completing row tile first
          constexpr int cu = SKEW ? ((c - ti) & 3) : c;          // the chunk (of ITS tile) row tile ti works on in this step
          constexpr int ks = 2 * cu + k2;
          bf16x8 acur[NS];
#pragma unroll
          for (int s_ = 0; s_ < NS; ++s_) acur[s_] = afrag[s_];
          if constexpr (u + 1 < NU) {
            constexpr int tn = SKEW ? ((((u + 1) >> 1) + c + 1) & 3) : ((u + 1) >> 1), kn = (u + 1) & 1;
            const char *ap = a_src(tn, kn);
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) afrag[s_] = *reinterpret_cast<const bf16x8 *>(ap + s_ * kImgStride);
          }
          if constexpr (F16 && k2 == 0) {
            // the exponent the finished accumulators are in: SKEW stores the row tile later in this very step and reads it
            // then (store_target); the other forms store in the next step, when the slot is being rewritten
            if constexpr (cu == NC - 1 && !SKEW) erow[ti] = *reinterpret_cast<const int *>(side_src(ti));
          }
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
              // the first unit of a tile starts from a ZERO C operand (an inline constant of the MFMA) instead of
              // re-zeroed accumulator registers: 32 moves per tile and wave less on the vector ALU
              const f32x4 cin = (cu == 0 && k2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : d[ct][ti];
              if constexpr (F16) {
                d[ct][ti] = gwen::mma_split_f16(bw[ct][ks], acur, cin);
              } else {
                d[ct][ti] = gwen::mma_split<8, NS>(bw[ct][ks], acur, cin);
              }
            }
        }
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k) {
          const int rl = k * NU / NSTAGE, rv = SHIFT ? (rl + 1 < NU ? rl + 1 : NU - 1) : rl;
          if (rv == u) stage_valu(k);
        }
#pragma unroll
        for (int k = 0; k < CT * (NS == 2 ? 3 : 6); ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, kWeaveValu, 0);     // a few VALU instructions behind it
        }
      });
      // what may stay in flight at the next wait: with two chunks of DMA in flight, everything this interval
      // issued (the DMA the next wait is for is older); with one, nothing (that DMA is among them)
      young = D == 3 ? n_ops + prev_ops : D == 2 ? n_ops : 0;
      prev_ops = n_ops;
      sb = sb + 1 == NSTG ? 0 : sb + 1;
      if constexpr (SKEW) {
        // one chunk in flight: only the stores behind the step's last DMA may stay; two: the step's DMAs as well (the
        // chunk the next wait is for is older than all of them).  A guarded store may not have issued: not counted
        young = (D == 2 ? n_ops : 0) + (s_whole ? late_ops : 0);
        rd0 = rd0 == 1 ? 0 : rd0 + 1; rd1 = rd1 == 2 ? 0 : rd1 + 1; rd2 = rd2 == 3 ? 0 : rd2 + 1; rd3 = rd3 == 4 ? 0 : rd3 + 1;
      }
    });
    eb = eb + 1 == 3 ? 0 : eb + 1;
#pragma unroll
    for (int k = 0; k + 1 < WN; ++k) { wm[k] = wm[k + 1]; wt[k] = wt[k + 1]; }
    split_tile(tile_of(i + 1 + HI), wm[WN - 1], wt[WN - 1]);
  }
  if constexpr (SKEW) {
    // ---- drain: row tiles 1 .. 3 of the last tile are 1 .. 3 chunks behind: three more steps of matrix work only (their
    // A slices were written before the last barrier), each completing and storing one row tile ----
    const int ms = wm[0], ts = wt[0];                   // (the last tile: shifted once more after the loop)
    const float *bl = reinterpret_cast<const float *>(lds + kOffBias);
    auto store_rt = [&](int rt) {
      const int r = ts * kRows + rt * 16 + mi;
      float *row = out + (int64_t)ms * mstride_o + (int64_t)r * ldo + (CT * jw * 16 + 4 * mh);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float4_t o = finish4(d[ct][rt], erow[rt], (CT * jw + ct) * 16 + 4 * mh,
                             *reinterpret_cast<const float4_t *>(bl + (CT * jw + ct) * 16 + 4 * mh));
        o = relu4(o, relu, floor_bits);
        if (r < N) *reinterpret_cast<float4_t *>(row + ct * 16) = o;
      }
    };
    gwen_static_for<3>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      gwen_static_for<3 - c>([&](auto tt) {
        constexpr int ti = c + 1 + decltype(tt)::value, cu = (c - ti) & 3;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const char *ap = lds + kOffA + ring_base(ti) + (ti == 1 ? rd1 : ti == 2 ? rd2 : rd3) * kSl + (mi * kPB + 8 * mh) * 2 + k2 * 64;
          bf16x8 af[NS];
#pragma unroll
          for (int s_ = 0; s_ < NS; ++s_) af[s_] = *reinterpret_cast<const bf16x8 *>(ap + s_ * kImgStride);
          if constexpr (F16) {
            if (k2 == 0) {                                    // (cu >= 1 in the drain: never a tile's first chunk)
              typedef int int2_t __attribute__((ext_vector_type(2)));
              const int2_t sd = *reinterpret_cast<const int2_t *>(
                  lds + kOffA + ring_base(ti) + (ti == 1 ? rd1 : ti == 2 ? rd2 : rd3) * kSl + mi * (kPB * 2) + kSide);
              if (__builtin_amdgcn_ballot_w64(sd[1] != 0) != 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                  for (int e = 0; e < 4; ++e) d[ct][ti][e] = __builtin_ldexpf(d[ct][ti][e], sd[1]);
              }
              if constexpr (cu == NC - 1) erow[ti] = sd[0];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) d[ct][ti] = gwen::mma_split_f16(bw[ct][2 * cu + k2], af, d[ct][ti]);
            continue;
          }
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) d[ct][ti] = gwen::mma_split<8, NS>(bw[ct][2 * cu + k2], af, d[ct][ti]);
        }
      });
      store_rt(c + 1);                                  // on its last chunk in this step
      rd0 = rd0 == 1 ? 0 : rd0 + 1; rd1 = rd1 == 2 ? 0 : rd1 + 1; rd2 = rd2 == 3 ? 0 : rd2 + 1; rd3 = rd3 == 4 ? 0 : rd3 + 1;
    });
  } else {
    store_tile(tile_of(ntl - 1));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // drain the DMAs issued past the last tile
}

template <int FIN, int FOUT, int NW, int D, int KU, bool EARLY, bool DENSE = false, int NS = 2, bool F16 = false>
int launch(const int32_t *t_rows, const uint16_t *t_lid, const float *t_val, const float *x,
           const float *W, const float *bias, float *out, int64_t N, int64_t ldo, int64_t members,
           int64_t msx, int64_t mso, int relu, hipStream_t st, int64_t ldx = FIN) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    GWEN_HIP_CHECK(hipGetDevice(&dev));
    GWEN_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = n < 8 ? 8 : n;
  }
  const int64_t T = (N + kRows - 1) / kRows, G = T * members;
  if (G >= (int64_t(1) << 31)) return GWEN_ERANGE;
  const int64_t blocks = G < cus ? G : cus;
  // output rows nobody re-reads before they leave the 256 MiB Infinity Cache anyway (in + out beyond it) are stored
  // non-temporally: they then do not push the halo rows the tiles share out of the caches (64 channels x 16
  // members: 204 -> 191 us per layer in the 4-layer stack); a working set that fits stays on plain stores (the
  // next layer reads its input from the cache: 256 channels x 1 member 69.7 vs 74.3 us with nt)
  const int nt = members * N * (int64_t)(FIN + FOUT) * 4 > (int64_t(300) << 20) ? 1 : 0;
  k_wide<FIN, FOUT, NW, D, KU, EARLY, DENSE, NS, F16><<<(unsigned)blocks, NW * 64, 0, st>>>(
      t_rows, t_lid, t_val, x, W, bias, out, (int32_t)N, (int32_t)T, (int32_t)G, ldo, msx, mso, relu, (int32_t)ldx, nt);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

constexpr bool fin_ok(int64_t f) { return f == 64 || f == 128 || f == 256; }
constexpr bool fout_ok(int64_t f) { return f == 64 || f == 128 || f == 256; }

}  // namespace

// K3's tall case on the K8 pipeline: h[rows, Fout] = act(x[rows, Fin (pitch ldx)] W^T + b).  Internal: called by
// gwen_gcn_linear_f32 (linear.hip), which has validated the pointers.  contract: bf16x3, or bf16x6 up to Fin = 128.
int gwen_wide_dense_f32(const float *x, const float *W, const float *bias, float *h, int64_t rows, int64_t Fin,
                        int64_t Fout, int64_t ldx, int64_t ldh, int relu, int contract, hipStream_t st) {
  if (!fin_ok(Fin) || !fout_ok(Fout) || rows * ldx * 4 >= (int64_t(1) << 32) || rows >= (int64_t(1) << 31) - 64)
    return GWEN_ERANGE;
  const bool x6 = contract == GWEN_CONTRACT_BF16X6;
  if (x6 && Fin > 128) return GWEN_ERANGE;
#define GWEN_D(FI, FO)                                                                                  \
  if (Fin == FI && Fout == FO) {                                                                        \
    constexpr int NWV = FI >= 256 ? 8 : 16;                                                             \
    if constexpr (FI <= 128) {                                                                          \
      if (x6)                                                                                           \
        return launch<FI, FO, NWV, 2, 64, true, true, 3>(nullptr, nullptr, nullptr, x, W, bias, h, rows, ldh, \
                                                         1, 0, 0, relu, st, ldx);                       \
    }                                                                                                   \
    return launch<FI, FO, NWV, 2, 64, true, true>(nullptr, nullptr, nullptr, x, W, bias, h, rows, ldh,  \
                                                  1, 0, 0, relu, st, ldx);                              \
  }
  GWEN_D(64, 64); GWEN_D(64, 128); GWEN_D(64, 256);
  GWEN_D(128, 64); GWEN_D(128, 128); GWEN_D(128, 256);
  GWEN_D(256, 64); GWEN_D(256, 128); GWEN_D(256, 256);
#undef GWEN_D
  return GWEN_ERANGE;
}

extern "C" int gwen_gcn_wide_supported(int64_t Fin, int64_t Fout) {
  return fin_ok(Fin) && fout_ok(Fout) ? 1 : 0;
}

extern "C" int gwen_gcn_wide_contract_supported(int64_t Fin, int64_t Fout, int contract) {
  if (!gwen_gcn_wide_supported(Fin, Fout)) return 0;
  if (contract == GWEN_CONTRACT_BF16X3) return 1;
  if (contract == GWEN_CONTRACT_F16X3) return 1;
  // bf16x6 at 256 -> 256: three images of W are 192 of the 256 registers a wave has there (hipcc spills 85-91 of
  // them: 141-157 us per pass on one member against K4's 127, measured); 256 -> 64 / 128 hold half / a quarter of
  // the columns per wave and fit, so 256 -> 256 runs as two 256 -> 128 launches (gwen_gcn_wide_layer_f32).
  return contract == GWEN_CONTRACT_BF16X6 ? 1 : 0;
}

extern "C" int gwen_gcn_wide_preferred(int64_t N, int64_t members, int64_t Fin, int64_t Fout) {
  if (!gwen_gcn_wide_supported(Fin, Fout) || N <= 0 || members <= 0) return 0;
  return Fin >= 128 || N * members >= kPrefRows ? 1 : 0;
}

extern "C" int gwen_gcn_wide_layer_f32(const int32_t *t_rows, const uint16_t *t_lid,
                                       const float *t_val, const float *x, const float *W,
                                       const float *bias, float *out, int64_t N, int64_t N_src,
                                       int64_t Fin, int64_t Fout, int64_t ldo, int64_t members,
                                       int64_t mstride_x, int64_t mstride_o, int relu,
                                       int64_t union_max, int contract, gwen_stream_t stream_) {
  if (N < 0 || N_src < 0 || members < 0 || ldo < Fout || union_max < 0 || union_max > kUCap)
    return GWEN_EINVAL;
  if (!gwen_gcn_wide_contract_supported(Fin, Fout, contract)) return GWEN_EINVAL;
  if (N == 0 || members == 0) return GWEN_OK;
  if (!t_rows || !t_lid || !t_val || !x || !W || !out || x == out) return GWEN_EINVAL;
  if (!gwen_aligned(x, 16) || !gwen_aligned(out, 16) || !gwen_aligned(W, 16) || mstride_x % 4 ||
      ldo % 4 || mstride_o % 4 || (bias && !gwen_aligned(bias, 16)) || !gwen_aligned(t_val, 16) ||
      !gwen_aligned(t_lid, 16))
    return GWEN_EINVAL;
  if (N_src * Fin * 4 >= (int64_t(1) << 32)) return GWEN_ERANGE;     // 32-bit row offsets
  hipStream_t st = gwen_stream(stream_);
  // 16 waves where the registers allow (Fin <= 128), 8 at Fin = 256 (W alone is 128 registers there).
  // Chunks of DMA in flight behind the chunk being aggregated: D = 1 from 128 channels up -- at 256 channels x 4
  // members a second chunk in flight measures within +-1 % (236-240 us either way, round 3; 246-252 in round 2: a CU
  // takes ~32 LDS-DMA wave instructions in flight before further issues stall, tools/experiments/reads/glds.hip) --
  // and D = 2 at 64 channels on bf16x3, where a step is one whole tile and short: 64 -> 64 x 16 members 193.9 -> 185.5
  // us in the 4-layer stack (A/B on one box; with the ROLES order of the halves on top: no further gain, so ROLES is
  // for D = 1 -- i.e. bf16x6, whose third A image takes the third stage buffer's LDS).
  // Unions <= 128 rows stage 128 slots per chunk.
  // bf16x6 keeps a third image of every A chunk: its LDS fits beside 128-slot stage buffers only, so unions
  // beyond 128 rows are refused (GWEN_ERANGE: the caller takes K4).
  const bool small_union = union_max <= 128;
  const bool x6 = contract == GWEN_CONTRACT_BF16X6;
  if (x6 && !small_union) return GWEN_ERANGE;
  if (contract == GWEN_CONTRACT_F16X3) {
    // fp32-class on two scaled fp16 images: bf16x3's pipeline (SKEW at 256 -> 128 / 256, unions up to 192 rows), ONE launch
#define GWEN_ARGS t_rows, t_lid, t_val, x, W, bias, out, N, ldo, members, mstride_x, mstride_o, relu, st
#define GWEN_H(FI, FO)                                                                                \
  if (Fin == FI && Fout == FO) {                                                                      \
    constexpr int NWV = FI >= 256 ? 8 : 16;                                                           \
    return small_union ? launch<FI, FO, NWV, dma_depth(FI), 128, true, false, 2, true>(GWEN_ARGS)     \
                       : launch<FI, FO, NWV, 1, 192, true, false, 2, true>(GWEN_ARGS);                \
  }
    GWEN_H(64, 64); GWEN_H(64, 128); GWEN_H(64, 256);
    GWEN_H(128, 64); GWEN_H(128, 128); GWEN_H(128, 256);
    GWEN_H(256, 64); GWEN_H(256, 128); GWEN_H(256, 256);
#undef GWEN_H
#undef GWEN_ARGS
    return GWEN_EINVAL;
  }
  if (x6 && Fin == 256 && Fout == 256) {
    // three images of W for all 256 output columns are 192 registers per wave; for 128 columns they fit.  The layer
    // runs as TWO launches of the 256 -> 128 kernel, each staging and aggregating the rows again and writing its half
    // of the output columns (row stride ldo): every output column still sums the same terms in the same order
    // (bitwise K4's bf16x6), at about twice a 256 -> 128 pass instead of K4's gathers through L1
    for (int half = 0; half < 2; ++half) {
      const int rc = launch<256, 128, 8, 1, 128, true, false, 3>(
          t_rows, t_lid, t_val, x, W + (int64_t)half * 128 * 256, bias ? bias + half * 128 : nullptr, out + half * 128,
          N, ldo, members, mstride_x, mstride_o, relu, st);
      if (rc != GWEN_OK) return rc;
    }
    return GWEN_OK;
  }
#define GWEN_ARGS t_rows, t_lid, t_val, x, W, bias, out, N, ldo, members, mstride_x, mstride_o, relu, st
#define GWEN_W(FI, FO)                                                                                \
  if (Fin == FI && Fout == FO) {                                                                      \
    constexpr int NWV = FI >= 256 ? 8 : 16;                                                           \
    if constexpr (FI <= 128 || FO <= 128) {                                                           \
      constexpr int NW6 = FI == 128 ? x6_waves_128(FO) : NWV;                                         \
      if (x6) return launch<FI, FO, NW6, dma_depth(FI), 128, true, false, 3>(GWEN_ARGS);              \
    }                                                                                                 \
    constexpr int DV = dma_depth(FI);                                                                 \
    return small_union ? launch<FI, FO, NWV, DV, 128, true>(GWEN_ARGS)                                \
                       : launch<FI, FO, NWV, 1, 192, true>(GWEN_ARGS);                                \
  }
  GWEN_W(64, 64); GWEN_W(64, 128); GWEN_W(64, 256);
  GWEN_W(128, 64); GWEN_W(128, 128); GWEN_W(128, 256);
  GWEN_W(256, 64); GWEN_W(256, 128); GWEN_W(256, 256);
#undef GWEN_W
#undef GWEN_ARGS
  return GWEN_EINVAL;
}
