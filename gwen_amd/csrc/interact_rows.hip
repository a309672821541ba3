// K6 at 64 and 256 channels, ROW-STATIONARY -- the same contract as k_mlp2 (interact.hip):
//
//     pre[r] = A[r] W1^T + G1[idx1[r]] + G2[idx2[r]] + b1 ;  y[r] = act(pre[r]) W2^T + b2
//     out[r] = res[r] + y[r] ;  agg[d] = sum_{r : dst(r) = d} y[r]      (rows sorted by d, stored order)
//
// At 256 channels the two weight matrices as 3xbf16 hi/lo fragments are 512 KB, the whole register file of a
// CU, so they cannot stay resident.  k_mlp2 kept the activations in LDS and let every wave stream ITS
// column slice of W from L2, one k-step ahead: per 64-row pass a CU pulled 512 KB of fragments through the
// vector memory path, each k-step (384 cycles of MFMA per wave) waited for an L2 round trip, and the hidden
// layer crossed LDS between two barriers (measured: 886 us for 600 000 edges = 0.21 of the bf16 MFMA peak;
// 690 us with the fragment loads removed).  Here the roles are swapped:
//   * a wave owns 16 ROWS and all F columns of them -- its A rows, the first contraction's accumulators
//     (= the hidden layer) and the second one's live in registers, three sets of F / 4; a lane holds, of row
//     lane % 16, the 4 columns 16 j + 4 (lane / 16) .. + 3 of every 16-column tile j;
//   * the k index of a contraction may be permuted freely as long as both operands agree, so a k-step takes
//     as "its" 32 channels the two column tiles 2 ks and 2 ks + 1 in exactly that register layout: the B
//     operand (8 values per lane) is read straight from the lane's registers -- the A rows for the first
//     contraction, the activated accumulators for the second -- and the hidden layer never touches LDS;
//     the weight images (k_split_wr) are written in the matching order;
//   * the weights are the shared operand.  256 channels: one k-step of a matrix is 32 KB of fragments for ALL 16
//     column tiles, brought by LDS-DMA (global_load_lds_dwordx4, no registers) into a ring of three slots, two
//     k-steps ahead, and read by all 8 waves -- 256 KB per 128-row pass and matrix instead of 512 KB per 64 --
//     with one barrier per k-step (it frees the slot the next DMA overwrites).  64 channels: both matrices'
//     fragments are 32 KB and stay in LDS for the whole launch: no ring, no per-step barrier, two blocks per CU;
//   * the aggregation goes through a 64-column fp32 tile in LDS (four times per pass at 256 channels), summed
//     per target row in stored order by one thread per (row, 4 columns) as in k_mlp2 -- no atomics, bitwise
//     reproducible, independent of the tiling;
//   * 64 channels: A rows are read and out rows written in a ROW layout -- 16 lanes on the 256 contiguous bytes
//     of a row -- and turned into / out of the MFMA layout through the wave's own 16 rows of that LDS tile: the
//     MFMA layout's natural access, 64-B pieces of 16 different rows per instruction, cost 15 % of the kernel.
// The ring's DMAs are inline asm (hipcc neither counts nor drains them); the waits for them are explicit and
// count the vector-memory instructions issued since (cdna_hip_programming.md, "Pipelining across barriers").
// Register roles rotate: the next pass's A rows are loaded into the hidden layer's registers as the second
// contraction releases them, and the finished pass's A registers receive the next pass's gathered G2 rows
// (the first contraction accumulates onto them), so the pass body exists twice with the two sets swapped.
#include "common.h"
#include "rows_common.h"
// (the lab version of this file -- timing ablations -DABL=1..7, per-phase s_memtime stamps -DGWEN_K6R_STAMPS, two row
//  tiles per wave -DK6R_RT=2 -- is tools/experiments/interact_rows_lab.patch; DESIGN.md section 4 / 7.3 has their numbers)
constexpr int K6R_RT = 1;      // 16-row tiles per wave (see RCfg)

namespace {

constexpr int kNone = 0, kSelf = 1, kIdx = 2;      // how a table is addressed (as interact.hip)
constexpr int kResNone = 0, kResA = 1, kResOther = 2;

// RT = 16-row tiles per wave: 1 -> 8 waves (two per SIMD, 256 registers each) -- what runs; 2 -> 4 waves (one per
// SIMD, 512 registers, every W fragment read from LDS feeds two row tiles): hipcc spills 130-230 registers of the
// three 128-register sets and the edge kernel takes 1 006 us instead of 679 (tools/experiments/abl_k6r.sh)
template <int F, int RT = 1>
struct RCfg {
  static constexpr int NJ = F / 16;                 // column tiles = float4 registers per row and set
  static constexpr int KS = F / 32;                 // k-steps per contraction
  static constexpr int NW = 8 / RT;                 // waves per block
  static constexpr int ROWS = NW * RT * 16;         // rows per pass: 128
  static constexpr int STEP = NJ * 2 * 1024;        // bytes of W fragments per k-step: (tile, hi/lo) x 1 KB
  // F <= 64: both matrices' fragments (2 KS steps, 32 KB at F = 64) stay RESIDENT in LDS for the whole launch --
  // no ring, no per-step waits or barriers, no inline-asm DMA in the loop -- and two blocks share a CU
  static constexpr bool RESIDENT = F <= 64;
  static constexpr int NSLOT = RESIDENT ? 2 * KS : 3;
  // A rows and out rows cross global memory in a ROW layout (16 lanes on 256 contiguous bytes) and are turned
  // into / out of the MFMA layout through LDS: at 64 channels -15 % (the MFMA layout's natural access is 64-B
  // pieces of 16 different rows per instruction); at 256 channels (1 KB rows) it measured 690 -> 739 us, off
  static constexpr bool ROWIO = RESIDENT;
  // ... the out rows alone, one 16-B piece at a time (no registers to spare), do pay at 256 channels: 653 -> 622 us
  static constexpr bool ROWST = true;
  static constexpr int GS = KS < 4 ? KS : 4;        // steps of the first contraction that load G1 rows
  static constexpr int MINW = RESIDENT ? 4 : 2;     // waves per SIMD to compile for
  static constexpr int DPW = STEP / 1024 / NW;      // DMA instructions per wave and k-step
  static constexpr int YC = 64, PY = YC + 4;        // aggregation chunk: columns, LDS pitch (floats)
  static constexpr int NCH = F / YC, JC = YC / 16;
  static constexpr int kOffY = NSLOT * STEP;
  static constexpr int kOffB = kOffY + ROWS * PY * 4;
  static constexpr int lds_bytes = kOffB + 2 * F * 4;
  static constexpr int Q = YC / 4;                  // 16-B pieces per chunk row
  static constexpr int SLOTS = NW * 64 / Q;         // target rows reduced at a time
  static constexpr int JG = 2 / RT;                 // column tiles per MFMA group (two accumulators a group)
  static_assert(RT == 1 || RT == 2, "one or two row tiles per wave");
  static_assert(STEP % (1024 * NW) == 0 && lds_bytes <= 160 * 1024, "ring must tile over the waves and fit");
};

__device__ inline float4_t activate(float4_t v, int act) {
  if (act == GWEN_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] < 0.0f ? 0.0f : v[i];
  } else if (act == GWEN_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i] * -1.44269504088896341f));
  }
  return v;
}

// W [F,F] fp32 (row = output column) -> the k-step-major fragment image: k-step ks, column tile jo, (hi, lo),
// lane l = (i = l % 16, g = l / 16): the 8 values W[16 jo + i][32 ks + 4 g .. + 3], W[16 jo + i][32 ks + 16 + 4 g .. + 3]
// (blockIdx.y: 0 = W1 -> the first KS chunks of the image, 1 = W2 -> the next KS: one launch for both)
template <int F>
__global__ __launch_bounds__(64) void k_split_wr(const float *__restrict__ W1, const float *__restrict__ W2,
                                                 bf16x8 *__restrict__ img) {
  using C = RCfg<F>;
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int jo = blockIdx.x / C::KS, ks = blockIdx.x % C::KS;
  const float *wp = (blockIdx.y ? W2 : W1) + (int64_t)(jo * 16 + i) * F + 32 * ks + 4 * g;
  bf16x8 hi, lo;
  split8(*reinterpret_cast<const float4_t *>(wp), *reinterpret_cast<const float4_t *>(wp + 16), hi, lo);
  bf16x8 *dst = img + (int64_t)(blockIdx.y * C::KS + ks) * (C::STEP / 16) + (jo * 2) * 64 + lane;
  dst[0] = hi;
  dst[64] = lo;
}

struct Pass {
  int32_t tile, r0, r1;
  int32_t w0, e0, e1;
};

// visible (hipcc-counted) vector-memory instructions step si of a pass issues after its DMAs: the first
// contraction's steps load the G1 rows (4 tiles a step, steps 0-3) and, with a residual other than A, its rows;
// the second one's load the next pass's A rows (2 tiles a step) and, in its first step, the next indices
// (HID: the activation step also stores the hidden layer, one store per column tile -- every lane, every time: the
//  counts must be exact)
template <int F, int RT, int M1, int M2, int RES, bool HID = false>
struct Vis {
  static constexpr int of(int si) {
    using C = RCfg<F, RT>;
    if (si < C::KS) return RT * (((M1 != kNone && si < C::GS) ? C::NJ / C::GS : 0) + (RES == kResOther ? 2 : 0));
    return RT * (2 + (si == C::KS ? (M1 == kIdx ? 1 : 0) + (M2 == kIdx ? 1 : 0) + (HID ? C::NJ : 0) : 0));
  }
};

// HID (the block's backward, gwen_mlp2_bwd_f32): the hidden layer is the PRODUCT (acc + b1) * G1[row] instead of
// act(acc + G1 + b1), and it is stored as well (`hid`, rows padded to whole passes: the stores are unconditional) --
//     g_pre1 = (ge W2 + T[dst]) * act'(pre1)  ->  hid ;   g_e = ge + g_pre1 We  ->  out
// with A = res = ge, "W1" = W2^T, G1 = act'(pre1) row for row, G2 = T gathered by target, "W2" = We^T.
template <int F, int RT, int M1, int M2, bool SEG, int RES, bool HID = false>
__global__ __launch_bounds__((RCfg<F, RT>::NW * 64), (RCfg<F, RT>::MINW)) void k_mlp2r(
    const float *__restrict__ A, const char *__restrict__ img, const float *__restrict__ W1f,
    const float *__restrict__ W2f, const float *__restrict__ G1,
    const int32_t *__restrict__ idx1, const float *__restrict__ G2, const int32_t *__restrict__ idx2,
    const float *__restrict__ b1, const float *__restrict__ b2, const float *__restrict__ res,
    float *__restrict__ out, int32_t R, int act, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ tile_row, int32_t n_tiles, float *__restrict__ agg, int mean,
    uint32_t ldb1, uint32_t ldb2, float *__restrict__ hid = nullptr) {
  using C = RCfg<F, RT>;
  constexpr int NR = RT * C::NJ;                                     // float4 registers per set
  __shared__ __attribute__((aligned(1024))) char lds[C::lds_bytes];
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  float *ytile = reinterpret_cast<float *>(lds + C::kOffY);
  float *bl = reinterpret_cast<float *>(lds + C::kOffB);           // b1 | b2
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mi = lane & 15, g = lane >> 4;
  auto prow = [&](int rt) { return (wave * RT + rt) * 16 + mi; };    // this lane's rows within a pass
  auto col = [&](int j) { return 16 * j + 4 * g; };                  // its 4 columns of column tile j

  for (int f = t; f < F; f += C::NW * 64) {
    bl[f] = b1 ? b1[f] : 0.0f;
    bl[F + f] = b2 ? b2[f] : 0.0f;
  }

  // tiles are dealt so that the blocks of one XCD (blockIdx % 8) walk ONE contiguous eighth of them
  const int nb = gridDim.x, xcd = blockIdx.x & 7;
  const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
  const int t_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int t_hi = t_lo + q8 + (xcd < r8 ? 1 : 0);
  const int stride = (nb + 7 - xcd) >> 3;                            // blocks on this XCD
  const int tile0 = t_lo + (blockIdx.x >> 3);
  if (tile0 >= t_hi) return;

  auto span = [&](int32_t tile, Pass &p) {                           // uniform: scalar loads
    p.tile = tile;
    if constexpr (SEG) {
      p.r0 = tile_row[tile];
      p.r1 = tile_row[tile + 1];
      p.e0 = rowptr[p.r0];
      p.e1 = rowptr[p.r1];
    } else {
      p.r0 = p.r1 = 0;
      p.e0 = tile * C::ROWS;
      p.e1 = p.e0 + C::ROWS < R ? p.e0 + C::ROWS : R;
    }
    p.w0 = p.e0;
  };
  // rows beyond the pass are clamped to its last row (always a valid row: R >= 1)
  auto clamp_row = [&](const Pass &p, int rt) {
    int32_t rr = p.w0 + prow(rt);
    const int32_t last = (p.e1 < R ? p.e1 : R) - 1;
    rr = rr < last ? rr : last;
    return rr < 0 ? 0 : rr;
  };

  // ---- the weight ring: chunk c (0 .. 2 KS - 1: W1's k-steps, then W2's) is the same for every pass ---------
  const char *img_w = uniform_ptr(img + (int64_t)wave * C::DPW * 1024);
  int slot = 0;                                                      // slot of the chunk the next step consumes
  const uint32_t lane16 = (uint32_t)lane * 16;
  // (the chunk's offset is added to the scalar base behind an opaque asm and the piece's goes into the
  //  instruction's immediate: left to hipcc, the 64 loop-invariant per-lane offsets -- or the 64 scalar bases --
  //  are hoisted out of the pass loop into registers of their own, and spilled)
  auto dma = [&](int chunk, int into) {                              // this wave's share of one chunk
    uint32_t off = (uint32_t)chunk * C::STEP;
    asm volatile("" : "+s"(off));
    const char *src = img_w + off;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + into * C::STEP + wave * C::DPW * 1024);
    static_for<C::DPW>([&](auto qq) {
      constexpr int q = decltype(qq)::value;
      // the immediate offset (< 4096) applies to the global AND the LDS address
      glds16<(q % 4) * 1024>(src + (q / 4) * 4096, lane16, dst + (q / 4) * 4096);
    });
  };

  float4_t ra[NR], rb[NR], rc[NR];
  // table rows of the CURRENT pass's rows and of the next pass's.  (The gathered G1 / G2 rows stay in the MFMA
  // layout's 64-B pieces: read in the row layout with four indices per lane and turned through LDS, the
  // 64-channel edge kernel measured 99 -> 108 us -- rows of one target share their G2 row anyway.)
  int32_t i1[RT], i2[RT], i1n[RT], i2n[RT];

  Pass cur, nxt;
  span(tile0, cur);
  {
    const int32_t tn = tile0 + stride < t_hi ? tile0 + stride : tile0;
    span(tn, nxt);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {                                  // first pass: rows, indices, G2 rows
    const int32_t rr = clamp_row(cur, rt);
    i1[rt] = M1 == kIdx ? idx1[rr] : rr;
    i2[rt] = M2 == kIdx ? idx2[rr] : rr;
    i1n[rt] = i2n[rt] = 0;
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
      if constexpr (C::ROWIO) {                                      // A rows: the row layout (see pass_body)
        int32_t last = (cur.e1 < R ? cur.e1 : R) - 1;
        last = last < 0 ? 0 : last;
        int32_t row = cur.w0 + (wave * RT + rt) * 16 + 4 * (j % 4) + g;
        row = row < last ? row : last;
        ra[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(A + (int64_t)row * F + (j / 4) * 64 + 4 * mi);
      } else {
        ra[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(A + (int64_t)rr * F + col(j));
      }
      if constexpr (M2 != kNone)
        rb[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(reinterpret_cast<const char *>(G2) +
                                                                 ((uint32_t)i2[rt] * ldb2 + col(j) * 4u));
      else
        rb[rt * C::NJ + j] = float4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  if constexpr (C::RESIDENT) {
    // every block splits the two fp32 matrices into its own LDS image itself (32 KB of L2-resident reads): no
    // pre-split pass, no workspace traffic -- two tiny launches in front of a 100 us kernel cost 4-5 % of it
    for (int p = wave; p < 2 * C::KS * C::NJ; p += C::NW) {          // fragment pairs (matrix, ks, jo)
      const int m = p / (C::KS * C::NJ), ks = (p / C::NJ) % C::KS, jo = p % C::NJ;
      const float *wp = (m ? W2f : W1f) + (int64_t)(jo * 16 + mi) * F + 32 * ks + 4 * g;
      bf16x8 hi, lo;
      split8(*reinterpret_cast<const float4_t *>(wp), *reinterpret_cast<const float4_t *>(wp + 16), hi, lo);
      char *dst = lds + (m * C::KS + ks) * C::STEP + (jo * 2) * 1024 + lane * 16;
      *reinterpret_cast<bf16x8 *>(dst) = hi;
      *reinterpret_cast<bf16x8 *>(dst + 1024) = lo;
    }
    __syncthreads();
  } else {
    dma(0, 0);
    dma(1, 1);
  }

  using V = Vis<F, RT, M1, M2, RES, HID>;

  bool last_pass = false;
  // One pass.  E: this pass's A rows (later the residual).  H: pre-loaded with the G2 rows (or zero) -- the first
  // contraction accumulates onto it, the activation turns it into the hidden layer, the second contraction
  // consumes it and the NEXT pass's A rows move in.  rc: G1 rows, then the second contraction's accumulators.
  auto pass_body = [&](float4_t (&E)[NR], float4_t (&H)[NR]) {
    const int32_t n_rows = cur.e1 - cur.w0;                          // valid rows of this pass (<= 0: none)
    int32_t seg_s = 0, seg_e = 0;                                    // phase 4's bounds, requested early
    if constexpr (SEG) {
      const int32_t r = cur.r0 + t / C::Q;
      seg_s = rowptr[r < cur.r1 ? r : cur.r1];
      seg_e = rowptr[r < cur.r1 ? r + 1 : cur.r1];
    }
    // the pass after this one
    Pass fol = cur;
    bool more = true;
    if (cur.w0 + C::ROWS < cur.e1) {
      fol.w0 = cur.w0 + C::ROWS;                                     // a tile longer than one pass
    } else if (nxt.tile != cur.tile) {
      fol = nxt;
    } else {
      more = false;                                                  // last pass: prefetch it again, unused
    }
    int32_t rr_next[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) rr_next[rt] = clamp_row(fol, rt);
    int32_t last_next = (fol.e1 < R ? fol.e1 : R) - 1;
    last_next = last_next < 0 ? 0 : last_next;
    const int64_t pass_off = (int64_t)cur.w0 * F;
    // E arrives in the row layout (see the loads in the second contraction's steps): through this wave's own 16
    // rows of the y tile -- free between two aggregations, touched by no other wave here -- into the MFMA layout
    auto to_mfma = [&](float4_t (&X)[NR]) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        int gt = g, mt = mi;                                         // opaque: no hoisted LDS addresses
        asm volatile("" : "+v"(gt), "+v"(mt));
        float *yown = ytile + (wave * RT + rt) * 16 * C::PY;
        float *yw = yown + gt * C::PY + 4 * mt;                      // row layout: row 4 k + g, columns 4 (lane % 16)
        const float *yr = yown + mt * C::PY + 4 * gt;                // MFMA layout: row lane % 16, columns 16 j + 4 g
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) {
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<float4_t *>(yw + 4 * k * C::PY) = X[rt * C::NJ + 4 * c + k];
#pragma unroll
          for (int j = 0; j < 4; ++j) X[rt * C::NJ + 4 * c + j] = *reinterpret_cast<const float4_t *>(yr + 16 * j);
        }
      }
    };
    if constexpr (C::ROWIO) to_mfma(E);

    static_for<2 * C::KS>([&](auto ss) {
      constexpr int si = decltype(ss)::value;
      constexpr bool first = si < C::KS;                             // which contraction
      constexpr int ks = first ? si : si - C::KS;
      // this step's view of the lane's column offset: opaque, or hipcc keeps 16 loop-invariant offsets per
      // table live across the whole pass loop (and spills them)
      int gs = g, ms = mi;
      asm volatile("" : "+v"(gs), "+v"(ms));
      auto col = [&](int j) { return 16 * j + 4 * gs; };
      if constexpr (!C::RESIDENT) {
        // ---- chunk si has landed (its DMA is two steps old); everything younger may stay in flight ----------
        if constexpr (si == 0) wait_vm<0>();
        else if constexpr (si == 1) wait_vm<C::DPW + V::of(0)>();
        else wait_vm<V::of(si - 2) + C::DPW + V::of(si - 1)>();
        __syncthreads();                                             // ... for every wave; slot - 1 is free
      }
      if constexpr (si == C::KS) {
        // the activation: hidden = act(acc + G1 rows + b1), in place; the second accumulators start at b2.
        // BEFORE this step's DMAs: hipcc waits for the G1 rows with its own count, which does not know the
        // DMAs and would drain the ones just issued with them.  (Both waves of a SIMD are here at once: ~8 000
        // cycles a pass.  Activating tile by tile in the steps that consume the tiles, beside the other wave's
        // MFMAs, needs ~10 more registers than the 256 there are: 150-220 spilled, slower.)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) {
            float4_t v = H[rt * C::NJ + j] + *reinterpret_cast<const float4_t *>(bl + col(j));
            if constexpr (HID) {
              v = v * rc[rt * C::NJ + j];
              H[rt * C::NJ + j] = v;
              // (every lane stores: rows past the launch's last row land in the buffer's padding)
              *reinterpret_cast<float4_t *>(hid + pass_off + (int64_t)prow(rt) * F + col(j)) = v;
            } else {
              if constexpr (M1 != kNone) v += rc[rt * C::NJ + j];
              H[rt * C::NJ + j] = activate(v, act);
            }
            rc[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(bl + F + col(j));
          }
      }
      if constexpr (!C::RESIDENT) {
        const int into = slot >= 1 ? slot - 1 : C::NSLOT - 1;        // (slot + 2) % 3
        dma((si + 2) % (2 * C::KS), into);
      }
      // ---- this step's B operands, from registers ----------------------------------------------------------
      bf16x8 bh[RT], bo[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if constexpr (first) {
          split8(E[rt * C::NJ + 2 * ks], E[rt * C::NJ + 2 * ks + 1], bh[rt], bo[rt]);
        } else {
          split8(H[rt * C::NJ + 2 * ks], H[rt * C::NJ + 2 * ks + 1], bh[rt], bo[rt]);
        }
      }
      // ---- the loads that ride along (issued after the DMAs, counted by vis()) ---------------------------------
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if constexpr (first) {
          if constexpr (M1 != kNone && si < C::GS) {
            const char *p1 = reinterpret_cast<const char *>(G1) + ((uint32_t)i1[rt] * ldb1 + 16u * gs);
#pragma unroll
            for (int j = si * (C::NJ / C::GS); j < (si + 1) * (C::NJ / C::GS); ++j)
              rc[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(p1 + 64 * j);
          }
          if constexpr (RES == kResOther) {                          // E's registers are free: the residual rows
            int row = prow(rt) < n_rows ? prow(rt) : (n_rows > 0 ? n_rows - 1 : 0);
            const char *pr = reinterpret_cast<const char *>(res + pass_off) + (uint32_t)(row * F * 4 + 16 * gs);
#pragma unroll
            for (int j = 2 * ks; j < 2 * ks + 2; ++j)
              E[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(pr + 64 * j);
          }
        } else {
          if constexpr (si == C::KS) {
            i1n[rt] = rr_next[rt];
            i2n[rt] = rr_next[rt];
            if constexpr (M1 == kIdx) i1n[rt] = idx1[rr_next[rt]];
            if constexpr (M2 == kIdx) i2n[rt] = idx2[rr_next[rt]];
          }
          // the next pass's A rows in the ROW layout (register 4 c + k: row 4 k + lane / 16, columns 64 c + 4 (lane
          // % 16) ..+3 -- 16 lanes read 256 contiguous bytes); the pass turns them into the MFMA layout at its top
          if constexpr (C::ROWIO) {
#pragma unroll
            for (int j = 2 * ks; j < 2 * ks + 2; ++j) {
              int32_t row = fol.w0 + (wave * RT + rt) * 16 + 4 * (j % 4) + gs;
              row = row < last_next ? row : last_next;
              H[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(A + (int64_t)row * F + (j / 4) * 64 + 4 * ms);
            }
          } else {
            const char *pa = reinterpret_cast<const char *>(A + (int64_t)rr_next[rt] * F) + 16 * gs;
#pragma unroll
            for (int j = 2 * ks; j < 2 * ks + 2; ++j)
              H[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(pa + 64 * j);
          }
        }
      }
      // ---- 16 column tiles x RT row tiles x 3 products; W fragments from the ring, two accumulators a group ----
      const char *wb = lds + slot * C::STEP + lane * 16;
#pragma unroll
      for (int jo = 0; jo < C::NJ; jo += C::JG) {
        bf16x8 wh[C::JG], wl[C::JG];
#pragma unroll
        for (int d = 0; d < C::JG; ++d) {
          const int jr = jo + d;
          wh[d] = *reinterpret_cast<const bf16x8 *>(wb + (jr * 2) * 1024);
          wl[d] = *reinterpret_cast<const bf16x8 *>(wb + (jr * 2 + 1) * 1024);
        }
#pragma unroll
        for (int kind = 0; kind < 3; ++kind)
#pragma unroll
          for (int d = 0; d < C::JG; ++d)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              const bf16x8 wa = kind == 1 ? wl[d] : wh[d];
              const bf16x8 xb = kind == 0 ? bo[rt] : bh[rt];
              const int ix = rt * C::NJ + jo + d;
              if constexpr (first) H[ix] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, H[ix], 0, 0, 0);
              else rc[ix] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, rc[ix], 0, 0, 0);
            }
      }
      slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
    });

    // ---- out = res + y stored from registers; E's registers then take the next pass's G2 rows (its first
    // contraction accumulates onto them).  (Issued in four parts inside the aggregation's chunks instead:
    // 679 -> 701 us.) --------------------------------------------------------------------------------------------
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int gs = g, ps = mi;
      asm volatile("" : "+v"(gs), "+v"(ps));
      if constexpr (!C::ROWST) {
        if (out && prow(rt) < n_rows) {
          char *po = reinterpret_cast<char *>(out + pass_off) + (uint32_t)(prow(rt) * F * 4 + 16 * gs) + 0 * ps;
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) {
            float4_t o = rc[rt * C::NJ + j];
            if constexpr (RES != kResNone) o += E[rt * C::NJ + j];
            *reinterpret_cast<float4_t *>(po + 64 * j) = o;
          }
        }
      } else if (out) {
        // through the wave's own rows of the y tile into the row layout: 16 lanes store 256 contiguous bytes
        float *yown = ytile + (wave * RT + rt) * 16 * C::PY;
        const int wrow = (wave * RT + rt) * 16;
        float *po = out + pass_off + (int64_t)(wrow + gs) * F + 4 * ps;       // ps: lane % 16 here
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4_t o = rc[rt * C::NJ + 4 * c + j];
            if constexpr (RES != kResNone) o += E[rt * C::NJ + 4 * c + j];
            *reinterpret_cast<float4_t *>(yown + ps * C::PY + 16 * j + 4 * gs) = o;
          }
          if constexpr (C::ROWIO) {
            float4_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4_t *>(yown + (4 * k + gs) * C::PY + 4 * ps);
            if (wrow + 16 <= n_rows) {                                 // the wave's 16 rows all exist: no per-row test
#pragma unroll
              for (int k = 0; k < 4; ++k) *reinterpret_cast<float4_t *>(po + (int64_t)(4 * k) * F + 64 * c) = v[k];
            } else {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (wrow + 4 * k + gs < n_rows) *reinterpret_cast<float4_t *>(po + (int64_t)(4 * k) * F + 64 * c) = v[k];
            }
          } else {                                                     // 256 channels: one piece at a time
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float4_t v = *reinterpret_cast<const float4_t *>(yown + (4 * k + gs) * C::PY + 4 * ps);
              if (wrow + 4 * k + gs < n_rows) *reinterpret_cast<float4_t *>(po + (int64_t)(4 * k) * F + 64 * c) = v;
              asm volatile("" ::: "memory");
            }
          }
        }
      }
      const char *p2 = reinterpret_cast<const char *>(G2) + ((uint32_t)i2n[rt] * ldb2 + 16u * gs);
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) {
        if constexpr (M2 != kNone)
          E[rt * C::NJ + j] = *reinterpret_cast<const float4_t *>(p2 + 64 * j);
        else
          E[rt * C::NJ + j] = float4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    // ---- the messages of each target row, summed in stored order, 64 columns at a time -------------------------
    if constexpr (SEG) {
      const int q = t % C::Q;
      const int32_t w0 = cur.w0;
      static_for<C::NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int j = 0; j < C::JC; ++j)
            *reinterpret_cast<float4_t *>(ytile + prow(rt) * C::PY + 16 * j + 4 * g) = rc[rt * C::NJ + c * C::JC + j];
        __syncthreads();
        int32_t s = seg_s, en = seg_e;
        for (int32_t r = cur.r0 + t / C::Q; r < cur.r1; r += C::SLOTS) {
          if (r != cur.r0 + t / C::Q) {                              // a pass with more target rows than slots
            s = rowptr[r];
            en = rowptr[r + 1];
          }
          const int32_t lo = s > w0 ? s : w0;
          const int32_t hi = en < w0 + C::ROWS ? en : w0 + C::ROWS;
          float *dst = agg + (int64_t)r * F + c * C::YC + 4 * q;
          if (lo < hi) {
            float4_t acc = {0.f, 0.f, 0.f, 0.f};
            if (s < w0) acc = *reinterpret_cast<const float4_t *>(dst);     // this block's own partial
            for (int32_t e = lo; e < hi; ++e)
              acc += *reinterpret_cast<const float4_t *>(ytile + (e - w0) * C::PY + 4 * q);
            if (mean && hi == en) {
              const float inv = 1.0f / (float)(en - s);
              acc *= float4_t{inv, inv, inv, inv};
            }
            *reinterpret_cast<float4_t *>(dst) = acc;
          } else if (s == en && w0 == cur.e0) {
            *reinterpret_cast<float4_t *>(dst) = float4_t{0.f, 0.f, 0.f, 0.f};     // no in-edges
          }
        }
        __syncthreads();
      });
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      i1[rt] = i1n[rt];
      i2[rt] = i2n[rt];
    }
    if (!more) {
      last_pass = true;
      return;
    }
    if (fol.tile != cur.tile) {                                      // moved on to the next tile: look one ahead
      const int32_t tn = fol.tile + stride < t_hi ? fol.tile + stride : fol.tile;
      span(tn, nxt);
    }
    cur = fol;
  };

  for (;;) {
    pass_body(ra, rb);
    if (last_pass) break;
    pass_body(rb, ra);
    if (last_pass) break;
  }
  if constexpr (!C::RESIDENT) wait_vm<0>();                          // the DMAs issued past the last chunk used
}

template <int F, int M1, int M2>
int launch_rows(const float *A, const float *W1, const float *G1, const int32_t *idx1, const float *G2,
                const int32_t *idx2, const float *b1, const float *W2, const float *b2, const float *res,
                float *out, int64_t R, int act, const int32_t *rowptr, const int32_t *tile_row,
                int64_t n_tiles, float *agg, int mean, void *workspace, uint32_t ldb1, uint32_t ldb2,
                hipStream_t st) {
  using C = RCfg<F, K6R_RT>;
  const bool seg = agg != nullptr;
  bf16x8 *img = reinterpret_cast<bf16x8 *>(workspace);               // W1's k-steps, then W2's
  if constexpr (!C::RESIDENT) {                                      // 64 channels: split inside the kernel
    k_split_wr<F><<<dim3(C::NJ * C::KS, 2), 64, 0, st>>>(W1, W2, img);
    GWEN_LAUNCH_CHECK();
  }
  const int64_t tiles = seg ? n_tiles : (R + C::ROWS - 1) / C::ROWS;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    GWEN_HIP_CHECK(hipGetDevice(&dev));
    GWEN_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = n < 8 ? 8 : n;
  }
  int64_t blocks = (int64_t)cus * (C::RESIDENT ? 2 : 1);              // 130 KB of LDS at 256 channels, 68 KB at 64
  if (blocks > tiles) blocks = tiles;
  const int r = !res ? kResNone : (res == A ? kResA : kResOther);
  const char *im = reinterpret_cast<const char *>(img);
#define GWEN_R(SEGV, RV)                                                                              \
  k_mlp2r<F, K6R_RT, M1, M2, SEGV, RV><<<(unsigned)blocks, C::NW * 64, 0, st>>>(                              \
      A, im, W1, W2, G1, idx1, G2, idx2, b1, b2, res, out, (int32_t)R, act, SEGV ? rowptr : nullptr,  \
      SEGV ? tile_row : nullptr, (int32_t)tiles, SEGV ? agg : nullptr, SEGV ? mean : 0, ldb1, ldb2)
  if (seg) {
    if (r == kResNone) GWEN_R(true, kResNone); else if (r == kResA) GWEN_R(true, kResA); else GWEN_R(true, kResOther);
  } else {
    if (r == kResNone) GWEN_R(false, kResNone); else if (r == kResA) GWEN_R(false, kResA); else GWEN_R(false, kResOther);
  }
#undef GWEN_R
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

// the block's edge-level backward (see k_mlp2r, HID): hid = (ge W2t^T + T[dst]) * d1 ; out = ge + hid Wet^T
template <int F>
int launch_rows_bwd(const float *ge, const float *W2t, const float *d1, const float *T, const int32_t *dst,
                    const float *Wet, float *hid, float *out, int64_t R, void *workspace, uint32_t ldbT, hipStream_t st) {
  using C = RCfg<F, K6R_RT>;
  bf16x8 *img = reinterpret_cast<bf16x8 *>(workspace);
  if constexpr (!C::RESIDENT) {
    k_split_wr<F><<<dim3(C::NJ * C::KS, 2), 64, 0, st>>>(W2t, Wet, img);
    GWEN_LAUNCH_CHECK();
  }
  const int64_t tiles = (R + C::ROWS - 1) / C::ROWS;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    GWEN_HIP_CHECK(hipGetDevice(&dev));
    GWEN_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = n < 8 ? 8 : n;
  }
  int64_t blocks = (int64_t)cus * (C::RESIDENT ? 2 : 1);
  if (blocks > tiles) blocks = tiles;
  k_mlp2r<F, K6R_RT, kSelf, kIdx, false, kResA, true><<<(unsigned)blocks, C::NW * 64, 0, st>>>(
      ge, reinterpret_cast<const char *>(img), W2t, Wet, d1, nullptr, T, dst, nullptr, nullptr, ge, out, (int32_t)R,
      GWEN_ACT_NONE, nullptr, nullptr, (int32_t)tiles, nullptr, 0, (uint32_t)(F * 4), ldbT, hid);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

}  // namespace

int gwen_mlp2_rows_f() { return RCfg<256, K6R_RT>::ROWS; }   // the same at every width this kernel takes

int gwen_mlp2_rows_bwd_launch(int F, const float *ge, const float *W2t, const float *d1, const float *T,
                              const int32_t *dst, const float *Wet, float *hid, float *out, int64_t R, void *workspace,
                              uint32_t ldbT, hipStream_t st) {
  if (F == 64) return launch_rows_bwd<64>(ge, W2t, d1, T, dst, Wet, hid, out, R, workspace, ldbT, st);
  if (F == 256) return launch_rows_bwd<256>(ge, W2t, d1, T, dst, Wet, hid, out, R, workspace, ldbT, st);
  return GWEN_EINVAL;
}

// interact.hip's dispatch for F = 256 (pointers validated there); m1 / m2 as interact.hip's kNone / kSelf / kIdx
int gwen_mlp2_rows_launch(int F, int m1, int m2, const float *A, const float *W1, const float *G1, const int32_t *idx1,
                          const float *G2, const int32_t *idx2, const float *b1, const float *W2,
                          const float *b2, const float *res, float *out, int64_t R, int act,
                          const int32_t *rowptr, const int32_t *tile_row, int64_t n_tiles, float *agg,
                          int mean, void *workspace, uint32_t ldb1, uint32_t ldb2, hipStream_t st) {
#define GWEN_MODE(FF, A1, A2)                                                                         \
  if (F == FF && m1 == A1 && m2 == A2)                                                                \
    return launch_rows<FF, A1, A2>(A, W1, G1, idx1, G2, idx2, b1, W2, b2, res, out, R, act, rowptr,   \
                                   tile_row, n_tiles, agg, mean, workspace, ldb1, ldb2, st)
  GWEN_MODE(256, kNone, kNone); GWEN_MODE(256, kSelf, kNone); GWEN_MODE(256, kIdx, kNone); GWEN_MODE(256, kIdx, kIdx);
  GWEN_MODE(64, kNone, kNone); GWEN_MODE(64, kSelf, kNone); GWEN_MODE(64, kIdx, kNone); GWEN_MODE(64, kIdx, kIdx);
#undef GWEN_MODE
  return GWEN_EINVAL;
}
