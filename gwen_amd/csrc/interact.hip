// K6 -- the InteractionNet block's two MLPs as ONE kernel shape (BUILD-DEFINED: the
// reference has no edge MLP -- BASELINE.json names it; semantics follow the published Interaction
// Network / encode-process-decode formulation, oracle/interaction_oracle.py restates them on the CPU):
//
//     pre[r] = A[r] W1^T + G1[idx1[r]] + G2[idx2[r]] + b1          (gathered addends = the node halves
//     y[r]   = act(pre[r]) W2^T + b2                                 of the first layer, pre-projected)
//     out[r] = res[r] + y[r]                                         (residual update)
//     agg[d] = sum_{r : dst(r) = d} y[r]      (rows sorted by d)     (aggregation, optional mean)
//
// With rows = edges in target-sorted (CSR) order this is  e' = e + MLP_e([e, x_src, x_dst]),
// agg = sum of messages; with rows = nodes, A = agg, G1 = x W^T it is  x' = x + MLP_n([x, agg]).
// Neither the [E,3F] concatenation, nor the hidden layer, nor the message tensor reaches HBM.
//
// Shape (wave64): a block = 8 waves that walk tiles of at most 64 rows persistently (one resident set
//   of blocks; the tiles of one XCD are one contiguous eighth of them);
//   a wave owns NC column tiles (16 NC output columns) of BOTH contractions and keeps its slices of W1
//   and W2 in registers as 3xbf16 fragments (hi/lo, see layer.hip) with W as the MFMA A operand, so a lane
//   ends up with 4 consecutive columns of one row (16-B gathers, 16-B stores).  At 256 channels the
//   fragments (512 KB) exceed the register file: that width runs on the row-stationary kernel of
//   interact_rows.hip instead (weights streamed through an LDS ring, activations in registers), and so does
//   64 channels (weights resident in LDS, rows read and written 256 contiguous bytes at a time): this kernel
//   serves 32 and 128 channels;
//   phase 1  A rows, prefetched during the previous pass in the OUTPUT layout (the residual stays in
//            registers) -> hi/lo bf16 LDS tile;
//   phase 2  pre = MFMA + gathered addends + b1, activation -> hi/lo hidden tile (its own LDS image);
//   phase 3  y = MFMA + b2; out = res + y stored from registers;
//   phase 4  (aggregation) y -> LDS fp32 tile (aliases the A tile); one thread per (target row, 4
//            columns) adds its rows' entries in stored order -- no atomics, bitwise reproducible.
// Tiles are ROW-ALIGNED (gwen_edge_tiles): tile c owns the target rows whose first edge lies in
//   [cT, (c+1)T), so every target row is summed by exactly one block; T = 64 - (max degree - 1) makes a
//   tile one pass of 64 edges on bounded-degree graphs, and a row longer than a pass is carried
//   through the block's own earlier partial sum (same thread, program order).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxRows = 128;             // most rows a pass takes (MCfg<F>::ROWS; 128 at 256 channels)

template <int F>
struct MCfg {
  static constexpr int NJ = F / 16;                              // 16-column output tiles
  // rows per pass.  (32 at 128 channels -- half the per-wave state, two blocks per CU -- measured
  // slower: 357 vs 259 us, twice the passes and their fixed costs.)
  static constexpr int ROWS = 64;
  static constexpr int NT = ROWS / 16;
  // A wave owns NC column tiles.  With one tile every hi/lo operand pair read from LDS feeds three
  // MFMAs, with two it would feed six: at 64 channels the doubled fragments and tiles spill, at 128 it is a wash.
  static constexpr int NC = 1;
  static constexpr int NJW = NJ / NC;                            // column groups
  // every column group is shared by TSTEP waves (each takes every TSTEP-th 16-row tile)
  static constexpr int TSTEP = F == 32 ? 4 : (F == 64 ? 2 : 1);
  static constexpr int NWB = NJW * TSTEP;                        // waves per block: 8 at every width
  static constexpr int MINW = F >= 128 ? 2 : 4;                  // waves per SIMD to compile for
  static constexpr int TPW = NT / TSTEP;                         // row tiles per wave
  static constexpr int KS = F / 32;                              // k-steps of v_mfma_f32_16x16x32_bf16
  static constexpr int PB = ((F / 2) % 16 == 8 ? F / 2 : F / 2 + 8) * 2;   // hi/lo tile pitch (bf16)
  static constexpr int PY = F + 4;                               // y tile pitch (floats)
  static constexpr size_t split_bytes = (size_t)2 * ROWS * PB * 2;
  static constexpr size_t y_bytes = (size_t)ROWS * PY * 4;
  // two hi/lo images: the A tile (reused as the fp32 y tile of phase 4) and the hidden tile
  static constexpr size_t a_bytes = split_bytes > y_bytes ? split_bytes : y_bytes;
  static constexpr size_t lds_bytes = a_bytes + split_bytes;
  static constexpr int Q = F / 4;                                // 16-B pieces per row
  static constexpr int SLOTS = NWB * 64 / Q;                     // target rows reduced at a time
  static_assert(NT % TSTEP == 0 && NJ % NC == 0, "tiles must divide over the waves");
};

__device__ inline void split4(const float4_t v, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// column tile jt (16 output columns) of W^T (W is [F,F] row-major [out,in]) as hi/lo A-operand fragments
template <int F>
__device__ inline void load_w(const float *__restrict__ W, int jt, int mi, int mh,
                              bf16x8 (&whi)[MCfg<F>::KS], bf16x8 (&wlo)[MCfg<F>::KS]) {
  const float *wrow = W + (int64_t)(jt * 16 + mi) * F;
#pragma unroll
  for (int ks = 0; ks < MCfg<F>::KS; ++ks) {
    const float4_t a = *reinterpret_cast<const float4_t *>(wrow + 8 * (4 * ks + mh));
    const float4_t b = *reinterpret_cast<const float4_t *>(wrow + 8 * (4 * ks + mh) + 4);
    bf16x4 h0, l0, h1, l1;
    split4(a, h0, l0);
    split4(b, h1, l1);
    whi[ks] = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    wlo[ks] = bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  }
}

// One k-step of TPW row tiles x NC column tiles: the operand pairs are read once, the products of
// one kind go to all TPW * NC independent accumulators before the next kind touches them again.
template <int F>
__device__ inline void mma_step(const bf16x8 (&ahi)[MCfg<F>::TPW], const bf16x8 (&alo)[MCfg<F>::TPW],
                                const bf16x8 (&whi)[MCfg<F>::NC], const bf16x8 (&wlo)[MCfg<F>::NC],
                                f32x4 (&d)[MCfg<F>::TPW][MCfg<F>::NC]) {
  using C = MCfg<F>;
#pragma unroll
  for (int n = 0; n < C::NC; ++n)
#pragma unroll
    for (int k = 0; k < C::TPW; ++k)
      d[k][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[n], alo[k], d[k][n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < C::NC; ++n)
#pragma unroll
    for (int k = 0; k < C::TPW; ++k)
      d[k][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[n], ahi[k], d[k][n], 0, 0, 0);
#pragma unroll
  for (int n = 0; n < C::NC; ++n)
#pragma unroll
    for (int k = 0; k < C::TPW; ++k)
      d[k][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[n], ahi[k], d[k][n], 0, 0, 0);
}

template <int F>
__device__ inline void read_tiles(const __bf16 *thi, const __bf16 *tlo, int tt0, int mi, int mh, int ks,
                                  bf16x8 (&ahi)[MCfg<F>::TPW], bf16x8 (&alo)[MCfg<F>::TPW]) {
  using C = MCfg<F>;
#pragma unroll
  for (int k = 0; k < C::TPW; ++k) {
    const int arow = ((tt0 + k * C::TSTEP) * 16 + mi) * C::PB + 8 * (4 * ks + mh);
    ahi[k] = *reinterpret_cast<const bf16x8 *>(thi + arow);
    alo[k] = *reinterpret_cast<const bf16x8 *>(tlo + arow);
  }
}

// weights resident in registers
template <int F>
__device__ inline void tiles_mma(const __bf16 *thi, const __bf16 *tlo, int tt0, int mi, int mh,
                                 const bf16x8 (&whi)[MCfg<F>::NC][MCfg<F>::KS],
                                 const bf16x8 (&wlo)[MCfg<F>::NC][MCfg<F>::KS],
                                 f32x4 (&d)[MCfg<F>::TPW][MCfg<F>::NC]) {
  using C = MCfg<F>;
#pragma unroll
  for (int k = 0; k < C::TPW; ++k)
#pragma unroll
    for (int n = 0; n < C::NC; ++n) d[k][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    bf16x8 ahi[C::TPW], alo[C::TPW], wh[C::NC], wl[C::NC];
    read_tiles<F>(thi, tlo, tt0, mi, mh, ks, ahi, alo);
#pragma unroll
    for (int n = 0; n < C::NC; ++n) {
      wh[n] = whi[n][ks];
      wl[n] = wlo[n][ks];
    }
    mma_step<F>(ahi, alo, wh, wl, d);
  }
}

// SiLU on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each): x / (1 + 2^(-x log2 e))
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

// M1 / M2: how table G1 / G2 is addressed -- 0 absent, 1 row r itself, 2 row idx[r]
constexpr int kNone = 0, kSelf = 1, kIdx = 2;

// One PASS = up to 64 consecutive rows [w0, min(w0 + 64, e1)) of tile `tile` (target rows r0 .. r1-1).
struct Pass {
  int32_t tile, r0, r1;
  int32_t w0, e0, e1;
};

template <int F, int M1, int M2, bool SEG>
__global__ __launch_bounds__(MCfg<F>::NWB * 64, MCfg<F>::MINW) void k_mlp2(
    const float *__restrict__ A, const float *__restrict__ W1, const float *__restrict__ G1,
    const int32_t *__restrict__ idx1, const float *__restrict__ G2,
    const int32_t *__restrict__ idx2, const float *__restrict__ b1, const float *__restrict__ W2,
    const float *__restrict__ b2, const float *__restrict__ res, float *__restrict__ out,
    int32_t R, int act, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ tile_row, int32_t n_tiles, float *__restrict__ agg, int mean,
    uint32_t ldb1, uint32_t ldb2) {                     // row strides of G1 / G2 in bytes
  using C = MCfg<F>;
  __shared__ __attribute__((aligned(16))) char lds_raw[C::lds_bytes];
  __bf16 *thi = reinterpret_cast<__bf16 *>(lds_raw);                    // A tile, hi / lo
  __bf16 *tlo = thi + C::ROWS * C::PB;
  float *ytile = reinterpret_cast<float *>(lds_raw);                     // aliases the A tile (phase 4)
  __bf16 *hhi = reinterpret_cast<__bf16 *>(lds_raw + C::a_bytes);        // hidden tile, hi / lo
  __bf16 *hlo = hhi + C::ROWS * C::PB;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int mi = lane & 15, mh = lane >> 4;
  const int jw = wave % C::NJW, tt0 = wave / C::NJW;
  // this lane's output columns: 4 consecutive ones in each of the wave's NC column tiles
  auto col = [&](int n) { return (jw * C::NC + n) * 16 + 4 * mh; };

  // resident: this wave's hi/lo fragments of both matrices
  bf16x8 w1hi[C::NC][C::KS], w1lo[C::NC][C::KS], w2hi[C::NC][C::KS], w2lo[C::NC][C::KS];
#pragma unroll
  for (int n = 0; n < C::NC; ++n) {
    load_w<F>(W1, jw * C::NC + n, mi, mh, w1hi[n], w1lo[n]);
    load_w<F>(W2, jw * C::NC + n, mi, mh, w2hi[n], w2lo[n]);
  }
  float4_t b1v[C::NC], b2v[C::NC];
#pragma unroll
  for (int n = 0; n < C::NC; ++n) {
    b1v[n] = b1 ? *reinterpret_cast<const float4_t *>(b1 + col(n)) : float4_t{0.f, 0.f, 0.f, 0.f};
    b2v[n] = b2 ? *reinterpret_cast<const float4_t *>(b2 + col(n)) : float4_t{0.f, 0.f, 0.f, 0.f};
  }

  // tiles are dealt so that the blocks of one XCD (blockIdx % 8) walk ONE contiguous eighth of them:
  // neighbouring tiles gather neighbouring table rows, which then meet in that XCD's L2
  const int nb = gridDim.x, xcd = blockIdx.x & 7;
  const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
  const int t_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int t_hi = t_lo + q8 + (xcd < r8 ? 1 : 0);
  const int stride = (nb + 7 - xcd) >> 3;                  // blocks on this XCD
  int tile0 = t_lo + (blockIdx.x >> 3);
  if (tile0 >= t_hi) return;

  auto span = [&](int32_t tile, Pass &p) {                 // uniform: scalar loads
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
  auto clamp_row = [&](const Pass &p, int row) {
    int32_t rr = p.w0 + row;
    const int32_t last = (p.e1 < R ? p.e1 : R) - 1;
    rr = rr < last ? rr : last;
    return rr < 0 ? 0 : rr;
  };
  // A is read in the OUTPUT layout (this lane: rows (tt0 + k TSTEP) 16 + mi, columns col(n) .. +3), so
  // the residual of out = A + y is already in this lane's registers
  float4_t areg[C::TPW][C::NC];
  int32_t i1[C::TPW], i2[C::TPW];
  auto prefetch = [&](const Pass &p) {                     // A rows and gather indices of pass p
#pragma unroll
    for (int k = 0; k < C::TPW; ++k) {
      const int32_t rr = clamp_row(p, (tt0 + k * C::TSTEP) * 16 + mi);
#pragma unroll
      for (int n = 0; n < C::NC; ++n)
        areg[k][n] = *reinterpret_cast<const float4_t *>(A + (int64_t)rr * F + col(n));
      i1[k] = M1 == kIdx ? idx1[rr] : rr;
      i2[k] = M2 == kIdx ? idx2[rr] : rr;
    }
  };
  const bool res_is_a = res == A;

  Pass cur, nxt;
  span(tile0, cur);
  {
    const int32_t tn = tile0 + stride < t_hi ? tile0 + stride : tile0;
    span(tn, nxt);
  }
  prefetch(cur);
  // The gathered addends of a pass: requested at its top (the indices came with its prefetch).
  float4_t add[C::TPW][C::NC];
  auto gather_addends = [&]() {
#pragma unroll
    for (int k = 0; k < C::TPW; ++k)
#pragma unroll
      for (int n = 0; n < C::NC; ++n) {
        add[k][n] = b1v[n];
        if constexpr (M1 != kNone)
          add[k][n] += *reinterpret_cast<const float4_t *>(
              reinterpret_cast<const char *>(G1) + ((uint32_t)i1[k] * ldb1 + col(n) * 4u));
        if constexpr (M2 != kNone)
          add[k][n] += *reinterpret_cast<const float4_t *>(
              reinterpret_cast<const char *>(G2) + ((uint32_t)i2[k] * ldb2 + col(n) * 4u));
      }
  };
  for (;;) {
    const int32_t n_rows = cur.e1 - cur.w0;                // valid rows of this pass (<= 0: none)
    // ---- the gathered addends of this lane's rows (indices came with the prefetch) ---------------
    gather_addends();
    // phase 4's row bounds for this thread's first target row, requested early
    int32_t seg_s = 0, seg_e = 0;
    if constexpr (SEG) {
      const int32_t r = cur.r0 + t / C::Q;
      seg_s = rowptr[r < cur.r1 ? r : cur.r1];
      seg_e = rowptr[r < cur.r1 ? r + 1 : cur.r1];
    }
    // ---- phase 1: prefetched A rows -> hi/lo tiles -----------------------------------------------
    float4_t rv[C::TPW][C::NC];
#pragma unroll
    for (int k = 0; k < C::TPW; ++k) {
      const int row = (tt0 + k * C::TSTEP) * 16 + mi;
#pragma unroll
      for (int n = 0; n < C::NC; ++n) {
        float4_t v = areg[k][n];
        if (row >= n_rows) v = float4_t{0.f, 0.f, 0.f, 0.f};
        rv[k][n] = v;
        bf16x4 h4, l4;
        split4(v, h4, l4);
        *reinterpret_cast<bf16x4 *>(thi + row * C::PB + col(n)) = h4;
        *reinterpret_cast<bf16x4 *>(tlo + row * C::PB + col(n)) = l4;
      }
    }
    // ---- the pass after this one: its rows and indices travel while this one computes -------------
    Pass fol = cur;
    bool more = true;
    if (cur.w0 + C::ROWS < cur.e1) {
      fol.w0 = cur.w0 + C::ROWS;                             // a tile longer than one pass
    } else if (nxt.tile != cur.tile) {
      fol = nxt;
    } else {
      more = false;                                         // last pass: prefetch it again, unused
    }
    prefetch(fol);
    __syncthreads();
    // ---- phase 2: first contraction, activation --------------------------------------------------
    // (the hidden tile has its own LDS image: the previous pass's second contraction, which read it,
    //  lies before the barrier above for every wave)
    f32x4 d[C::TPW][C::NC];
    tiles_mma<F>(thi, tlo, tt0, mi, mh, w1hi, w1lo, d);
#pragma unroll
    for (int k = 0; k < C::TPW; ++k) {
      const int row = (tt0 + k * C::TSTEP) * 16 + mi;
#pragma unroll
      for (int n = 0; n < C::NC; ++n) {
        const float4_t hk =
            activate(float4_t{d[k][n][0], d[k][n][1], d[k][n][2], d[k][n][3]} + add[k][n], act);
        bf16x4 h4, l4;
        split4(hk, h4, l4);
        *reinterpret_cast<bf16x4 *>(hhi + row * C::PB + col(n)) = h4;
        *reinterpret_cast<bf16x4 *>(hlo + row * C::PB + col(n)) = l4;
      }
    }
    // residual rows other than A: requested before the second contraction, used after it
    const int64_t pass_off = (int64_t)cur.w0 * F;
    if (res && !res_is_a) {
#pragma unroll
      for (int k = 0; k < C::TPW; ++k) {
        int row = (tt0 + k * C::TSTEP) * 16 + mi;
        row = row < n_rows ? row : (n_rows > 0 ? n_rows - 1 : 0);
#pragma unroll
        for (int n = 0; n < C::NC; ++n)
          rv[k][n] = *reinterpret_cast<const float4_t *>(
              reinterpret_cast<const char *>(res + pass_off) + (uint32_t)(row * F * 4 + col(n) * 4));
      }
    }
    __syncthreads();
    // ---- phase 3: second contraction, residual, store ---------------------------------------------
    float4_t y[C::TPW][C::NC];
    tiles_mma<F>(hhi, hlo, tt0, mi, mh, w2hi, w2lo, d);
#pragma unroll
    for (int k = 0; k < C::TPW; ++k) {
      const int row = (tt0 + k * C::TSTEP) * 16 + mi;
#pragma unroll
      for (int n = 0; n < C::NC; ++n) {
        y[k][n] = float4_t{d[k][n][0], d[k][n][1], d[k][n][2], d[k][n][3]} + b2v[n];
        if (out && row < n_rows) {
          float4_t o = y[k][n];
          if (res) o += rv[k][n];
          *reinterpret_cast<float4_t *>(reinterpret_cast<char *>(out + pass_off) +
                                        (uint32_t)(row * F * 4 + col(n) * 4)) = o;
        }
      }
    }
    // ---- phase 4: sum the messages of each target row, in stored order ----------------------------
    if constexpr (SEG) {
      // the A tile is dead (every wave is past the barrier that followed its first contraction)
#pragma unroll
      for (int k = 0; k < C::TPW; ++k) {
        const int row = (tt0 + k * C::TSTEP) * 16 + mi;
#pragma unroll
        for (int n = 0; n < C::NC; ++n)
          *reinterpret_cast<float4_t *>(ytile + row * C::PY + col(n)) = y[k][n];
      }
      __syncthreads();
      const int q = t % C::Q;
      const int32_t w0 = cur.w0;
      for (int32_t r = cur.r0 + t / C::Q; r < cur.r1; r += C::SLOTS) {
        const int32_t s = seg_s, en = seg_e;
        const int32_t lo = s > w0 ? s : w0;
        const int32_t hi = en < w0 + C::ROWS ? en : w0 + C::ROWS;
        float *dst = agg + (int64_t)r * F + 4 * q;
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
        if (r + C::SLOTS < cur.r1) {                       // a tile with more rows than slots
          seg_s = rowptr[r + C::SLOTS];
          seg_e = rowptr[r + C::SLOTS + 1];
        }
      }
    }
    if (!more) break;
    if (fol.tile != cur.tile) {                            // moved on to the next tile: look one ahead
      const int32_t tn = fol.tile + stride < t_hi ? fol.tile + stride : fol.tile;
      span(tn, nxt);
    }
    cur = fol;
    if constexpr (SEG) __syncthreads();                    // the y tile aliases the next pass's A tile
  }
}

// tile_row[c] = first target row whose first edge is at or after c * T  (tile_row[n_tiles] = N),
// dst[e] = target row of stored entry e
__global__ void k_edge_tiles(const int32_t *__restrict__ rowptr, int32_t N, int32_t T,
                             int32_t n_tiles, int32_t *__restrict__ tile_row,
                             int32_t *__restrict__ dst) {
  const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;     // r = 0 .. N
  if (r > N) return;
  if (r == 0) {
    tile_row[0] = 0;
    tile_row[n_tiles] = N;
  }
  if (r == N) return;
  const int32_t s = rowptr[r], en = rowptr[r + 1];
  if (dst)
    for (int32_t e = s; e < en; ++e) dst[e] = r;
  // row r + 1 starts at `en`: it opens every tile c with  s < c T <= en
  for (int32_t c = s / T + 1; c < n_tiles && (int64_t)c * T <= en; ++c) tile_row[c] = r + 1;
}

template <int F, int M1, int M2>
int launch(const float *A, const float *W1, const float *G1, const int32_t *idx1, const float *G2,
           const int32_t *idx2, const float *b1, const float *W2, const float *b2, const float *res,
           float *out, int64_t R, int act, const int32_t *rowptr, const int32_t *tile_row,
           int64_t n_tiles, float *agg, int mean, void *workspace, uint32_t ldb1, uint32_t ldb2,
           hipStream_t st) {
  using C = MCfg<F>;
  const bool seg = agg != nullptr;
  (void)workspace;
  const int64_t tiles = seg ? n_tiles : (R + C::ROWS - 1) / C::ROWS;
  // W1 and W2 (2 x 16 F^2 bytes per block) are fetched once per block: one resident set of blocks
  // walks the tiles, each prefetching its next pass while it computes the current one
  static int per_cu = 0;
  if (per_cu == 0) {
    int nbk = 0;
    GWEN_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nbk, reinterpret_cast<const void *>(&k_mlp2<F, M1, M2, true>), C::NWB * 64, 0));
    per_cu = nbk < 1 ? 1 : nbk;
  }
  int64_t blocks = (int64_t)256 * per_cu;
  if (blocks > tiles) blocks = tiles;
  if (seg)
    k_mlp2<F, M1, M2, true><<<(unsigned)blocks, C::NWB * 64, 0, st>>>(
        A, W1, G1, idx1, G2, idx2, b1, W2, b2, res, out, (int32_t)R, act, rowptr, tile_row,
        (int32_t)tiles, agg, mean, ldb1, ldb2);
  else
    k_mlp2<F, M1, M2, false><<<(unsigned)blocks, C::NWB * 64, 0, st>>>(
        A, W1, G1, idx1, G2, idx2, b1, W2, b2, res, out, (int32_t)R, act, nullptr, nullptr,
        (int32_t)tiles, nullptr, 0, ldb1, ldb2);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

template <int F>
int launch_mode(int m1, int m2, const float *A, const float *W1, const float *G1, const int32_t *idx1,
                const float *G2, const int32_t *idx2, const float *b1, const float *W2,
                const float *b2, const float *res, float *out, int64_t R, int act,
                const int32_t *rowptr, const int32_t *tile_row, int64_t n_tiles, float *agg, int mean,
                void *workspace, uint32_t ldb1, uint32_t ldb2, hipStream_t st) {
#define GWEN_MODE(A1, A2)                                                                        \
  if (m1 == A1 && m2 == A2)                                                                      \
    return launch<F, A1, A2>(A, W1, G1, idx1, G2, idx2, b1, W2, b2, res, out, R, act, rowptr,    \
                             tile_row, n_tiles, agg, mean, workspace, ldb1, ldb2, st)
  GWEN_MODE(kNone, kNone); GWEN_MODE(kSelf, kNone); GWEN_MODE(kIdx, kNone); GWEN_MODE(kIdx, kIdx);
#undef GWEN_MODE
  return GWEN_EINVAL;
}

}  // namespace

// interact_rows.hip: the row-stationary kernel 256 channels run on
int gwen_mlp2_rows_f();
int gwen_mlp2_rows_launch(int F, int m1, int m2, const float *A, const float *W1, const float *G1, const int32_t *idx1,
                          const float *G2, const int32_t *idx2, const float *b1, const float *W2,
                          const float *b2, const float *res, float *out, int64_t R, int act,
                          const int32_t *rowptr, const int32_t *tile_row, int64_t n_tiles, float *agg,
                          int mean, void *workspace, uint32_t ldb1, uint32_t ldb2, hipStream_t st);

extern "C" int gwen_mlp2_supported(int64_t F) {
  return F == 32 || F == 64 || F == 128 || F == 256 ? 1 : 0;
}

extern "C" int gwen_mlp2_rows(int64_t F) {
  return F == 32 ? MCfg<32>::ROWS : F == 64 ? gwen_mlp2_rows_f() : F == 128 ? MCfg<128>::ROWS
       : F == 256 ? gwen_mlp2_rows_f() : GWEN_EINVAL;
}

extern "C" int64_t gwen_mlp2_workspace_bytes(int64_t F) {
  if (!gwen_mlp2_supported(F)) return GWEN_EINVAL;
  return F > 128 ? 2 * 2 * F * F * 2 : 0;      // two matrices x (hi, lo) x bf16
}

extern "C" int64_t gwen_edge_tiles_count(int64_t E, int64_t T) {
  if (E < 0 || T < 1) return GWEN_EINVAL;
  return E == 0 ? 1 : (E + T - 1) / T;
}

extern "C" int gwen_edge_tiles(const int32_t *rowptr, int64_t N, int64_t E, int64_t T,
                               int32_t *tile_row, int32_t *dst, gwen_stream_t stream_) {
  if (N < 0 || E < 0 || T < 1 || T > kMaxRows) return GWEN_EINVAL;
  if (N >= (int64_t(1) << 31) - 1 || E >= (int64_t(1) << 31) - kMaxRows) return GWEN_ERANGE;
  if (!rowptr || !tile_row || (E > 0 && !dst)) return GWEN_EINVAL;
  const int64_t n_tiles = gwen_edge_tiles_count(E, T);
  hipStream_t st = gwen_stream(stream_);
  const int64_t threads = N + 1;
  k_edge_tiles<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(
      rowptr, (int32_t)N, (int32_t)T, (int32_t)n_tiles, tile_row, E > 0 ? dst : nullptr);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

// The edge-level half of the block's backward as ONE launch of the row-stationary kernel (interact_rows.hip, HID):
//     g_pre1 = (ge W2 + T[dst]) * d1        T = (g_agg, scaled for the mean) W2 per NODE: ge + g_agg[dst] never exists
//     g_e    = ge + g_pre1 We
// W2t = W2^T and Wet = We^T, [F, F] row-major (row = output column of the contraction); d1 = act'(pre1) [R, F];
// g_pre1 has gwen_mlp2_bwd_rows(R) rows (whole passes: the kernel stores every lane), g_e has R.  F in {64, 256}.
int gwen_mlp2_rows_bwd_launch(int F, const float *ge, const float *W2t, const float *d1, const float *T,
                              const int32_t *dst, const float *Wet, float *hid, float *out, int64_t R, void *workspace,
                              uint32_t ldbT, hipStream_t st);
extern "C" int64_t gwen_mlp2_bwd_rows(int64_t R) {
  const int64_t rows = gwen_mlp2_rows_f();
  return R <= 0 ? 0 : (R + rows - 1) / rows * rows;
}
extern "C" int gwen_mlp2_bwd_supported(int64_t F) { return F == 64 || F == 256 ? 1 : 0; }
extern "C" int gwen_mlp2_bwd_f32(const float *ge, const float *W2t, const float *d1, const float *T, const int32_t *dst,
                                 int64_t T_rows, int64_t ldT, const float *Wet, float *g_pre1, float *g_e, int64_t R,
                                 int64_t F, void *workspace, size_t workspace_bytes, gwen_stream_t stream_) {
  if (R < 0 || T_rows < 0 || !gwen_mlp2_bwd_supported(F)) return GWEN_EINVAL;
  if (R == 0) return GWEN_OK;
  if (!ge || !W2t || !d1 || !T || !dst || !Wet || !g_pre1 || !g_e || ldT < F || ldT % 4) return GWEN_EINVAL;
  if (g_pre1 == ge || g_pre1 == d1 || g_pre1 == g_e || g_e == d1) return GWEN_EINVAL;      // g_e may alias ge row for row
  if (R >= (int64_t(1) << 31) - kMaxRows || T_rows * ldT * 4 >= (int64_t(1) << 32) || R * F * 4 >= (int64_t(1) << 32))
    return GWEN_ERANGE;              // 32-bit byte offsets into the tables (d1 is read as a table, row for row)
  const int64_t need = gwen_mlp2_workspace_bytes(F);
  if (need > 0 && (!workspace || (int64_t)workspace_bytes < need)) return GWEN_ENOSPACE;
  const void *al[] = {ge, W2t, d1, T, Wet, g_pre1, g_e, need > 0 ? workspace : nullptr};
  for (const void *p : al)
    if (p && !gwen_aligned(p, 16)) return GWEN_EINVAL;
  return gwen_mlp2_rows_bwd_launch((int)F, ge, W2t, d1, T, dst, Wet, g_pre1, g_e, R, workspace, (uint32_t)(ldT * 4),
                                   gwen_stream(stream_));
}

extern "C" int gwen_mlp2_f32(const float *A, const float *W1, const float *G1, const int32_t *idx1,
                             int64_t G1_rows, int64_t ldg1, const float *G2, const int32_t *idx2,
                             int64_t G2_rows, int64_t ldg2,
                             const float *b1, const float *W2, const float *b2, const float *res,
                             float *out, int64_t R, int64_t F, int act, const int32_t *rowptr,
                             const int32_t *tile_row, int64_t n_tiles, float *agg, int64_t N_agg,
                             int mean, void *workspace, size_t workspace_bytes,
                             gwen_stream_t stream_) {
  if (R < 0 || N_agg < 0 || n_tiles < 0 || G1_rows < 0 || G2_rows < 0) return GWEN_EINVAL;
  if (!gwen_mlp2_supported(F)) return GWEN_EINVAL;
  if (act != GWEN_ACT_NONE && act != GWEN_ACT_RELU && act != GWEN_ACT_SILU) return GWEN_EINVAL;
  if (agg && (!rowptr || !tile_row || n_tiles < 1)) return GWEN_EINVAL;
  if (!agg && !out) return R == 0 ? GWEN_OK : GWEN_EINVAL;
  if (R >= (int64_t(1) << 31) - kMaxRows || N_agg >= (int64_t(1) << 31) - 1) return GWEN_ERANGE;
  hipStream_t st = gwen_stream(stream_);
  if (R == 0) {                      // no rows: every target's sum is empty
    if (agg && N_agg > 0) GWEN_HIP_CHECK(hipMemsetAsync(agg, 0, (size_t)N_agg * F * 4, st));
    return GWEN_OK;
  }
  if (!A || !W1 || !W2 || (idx1 && !G1) || (idx2 && !G2) || (G2 && !G1)) return GWEN_EINVAL;
  // a table without an index is read row for row; the supported pairs are listed in the header
  const int m1 = !G1 ? kNone : (idx1 ? kIdx : kSelf), m2 = !G2 ? kNone : (idx2 ? kIdx : kSelf);
  if ((m1 == kSelf && G1_rows < R) || (m2 == kSelf && G2_rows < R)) return GWEN_EINVAL;
  if ((G1 && (ldg1 < F || ldg1 % 4)) || (G2 && (ldg2 < F || ldg2 % 4))) return GWEN_EINVAL;
  if ((G1 && G1_rows * ldg1 * 4 >= (int64_t(1) << 32)) || (G2 && G2_rows * ldg2 * 4 >= (int64_t(1) << 32)))
    return GWEN_ERANGE;              // 32-bit byte offsets into the tables
  if (out && (out == G1 || out == G2)) return GWEN_EINVAL;      // out may alias A / res row for row
  const int64_t need = gwen_mlp2_workspace_bytes(F);
  if (need > 0 && (!workspace || (int64_t)workspace_bytes < need)) return GWEN_ENOSPACE;
  const void *al[] = {A, W1, G1, G2, b1, W2, b2, res, out, agg, need > 0 ? workspace : nullptr};
  for (const void *p : al)
    if (p && !gwen_aligned(p, 16)) return GWEN_EINVAL;
#define GWEN_M(FF)                                                                                \
  if (F == FF)                                                                                    \
    return launch_mode<FF>(m1, m2, A, W1, G1, idx1, G2, idx2, b1, W2, b2, res, out, R, act, rowptr, \
                           tile_row, n_tiles, agg, mean, workspace, (uint32_t)(ldg1 * 4),       \
                           (uint32_t)(ldg2 * 4), st)
  if (F == 256 || F == 64)
    return gwen_mlp2_rows_launch((int)F, m1, m2, A, W1, G1, idx1, G2, idx2, b1, W2, b2, res, out, R, act, rowptr, tile_row,
                                 n_tiles, agg, mean, workspace, (uint32_t)(ldg1 * 4), (uint32_t)(ldg2 * 4), st);
  GWEN_M(32); GWEN_M(128);
#undef GWEN_M
  return GWEN_EINVAL;
}
