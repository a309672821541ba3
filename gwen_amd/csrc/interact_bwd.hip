// K6^T -- element-wise and gather pieces of the InteractionNet block's BACKWARD (gwen_amd/interaction.py).
//
// BUILD-DEFINED like the block itself (the reference has no edge MLP: its only graph layer is GCNConv,
// /root/reference/src/gwen/models_gnn.py:118-130; its training step -- loss.backward(), :372-373 -- is what this
// path serves for the InteractionNet forecaster).  The backward is assembled on the host from launches that are all
// atomic-free and fixed-order, so two runs are bitwise equal:
//   dense products            K3 (gwen_gcn_linear_f32)
//   sums over the edges of a target / of a source      K2 (gwen_gcn_propagate_f32) over a CSR whose columns are
//                             EDGE positions (unit or 1/degree weights): every row summed in stored order
//   weight / bias gradients   gwen_gcn_grad_weight_f32 / gwen_gcn_grad_bias_f32 (fixed-order two-stage reductions)
//   and the three kernels below: the activation with its derivative behind gathered addends, the gathered add
//   that forms the message gradient, and plain element-wise products / sums.
#include "common.h"

namespace {

__device__ inline float act_fwd(float x, int act, float &d) {
  if (act == GWEN_ACT_RELU) {
    d = x > 0.0f ? 1.0f : 0.0f;
    return x > 0.0f ? x : 0.0f;
  }
  if (act == GWEN_ACT_SILU) {
    const float s = 1.0f / (1.0f + __expf(-x));          // sigmoid
    d = s * (1.0f + x * (1.0f - s));
    return x * s;
  }
  d = 1.0f;
  return x;
}

// h = act(pre), dact = act'(pre), pre = a + g1[idx1 or row] + g2[idx2 or row]; one thread per 4 columns.
// h may be a (in place: every thread loads its 4 elements before it stores them) -- so neither is __restrict__
__global__ __launch_bounds__(256) void k_act_pair(const float *a, const float *__restrict__ g1,
                                                  const int32_t *__restrict__ idx1, int64_t ld1,
                                                  const float *__restrict__ g2, const int32_t *__restrict__ idx2,
                                                  int64_t ld2, float *h, float *__restrict__ dact,
                                                  int64_t rows, int F4, int act) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows * F4) return;
  const int64_t r = i / F4;
  const int c = (int)(i - r * F4) * 4;
  float4_t v = *reinterpret_cast<const float4_t *>(a + r * (int64_t)(F4 * 4) + c);
  if (g1) v += *reinterpret_cast<const float4_t *>(g1 + (idx1 ? (int64_t)idx1[r] : r) * ld1 + c);
  if (g2) v += *reinterpret_cast<const float4_t *>(g2 + (idx2 ? (int64_t)idx2[r] : r) * ld2 + c);
  float4_t o, d;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float de;
    o[e] = act_fwd(v[e], act, de);
    d[e] = de;
  }
  *reinterpret_cast<float4_t *>(h + r * (int64_t)(F4 * 4) + c) = o;
  if (dact) *reinterpret_cast<float4_t *>(dact + r * (int64_t)(F4 * 4) + c) = d;
}

// k_act_pair over the edges of ONE TARGET per thread (edges are stored by target: rows rowptr[d] .. rowptr[d + 1]) with the
// per-target sum of h in stored order, hsum[d] = sum_r h[r] -- the backward's "hidden layer summed over a target's edges"
// without a second pass over [E, F]; g2 is the TARGET's row (read once per node).  Same values as k_act_pair followed by
// K2 over the edge-position CSR, bit for bit (same order of additions).  Four rows in flight per thread.
__global__ __launch_bounds__(256) void k_act_pair_seg(const float *a, const float *__restrict__ g1,
                                                      const int32_t *__restrict__ idx1, int64_t ld1,
                                                      const float *__restrict__ g2, int64_t ld2,
                                                      const int32_t *__restrict__ rowptr, float *h, float *__restrict__ dact,
                                                      float *__restrict__ hsum, int64_t n_dst, int F4, int act) {
  constexpr int U = 4;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n_dst * F4) return;
  const int64_t d_ = i / F4;
  const int c = (int)(i - d_ * F4) * 4;
  const int64_t F = (int64_t)F4 * 4;
  float4_t base = {0.f, 0.f, 0.f, 0.f};
  if (g2) base = *reinterpret_cast<const float4_t *>(g2 + d_ * ld2 + c);
  const int32_t s0 = rowptr[d_], s1 = rowptr[d_ + 1];
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int32_t s = s0; s < s1; s += U) {
    float4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = s + u < s1 ? s + u : s1 - 1;
      v[u] = *reinterpret_cast<const float4_t *>(a + r * F + c);
      if (g1) v[u] += *reinterpret_cast<const float4_t *>(g1 + (idx1 ? (int64_t)idx1[r] : r) * ld1 + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s + u < s1) {
        const int64_t r = s + u;
        float4_t pre = v[u];
        if (g2) pre += base;
        float4_t o, dd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float de;
          o[e] = act_fwd(pre[e], act, de);
          dd[e] = de;
        }
        *reinterpret_cast<float4_t *>(h + r * F + c) = o;
        if (dact) *reinterpret_cast<float4_t *>(dact + r * F + c) = dd;
        acc = acc + o;
      }
    }
  }
  *reinterpret_cast<float4_t *>(hsum + d_ * F + c) = acc;
}

// out[r] = (a ? a[r] : 0) + t[idx[r]] * (scale ? scale[idx[r]] : 1); out may be a (in place, as above)
__global__ __launch_bounds__(256) void k_gather_add(const float *a, const float *__restrict__ t,
                                                    const int32_t *__restrict__ idx,
                                                    const float *__restrict__ scale, float *out,
                                                    int64_t rows, int F4) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= rows * F4) return;
  const int64_t r = i / F4;
  const int c = (int)(i - r * F4) * 4;
  const int64_t j = idx[r];
  float4_t v = *reinterpret_cast<const float4_t *>(t + j * (int64_t)(F4 * 4) + c);
  if (scale) {
    const float s = scale[j];
    v = v * float4_t{s, s, s, s};
  }
  if (a) v += *reinterpret_cast<const float4_t *>(a + r * (int64_t)(F4 * 4) + c);
  *reinterpret_cast<float4_t *>(out + r * (int64_t)(F4 * 4) + c) = v;
}

__global__ __launch_bounds__(256) void k_ew(int op, const float *a, const float *b, float *out, int64_t n4) {   // out may be a or b
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4_t x = reinterpret_cast<const float4_t *>(a)[i], y = reinterpret_cast<const float4_t *>(b)[i];
  reinterpret_cast<float4_t *>(out)[i] = op == GWEN_EW_MUL ? x * y : x + y;
}

inline bool grid_ok(int64_t threads) { return (threads + 255) / 256 < (int64_t(1) << 31); }

}  // namespace

extern "C" int gwen_act_pair_f32(const float *a, const float *g1, const int32_t *idx1, int64_t ld1,
                                 const float *g2, const int32_t *idx2, int64_t ld2, float *h, float *dact,
                                 int64_t rows, int64_t F, int act, gwen_stream_t stream) {
  if (rows < 0 || F <= 0 || F % 4 || act < 0 || act > GWEN_ACT_SILU) return GWEN_EINVAL;
  if (rows == 0) return GWEN_OK;
  if (!a || !h || (idx1 && !g1) || (idx2 && !g2) || (g1 && (ld1 < F || ld1 % 4)) || (g2 && (ld2 < F || ld2 % 4)))
    return GWEN_EINVAL;
  const void *al[] = {a, g1, g2, h, dact};
  for (const void *p : al)
    if (p && !gwen_aligned(p, 16)) return GWEN_EINVAL;
  if (!grid_ok(rows * (F / 4))) return GWEN_ERANGE;
  const int64_t n = rows * (F / 4);
  k_act_pair<<<(unsigned)((n + 255) / 256), 256, 0, gwen_stream(stream)>>>(a, g1, idx1, ld1, g2, idx2, ld2, h, dact,
                                                                           rows, (int)(F / 4), act);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int gwen_act_pair_seg_f32(const float *a, const float *g1, const int32_t *idx1, int64_t ld1, const float *g2,
                                     int64_t ld2, const int32_t *rowptr, float *h, float *dact, float *hsum,
                                     int64_t rows, int64_t n_dst, int64_t F, int act, gwen_stream_t stream) {
  if (rows < 0 || n_dst < 0 || F <= 0 || F % 4 || act < 0 || act > GWEN_ACT_SILU || rows >= (int64_t(1) << 31))
    return GWEN_EINVAL;
  if (n_dst == 0) return GWEN_OK;
  if (!rowptr || !hsum || (rows > 0 && (!a || !h)) || (idx1 && !g1) || (g1 && (ld1 < F || ld1 % 4)) ||
      (g2 && (ld2 < F || ld2 % 4)))
    return GWEN_EINVAL;
  const void *al[] = {a, g1, g2, h, dact, hsum};
  for (const void *p : al)
    if (p && !gwen_aligned(p, 16)) return GWEN_EINVAL;
  if (!grid_ok(n_dst * (F / 4))) return GWEN_ERANGE;
  const int64_t n = n_dst * (F / 4);
  k_act_pair_seg<<<(unsigned)((n + 255) / 256), 256, 0, gwen_stream(stream)>>>(a, g1, idx1, ld1, g2, ld2, rowptr, h, dact,
                                                                               hsum, n_dst, (int)(F / 4), act);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int gwen_gather_add_f32(const float *a, const float *t, const int32_t *idx, const float *scale,
                                   float *out, int64_t rows, int64_t F, gwen_stream_t stream) {
  if (rows < 0 || F <= 0 || F % 4) return GWEN_EINVAL;
  if (rows == 0) return GWEN_OK;
  if (!t || !idx || !out) return GWEN_EINVAL;
  if ((a && !gwen_aligned(a, 16)) || !gwen_aligned(t, 16) || !gwen_aligned(out, 16)) return GWEN_EINVAL;
  if (!grid_ok(rows * (F / 4))) return GWEN_ERANGE;
  const int64_t n = rows * (F / 4);
  k_gather_add<<<(unsigned)((n + 255) / 256), 256, 0, gwen_stream(stream)>>>(a, t, idx, scale, out, rows,
                                                                             (int)(F / 4));
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}

extern "C" int gwen_ew_f32(int op, const float *a, const float *b, float *out, int64_t n, gwen_stream_t stream) {
  if (n < 0 || n % 4 || (op != GWEN_EW_MUL && op != GWEN_EW_ADD)) return GWEN_EINVAL;
  if (n == 0) return GWEN_OK;
  if (!a || !b || !out || !gwen_aligned(a, 16) || !gwen_aligned(b, 16) || !gwen_aligned(out, 16)) return GWEN_EINVAL;
  if (!grid_ok(n / 4)) return GWEN_ERANGE;
  k_ew<<<(unsigned)((n / 4 + 255) / 256), 256, 0, gwen_stream(stream)>>>(op, a, b, out, n / 4);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
