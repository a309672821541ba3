/*
 * Plain-C restatement of the GCNConv forward  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Second, independent oracle beside oracle/gcn_oracle.py (the two are cross-checked in
 * tests/test_oracle_kat.py).  PARITY UNPINNED: the reference's arithmetic lives in the un-vendored
 * dependency torch-geometric==2.3.1 (/root/reference/requirements/environment.yml:552) and the
 * reference's tests hold no golden vector for it; this restates the published rule
 *     X' = D^-1/2 (A + I) D^-1/2 X W^T + b
 * with the options the reference's call sites fix
 * (/root/reference/src/gwen/models_gnn.py:118-130,172-184 constructors; :147-149,204-206 calls).
 *
 * Summation order is the order the reference's CPU path uses: kept (non-loop) edges in their
 * original order, then the N self-loops; per destination the terms are added sequentially in that
 * order, each product rounded to fp32 before the add (message = w * x_j, then scatter-add).
 * Build with -ffp-contract=off so that order/rounding is what the source says.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* gcn_norm + add_remaining_self_loops.  out_* must hold E + N entries.  Returns E' or -1 on a
 * node index outside [0,N). */
int64_t gcn_ref_norm(const int64_t *edge_index, const float *edge_weight, int64_t N, int64_t E,
                     int add_self_loops, float fill, int64_t *out_src, int64_t *out_dst,
                     float *out_w) {
  const int64_t *src = edge_index, *dst = edge_index + E;
  int64_t m = 0;
  float *loop_w = NULL;
  if (add_self_loops) {
    loop_w = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < N; ++i) loop_w[i] = fill;
  }
  for (int64_t e = 0; e < E; ++e) {
    if (src[e] < 0 || src[e] >= N || dst[e] < 0 || dst[e] >= N) { free(loop_w); return -1; }
    float w = edge_weight ? edge_weight[e] : 1.0f;
    if (add_self_loops && src[e] == dst[e]) {
      if (edge_weight) loop_w[src[e]] = w;   /* existing loop keeps its weight, last wins */
      continue;                              /* without weights: replaced by the unit loop */
    }
    out_src[m] = src[e]; out_dst[m] = dst[e]; out_w[m] = w; ++m;
  }
  if (add_self_loops) {
    for (int64_t i = 0; i < N; ++i) { out_src[m] = i; out_dst[m] = i; out_w[m] = loop_w[i]; ++m; }
    free(loop_w);
  }
  float *deg = (float *)calloc((size_t)(N > 0 ? N : 1), sizeof(float));
  for (int64_t e = 0; e < m; ++e) deg[out_dst[e]] += out_w[e];            /* degree over TARGETS */
  for (int64_t i = 0; i < N; ++i) {
    float d = 1.0f / sqrtf(deg[i]);                                      /* deg.pow(-0.5) */
    if (isinf(d) && d > 0) d = 0.0f;                                     /* inf -> 0 */
    deg[i] = d;
  }
  for (int64_t e = 0; e < m; ++e) out_w[e] = deg[out_src[e]] * out_w[e] * deg[out_dst[e]];
  free(deg);
  return m;
}

/* h = x @ W^T ; W is [Fout, Fin] row-major (lin.weight).  Sequential k order, mul then add. */
void gcn_ref_linear_f32(const float *x, const float *W, float *h, int64_t N, int64_t Fin,
                        int64_t Fout) {
  for (int64_t i = 0; i < N; ++i)
    for (int64_t o = 0; o < Fout; ++o) {
      float acc = 0.0f;
      for (int64_t k = 0; k < Fin; ++k) acc = acc + x[i * Fin + k] * W[o * Fin + k];
      h[i * Fout + o] = acc;
    }
}

/* Same contraction as a k-ordered fmaf chain (what an fp32 MFMA accumulates bit for bit). */
void gcn_ref_linear_fma_f32(const float *x, const float *W, float *h, int64_t N, int64_t Fin,
                            int64_t Fout) {
  for (int64_t i = 0; i < N; ++i)
    for (int64_t o = 0; o < Fout; ++o) {
      float acc = 0.0f;
      for (int64_t k = 0; k < Fin; ++k) acc = fmaf(x[i * Fin + k], W[o * Fin + k], acc);
      h[i * Fout + o] = acc;
    }
}

/* out[i] = sum_{e: dst=i} w_e * h[src_e]  (+ bias, optional ReLU).  out is overwritten. */
void gcn_ref_propagate_f32(const int64_t *src, const int64_t *dst, const float *w, int64_t Ep,
                           const float *h, const float *bias, int relu, float *out, int64_t N,
                           int64_t F) {
  memset(out, 0, sizeof(float) * (size_t)(N * F));
  for (int64_t e = 0; e < Ep; ++e) {
    const float *hs = h + src[e] * F;
    float *o = out + dst[e] * F;
    float we = w[e];
    for (int64_t f = 0; f < F; ++f) {
      float msg = we * hs[f];
      o[f] = o[f] + msg;
    }
  }
  for (int64_t i = 0; i < N; ++i)
    for (int64_t f = 0; f < F; ++f) {
      float v = out[i * F + f];
      if (bias) v = v + bias[f];
      if (relu) v = v > 0.0f ? v : 0.0f;
      out[i * F + f] = v;
    }
}

/* One whole layer in fp32.  Returns 0, or -1 on a bad index. */
int gcn_ref_conv_f32(const float *x, const int64_t *edge_index, const float *edge_weight,
                     const float *W, const float *bias, int relu, int add_self_loops, float fill,
                     float *out, int64_t N, int64_t E, int64_t Fin, int64_t Fout) {
  int64_t cap = E + N + 1;
  int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
  int64_t *d = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
  float *w = (float *)malloc(sizeof(float) * (size_t)cap);
  float *h = (float *)malloc(sizeof(float) * (size_t)(N * Fout + 1));
  int64_t m = gcn_ref_norm(edge_index, edge_weight, N, E, add_self_loops, fill, s, d, w);
  int rc = 0;
  if (m < 0) rc = -1;
  else {
    gcn_ref_linear_f32(x, W, h, N, Fin, Fout);
    gcn_ref_propagate_f32(s, d, w, m, h, bias, relu, out, N, Fout);
  }
  free(s); free(d); free(w); free(h);
  return rc;
}

/* The same layer with every intermediate in fp64 (error budgeting: inputs are fp32 values). */
int gcn_ref_conv_f64(const float *x, const int64_t *edge_index, const float *edge_weight,
                     const float *W, const float *bias, int relu, int add_self_loops, float fill,
                     double *out, int64_t N, int64_t E, int64_t Fin, int64_t Fout) {
  const int64_t *src = edge_index, *dst = edge_index + E;
  double *deg = (double *)calloc((size_t)(N + 1), sizeof(double));
  double *loop_w = (double *)malloc(sizeof(double) * (size_t)(N + 1));
  double *h = (double *)malloc(sizeof(double) * (size_t)(N * Fout + 1));
  for (int64_t i = 0; i < N; ++i) loop_w[i] = add_self_loops ? (double)fill : 0.0;
  for (int64_t e = 0; e < E; ++e) {
    if (src[e] < 0 || src[e] >= N || dst[e] < 0 || dst[e] >= N) { free(deg); free(loop_w); free(h); return -1; }
    double w = edge_weight ? (double)edge_weight[e] : 1.0;
    if (add_self_loops && src[e] == dst[e]) { if (edge_weight) loop_w[src[e]] = w; continue; }
    deg[dst[e]] += w;
  }
  for (int64_t i = 0; i < N; ++i) {
    deg[i] += loop_w[i];
    double dd = 1.0 / sqrt(deg[i]);
    deg[i] = (isinf(dd) && dd > 0) ? 0.0 : dd;
  }
  for (int64_t i = 0; i < N; ++i)
    for (int64_t o = 0; o < Fout; ++o) {
      double acc = 0.0;
      for (int64_t k = 0; k < Fin; ++k) acc += (double)x[i * Fin + k] * (double)W[o * Fin + k];
      h[i * Fout + o] = acc;
    }
  for (int64_t i = 0; i < N * Fout; ++i) out[i] = 0.0;
  for (int64_t e = 0; e < E; ++e) {
    if (add_self_loops && src[e] == dst[e]) continue;
    double w = (edge_weight ? (double)edge_weight[e] : 1.0) * deg[src[e]] * deg[dst[e]];
    for (int64_t f = 0; f < Fout; ++f) out[dst[e] * Fout + f] += w * h[src[e] * Fout + f];
  }
  for (int64_t i = 0; i < N; ++i) {
    double w = loop_w[i] * deg[i] * deg[i];
    for (int64_t f = 0; f < Fout; ++f) {
      double v = out[i * Fout + f] + w * h[i * Fout + f];
      if (bias) v += (double)bias[f];
      if (relu && v < 0.0) v = 0.0;
      out[i * Fout + f] = v;
    }
  }
  free(deg); free(loop_w); free(h);
  return 0;
}
