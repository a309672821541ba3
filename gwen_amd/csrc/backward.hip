// Whole-stack backward: every kernel of GNNModel's backward issued back to back from one host call.
//
// The reference trains through the six GCNConv layers with loss.backward()
// (/root/reference/src/gwen/models_gnn.py:372): per layer torch-geometric's autograd runs the transposed
// scatter (index_select + scatter-add with float atomics), two GEMMs, the bias reduction and the ReLU mask as
// ~10 eager launches.  Here a layer's backward is
//     [gh = A~^T g ,  gx = (gh W) masked by the ReLU below]   ONE launch (K4's kernel on the transposed graph)
//     grad_W = gh^T x                                           one reduction launch (+ its fixed-order finish)
//     grad_b = column sums of g                                 one reduction launch
// with g arriving already masked from the launch above it, all of it enqueued by this one call: no
// per-layer Python, no allocation, no synchronisation (hipGraph-capturable).
#include "common.h"

namespace {

inline int64_t round4(int64_t v) { return (v + 3) / 4 * 4; }

struct BPlan {
  int64_t g0, g1, gh, wt, part, total;
  int64_t wt_off[GWEN_MAX_REDUCE_TASKS], pw_off[GWEN_MAX_REDUCE_TASKS], pb_off[GWEN_MAX_REDUCE_TASKS];
};

// scratch: two gradient buffers, gh, every layer's W^T, every layer's grad_W / grad_b partial sums (the
// reductions finish together at the end, so each needs its own region)
int make_bplan(int64_t N, int64_t members, const gwen_layer_desc *layers, int32_t n, BPlan *P) {
  const int64_t rows = members * N;
  if (2 * n > GWEN_MAX_REDUCE_TASKS) return GWEN_EINVAL;
  const int64_t nc = gwen_gcn_grad_chunks(rows);
  int64_t fmax = 0, wsum = 0, psum = 0;
  for (int32_t i = 0; i < n; ++i) {
    const gwen_layer_desc &L = layers[i];
    if (L.fin <= 0 || L.fout <= 0) return GWEN_EINVAL;
    if (i > 0 && layers[i - 1].fout != L.fin) return GWEN_EINVAL;
    if (L.fin > fmax) fmax = L.fin;
    if (L.fout > fmax) fmax = L.fout;
    P->wt_off[i] = wsum;
    wsum += round4((int64_t)L.fin * L.fout);
    P->pw_off[i] = psum;
    psum += round4(nc * L.fin * L.fout);
    P->pb_off[i] = psum;                           // (its column sums may come per chunk of the backward kernel above it)
    const int64_t nb = gwen_gcn_layer_bwd_bias_rows(N, members);
    psum += round4((nb > nc ? nb : nc) * L.fout);
  }
  P->g0 = 0;
  P->g1 = round4(rows * fmax);
  P->gh = P->g1 + round4(rows * fmax);
  P->wt = P->gh + round4(rows * fmax);
  P->part = P->wt + wsum;
  P->total = P->part + psum + 4;
  return GWEN_OK;
}

}  // namespace

extern "C" int64_t gwen_gnn_backward_scratch_floats(int64_t N, int64_t members,
                                                    const gwen_layer_desc *layers, int32_t n_layers) {
  BPlan P;
  if (N < 0 || members < 0 || n_layers <= 0 || !layers) return GWEN_EINVAL;
  if (make_bplan(N, members, layers, n_layers, &P) != GWEN_OK) return GWEN_EINVAL;
  return P.total;
}

extern "C" int gwen_gnn_backward_f32(const gwen_graph *graph_t, const gwen_layer_desc *layers,
                                     int32_t n_layers, const float *x, const float *const *acts,
                                     const float *grad_out, float *grad_x, float *const *grad_W,
                                     float *const *grad_b, float *scratch, int64_t scratch_floats,
                                     int64_t members, gwen_stream_t stream) {
  if (!graph_t || !layers || n_layers <= 0 || !x || !acts || !grad_out) return GWEN_EINVAL;
  const int64_t N = graph_t->N;
  if (N < 0 || members < 0) return GWEN_EINVAL;
  BPlan P;
  int rc = make_bplan(N, members, layers, n_layers, &P);
  if (rc != GWEN_OK) return rc;
  if (!scratch || scratch_floats < P.total) return GWEN_ENOSPACE;
  if (!gwen_aligned(scratch, 16)) return GWEN_EINVAL;
  if (N == 0 || members == 0) return GWEN_OK;
  const int64_t rows = members * N;
  float *gbuf[2] = {scratch + P.g0, scratch + P.g1};
  float *gh = scratch + P.gh;
  const int64_t nc = gwen_gcn_grad_chunks(rows);
  const bool have_grouped = graph_t->g_col && graph_t->g_val;
#define GWEN_TRY(expr) do { int _r = (expr); if (_r != GWEN_OK) return _r; } while (0)

  {   // every W^T the gx contractions need, one launch
    const float *w[GWEN_MAX_REDUCE_TASKS];
    float *wt[GWEN_MAX_REDUCE_TASKS];
    int32_t r_[GWEN_MAX_REDUCE_TASKS], c_[GWEN_MAX_REDUCE_TASKS], k = 0;
    for (int32_t l = 0; l < n_layers; ++l) {
      if (l == 0 && !grad_x) continue;
      if (!layers[l].W) return GWEN_EINVAL;
      w[k] = layers[l].W; wt[k] = scratch + P.wt + P.wt_off[l];
      r_[k] = layers[l].fout; c_[k] = layers[l].fin; ++k;
    }
    GWEN_TRY(gwen_transpose_batched(w, wt, r_, c_, k, stream));
  }
  gwen_reduce_task tasks[GWEN_MAX_REDUCE_TASKS];
  int32_t n_tasks = 0;

  const float *g = grad_out;
  bool bias_done[GWEN_MAX_REDUCE_TASKS] = {};
  int nb = 0;
  {
    const gwen_layer_desc &T = layers[n_layers - 1];
    if (T.relu) {           // the stack's last layer has an activation: mask the incoming gradient once
      float *dst = gbuf[nb++ & 1];
      GWEN_TRY(gwen_relu_backward_f32(acts[n_layers - 1], grad_out, dst, rows * T.fout, stream));
      g = dst;
    }
  }
  for (int32_t l = n_layers - 1; l >= 0; --l) {
    const gwen_layer_desc &L = layers[l];
    const int64_t fi = L.fin, fo = L.fout;
    const float *xin = l == 0 ? x : acts[l - 1];
    const bool need_gx = l > 0 || grad_x != nullptr;
    const float *mask = (l > 0 && layers[l - 1].relu) ? acts[l - 1] : nullptr;
    float *gx = l == 0 ? grad_x : gbuf[nb++ & 1];
    const float *wt = scratch + P.wt + P.wt_off[l];
    if (need_gx && gx == g) return GWEN_EINVAL;
    if (grad_b && grad_b[l] && !bias_done[l]) {
      float *pb = scratch + P.part + P.pb_off[l];
      GWEN_TRY(gwen_gcn_grad_bias_partial_f32(g, pb, rows, fo, fo, stream));
      tasks[n_tasks++] = gwen_reduce_task{pb, grad_b[l], fo, nc};
    }
    const bool fused = need_gx && have_grouped && gwen_gcn_layer_supported(fo, fi) &&
                       (L.order == GWEN_ORDER_AUTO || L.order == GWEN_ORDER_FUSED);
    // the layer's own precision: bf16x3 layers contract gx on bf16x3, fp32-class layers (bf16x6,
    // f16x3) on bf16x6, explicit / fp32 orders on the fp32-input MFMA (K3 below)
    const int cc = gwen_dense_contract(gwen_contract_of(L));
    if (fused) {
      // gx is the incoming gradient of layer l - 1: the launch leaves stage 1 of that layer's grad_b as well
      float *pbelow = (l > 0 && grad_b && grad_b[l - 1]) ? scratch + P.part + P.pb_off[l - 1] : nullptr;
      int64_t bchunks = 0;
      GWEN_TRY(gwen_gcn_layer_bwd_bias_f32(graph_t->g_rowptr, graph_t->g_col, graph_t->g_val, g, wt, mask, gh,
                                           gx, N, fo, fi, members, cc, pbelow, pbelow ? &bchunks : nullptr, stream));
      if (pbelow && bchunks > 0) {
        tasks[n_tasks++] = gwen_reduce_task{pbelow, grad_b[l - 1], fi, bchunks};
        bias_done[l - 1] = true;
      }
    } else {
      GWEN_TRY(gwen_gcn_propagate_f32(graph_t->rowptr, graph_t->col, graph_t->val, g, nullptr, gh, N, fo,
                                      fo, fo, members, N * fo, N * fo, 0, stream));
      if (need_gx) {
        GWEN_TRY(gwen_gcn_linear_f32(gh, wt, nullptr, gx, rows, fo, fi, fo, fi, 0, cc, nullptr, 0, stream));
        if (mask) GWEN_TRY(gwen_relu_backward_f32(mask, gx, gx, rows * fi, stream));
      }
    }
    if (grad_W && grad_W[l]) {
      float *pw = scratch + P.part + P.pw_off[l];
      GWEN_TRY(gwen_gcn_grad_weight_partial_f32(gh, xin, pw, rows, fi, fo, fo, fi, cc, stream));
      tasks[n_tasks++] = gwen_reduce_task{pw, grad_W[l], fi * fo, gwen_gcn_grad_weight_chunks(rows, fi, fo, cc)};
    }
    g = gx;
  }
  GWEN_TRY(gwen_reduce_chunks_batched(tasks, n_tasks, stream));      // every grad_W / grad_b finishes here
#undef GWEN_TRY
  return GWEN_OK;
}
