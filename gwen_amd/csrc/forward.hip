// Whole-stack forward: every kernel of GNNModel.forward issued back to back from one host call.
//
// Replaces the host side of /root/reference/src/gwen/models_gnn.py:292-303 -> :241-258 ->
// :135-157 / :189-212 (six GCNConv.__call__ + five torch.relu, ~15 eager launches per layer with
// the normalisation recomputed each time): here one call enqueues 6..12 launches on one stream
// with no allocation and no synchronisation, so it is hipGraph-capturable and the host stays ahead
// of the device.  Optional hipEvents bracket every launch for in-situ kernel timing.
#include "common.h"

namespace {

struct Plan {
  int64_t ping, pong, tmp, lin;   // element offsets into scratch (lin: split-K workspace of K3)
  int64_t lin_floats, total;
};

inline int resolve_order(const gwen_layer_desc &L) {
  if (L.order == GWEN_ORDER_AUTO) {
    if (gwen_gcn_layer_supported(L.fin, L.fout)) return GWEN_ORDER_FUSED;
    return L.fin < L.fout ? GWEN_ORDER_AGGREGATE_FIRST : GWEN_ORDER_TRANSFORM_FIRST;
  }
  return L.order;
}

inline int contract_of(const gwen_layer_desc &L) { return gwen_contract_of(L); }           // (common.h)
inline int dense_contract(int c) { return gwen_dense_contract(c); }
inline int wide_contract(int c, int64_t fi, int64_t fo) {
  return c == GWEN_CONTRACT_F16X3 && !gwen_gcn_wide_contract_supported(fi, fo, c) ? GWEN_CONTRACT_BF16X6 : c;
}

inline int64_t round4(int64_t v) { return (v + 3) / 4 * 4; }

int make_plan(int64_t N, int64_t members, const gwen_layer_desc *layers, int32_t n, Plan *P) {
  int64_t out_w = 0, tmp_w = 0;
  for (int32_t i = 0; i < n; ++i) {
    const gwen_layer_desc &L = layers[i];
    if (L.fin < 0 || L.fout < 0) return GWEN_EINVAL;
    if (L.contract != GWEN_CONTRACT_BF16X3 && L.contract != GWEN_CONTRACT_BF16X6 && L.contract != GWEN_CONTRACT_F16X3)
      return GWEN_EINVAL;
    if (i > 0 && layers[i - 1].fout != L.fin) return GWEN_EINVAL;
    const int o = resolve_order(L);
    if ((o == GWEN_ORDER_FUSED || o == GWEN_ORDER_FUSED_EXACT) &&
        !gwen_gcn_layer_supported(L.fin, L.fout))
      return GWEN_EINVAL;
    if (L.fout > out_w) out_w = L.fout;      // activations and pre-projected inputs of later layers
    if (o == GWEN_ORDER_TRANSFORM_FIRST && L.fout > tmp_w) tmp_w = L.fout;
    if (o == GWEN_ORDER_AGGREGATE_FIRST && L.fin > tmp_w) tmp_w = L.fin;
  }
  const int64_t rows = members * N;
  int64_t lin_w = 0;
  for (int32_t i = 0; i < n; ++i) {
    const int64_t w = gwen_gcn_linear_workspace_floats(rows, layers[i].fin, layers[i].fout);
    if (w > lin_w) lin_w = w;
    const int64_t w7 = gwen_gcn_small_workspace_floats(N, members, layers[i].fin, layers[i].fout);
    if (w7 > lin_w) lin_w = w7;               // K7's split-K partials share the region
  }
  P->ping = 0;
  P->pong = round4(rows * out_w);
  P->tmp = P->pong + round4(rows * out_w);
  P->lin = P->tmp + round4(rows * tmp_w);
  P->lin_floats = lin_w;
  P->total = P->lin + round4(lin_w) + 4;
  return GWEN_OK;
}

}  // namespace

extern "C" int64_t gwen_gnn_forward_scratch_floats(int64_t N, int64_t members,
                                                   const gwen_layer_desc *layers, int32_t n_layers) {
  Plan P;
  if (N < 0 || members < 0 || n_layers < 0 || (n_layers > 0 && !layers)) return GWEN_EINVAL;
  if (make_plan(N, members, layers, n_layers, &P) != GWEN_OK) return GWEN_EINVAL;
  return P.total;
}

extern "C" int gwen_gnn_forward_f32(const gwen_graph *graph, const gwen_layer_desc *layers,
                                    int32_t n_layers, const float *x, float *out, float *scratch,
                                    int64_t scratch_floats, int64_t members, gwen_stream_t stream,
                                    void **events, gwen_launch_info *info, int32_t max_launches,
                                    int32_t *n_launches, float *const *acts) {
  if (!graph) return GWEN_EINVAL;
  const int64_t N = graph->N;
  const int32_t *rowptr = graph->rowptr, *col = graph->col, *g_rowptr = graph->g_rowptr,
                *g_col = graph->g_col;
  const float *val = graph->val, *g_val = graph->g_val, *dense = graph->dense;
  const bool have_tiles = graph->t_rows && graph->t_lid && graph->t_val;
  if (N < 0 || members < 0 || n_layers <= 0 || !layers) return GWEN_EINVAL;
  Plan P;
  int rc = make_plan(N, members, layers, n_layers, &P);
  if (rc != GWEN_OK) return rc;
  if (scratch_floats < P.total || (P.total > 4 && !scratch)) return GWEN_ENOSPACE;
  if (scratch && !gwen_aligned(scratch, 16)) return GWEN_EINVAL;
  if ((events || info) && max_launches < 2 * n_layers) return GWEN_EINVAL;
  if (acts)
    for (int32_t i = 0; i < n_layers; ++i)
      if (!acts[i]) return GWEN_EINVAL;
  hipStream_t st = gwen_stream(stream);
  const int64_t rows = members * N;
  float *buf[2] = {scratch + P.ping, scratch + P.pong};
  float *tmp = scratch + P.tmp;
  float *lin_ws = scratch + P.lin;
  int32_t nl = 0;

  auto before = [&](int kind, int layer, int fin, int fout) -> int {
    if (info) info[nl] = gwen_launch_info{kind, layer, fin, fout};
    if (events) GWEN_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(events[2 * nl]), st));
    return GWEN_OK;
  };
  auto after = [&]() -> int {
    if (events) GWEN_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(events[2 * nl + 1]), st));
    ++nl;
    return GWEN_OK;
  };
#define GWEN_TRY(expr) do { int _r = (expr); if (_r != GWEN_OK) return _r; } while (0)

  // AUTO layer whose widths K4 takes and that shrinks: worth gathering at fout, i.e. transform-first
  auto shrinking_auto = [&](int32_t i, int contract) {   // never while training: every layer's output must exist
    return !acts && i < n_layers && layers[i].order == GWEN_ORDER_AUTO && layers[i].fout < layers[i].fin &&
           dense_contract(contract_of(layers[i])) == contract;               // one contraction per chained kernel
  };
  const bool have_grouped = g_col && g_val;          // g_rowptr NULL = uniform layout

  const float *cur = x;
  // small graph, wide features (the reference's own shape): every layer is one K7 launch
  bool small = dense != nullptr;
  for (int32_t i = 0; i < n_layers && small; ++i)
    small = layers[i].order == GWEN_ORDER_AUTO &&
            gwen_gcn_small_supported(N, layers[i].fin, layers[i].fout, dense_contract(contract_of(layers[i])));
  if (small) {
    for (int32_t i = 0; i < n_layers; ++i) {
      const gwen_layer_desc &L = layers[i];
      float *dst = acts ? acts[i] : (i + 1 == n_layers ? out : buf[i & 1]);
      GWEN_TRY(before(GWEN_KIND_SMALL, i, L.fin, L.fout));
      GWEN_TRY(gwen_gcn_small_layer_f32(dense, cur, L.W, L.packed, L.bias, dst, N, L.fin, L.fout, members,
                                        N * L.fin, N * L.fout, L.relu, lin_ws, P.lin_floats,
                                        dense_contract(contract_of(L)), stream));
      GWEN_TRY(after());
      cur = dst;
    }
    if (n_launches) *n_launches = nl;
    return GWEN_OK;
  }
  bool projected = false;          // cur holds h_i = a_i W_i^T (layer i's bias/ReLU still pending)
  int32_t nbuf = 0;
  for (int32_t i = 0; i < n_layers; ++i) {
    const gwen_layer_desc &L = layers[i];
    const bool last = i + 1 == n_layers;
    const int64_t fi = L.fin, fo = L.fout;
    if (projected) {
      // layer i: propagate at width fo with its bias/ReLU; chain layer i+1's projection if it shrinks
      const int cn = last ? 0 : dense_contract(contract_of(layers[i + 1]));
      const bool chain = !last && have_grouped && shrinking_auto(i + 1, cn) &&
                         gwen_gcn_chain_supported(fo, layers[i + 1].fout, 0, 1, cn);
      float *dst = acts ? acts[i] : ((last) ? out : buf[nbuf++ & 1]);
      if (chain) {
        GWEN_TRY(before(GWEN_KIND_CHAIN, i, fo, layers[i + 1].fout));
        GWEN_TRY(gwen_gcn_chain_f32(g_rowptr, g_col, g_val, cur, layers[i + 1].W, nullptr, L.bias,
                                    dst, N, fo, layers[i + 1].fout, 0, 1, L.relu, members, N * fo,
                                    N * layers[i + 1].fout, cn, stream));
        GWEN_TRY(after());
        projected = true;
      } else if (have_grouped && gwen_gcn_chain_supported(fo, 0, 0, 1, GWEN_CONTRACT_BF16X3)) {
        GWEN_TRY(before(GWEN_KIND_CHAIN, i, fo, fo));
        GWEN_TRY(gwen_gcn_chain_f32(g_rowptr, g_col, g_val, cur, nullptr, nullptr, L.bias, dst, N,
                                    fo, 0, 0, 1, L.relu, members, N * fo, N * fo, GWEN_CONTRACT_BF16X3,
                                    stream));           // no contraction in this form
        GWEN_TRY(after());
        projected = false;
      } else {
        GWEN_TRY(before(GWEN_KIND_PROPAGATE, i, fo, fo));
        GWEN_TRY(gwen_gcn_propagate_f32(rowptr, col, val, cur, L.bias, dst, N, fo, fo, fo, members,
                                        N * fo, N * fo, L.relu, stream));
        GWEN_TRY(after());
        projected = false;
      }
      cur = dst;
      continue;
    }
    const int o = resolve_order(L);
    const int cw = wide_contract(contract_of(L), fi, fo);     // K8's
    const int cc = dense_contract(contract_of(L));             // every other kernel's
    if (o == GWEN_ORDER_FUSED || o == GWEN_ORDER_FUSED_EXACT) {
      const bool chain = have_grouped && L.order == GWEN_ORDER_AUTO && !last && shrinking_auto(i + 1, cc) &&
                         gwen_gcn_chain_supported(fi, fo, layers[i + 1].fout, 0, cc);
      if (!chain && L.order == GWEN_ORDER_AUTO && have_tiles &&
          gwen_gcn_wide_preferred(N, members, fi, fo) && gwen_gcn_wide_contract_supported(fi, fo, cw) &&
          (cw != GWEN_CONTRACT_BF16X6 || graph->union_max <= 128)) {   // K8: the layer tile-staged
        float *dst = acts ? acts[i] : (last ? out : buf[nbuf++ & 1]);
        GWEN_TRY(before(GWEN_KIND_WIDE, i, fi, fo));
        GWEN_TRY(gwen_gcn_wide_layer_f32(graph->t_rows, graph->t_lid, graph->t_val, cur, L.W, L.bias, dst,
                                         N, N, fi, fo, fo, members, N * fi, N * fo, L.relu,
                                         graph->union_max, cw, stream));
        GWEN_TRY(after());
        cur = dst;
        continue;
      }
      if (!have_grouped) return GWEN_EINVAL;
      // a chained kernel stores the NEXT layer's input, never the stack's output
      float *dst = acts ? acts[i] : ((last && !chain) ? out : buf[nbuf++ & 1]);
      if (chain) {
        GWEN_TRY(before(GWEN_KIND_CHAIN, i, fi, layers[i + 1].fout));
        GWEN_TRY(gwen_gcn_chain_f32(g_rowptr, g_col, g_val, cur, L.W, layers[i + 1].W, L.bias, dst,
                                    N, fi, fo, layers[i + 1].fout, 0, L.relu, members, N * fi,
                                    N * layers[i + 1].fout, cc, stream));
        GWEN_TRY(after());
        projected = true;
      } else {
        GWEN_TRY(before(GWEN_KIND_LAYER, i, fi, fo));
        GWEN_TRY(gwen_gcn_layer_f32(g_rowptr, g_col, g_val, cur, L.W, L.bias, dst, N, fi, fo, fi, fo,
                                    members, N * fi, N * fo, L.relu,
                                    o == GWEN_ORDER_FUSED_EXACT ? GWEN_CONTRACT_F32 : cc, stream));
        GWEN_TRY(after());
      }
      cur = dst;
    } else if (o == GWEN_ORDER_TRANSFORM_FIRST) {
      float *dst = acts ? acts[i] : (last ? out : buf[nbuf++ & 1]);
      GWEN_TRY(before(GWEN_KIND_LINEAR, i, fi, fo));
      GWEN_TRY(gwen_gcn_linear_f32(cur, L.W, nullptr, tmp, rows, fi, fo, fi, fo, 0, cc, lin_ws,
                                   P.lin_floats, stream));
      GWEN_TRY(after());
      GWEN_TRY(before(GWEN_KIND_PROPAGATE, i, fo, fo));
      GWEN_TRY(gwen_gcn_propagate_f32(rowptr, col, val, tmp, L.bias, dst, N, fo, fo, fo, members,
                                      N * fo, N * fo, L.relu, stream));
      GWEN_TRY(after());
      cur = dst;
    } else if (o == GWEN_ORDER_AGGREGATE_FIRST) {
      float *dst = acts ? acts[i] : (last ? out : buf[nbuf++ & 1]);
      GWEN_TRY(before(GWEN_KIND_PROPAGATE, i, fi, fi));
      GWEN_TRY(gwen_gcn_propagate_f32(rowptr, col, val, cur, nullptr, tmp, N, fi, fi, fi, members,
                                      N * fi, N * fi, 0, stream));
      GWEN_TRY(after());
      GWEN_TRY(before(GWEN_KIND_LINEAR, i, fi, fo));
      GWEN_TRY(gwen_gcn_linear_f32(tmp, L.W, L.bias, dst, rows, fi, fo, fi, fo, L.relu, cc,
                                   lin_ws, P.lin_floats, stream));
      GWEN_TRY(after());
      cur = dst;
    } else {
      return GWEN_EINVAL;
    }
  }
#undef GWEN_TRY
  if (n_launches) *n_launches = nl;
  return GWEN_OK;
}

extern "C" int gwen_event_create(void **event) {
  if (!event) return GWEN_EINVAL;
  hipEvent_t e;
  GWEN_HIP_CHECK(hipEventCreate(&e));
  *event = e;
  return GWEN_OK;
}
extern "C" int gwen_event_destroy(void *event) {
  if (!event) return GWEN_OK;
  GWEN_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return GWEN_OK;
}
extern "C" int gwen_event_record(void *event, gwen_stream_t stream) {
  if (!event) return GWEN_EINVAL;
  GWEN_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(event), gwen_stream(stream)));
  return GWEN_OK;
}
extern "C" int gwen_event_synchronize(void *event) {
  if (!event) return GWEN_EINVAL;
  GWEN_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return GWEN_OK;
}
extern "C" int gwen_event_elapsed_ms(void *start, void *stop, float *ms) {
  if (!start || !stop || !ms) return GWEN_EINVAL;
  GWEN_HIP_CHECK(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return GWEN_OK;
}
