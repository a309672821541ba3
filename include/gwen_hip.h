/*
 * gwen_hip.h -- C ABI of libgwen_hip.so: the MI355X (gfx950) implementation of GWEN's GCNConv
 * hot path.  Plain pointers and sizes only; no torch types.  Every pointer is a DEVICE pointer
 * unless the comment says "host".  All launchers are asynchronous on `stream` (a hipStream_t
 * passed as void*), allocate nothing, synchronise nothing and keep no global state, so they can be
 * captured into a hipGraph.
 *
 * The reference (MeteoSwiss/GWEN) has no C/FFI layer: its hot path is the Python call
 * `GCNConv.__call__(x, edge_index)` into torch-geometric 2.3.1
 * (/root/reference/src/gwen/models_gnn.py:19 import; :118-130,:172-184 constructors;
 * :147-149,:204-206 calls).  Each entry point below names the piece of that call it replaces.
 * The Python binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Return value: 0 on success; > 0 a hipError_t; < 0 one of the GWEN_E* codes.
 */
#ifndef GWEN_HIP_H
#define GWEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GWEN_OK 0
#define GWEN_EINVAL (-1)    /* bad size / null pointer / misaligned argument            */
#define GWEN_ERANGE (-2)    /* N, E or a byte count does not fit the int32 CSR          */
#define GWEN_ENOSPACE (-3)  /* workspace smaller than gwen_gcn_prep_workspace_bytes()   */

typedef void *gwen_stream_t; /* hipStream_t */

/* Library identification: "gwen_hip <semver> gfx950". */
const char *gwen_hip_version(void);
/* Text for a return code of any function below (hipGetErrorString for positive codes). */
const char *gwen_hip_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * K1  graph preparation == gcn_norm + add_remaining_self_loops, hoisted out of the layer
 * (the reference recomputes it inside each of the 6 GCNConv calls of every forward:
 *  /root/reference/src/gwen/models_gnn.py:147-149,:204-206; constructors pass cached=False).
 *
 * edge_index : int64 [2, E] row-major; row 0 = source j, row 1 = target i.
 * edge_weight: fp32 [E] or NULL (unit weights; the reference never passes weights).
 * Result: CSR by TARGET node with the N self-loops completed,
 *   rowptr int32 [N+1], col int32 [cap], val fp32 [cap], eid int32 [cap], cap = E + N,
 *   where row i holds its kept (non-loop) in-edges in ORIGINAL edge order followed by the
 *   self-loop -- the order in which the reference's CPU scatter-add visits them --
 *   col = source node, val = d^-1/2[src] * w * d^-1/2[dst] (or the raw w if !normalize),
 *   eid = index of the edge in edge_index, -1 for a completed self-loop.
 *   dis fp32 [N] = deg^-1/2 (inf -> 0).  rowptr[N] = number of stored entries E'.
 * status: int32 [2]; status[0] is OR-ed with 1 if any node index is outside [0, N)
 *   (such edges are dropped); status[1] = E'.  The caller zeroes nothing: prep does.
 * ------------------------------------------------------------------------------------------- */
int gwen_gcn_prep_workspace_bytes(int64_t N, int64_t E, size_t *bytes /* host */);
int gwen_gcn_prep(const int64_t *edge_index, const float *edge_weight, int64_t N, int64_t E,
                  int add_self_loops, float fill_value, int normalize, int32_t *rowptr,
                  int32_t *col, float *val, int32_t *eid, float *dis, int32_t *status,
                  void *workspace, size_t workspace_bytes, gwen_stream_t stream);

/* Rectangular (bipartite) graph for the grid->mesh / mesh->grid maps of SURVEY 8(f) f2 (BUILD-DEFINED:
 * the reference has no such graphs): edges run from N_src source nodes to N_dst target
 * nodes, no self-loops; CSR by target as above with rowptr [N_dst+1].  mean = 1: val = w / (sum of the
 * row's weights) (mean aggregation); mean = 0: val = w.  K2/K4 take it as is (x has N_src rows).
 * workspace: gwen_gcn_prep_workspace_bytes(N_dst, E). */
int gwen_gcn_prep_rect(const int64_t *edge_index, const float *edge_weight, int64_t N_src,
                       int64_t N_dst, int64_t E, int mean, int32_t *rowptr, int32_t *col, float *val,
                       int32_t *eid, int32_t *status, void *workspace, size_t workspace_bytes,
                       gwen_stream_t stream);

/* Transposed structure (CSR by SOURCE) of a prepared graph, for the backward pass
 * (grad_h = A~^T grad_out).  Same value array semantics; rows keep target order ascending.
 * workspace: same size as for gwen_gcn_prep with E := rowptr[N] upper bound (E + N). */
int gwen_gcn_transpose(const int32_t *rowptr, const int32_t *col, const float *val, int64_t N,
                       int64_t cap, int32_t *t_rowptr, int32_t *t_col, float *t_val,
                       void *workspace, size_t workspace_bytes, gwen_stream_t stream);
/* The same for a rectangular CSR (gwen_gcn_prep_rect: N target rows, columns in [0, N_t)): the transpose
 * has N_t rows (t_rowptr [N_t + 1]).  workspace: gwen_gcn_prep_workspace_bytes(max(N, N_t), cap). */
int gwen_gcn_transpose_rect(const int32_t *rowptr, const int32_t *col, const float *val, int64_t N,
                            int64_t N_t, int64_t cap, int32_t *t_rowptr, int32_t *t_col, float *t_val,
                            void *workspace, size_t workspace_bytes, gwen_stream_t stream);

/* Grouped layout of a prepared graph for K4: every row padded to a whole number of 8-entry groups
 * (padding entries carry weight 0 and the column of the row's first entry), so the kernel reads
 * indices and weights as aligned 32-byte groups and needs no per-entry bounds logic.
 *   g_rowptr int32 [N+1] (entry offsets, multiples of 8), g_col / g_val [gwen_gcn_group8_capacity()],
 *   followed at g_rowptr[N] by one all-zero "null group" (col 0, weight 0).
 * uniform (int32 [1], may be NULL) is set to 1 when EVERY row is exactly one group (bounded-degree
 *   meshes: 1 <= entries <= 8), i.e. g_rowptr[r] == 8 r.  K4/K5 then accept rowptr == NULL and skip the
 *   row-pointer lookup (one dependent load less per gathered row).
 * workspace: as for gwen_gcn_prep. */
int64_t gwen_gcn_group8_capacity(int64_t N, int64_t cap);
int gwen_gcn_group8(const int32_t *rowptr, const int32_t *col, const float *val, int64_t N,
                    int64_t cap, int32_t *g_rowptr, int32_t *g_col, float *g_val, int32_t *uniform,
                    void *workspace, size_t workspace_bytes, gwen_stream_t stream);

/* Content checksum of a device buffer (128 bits: two independent position-dependent 64-bit sums over its
 * 8-byte words), the key under which a caller caches prepared graphs: the reference's loaders hand every
 * forward a NEW edge_index tensor with the same edges (/root/reference/src/gwen/models_gnn.py:351-360).
 * data: 8-byte aligned; out: uint64 [2] (device); workspace: gwen_checksum_workspace_bytes() bytes.
 * Deterministic; two small launches; nothing is read back here. */
int64_t gwen_checksum_workspace_bytes(void);
int gwen_checksum128(const void *data, int64_t bytes, uint64_t *out, void *workspace,
                     int64_t workspace_bytes, gwen_stream_t stream);

/* Row segmentation for LONG rows (graphs beyond K7's 256 nodes whose rows are much longer than the 8 entries
 * one lane group gathers per round trip: complete graphs over more than 256 members, hub nodes).  Row r of
 * the CSR (length len) is cut into max(1, ceil(len / S)) segments of S entries:
 *     seg_rowptr int32 [n_seg + 1]   a REFINEMENT of rowptr over the same col / val arrays: K2 on
 *                                    (seg_rowptr, col, val) gives one partial sum per segment, every segment
 *                                    summed in stored order by its own lane group (edge-parallel);
 *     rowptr2 [N + 1], col2 = 0 .. n_seg - 1, val2 = 1   the CSR that adds each row's partial sums in
 *                                    segment order: K2 on it (bias / ReLU there) finishes the row.
 * Deterministic (fixed order, no atomics); differs from K2's sequential sum by rounding order only.
 * rowptr2 itself may be segmented again when a row has more than S segments.
 * status int32 [2]: [0] = most segments of a row, [1] = longest row.  n_seg = rowptr2[N] <=
 * gwen_gcn_segments_capacity(N, nnz, S).  workspace: gwen_gcn_prep_workspace_bytes(N, 0) bytes. */
int64_t gwen_gcn_segments_capacity(int64_t N, int64_t nnz, int64_t S);
int gwen_gcn_segments(const int32_t *rowptr, int64_t N, int64_t S, int32_t *rowptr2, int32_t *seg_rowptr,
                      int32_t *col2, float *val2, int32_t *status, void *workspace,
                      size_t workspace_bytes, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  fused propagate == MessagePassing.propagate (message w~ * x_j, aggregate add at target)
 *     + bias add (GCNConv.forward) + torch.relu (/root/reference/src/gwen/models_gnn.py:147-149,
 *     :204-205) in one pass:   out[m,i,:] = act( sum_s val[s] * h[m, col[s], :] + bias ).
 * h   : fp32 [members, N, F] with row stride ldh (>= F) and member stride mstride_h (elements).
 * out : fp32 [members, N, F] with row stride ldo and member stride mstride_o.  out != h.
 * bias: fp32 [F] or NULL.  relu: 0/1.
 * Terms are added sequentially in stored order with a rounded product per term (no FMA), so the
 * result is bitwise reproducible and equals a sequential CPU scatter-add in edge order.
 * ------------------------------------------------------------------------------------------- */
int gwen_gcn_propagate_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                           const float *h, const float *bias, float *out, int64_t N, int64_t F,
                           int64_t ldh, int64_t ldo, int64_t members, int64_t mstride_h,
                           int64_t mstride_o, int relu, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Contractions.  The reference's `lin` is an fp32 GEMM (/root/reference/src/gwen/models_gnn.py:118-130 -> PyG
 * Linear).  Every kernel with a dense contraction (K3, K4, K5, K8) takes one of these codes (the parameter is
 * called `exact` in K3 / K4 for history, `contract` elsewhere):
 *   GWEN_CONTRACT_BF16X3 (0)  operands cut into two bf16 images (x = hi + lo), 3 MFMAs per k-step, fp32
 *                             accumulation: ~17 bits per product, 7e-6 relative on the 6-layer model;
 *   GWEN_CONTRACT_F32    (1)  fp32-input MFMA: bit-identical to a k-ordered fp32 fmaf chain, 1/16 of the bf16 rate;
 *   GWEN_CONTRACT_BF16X6 (2)  three bf16 images per operand, 6 MFMAs per k-step: 24 bits per operand, fp32-class
 *                             (<= 2e-6 on the 6-layer model) at 6/16 of the fp32 MFMA's cost;
 *   GWEN_CONTRACT_F16X3  (3)  two fp16 images per operand, both operands scaled by exact powers of two (W per output
 *                             column, the rows per 64-feature chunk) so that fp16's range holds them: operands kept to
 *                             2^-24, 3 MFMAs per k-step -- fp32-class at bf16x3's cost.  K8 has it at every
 *                             width (gwen_gcn_wide_layer_f32).  On a layer of the stack launcher
 *                             (gwen_layer_desc.contract) it means "fp32-class on the kernel's own split": K8 runs
 *                             f16x3, K3 / K4 / K5 / K7 run bf16x6 -- the host API's default.
 * ------------------------------------------------------------------------------------------- */
#define GWEN_CONTRACT_BF16X3 0
#define GWEN_CONTRACT_F32 1
#define GWEN_CONTRACT_BF16X6 2
#define GWEN_CONTRACT_F16X3 3

/* ---------------------------------------------------------------------------------------------
 * K3  dense projection == GCNConv.lin (PyG Linear(Fin, Fout, bias=False)):  h = x @ W^T.
 * x [rows, Fin] (row stride ldx), W [Fout, Fin] contiguous (lin.weight), h [rows, Fout]
 * (row stride ldh).  Optional epilogue: + bias[Fout] (NULL = none), ReLU.
 * exact: a GWEN_CONTRACT_* code (0: bf16x3 split, fp32 accumulate, as K4; 1: fp32-input MFMA
 *   v_mfma_f32_32x32x2_f32, bit-identical to a k-ordered fp32 fmaf chain; 2: bf16x6 split).
 * workspace (optional, gwen_gcn_linear_workspace_floats() elements; 0 = not needed): lets the split
 *   variant cut a long K over several blocks when there are few output tiles (few rows x wide input,
 *   the reference's own C -> 1024 projection on ~125 nodes) and add the partial products in a fixed
 *   order; without it that shape runs on a handful of CUs.
 * ------------------------------------------------------------------------------------------- */
int64_t gwen_gcn_linear_workspace_floats(int64_t rows, int64_t Fin, int64_t Fout);
int gwen_gcn_linear_f32(const float *x, const float *W, const float *bias, float *h, int64_t rows,
                        int64_t Fin, int64_t Fout, int64_t ldx, int64_t ldh, int relu, int exact,
                        float *workspace, int64_t workspace_floats, gwen_stream_t stream);
/* The same product with the weight operand given as Wt [Fin, Fout] row-major ("NN"): h = act(x Wt + bias).  The
 * backward's gx = gh W with W as the layer stores it ([out, in] of the forward = [K, N] here), without a transposing
 * copy; out = D g for a dense adjacency D.  Split contractions only (GWEN_CONTRACT_BF16X3 / BF16X6; F16X3 = BF16X6
 * here), ldw == Fout, x != h; workspace as gwen_gcn_linear_f32 (split-K). */
int gwen_gcn_linear_nn_f32(const float *x, const float *Wt, const float *bias, float *h, int64_t rows, int64_t Fin,
                           int64_t Fout, int64_t ldx, int64_t ldw, int64_t ldh, int relu, int contract,
                           float *workspace, int64_t workspace_floats, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4  one whole GCNConv layer (+ReLU) in a single launch, aggregate-first:
 *        out = act( (A~ x) W^T + bias )        (== A~ (x W^T) + bias, A~ linear)
 * Gathers x at width Fin, keeps the aggregated tile in LDS, contracts it with W on the matrix cores
 * and stores once at width Fout: no [N, Fout] intermediate `h` goes through HBM.
 * exact = 0: the contraction runs as a 3-term bf16 split (x = hi + lo, fp32 accumulate; < 2^-15
 *   relative per product, 8e-6 measured on the 6-layer model against a 1e-4 tolerance);
 * exact = 1: fp32-input MFMA, bit-exact fp32 fmaf chains (about 1.3x slower at 64 -> 64);
 * exact = 2: bf16x6 split (GWEN_CONTRACT_BF16X6), one more LDS image of the tile.
 * rowptr/col/val here are the GROUPED arrays of gwen_gcn_group8() (rows in whole groups of 8, null
 * group at rowptr[N]; rowptr may be NULL for a uniform layout, see gwen_gcn_group8); x rows must be contiguous (ldx == Fin) and N * Fin * 4 < 2^32.
 * Supported widths: Fin, Fout in {16, 32, 64, 128, 256} (gwen_gcn_layer_supported() says; otherwise
 * GWEN_EINVAL: use K3 + K2).  Same alignment rules as K2.
 * ------------------------------------------------------------------------------------------- */
int gwen_gcn_layer_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
                       const float *W, const float *bias, float *out, int64_t N, int64_t Fin,
                       int64_t Fout, int64_t ldx, int64_t ldo, int64_t members, int64_t mstride_x,
                       int64_t mstride_o, int relu, int exact, gwen_stream_t stream);
int gwen_gcn_layer_supported(int64_t Fin, int64_t Fout);

/* ---------------------------------------------------------------------------------------------
 * K8  K4's contract for WIDE layers on locality-ordered bounded-degree graphs (meshes), tile-staged:
 *        out = act( (A~ x) W^T + bias )
 * (the same piece of /root/reference/src/gwen/models_gnn.py:147-149,:204-206 as K4).  Destination rows
 * are cut into tiles of GWEN_TILE_ROWS; per tile the UNION of the source rows its entries name is
 * staged once in LDS by LDS-DMA, 64 features at a time, and the 7-8 gathers per destination row are
 * served from there, overlapped with the 3xbf16 MFMA contraction of the previous chunk.
 *
 * gwen_gcn_tiles64 (part of K1, once per graph; plain CSR in -- rowptr/col/val of gwen_gcn_prep or
 * gwen_gcn_prep_rect, every row at most 8 entries):
 *   t_rows int32 [T * GWEN_TILE_UNION]  the tile's distinct source rows, ascending; a group of 4 slots that
 *                                      starts past the union holds -1, other slots past it the first row
 *   t_lid  u16   [T * 512]             per entry slot (row-in-tile * 8 + k) the rank of its source row in
 *                                      the union; padding slots repeat the row's first entry
 *   t_val  fp32  [T * 512]             per entry slot its weight, 0 for padding slots and rows >= N
 *   status int32 [2]                   status[0] = 1 if some row has more than 8 entries or some tile's
 *                                      union exceeds GWEN_TILE_UNION (K8 must not be used: K4 or K3 + K2
 *                                      instead); status[1] = largest union.  T = gwen_gcn_tiles64_count(N).
 * gwen_gcn_wide_layer_f32: x [members, N_src, Fin] contiguous rows, out [members, N, Fout] row stride ldo;
 *   Fin, Fout in {64, 128, 256} (gwen_gcn_wide_supported); N_src * Fin * 4 < 2^32.  Split contraction `contract`,
 *   fp32 accumulation; the bf16 splits are term for term the arithmetic of gwen_gcn_layer_f32(exact = contract).
 *   union_max = status[1] of gwen_gcn_tiles64 (any upper bound <= GWEN_TILE_UNION is correct): with unions
 *   of at most 128 rows the kernel keeps two chunks of DMA in flight instead of one.
 * ------------------------------------------------------------------------------------------- */
#define GWEN_TILE_ROWS 64
#define GWEN_TILE_UNION 192
int64_t gwen_gcn_tiles64_count(int64_t N);
int gwen_gcn_tiles64(const int32_t *rowptr, const int32_t *col, const float *val, int64_t N,
                     int32_t *t_rows, uint16_t *t_lid, float *t_val, int32_t *status,
                     gwen_stream_t stream);
/* HOST arrays in, HOST array out (no device work): a locality order of the rows of a square prepared CSR whose own
 * numbering has none -- breadth-first balls of GWEN_TILE_ROWS rows grown next to each other over the in-neighbour
 * lists.  perm [N]: new position -> old row.  Relabel the CSR with it (rows permuted, columns mapped through the
 * inverse, entry order inside a row kept) and K8 tiles what it could not before; results stay bitwise those of the
 * unpermuted kernels because every row still sums its entries in stored order. */
int gwen_cluster_rows64_host(const int32_t *rowptr, const int32_t *col, int64_t N, int64_t N_src, int32_t *perm);
int gwen_gcn_wide_supported(int64_t Fin, int64_t Fout);
/* contract: GWEN_CONTRACT_BF16X3 or GWEN_CONTRACT_BF16X6 for every supported width pair (bf16x6 needs tile unions
 * within 128 rows; at 256 -> 256 it runs as two 256 -> 128 launches: three images of W for all 256 output columns
 * exceed the registers of the 8 waves that hold them); GWEN_CONTRACT_F16X3: one launch at every width pair, unions up to
 * GWEN_TILE_UNION rows. */
int gwen_gcn_wide_contract_supported(int64_t Fin, int64_t Fout, int contract);
/* 1 when an AUTO layer of these widths over N rows x members is issued as K8 rather than K4 (given a graph
 * that tiles): every supported width pair with Fin >= 128 (measured 1.15x - 1.6x K4), and Fin = 64 once the
 * layer's input no longer sits in the caches (members * N >= 300 000 rows; equal to K4 below that). */
int gwen_gcn_wide_preferred(int64_t N, int64_t members, int64_t Fin, int64_t Fout);
int gwen_gcn_wide_layer_f32(const int32_t *t_rows, const uint16_t *t_lid, const float *t_val,
                            const float *x, const float *W, const float *bias, float *out, int64_t N,
                            int64_t N_src, int64_t Fin, int64_t Fout, int64_t ldo, int64_t members,
                            int64_t mstride_x, int64_t mstride_o, int relu, int64_t union_max,
                            int contract, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5  K4 with the NEXT layer's projection chained on (A~ is linear, so a layer may be gathered at
 * min(Fin, Fout); for a shrinking layer its projection must exist before its gather starts):
 *   pre = 0:  out = act( (A~ x) W1^T + bias ) W2^T     x [.,Fin], W1 [F1,Fin], W2 [F2,F1], F2 < F1
 *   pre = 1:  out = act( A~ h + bias ) W1^T            h [.,Fin] already projected, bias [Fin], F2 = 0
 * rowptr/col/val: GROUPED arrays (rowptr NULL = uniform layout); x, out contiguous rows; widths in {16, 32, 64, 128};
 * contract: GWEN_CONTRACT_BF16X3 or GWEN_CONTRACT_BF16X6, as K4.  The re-bracketing changes fp32 rounding order only.
 * ------------------------------------------------------------------------------------------- */
int gwen_gcn_chain_supported(int64_t Fin, int64_t F1, int64_t F2, int pre, int contract);
int gwen_gcn_chain_f32(const int32_t *rowptr, const int32_t *col, const float *val, const float *x,
                       const float *W1, const float *W2, const float *bias, float *out, int64_t N,
                       int64_t Fin, int64_t F1, int64_t F2, int pre, int relu, int64_t members,
                       int64_t mstride_x, int64_t mstride_o, int contract, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-stack forward == GNNModel.forward (/root/reference/src/gwen/models_gnn.py:292-303 ->
 * :241-258 -> :135-157, :189-212): every layer of the stack issued back to back from one host
 * call, so the host never limits the device (the reference pays ~15 eager launches per layer).
 *
 * layers (HOST array): per layer W [fout, fin], bias [fout] or NULL (device pointers), relu 0/1 and
 *   order: GWEN_ORDER_AUTO picks K4 when gwen_gcn_layer_supported(fin, fout) -- and chains the next
 *   layer's projection (K5) when that layer is AUTO too and shrinks, so that it is gathered at its
 *   narrow width -- otherwise transform-first (K3 then K2) when fout <= fin, aggregate-first (K2
 *   then K3) when fin < fout.  Explicit orders are taken literally, layer by layer.
 * graph (HOST struct of device pointers): rowptr/col/val = the prepared CSR (K2 layers); g_rowptr/g_col/g_val
 *   = its grouped form (K4/K5 layers; g_col/g_val may be NULL when no layer resolves to K4/K5, g_rowptr
 *   NULL = uniform layout); dense = the graph as a dense padded matrix (gwen_gcn_dense_f32) or NULL -- when
 *   given and every layer is AUTO and gwen_gcn_small_supported(N, fin, fout), every layer is ONE K7 launch;
 *   t_rows/t_lid/t_val/union_max = the tile layout of gwen_gcn_tiles64 or NULL -- when given, AUTO layers
 *   with gwen_gcn_wide_preferred() run as K8.
 * x [members, N, layers[0].fin] and out [members, N, layers[n-1].fout] contiguous; out != x.
 * scratch: fp32 workspace of gwen_gnn_forward_scratch_floats() elements (16-byte aligned).
 * acts (HOST array of n_layers device pointers, or NULL): the TRAINING forward -- layer l's output is
 *   stored in acts[l] ([members, N, layers[l].fout]; acts[n_layers-1] may be `out`) and no projection is
 *   chained across layers (every activation the backward needs exists).
 * events (HOST array of hipEvent_t, or NULL): if given, events[2i] / events[2i+1] are recorded on
 *   `stream` right before / after kernel launch i; info[i] (HOST, or NULL) says what launch i was.
 *   n_launches (HOST, or NULL) receives the number of launches; max_launches bounds both arrays.
 * ------------------------------------------------------------------------------------------- */
#define GWEN_ORDER_AUTO (-1)
#define GWEN_ORDER_TRANSFORM_FIRST 0
#define GWEN_ORDER_AGGREGATE_FIRST 1
#define GWEN_ORDER_FUSED 2         /* K4, split contraction (gwen_layer_desc.contract) */
#define GWEN_ORDER_FUSED_EXACT 3   /* K4, fp32 MFMA contraction */
#define GWEN_KIND_LINEAR 3      /* K3 */
#define GWEN_KIND_PROPAGATE 2   /* K2 */
#define GWEN_KIND_LAYER 4       /* K4 */
#define GWEN_KIND_CHAIN 5       /* K5: info.fin = gathered width, info.fout = stored width */
#define GWEN_KIND_SMALL 6       /* K7: whole layer on a small graph (dense adjacency) */
#define GWEN_KIND_WIDE 8        /* K8: whole layer, tile-staged */

typedef struct gwen_graph {
  int64_t N;                          /* nodes (rows of x and out) */
  const int32_t *rowptr, *col;        /* gwen_gcn_prep */
  const float *val;
  const int32_t *g_rowptr, *g_col;    /* gwen_gcn_group8 (g_rowptr NULL = uniform) or NULL */
  const float *g_val;
  const float *dense;                 /* gwen_gcn_dense_f32 or NULL */
  const int32_t *t_rows;              /* gwen_gcn_tiles64 or NULL */
  const uint16_t *t_lid;
  const float *t_val;
  int64_t union_max;
} gwen_graph;

typedef struct gwen_layer_desc {
  const float *W;
  const float *bias;
  int32_t fin, fout, relu, order;
  const void *packed;  /* gwen_gcn_small_pack_f32 image of W for K7, or NULL */
  int32_t contract;    /* AUTO / FUSED layers: GWEN_CONTRACT_BF16X3, _BF16X6 or _F16X3 (fp32-class on the kernel's own
                          split, see above); explicit transform-first / aggregate-first / FUSED_EXACT orders
                          contract in fp32 */
  int32_t reserved;
} gwen_layer_desc;

typedef struct gwen_launch_info {
  int32_t kind, layer, fin, fout;
} gwen_launch_info;

int64_t gwen_gnn_forward_scratch_floats(int64_t N, int64_t members, const gwen_layer_desc *layers,
                                        int32_t n_layers);
int gwen_gnn_forward_f32(const gwen_graph *graph, const gwen_layer_desc *layers, int32_t n_layers,
                         const float *x, float *out, float *scratch, int64_t scratch_floats,
                         int64_t members, gwen_stream_t stream, void **events,
                         gwen_launch_info *info, int32_t max_launches, int32_t *n_launches,
                         float *const *acts);

/* ---------------------------------------------------------------------------------------------
 * K7  a whole GCNConv layer on a SMALL graph (N <= 256) with wide features -- the reference's own
 * shape: the complete graph over ~125-150 ensemble members (/root/reference/src/gwen/utils.py:175-176),
 * features = flattened fields, hidden 1024 (/root/reference/src/gwen/config.json:9,12).
 *   gwen_gcn_dense_f32: dense[i*NP + j] = sum of the stored weights of entries (i <- j) of a prepared
 *     square CSR (K1), zero elsewhere; NP = gwen_gcn_small_pad(N) = 128 or 256; dense: fp32 [NP*NP].
 *   gwen_gcn_small_layer_f32: out[m] = act( dense (x[m] W^T) + bias ), x [members, N, Fin] contiguous
 *     rows (member stride mstride_x), W [Fout, Fin], out [members, N, Fout]; Fin % 32 == 0,
 *     Fout % 16 == 0 (gwen_gcn_small_supported).  Both contractions on the bf16 split `contract` names
 *     (GWEN_CONTRACT_BF16X3 or _BF16X6; the packed images serve bf16x3 only: with bf16x6 W is read and split
 *     in the kernel), fp32 accumulation; the aggregation is a dense contraction, so only the summation order
 *     differs from K2's.
 *     workspace: gwen_gcn_small_workspace_floats() fp32 elements (0 unless K is cut over blocks).
 *   gwen_gcn_small_pack_f32: W -> its 3xbf16 hi / lo images in MFMA fragment order
 *     (gwen_gcn_small_pack_bytes() bytes, as many as W itself); pass the result as `packed` (W may then
 *     be NULL) and the kernel streams the weights as one contiguous run per column tile -- worth it
 *     when the weights are reused (inference): the C -> 1024 and 1024 -> C layers are bound by reading W.
 * ------------------------------------------------------------------------------------------- */
int gwen_gcn_small_pad(int64_t N);
int gwen_gcn_small_supported(int64_t N, int64_t Fin, int64_t Fout, int contract);
int64_t gwen_gcn_small_workspace_floats(int64_t N, int64_t members, int64_t Fin, int64_t Fout);
int gwen_gcn_dense_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t N,
                       float *dense, gwen_stream_t stream);
int64_t gwen_gcn_small_pack_bytes(int64_t Fin, int64_t Fout);
int gwen_gcn_small_pack_f32(const float *W, int64_t Fin, int64_t Fout, void *packed,
                            gwen_stream_t stream);
int gwen_gcn_small_layer_f32(const float *dense, const float *x, const float *W, const void *packed,
                             const float *bias, float *out, int64_t N, int64_t Fin, int64_t Fout,
                             int64_t members,
                             int64_t mstride_x, int64_t mstride_o, int relu, float *workspace,
                             int64_t workspace_floats, int contract, gwen_stream_t stream);

/* hipEvent plumbing for callers without a HIP binding (bench.py times kernels with these, on the
 * stream the kernels are launched on). */
int gwen_event_create(void **event /* host out */);
int gwen_event_destroy(void *event);
int gwen_event_record(void *event, gwen_stream_t stream);
int gwen_event_synchronize(void *event);
int gwen_event_elapsed_ms(void *start, void *stop, float *ms /* host out */);

/* ---------------------------------------------------------------------------------------------
 * Backward pieces (autograd of the layer; the reference trains through it:
 * /root/reference/src/gwen/models_gnn.py:372 loss.backward()).
 *   grad_W[Fout,Fin] = g^T @ x   (g [rows,Fout], x [rows,Fin]); deterministic two-stage reduce.  contract: the
 *                      contraction of the layer the gradient belongs to -- GWEN_CONTRACT_F32: fp32-input MFMA (exact
 *                      fp32 products); _BF16X3 / _BF16X6 (_F16X3 = _BF16X6 here): the split contractions on 64 x 64
 *                      tiles of grad_W, operands staged through LDS, where Fin and Fout are multiples of 64 (at 256
 *                      channels 2.7 - 3.7 x the fp32 MFMA's rate), the fp32 MFMA elsewhere.
 *   grad_b[F]        = column sums of g.
 *   relu mask        : g *= (y > 0).
 * partial: fp32 workspace of gwen_gcn_grad_workspace_floats(rows, Fin, Fout) elements.
 * ------------------------------------------------------------------------------------------- */
int64_t gwen_gcn_grad_workspace_floats(int64_t rows, int64_t Fin, int64_t Fout);
int gwen_gcn_grad_weight_f32(const float *g, const float *x, float *grad_W, int64_t rows,
                             int64_t Fin, int64_t Fout, int64_t ldg, int64_t ldx, float *partial,
                             int contract, gwen_stream_t stream);
int gwen_gcn_grad_bias_f32(const float *g, float *grad_b, int64_t rows, int64_t F, int64_t ldg,
                           float *partial, gwen_stream_t stream);
int gwen_relu_backward_f32(const float *y, const float *g, float *gin, int64_t count,
                           gwen_stream_t stream);
/* Building blocks of gwen_gnn_backward_f32: stage 1 of the two reductions alone (per-chunk partial sums,
 * [gwen_gcn_grad_weight_chunks(rows, Fin, Fout, contract) <= gwen_gcn_grad_chunks(rows), Fout * Fin] and
 * [gwen_gcn_grad_chunks(rows), F]), the fixed-order finish of up to
 * GWEN_MAX_REDUCE_TASKS such reductions in ONE launch (dst[j] = sum over chunks of partial[c * count + j]),
 * and up to GWEN_MAX_REDUCE_TASKS small transposes (wt [cols, rows] = w [rows, cols]^T) in one launch.
 * tasks / w / wt / rows / cols are HOST arrays (they travel as kernel arguments). */
#define GWEN_MAX_REDUCE_TASKS 32
typedef struct gwen_reduce_task {
  const float *partial;
  float *dst;
  int64_t count, nchunks;
} gwen_reduce_task;
int64_t gwen_gcn_grad_chunks(int64_t rows);
int64_t gwen_gcn_grad_weight_chunks(int64_t rows, int64_t Fin, int64_t Fout, int contract);
int gwen_gcn_grad_weight_partial_f32(const float *g, const float *x, float *partial, int64_t rows,
                                     int64_t Fin, int64_t Fout, int64_t ldg, int64_t ldx, int contract,
                                     gwen_stream_t stream);
int gwen_gcn_grad_bias_partial_f32(const float *g, float *partial, int64_t rows, int64_t F, int64_t ldg,
                                   gwen_stream_t stream);
/* Stage 1 of grad_W = g^T x AND of grad_b = column sums of g in ONE launch (the g tile is staged anyway), where the
 * weight gradient runs on the LDS-staged split kernel: gwen_gcn_grad_weight_bias_supported(Fin, Fout, contract) -- widths
 * that are multiples of 64 on a split contraction; g, x 16-byte aligned with ldg, ldx multiples of 4 (GWEN_EINVAL
 * otherwise: use the two separate launches).  partial_w as above; partial_b [gwen_gcn_grad_weight_chunks(...), Fout]. */
int gwen_gcn_grad_weight_bias_supported(int64_t Fin, int64_t Fout, int contract);
int gwen_gcn_grad_weight_bias_partial_f32(const float *g, const float *x, float *partial_w, float *partial_b,
                                          int64_t rows, int64_t Fin, int64_t Fout, int64_t ldg, int64_t ldx,
                                          int contract, gwen_stream_t stream);
int gwen_reduce_chunks_batched(const gwen_reduce_task *tasks, int32_t n_tasks, gwen_stream_t stream);
int gwen_transpose_batched(const float *const *w, float *const *wt, const int32_t *rows,
                           const int32_t *cols, int32_t n, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of a whole layer / of the whole stack (the reference's loss.backward(),
 * /root/reference/src/gwen/models_gnn.py:372, through six GCNConv layers).
 *
 * gwen_gcn_layer_bwd_f32: K4's kernel on the TRANSPOSED graph (t_* = gwen_gcn_group8 of gwen_gcn_transpose):
 *     gh = A~^T g                         stored (the operand of grad_W = gh^T x); gh may be NULL
 *     gx = (gh Wt^T) masked by mask > 0   Wt [Fx, Fg] = the layer's lin.weight ([Fg, Fx]) transposed;
 *                                         mask = the output of the layer below when it has a ReLU, or NULL
 *   g, gh [members, N, Fg]; gx, mask [members, N, Fx], contiguous; widths as gwen_gcn_layer_supported(Fg, Fx);
 *   contract: GWEN_CONTRACT_BF16X3 or GWEN_CONTRACT_BF16X6 -- the split of the gx contraction.
 * gwen_gnn_backward_f32: all layers, last to first, from one host call: per layer grad_b (column sums of
 *   the incoming, already masked gradient), the launch above (or K2^T + K3 + mask where the widths are not
 *   K4's) on the LAYER'S OWN precision (bf16x3 layers: bf16x3; bf16x6 / f16x3 layers: bf16x6; explicit fp32
 *   orders: the fp32-input MFMA), grad_W (fp32-input MFMA reductions).  graph_t: the views of the TRANSPOSED graph (rowptr/col/val and, for the fused launch,
 *   g_rowptr/g_col/g_val).  acts (HOST array of n_layers device pointers): every layer's output as the
 *   training forward stored it (gwen_gnn_forward_f32 with acts); x: the stack's input; grad_out: gradient of
 *   the last layer's output.  grad_x may be NULL; grad_W / grad_b: HOST arrays of device pointers (NULL array
 *   or NULL entry = not wanted).  scratch: gwen_gnn_backward_scratch_floats() fp32 elements.
 * ------------------------------------------------------------------------------------------- */
int gwen_gcn_layer_bwd_f32(const int32_t *t_rowptr, const int32_t *t_col, const float *t_val,
                           const float *g, const float *Wt, const float *mask, float *gh, float *gx,
                           int64_t N, int64_t Fg, int64_t Fx, int64_t members, int contract, gwen_stream_t stream);
/* The same launch with stage 1 of the grad_b of the layer BELOW as well: gx is that layer's incoming gradient, and its
 * column sums per (member, chunk of rows) cost the kernel one LDS turn -- *bias_chunks (a HOST integer, written by
 * this call) partial rows of Fx floats go to bias_partial (room for gwen_gcn_layer_bwd_bias_rows(N, members) rows),
 * every one written exactly once; gwen_reduce_chunks_batched finishes them in a fixed order.  bias_partial and
 * bias_chunks both NULL: gwen_gcn_layer_bwd_f32. */
int64_t gwen_gcn_layer_bwd_bias_rows(int64_t N, int64_t members);
int gwen_gcn_layer_bwd_bias_f32(const int32_t *t_rowptr, const int32_t *t_col, const float *t_val, const float *g,
                                const float *Wt, const float *mask, float *gh, float *gx, int64_t N, int64_t Fg,
                                int64_t Fx, int64_t members, int contract, float *bias_partial,
                                int64_t *bias_chunks, gwen_stream_t stream);
int64_t gwen_gnn_backward_scratch_floats(int64_t N, int64_t members, const struct gwen_layer_desc *layers,
                                         int32_t n_layers);
int gwen_gnn_backward_f32(const struct gwen_graph *graph_t, const struct gwen_layer_desc *layers,
                          int32_t n_layers, const float *x, const float *const *acts,
                          const float *grad_out, float *grad_x, float *const *grad_W,
                          float *const *grad_b, float *scratch, int64_t scratch_floats, int64_t members,
                          gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K6 -- InteractionNet block: edge MLP + sum to target nodes, and the node MLP  (SURVEY 8(f) f2).
 * BUILD-DEFINED: the reference has no edge MLP (its layers are the GCNConv calls at
 * /root/reference/src/gwen/models_gnn.py:147-149,:204-206); BASELINE.json's north_star names the
 * block, semantics follow the published Interaction Network formulation restated in
 * oracle/interaction_oracle.py (PARITY UNPINNED).
 *
 * gwen_mlp2_f32, for rows r = 0 .. R-1 of width F (gwen_mlp2_supported):
 *     pre[r] = A[r] W1^T + G1[idx1 ? idx1[r] : r] + G2[idx2 ? idx2[r] : r] + b1    (G1, G2 optional)
 *     y[r]   = act(pre[r]) W2^T + b2                       act: GWEN_ACT_NONE / RELU / SILU
 *     out[r] = res[r] + y[r]                               (out, res optional; out may alias A / res)
 *     agg[d] = sum (or mean) of y[r] over rows r with rowptr[d] <= r < rowptr[d+1]   (agg optional)
 *   A, G*, res, out: fp32 row-major, contiguous rows of F; W1, W2: [F,F] row-major [out,in].
 *   G1 / G2 have G1_rows / G2_rows rows with row stride ldg1 / ldg2 floats (>= F, multiple of 4;
 *   rows * ld * 4 < 2^32 -- a table may be a column block of a wider matrix, e.g. one of several
 *   projections computed by a single K3 launch); supported table pairs: none; G1 alone
 *   (with or without idx1); G1 and G2 both indexed.  GWEN_EINVAL otherwise.
 *   With agg: rows are the stored entries of a target-sorted CSR (rowptr int32 [N_agg+1],
 *   rowptr[N_agg] == R) and tile_row [n_tiles+1] comes from gwen_edge_tiles; every target row is
 *   summed by one block in stored order (no atomics; rows without entries get 0).
 *   The contractions use the 3xbf16 split (see gwen_gcn_layer_f32, exact = 0).
 *   workspace: gwen_mlp2_workspace_bytes(F) bytes, 16-byte aligned (GWEN_ENOSPACE if smaller): only F = 256
 *   needs one -- its weights are streamed from pre-split bf16 fragment images (F = 64 splits them in-kernel).
 *
 * gwen_edge_tiles: row-aligned tiling of a CSR's entries.  tile c owns the target rows whose first
 *   entry lies in [cT, (c+1)T); n_tiles = gwen_edge_tiles_count(E, T) = max(1, ceil(E/T)); T <= 128.
 *   T = gwen_mlp2_rows(F) - (max row length - 1) (at least 1) keeps each tile a single pass of the
 *   kernel at width F; any T is correct (longer tiles take several passes).
 *   Also writes dst[e] = target row of entry e (int32 [E]) -- idx2 of the edge MLP.
 * ------------------------------------------------------------------------------------------- */
#define GWEN_ACT_NONE 0
#define GWEN_ACT_RELU 1
#define GWEN_ACT_SILU 2
int gwen_mlp2_supported(int64_t F);          /* F in {32, 64, 128, 256} */
int gwen_mlp2_rows(int64_t F);                /* rows one pass of the kernel takes at width F (128 at F = 64 and 256, else 64) */
int64_t gwen_mlp2_workspace_bytes(int64_t F); /* F = 256: room for the pre-split weight images; else 0 */
int64_t gwen_edge_tiles_count(int64_t E, int64_t T);
int gwen_edge_tiles(const int32_t *rowptr, int64_t N, int64_t E, int64_t T, int32_t *tile_row,
                    int32_t *dst, gwen_stream_t stream);
int gwen_mlp2_f32(const float *A, const float *W1, const float *G1, const int32_t *idx1,
                  int64_t G1_rows, int64_t ldg1, const float *G2, const int32_t *idx2,
                  int64_t G2_rows, int64_t ldg2,
                  const float *b1, const float *W2, const float *b2, const float *res, float *out,
                  int64_t R, int64_t F, int act, const int32_t *rowptr, const int32_t *tile_row,
                  int64_t n_tiles, float *agg, int64_t N_agg, int mean, void *workspace,
                  size_t workspace_bytes, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K6^T  pieces of the InteractionNet block's BACKWARD (build-defined like K6; serves the training step of the
 * reference's loop shape, /root/reference/src/gwen/models_gnn.py:372-373, for the InteractionNet forecaster).
 * The host (gwen_amd/interaction.py) assembles the backward from K3 (dense products), K2 over CSRs whose columns
 * are edge positions (sums over the edges of a target / source, in stored order), the weight / bias gradient
 * reductions below K4's backward, and these three: every launch is atomic-free with a fixed summation order.
 *   gwen_act_pair_f32:   pre[r] = a[r] + g1[idx1 ? idx1[r] : r] + g2[idx2 ? idx2[r] : r]  (tables optional, row
 *                        strides ld1 / ld2 >= F);  h[r] = act(pre[r]);  dact[r] = act'(pre[r])  (dact may be NULL;
 *                        h may alias a).  a, h, dact: [rows, F] contiguous, F % 4 == 0.
 *   gwen_act_pair_seg_f32: the same for edges STORED BY TARGET (rows rowptr[d] .. rowptr[d + 1] belong to target d, g2 is
 *                        the target's own row: g2[d]) with the per-target sums of h in stored order as well,
 *                        hsum[d] = sum_r h[r]  [n_dst, F] -- gwen_act_pair_f32 followed by K2 over the edge-position
 *                        CSR, bit for bit, without the second pass over [rows, F].
 *   gwen_gather_add_f32: out[r] = (a ? a[r] : 0) + t[idx[r]] * (scale ? scale[idx[r]] : 1)   t: [*, F] contiguous.
 *   gwen_ew_f32:         out = a * b (GWEN_EW_MUL) or a + b (GWEN_EW_ADD), n % 4 == 0; out may alias a or b.
 *   gwen_mlp2_bwd_f32 (round 4): the edge-level half of the backward as ONE launch of K6's row-stationary kernel, at
 *                        64 and 256 channels (gwen_mlp2_bwd_supported):
 *                            g_pre1[r] = (ge[r] W2 + T[dst[r]]) * d1[r]        g_e[r] = ge[r] + g_pre1[r] We
 *                        with T = (g_agg, scaled by 1 / degree for the mean) W2 computed per NODE, so that the message
 *                        gradient ge + g_agg[dst] is never formed; W2t = W2^T, Wet = We^T ([F, F] row-major: row =
 *                        output column); d1 = act'(pre1) [R, F]; T [T_rows, F] with row stride ldT.  g_pre1 has
 *                        gwen_mlp2_bwd_rows(R) rows (R rounded up to whole passes: the kernel stores every lane), g_e
 *                        [R, F] may be ge itself.  workspace as gwen_mlp2_f32.  3xbf16 contractions, as K6.
 * ------------------------------------------------------------------------------------------- */
#define GWEN_EW_MUL 0
#define GWEN_EW_ADD 1
int gwen_act_pair_f32(const float *a, const float *g1, const int32_t *idx1, int64_t ld1, const float *g2,
                      const int32_t *idx2, int64_t ld2, float *h, float *dact, int64_t rows, int64_t F, int act,
                      gwen_stream_t stream);
int gwen_act_pair_seg_f32(const float *a, const float *g1, const int32_t *idx1, int64_t ld1, const float *g2,
                          int64_t ld2, const int32_t *rowptr, float *h, float *dact, float *hsum, int64_t rows,
                          int64_t n_dst, int64_t F, int act, gwen_stream_t stream);
int gwen_gather_add_f32(const float *a, const float *t, const int32_t *idx, const float *scale, float *out,
                        int64_t rows, int64_t F, gwen_stream_t stream);
int gwen_ew_f32(int op, const float *a, const float *b, float *out, int64_t n, gwen_stream_t stream);
int gwen_mlp2_bwd_supported(int64_t F);
int64_t gwen_mlp2_bwd_rows(int64_t R);
int gwen_mlp2_bwd_f32(const float *ge, const float *W2t, const float *d1, const float *T, const int32_t *dst,
                      int64_t T_rows, int64_t ldT, const float *Wet, float *g_pre1, float *g_e, int64_t R, int64_t F,
                      void *workspace, size_t workspace_bytes, gwen_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Masked L1 loss, value and gradient in one pass -- the reference's training objective
 * (/root/reference/src/gwen/models_gnn.py:261-265: F.l1_loss(output[mask], target[mask]); backward :372):
 *     loss[0] = sum_{m, r : mask[r], c} |out[m,r,c] - target[m,r,c]| / (picked * C * members)
 *     grad[m,r,c] = sign(out - target) * mask[r] / (picked * C * members)      (grad may be NULL)
 * out, target, grad: fp32 [members, N, C] contiguous, C % 4 == 0, 16-byte aligned; mask: one byte per row [N]
 * (non-zero = selected); picked = number of selected rows (0 => NaN, as torch).  Three launches, sums in a
 * fixed order (bitwise reproducible).  workspace: gwen_masked_l1_workspace_floats() floats.
 * ------------------------------------------------------------------------------------------- */
int64_t gwen_masked_l1_workspace_floats(void);
int gwen_masked_l1_f32(const float *out, const float *target, const uint8_t *mask, int64_t members, int64_t N,
                       int64_t C, float *grad, float *loss, float *workspace, int64_t workspace_floats,
                       gwen_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GWEN_HIP_H */
