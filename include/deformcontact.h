/*
 * deformcontact.h -- C ABI of libdeformcontact_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of the DeformContact message-passing hot path.  The
 * reference has no FFI of its own: its boundary is the PyG operator surface
 * called from /root/reference/models/model.py:45,49,71,77
 * (`conv_layer(in, out)`, `conv(x, edge_index)`), whose arithmetic is PyG 2.5.2's
 * gcn_norm + MessagePassing.propagate + Linear.  Each entry point below names
 * the PyG/ATen step it replaces.  The Python host (deformcontact_amd/) binds
 * these with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the
 *     library allocates nothing and keeps no mutable global state except the
 *     thread-local last-error string;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t); no call
 *     synchronises, so all of them are legal inside hipGraph capture;
 *   - return value 0 = enqueued, <0 = DC_E* (nothing enqueued), message in
 *     dc_last_error();
 *   - matrices are row-major fp32 with an explicit leading dimension (in
 *     elements); index arrays produced by the library are int32; the only int64
 *     input is PyG's `edge_index` [2,E].
 */
#ifndef DEFORMCONTACT_H
#define DEFORMCONTACT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_OK 0
#define DC_EINVAL (-1)   /* bad argument (null pointer, negative size, alignment) */
#define DC_ELAUNCH (-2)  /* hipLaunch / runtime error, text in dc_last_error()   */
#define DC_EWORKSPACE (-3)

typedef void *dc_stream_t; /* hipStream_t */

#define DC_MAX_PARTS 4     /* edge_index parts of one merged (block-diagonal) adjacency */
#define DC_MAX_GROUPS 4    /* row groups with their own weights in one grouped dense launch */
#define DC_GROUP_ALIGN 256 /* every group of a grouped launch starts at a multiple of this many rows */

int dc_version(void);
const char *dc_last_error(void);
/* Sequence id of the hipGraph capture `stream` currently takes part in, 0 when it is not
 * capturing.  The host's adjacency cache uses it: a sorted adjacency is only reused by the
 * capture that enqueued its build (the build has not RUN while a capture is open). */
int64_t dc_stream_capture_id(dc_stream_t stream);

/* ---- topology: edge_index -> sorted adjacency ("CSR") ---------------------
 * Replaces the implicit ordering of PyG's scatter_add_ (utils/_scatter.py) and,
 * with `w != NULL`, gcn_norm (nn/conv/gcn_conv.py) which the reference
 * recomputes in every conv call (model.py:71,77 -> TAGConv.forward).
 *
 * Groups the E edges by edge_index[key_row] (key_row = 1: by destination, the
 * forward CSR; key_row = 0: by source, the transposed set used by backward),
 * STABLY: inside a group edges keep their input order, i.e. the result equals
 * np.argsort(edge_index[key_row], kind="stable").  Bit-exact, deterministic.
 *
 *   self_loops = 0 : edge set used as is (TAGConv: gcn_norm(add_self_loops=False))
 *   self_loops = 1 : existing self loops dropped, one loop per node appended
 *                    with edge ids E..E+N-1 (GCNConv add_remaining_self_loops,
 *                    GATConv remove_self_loops + add_self_loops; utils/loop.py).
 *                    Output arrays must then hold E+N entries; the actual edge
 *                    count is ptr[N].
 *
 *   ptr   [N+1]  group offsets
 *   other [E']   the other endpoint of each sorted edge (source for key_row=1)
 *   perm  [E']   original edge id of each sorted edge
 *   w     [E']   optional: deg^-1/2[src] * deg^-1/2[dst], deg = in-degree
 *                counting duplicates, 0 for deg = 0.  `deg_ptr` = offsets whose
 *                differences give that in-degree: NULL = this call's own `ptr`
 *                (valid for key_row = 1), else the key_row=1 ptr of the same
 *                edge set.
 *   status[1]    device int32, SET by the call: 1 if any endpoint is outside [0,N)
 *                (such edges are skipped), else 0.  Read it back after synchronising.
 *   workspace    dc_csr_workspace_bytes(E, N) bytes, 16-byte aligned.
 *
 * Groups of any length are handled in O(S log^2 S) (hubs are sorted by one workgroup, short
 * groups ranked by counting): no quadratic walk on high-degree nodes.
 */
int64_t dc_csr_workspace_bytes(int64_t E, int64_t N);
int dc_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int key_row,
                 int self_loops, int32_t *ptr, int32_t *other, int32_t *perm,
                 const int32_t *deg_ptr, float *w, int32_t *status, void *workspace,
                 int64_t workspace_bytes, dc_stream_t stream);

/* Both sides of one edge set in ONE pipeline (5 launches for N <= 65,536, 7 beyond): the
 * key_row = 1 set (`*_f`: by destination, the forward hop) and the key_row = 0 set (`*_b`: by
 * source, the transposed hop of the backward pass), identical to two dc_csr_build calls with
 * deg_ptr = ptr_f for the second.  w_f / w_b both NULL or both set.  This is what one
 * `conv(x, edge_index)` of the reference needs per NEW edge_index (model.py:71,77); it is cheap
 * enough (tens of microseconds at the batch-32 shape) to sit inside a captured training step.
 * Edge lists of 2^19 slots and more (E, + N with self loops) take the bucketed build - partition by
 * node bucket, one workgroup per bucket, no atomic per edge: the same arrays whatever the order of
 * the edges (a relabelled radius graph; env DC_CSR_BUCKETS = 0 / 1 forces the choice). */
int64_t dc_graph_workspace_bytes(int64_t E, int64_t N);
int dc_graph_build(const int64_t *edge_index, int64_t E, int64_t N, int self_loops,
                   int32_t *ptr_f, int32_t *other_f, int32_t *perm_f, float *w_f,
                   int32_t *ptr_b, int32_t *other_b, int32_t *perm_b, float *w_b,
                   int32_t *status, void *workspace, int64_t workspace_bytes, dc_stream_t stream);

/* dc_graph_build over a MERGED edge set: `nparts` (<= DC_MAX_PARTS) edge_index arrays
 * (edge_index_parts[p] = int64 [2, E_parts[p]]) read as one concatenated edge list - edge ids run
 * through the parts in order - over ONE node space of N nodes in which part p's node ids are
 * shifted by node_offset_parts[p] (ascending; part p must stay inside
 * [node_offset_parts[p], node_offset_parts[p] + nodes_parts[p]), which is also the range its ids
 * are checked against).  This is the block-diagonal union PyG's Batch.from_data_list builds per
 * graph type, taken one step further: the soft and the rigid graph of a batch (train.py:36-38)
 * in one adjacency, so that the conv layers of both encoder branches (models/model.py:69-78) run
 * as single launches.  Nodes that belong to no part (padding between parts) are isolated.  Rows
 * of part p come out exactly as dc_graph_build over that part alone would emit them (same stable
 * order, same gcn_norm weights), with `other` shifted by the part's offset and `perm` by the
 * part's first edge id. */
int dc_graph_build_parts(const int64_t *const *edge_index_parts, const int64_t *E_parts,
                         const int64_t *node_offset_parts, const int64_t *nodes_parts, int nparts,
                         int64_t N, int self_loops,
                         int32_t *ptr_f, int32_t *other_f, int32_t *perm_f, float *w_f,
                         int32_t *ptr_b, int32_t *other_b, int32_t *perm_b, float *w_b,
                         int32_t *status, void *workspace, int64_t workspace_bytes, dc_stream_t stream);

/* Which pipeline dc_csr_build / dc_graph_build / dc_graph_build_parts take for an edge set of this size (host only,
 * no HIP call): returns 1 = bucketed build, with *bucket_shift = log2 of the nodes per bucket and *buckets = buckets
 * per side; 0 = windowed pipeline (outputs untouched); DC_EINVAL for negative sizes.  The rule: at least
 * DC_CSR_BUCKETS_MIN slots (env, default 2^19; slots = E, + N with self loops), buckets such that an average
 * bucket fills at most a quarter of one 12,288-slot LDS pass, at most 2,048 nodes per bucket and 8,192 buckets.
 * (Diagnostic / test hook: nothing in the reference corresponds to it.) */
int dc_graph_build_plan(int64_t E, int64_t N, int self_loops, int *bucket_shift, int *buckets);

/* dc_graph_build for a BATCH of graphs whose layout is known (Batch.from_data_list, the loaders of
 * /root/reference/loaders/everyday.py:96 via torch_geometric.data.Batch): graph i owns nodes
 * [node_ptr[i], node_ptr[i+1]) and input edges [edge_ptr[i], edge_ptr[i+1]) and no edge leaves its
 * graph.  node_ptr / edge_ptr are HOST arrays of nseg + 1 int64 (ascending, from 0 to N / E; checked
 * here: DC_EINVAL otherwise, and when a graph exceeds DC_SEG_MAX_NODES / DC_SEG_MAX_EDGES); they
 * travel in the kernel arguments, so under hipGraph capture they are part of the captured launch.
 * One launch per 96 graphs - workgroup (graph, side) sorts the graph's edges in LDS - no workspace,
 * no global atomics; writes exactly the arrays dc_graph_build (self_loops = 0) writes, bit for bit.
 * status is only OR-ed into (the caller zeroes it once): bit 0 = an edge leaves its graph (the arrays
 * then hold in-range but meaningless entries, as with an out-of-range id); a flag survives rebuilds
 * until the caller clears it. */
#define DC_SEG_MAX_NODES 4096
#define DC_SEG_MAX_EDGES 16384
int dc_graph_build_segmented(const int64_t *edge_index, int64_t E, int64_t N,
                             const int64_t *node_ptr_host, const int64_t *edge_ptr_host, int nseg,
                             int32_t *ptr_f, int32_t *other_f, int32_t *perm_f, float *w_f,
                             int32_t *ptr_b, int32_t *other_b, int32_t *perm_b, float *w_b,
                             int32_t *status, dc_stream_t stream);

/* 30-bit Morton (Z-order) code of every point of pos [n, >=3] (fp32, leading dimension ld): each
 * axis is mapped from [lo[a], lo[a] + 1 / inv_extent[a]) to 10 bits (lo / inv_extent are HOST
 * arrays of 3 floats).  The host sorts nodes by it (graph.NodeOrder.morton) so that the rows a
 * radius-graph hop gathers (utils/pointcloud_utils.py:10) are neighbours in memory. */
int dc_morton_codes(const float *pos, int64_t ld, int64_t n, const float *lo_host,
                    const float *inv_extent_host, int64_t *codes, dc_stream_t stream);

/* ---- node reordering of one big graph, on the device (dc_order.hip; BASELINE.json configs[4]) ----
 * dc_morton_order: Z-order permutation of a point cloud pos [n, >= 3] (fp32, leading dimension ld): the bounding
 *   box, the 30-bit codes (10 bits per axis over its extent), a STABLE key sort (rocPRIM device radix sort) and the
 *   inverse - perm[new] = old, inv[old] = new (int32) - without a host synchronisation (hipGraph-capturable).
 *   Running a conv stack on gather_rows(x, perm) / relabel(edge_index) and putting the result back with
 *   gather_rows(y, inv) turns the hops' neighbour gathers of a radius graph (utils/pointcloud_utils.py:10) into reads
 *   of nearby rows; outputs are bit-identical to the unordered run (a row's neighbours stay in edge order).
 * dc_relabel_edges: out[i] = inv[ei[i]] over `count` int64 ids (both rows of edge_index at once).
 * dc_gather_rows  : out row i = x row idx[i], rows of row_bytes bytes (a multiple of 16; any element type). */
int64_t dc_morton_order_workspace_bytes(int64_t n);
int dc_morton_order(const float *pos, int64_t ld, int64_t n, int32_t *perm, int32_t *inv, void *workspace,
                    int64_t workspace_bytes, dc_stream_t stream);
int dc_relabel_edges(const int64_t *ei, int64_t count, const int32_t *inv, int64_t n, int64_t *out,
                     dc_stream_t stream);
int dc_gather_rows(const void *x, int64_t ldx_bytes, const int32_t *idx, void *out, int64_t ldo_bytes, int64_t n,
                   int64_t row_bytes, dc_stream_t stream);

/* Order-dependent 64-bit content hash of an int64 device array (e.g. edge_index) into
 * out[1] (device): the key of the host's per-topology cache (loaders.TopologyCache), so a batch
 * whose edge_index has been seen before reuses its sorted adjacency. */
int dc_hash_i64(const int64_t *v, int64_t n, uint64_t *out, dc_stream_t stream);

/* ---- the hop: fused gather - scale - segment-sum --------------------------
 * Replaces MessagePassing.propagate for aggr="add" with message = w_e * x_j:
 * index_select(x, 0, row) -> mul -> zeros(N,F).scatter_add_(0, col) (three ATen
 * ops and two [E,F] temporaries per hop in PyG).
 *
 *   y[i, 0:F] = (addend ? addend[i, 0:F] : 0)
 *             + sum over p in [ptr[i], ptr[i+1]) of  w[p] * x[other[p], 0:F]
 *
 * summed in p order with separately rounded multiply and add, so the result is
 * bit-identical to a serial walk over the stably sorted edges.  w == NULL means
 * weight 1.  Forward passes the key_row=1 set, backward the key_row=0 set
 * (y = A_hat^T g + addend).  y must not alias x; y may alias addend.
 */
int dc_spmm_f32(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                int64_t ldx, const float *addend, int64_t ldadd, float *y, int64_t ldy,
                int64_t N, int64_t F, dc_stream_t stream);

/* dc_tag_pack_input + the FIRST hop of a narrow layer's own input in one launch (F <= 32: the encoder's first TAGConv,
 * models/model.py:71,77 with the raw graph.x): slab[i, 0:F] = x[i, 0:F], slab[i, F:2F] = sum_p w[p] x[other[p], 0:F]
 * (terms and order of dc_spmm_f32), slab[i, width:wpad] = 0.  One launch less on the chain of small dependent kernels
 * a new batch starts with. */
int dc_spmm_f32_pack(const int32_t *ptr, const int32_t *other, const float *w, const float *x, int64_t ldx,
                     float *slab, int64_t lds, int64_t N, int64_t F, int64_t width, int64_t wpad,
                     dc_stream_t stream);

/* dc_spmm_f32 over a ROW WINDOW of a merged adjacency (dc_graph_build_parts): `ptr` points at the
 * window's first row (N + 1 entries; its values index the full other / w arrays), neighbour ids are
 * ids of the merged node space, and x / addend / y hold only the window's N rows: the neighbour
 * row read is other[p] - row_offset.  Lets one part of a merged graph (e.g. the first, narrow
 * layer of one encoder branch) hop over its own feature matrix without a second adjacency. */
int dc_spmm_f32_window(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                       int64_t ldx, const float *addend, int64_t ldadd, float *y, int64_t ldy,
                       int64_t N, int64_t F, int64_t row_offset, dc_stream_t stream);
/* ... and with the row maxima of dc_spmm_f32_rowmax (rowmax holds the window's N rows). */
int dc_spmm_f32_rowmax_window(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                              int64_t ldx, const float *addend, int64_t ldadd, float *y, int64_t ldy,
                              int64_t N, int64_t F, float *rowmax, int mode, int64_t row_offset,
                              dc_stream_t stream);

/* The same hop over bf16-STORED feature rows ("bf16 features, fp32
 * accumulate"; PyG reaches it as conv(x.bfloat16(), edge_index) under autocast, where
 * index_select/mul/scatter_add_ run in bf16 -- this entry keeps the sum in fp32 instead).
 * x is bf16 (uint16 bit patterns, row-major, ldx in elements).  Every term is
 * fp32(w[p]) * fp32(x[other[p]]) with the running sum in fp32, rounded and ordered exactly as
 * dc_spmm_f32.  y_is_f32 != 0: y / addend are fp32 and receive the fp32 sum; y_is_f32 == 0:
 * y / addend are bf16 and the sum is rounded ONCE, to nearest even, on store.  Half the gather
 * bytes of the fp32 hop: algorithmic bytes E*(8 + 2F) + N*(sizeof(y)*F + 4). */
int dc_spmm_bf16(const int32_t *ptr, const int32_t *other, const float *w, const uint16_t *x,
                 int64_t ldx, const void *addend, int64_t ldadd, void *y, int64_t ldy, int64_t N,
                 int64_t F, int y_is_f32, dc_stream_t stream);

/* dc_spmm_f32 that also records the row maxima the fp16x2 dense block scales by:
 * rowmax[i] = max |y[i,:]|, joined with max |x[i,:]| (the row's own input) when mode & 1 and with
 * the value rowmax[i] already holds when mode & 2.  y is bit-identical to dc_spmm_f32. */
int dc_spmm_f32_rowmax(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                       int64_t ldx, const float *addend, int64_t ldadd, float *y, int64_t ldy,
                       int64_t N, int64_t F, float *rowmax, int mode, dc_stream_t stream);

/* ---- K chained hops of a batch of small graphs in ONE launch (dc_hopchain.hip) ---------------
 * TAGConv.forward's K dependent propagate calls (PyG nn/conv/tag_conv.py, reached from
 * models/model.py:71,77) over a batch whose layout is known (Batch.from_data_list, train.py:36-38):
 * graph i owns the contiguous node range [node_ptr_host[i], node_ptr_host[i+1]) and none of its
 * edges leaves it (the layout dc_graph_build_segmented takes).  `slab` is the [N, ld] hop slab of a
 * layer, column block j = slab[:, j*F : (j+1)*F]; the call computes, for k = 1..K,
 *     block (src_block + k*dir)  =  A_hat . block (src_block + (k-1)*dir)
 * with every graph's 32-column slice resident in LDS for the whole chain - memory traffic: block
 * src_block in once, K blocks out once - and terms, order and rounding of dc_spmm_f32, so every
 * block is bit-identical to K dc_spmm_f32 calls.  rowmax != NULL: rowmax[i] = max over the K
 * produced blocks of max |block[i, :]|, joined with block src_block's when mode & 1 and with the
 * value rowmax[i] already holds when mode & 2 (what K dc_spmm_f32_rowmax calls leave there).
 * `cap` = number of elements `other` / `w` hold (16-byte id / weight loads are range-checked
 * against it).  deg_ptr != NULL (int32 [N+1]) states that the weights are gcn_norm's without self
 * loops, w[p] = d(source)^-1/2 * d(destination)^-1/2 with d(v) = deg_ptr[v+1] - deg_ptr[v] (the
 * in-degree: ptr of the by-destination set, for BOTH sets of dc_graph_build*): the kernel then
 * re-forms them from an LDS-resident table instead of loading them (same bits) and its hop loop
 * is free of vector-memory loads.  Needs F % 32 == 0, ld % 4 == 0, a 16-byte aligned slab and at most
 * dc_hop_chain_max_nodes() (4096: 32-column slices up to 1,024 nodes, 16-column ones up to 2,048, 8-column ones beyond; the
 * LDS-table form only with 32-column slices) nodes per graph; DC_EINVAL otherwise (use dc_spmm_f32). */
int64_t dc_hop_chain_max_nodes(void);
/* Bytes of LDS every dc_hop_chain_f32 workgroup requests: the WHOLE LDS of a compute unit (160 KiB), whatever its slice
 * needs, so that no LDS-using workgroup of another kernel can be resident beside it (round 5: with LDS left free, a
 * co-resident workgroup of a stock attention kernel on another stream made small chain workgroups return wrong elements
 * in 0.3 - 1.75 % of two-stream train steps; profiles/r05/README.md). */
int64_t dc_hop_chain_lds_request(void);
int dc_hop_chain_f32(const int32_t *ptr, const int32_t *other, const float *w, const int32_t *deg_ptr,
                     int64_t cap, const int64_t *node_ptr_host, int nseg, float *slab, int64_t ld, int64_t N,
                     int64_t F, int K, int src_block, int dir, float *rowmax, int mode,
                     dc_stream_t stream);

/* ---- the dense block over bf16-STORED features (BASELINE.json configs[4]) ------------------
 * out[N,Fo] = act(A[N,K] . W[Fo,K]^T + bias) with A and W bf16 (uint16 bit patterns, K-contiguous,
 * K % 32 == 0, 16-byte aligned, lda % 8 == 0), fp32 accumulate on v_mfma_f32_32x32x16_bf16, out
 * fp32 (out_is_bf16 = 0) or bf16 rounded to nearest even (the next layer's hop slab).  A is the hop
 * slab dc_spmm_bf16 leaves, W the layer's lins[k].weight concatenated along K by dc_to_bf16.  This is
 * the precision PyG reaches under torch.autocast(bfloat16) (nn/dense/linear.py -> F.linear in bf16),
 * not the fp32-accurate split forms above. */
int dc_tag_linear_fwd_bf16(const uint16_t *a, int64_t lda, const uint16_t *w, const float *bias,
                           int relu, void *out, int64_t ldo, int out_is_bf16, int64_t N, int64_t K,
                           int64_t Fo, dc_stream_t stream);
/* Backward of that layer (configs[4] is fwd + bwd): the input gradient runs in the forward's shape -
 * dc_tag_mask_grad_bf16 writes gm = g * (out > 0) as bf16 into block 0 of a gradient slab (g / out_for_mask bf16 or
 * fp32 as flagged), K transposed dc_spmm_bf16 hops fill the other blocks, and dc_tag_linear_fwd_bf16 over that slab
 * with the transposed weights (dc_tag_transpose_weights + dc_to_bf16) gives gx.  dc_tag_linear_bwd_dw_bf16: the weight
 * gradients gws[s] [Fo, Fi] fp32 (the master weights stay fp32) = gm^T . x_s over the bf16 hop slab x [N, nseg * Fi]
 * (+ gbias [Fo] = column sums of gm), fp32 accumulate on v_mfma_f32_32x32x16_bf16, per node chunk with the partial
 * slabs summed in chunk order (deterministic); accumulate != 0 adds into gws / gbias.  Needs Fo % 128 == 0,
 * Fi % 256 == 0, 16-byte aligned rows; any N. */
int dc_tag_mask_grad_bf16(const void *g, int64_t ldg, int g_is_bf16, const void *out_for_mask, int64_t ldo,
                          int mask_is_bf16, uint16_t *gm, int64_t ldgm, int64_t N, int64_t F,
                          dc_stream_t stream);
int64_t dc_tag_linear_bwd_dw_bf16_workspace_bytes(int64_t N, int64_t Fi, int64_t Fo, int nseg);
int dc_tag_linear_bwd_dw_bf16(const uint16_t *g, int64_t ldg, const uint16_t *x, int64_t ldx, int nseg,
                              float *const *gws, float *gbias, int accumulate, void *partials,
                              int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo, dc_stream_t stream);
/* dst[r, s*cols + c] = bf16(srcs[s][r, c]) (round to nearest even): fp32 matrices [rows, cols]
 * (leading dimension ld_src) concatenated along the columns into one bf16 matrix. */
int dc_to_bf16(const float *const *srcs, int nseg, int64_t rows, int64_t cols, int64_t ld_src,
               uint16_t *dst, int64_t ld_dst, dc_stream_t stream);

/* ---- the two training losses (train.py:51-53) ------------------------------------------------
 * losses[0] = L1Loss(pred, target) = mean |pred - target| over the [N,3] positions (train.py:51);
 * losses[1] = GradientConsistencyLoss (models/losses.py:7-19) = (1/E) sum over edges of
 *             || (target[dst] - target[src]) - (pred[dst] - pred[src]) ||_2 ;
 * grad_l1 / grad_gcl [N,3] = their gradients w.r.t. pred (zero gradient at a zero edge norm, as
 * torch's norm backward).  One pass over the nodes through BOTH sorted adjacencies of the edge set
 * (the *_f / *_b arrays of dc_graph_build; weights unused), deterministic two-stage sums, no float
 * atomics.  Replaces ~20 ATen launches (4 row gathers, norm, sums, L1 and their backward). */
int64_t dc_contact_loss_workspace_bytes(int64_t N);
int dc_contact_loss(const int32_t *ptr_f, const int32_t *other_f, const int32_t *ptr_b,
                    const int32_t *other_b, const float *pred, int64_t ld_pred, const float *target,
                    int64_t ld_target, int64_t N, int64_t E, float *grad_l1, float *grad_gcl,
                    float *losses /* [2] */, void *workspace, int64_t workspace_bytes,
                    dc_stream_t stream);

/* ---- dense block of TAGConv ------------------------------------------------
 * Replaces `out = lins[0](x); out = out + lins[k](x_k) ...; out = out + bias`
 * (nn/conv/tag_conv.py forward) = up to DC_MAX_SEG bias-free F.linear calls,
 * adds, and the optional ReLU of model.py:71,77, as ONE fp32-MFMA kernel:
 *
 *   out[N,Fo] = act( sum_s  xs[s][N,Fi] . ws[s][Fo,Fi]^T  + bias )
 *
 * xs[s] has leading dimension ldxs[s]; ws[s] is PyG's Linear.weight layout
 * [Fo,Fi] (ld = Fi).  Exact fp32 (v_mfma_f32_32x32x2_f32 is an fmaf chain).
 */
#define DC_MAX_SEG 4
int dc_tag_linear_fwd(const float *const *xs, const int64_t *ldxs, const float *const *ws,
                      int nseg, const float *bias, int relu, float *out, int64_t ldo,
                      int64_t N, int64_t Fi, int64_t Fo, dc_stream_t stream);

/* dX-side of the backward:  gxs[s][N,Fi] = g[N,Fo] . ws[s][Fo,Fi]   (s = 0..nseg-1),
 * with g optionally masked by the ReLU of the forward output (`out_for_mask`:
 * g is taken as 0 where out <= 0; NULL = no mask). */
int dc_tag_linear_bwd_dx(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo,
                         const float *const *ws, int nseg, float *const *gxs,
                         const int64_t *ldgxs, int64_t N, int64_t Fi, int64_t Fo,
                         dc_stream_t stream);

/* Variants on the bf16 matrix cores.  Every fp32 operand is split exactly into bf16 terms
 * (hi + mid + lo) while it is staged into LDS and a product tile is `products`
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
 *   products = 6 : hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid - fp32-accurate (split error
 *                  ~6e-9 relative, below the rounding of the fp32 accumulation), 2.7x the
 *                  fp32 MFMA peak;
 *   products = 3 : hi*hi, hi*mid, mid*hi (~4e-6 relative);
 *   products = 1 : plain bf16 operands, fp32 accumulate (BASELINE.json configs[4]).
 * Same arguments and semantics as the functions above; shapes the split kernels do not cover
 * (unaligned operands, reduction extent not a multiple of 16) silently take the fp32 path.
 * The dX variant needs a scratch of ..._workspace_bytes() for transposed weights. */
int dc_tag_linear_fwd_split(const float *const *xs, const int64_t *ldxs, const float *const *ws,
                            int nseg, const float *bias, int relu, float *out, int64_t ldo,
                            int64_t N, int64_t Fi, int64_t Fo, int products, dc_stream_t stream);
/* The forward dense block of a TAGConv layer with a SHORT reduction - the first layer of each encoder branch
 * (/root/reference/models/model.py:44-50: TAGConv(21, 256), TAGConv(25, 256); PyG tag_conv.py: sum_k lins[k](x_k) + bias,
 * followed by the F.relu of models/model.py:71,77):
 *     out[N, 256] = act(slab[N, wpad] . [W_0 | ... | W_{nseg-1} | 0]^T + bias)
 * slab = the layer's hop slab, (nseg * fi) data columns zero-padded to wpad = 96 / 112 / 128; ws[k] = lins[k].weight
 * [256, fi], read as they are (no packing launch).  Six-product bf16 arithmetic, bit-identical to dc_tag_linear_fwd_split
 * (products = 6) on the packed weights.  One persistent launch, weights resident in registers, whole reduction in one
 * stage, 1-KiB row stores (dc_dense_narrow.hip).  _ok: host-side query whether the entry takes a shape. */
int dc_tag_linear_fwd_narrow_ok(int64_t fi, int nseg, int64_t wpad, int64_t Fo);
int dc_tag_linear_fwd_narrow(const float *slab, int64_t ld, const float *const *ws, int nseg, int64_t fi,
                             const float *bias, int relu, float *out, int64_t ldo, int64_t N, int64_t wpad,
                             int64_t Fo, dc_stream_t stream);
int64_t dc_tag_linear_bwd_dx_split_workspace_bytes(int64_t Fi, int64_t Fo, int nseg);
int dc_tag_linear_bwd_dx_split(const float *g, int64_t ldg, const float *out_for_mask,
                               int64_t ldo, const float *const *ws, int nseg, float *const *gxs,
                               const int64_t *ldgxs, void *workspace, int64_t workspace_bytes,
                               int64_t N, int64_t Fi, int64_t Fo, int products, dc_stream_t stream);

/* dW-side: (g*relu')^T[Fo,N] . xs[s][N,Fi] (same optional ReLU mask), gbias[Fo] = column
 * sums of g*relu' (NULL to skip).  The result is delivered as `ngw` output blocks (ngw a
 * multiple of nseg): block j = columns [(j % bps)*gw_cols, +gw_cols) of segment j / bps,
 * bps = ngw / nseg, written to gws[j] as a contiguous [Fo, gw_cols] matrix - i.e. one block
 * per PyG `lins[k].weight` whether the layer ran as K+1 segments (ngw = nseg, gw_cols = Fi)
 * or as one concatenated segment (nseg = 1, ngw = K+1, gw_cols = F_in).  accumulate != 0 adds
 * into gws / gbias (fused gradient accumulation) instead of overwriting.
 * `partials` is a caller-owned scratch of dc_tag_linear_bwd_dw_workspace_bytes() bytes
 * (split-N partial slabs, summed in chunk order by a second kernel: deterministic, no float
 * atomics). */
int64_t dc_tag_linear_bwd_dw_workspace_bytes(int64_t N, int64_t Fi, int64_t Fo, int nseg);
int dc_tag_linear_bwd_dw(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo,
                         const float *const *xs, const int64_t *ldxs, int nseg,
                         float *const *gws, int ngw, int64_t gw_cols, float *gbias,
                         int accumulate, void *partials, int64_t partials_bytes, int64_t N,
                         int64_t Fi, int64_t Fo, dc_stream_t stream);
/* split-bf16x6 variant of the above (operands read through ds_read_b64_tr_b16) */
int dc_tag_linear_bwd_dw_split(const float *g, int64_t ldg, const float *out_for_mask,
                               int64_t ldo, const float *const *xs, const int64_t *ldxs, int nseg,
                               float *const *gws, int ngw, int64_t gw_cols, float *gbias,
                               int accumulate, void *partials, int64_t partials_bytes, int64_t N,
                               int64_t Fi, int64_t Fo, int products, dc_stream_t stream);

/* ---- fp16x2 ("h2") dense block: two scaled fp16 planes per operand, 3 MFMA products ---------
 * Same results contract as the *_split entries with products = 6 (fp32-accurate: simulated and
 * measured error at the level of fp32 accumulation) at half the matrix work.  Every operand is
 * multiplied by an exact power of two that puts its reference magnitude in [2^14, 2^15) before it
 * is written as h1 + h2 (fp16), and the accumulator is multiplied back:
 *   forward : row i of x by x_rowmax[i] >= max_k |x_k[i,:]| (dc_spmm_f32_rowmax records it for
 *             free), row o of the weights by w_rowmax[o] = max_s,f |W_s[o,f]| (dc_tag_weight_rowmax);
 *   dX      : row i of g * relu' by g_rowmax[i] >= max |g[i,:]| (dc_rowabsmax_f32), all weights by
 *             max_o w_rowmax[o] (reduced inside the kernel);
 *   dW      : the contraction runs over nodes, so g * relu' and x get ONE scale per node chunk of
 *             the split-N plan: the maxima of g_rowmax / x_rowmax over the chunk (inside the kernel).
 * Any upper bound is a valid reference.  fp16 denormals are honoured by the matrix core: elements
 * down to 2^-39 of their reference keep absolute accuracy.  Shapes: Fi % 16 == 0 (forward),
 * Fo % 16 == 0 (dX), N % 16 == 0 (dW), 16-byte aligned rows; otherwise DC_EINVAL (use *_split). */
int dc_tag_linear_fwd_h2(const float *const *xs, const int64_t *ldxs, const float *const *ws,
                         int nseg, const float *bias, int relu, float *out, int64_t ldo, int64_t N,
                         int64_t Fi, int64_t Fo, const float *x_rowmax, const float *w_rowmax,
                         dc_stream_t stream);
int dc_tag_linear_bwd_dx_h2(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo,
                            const float *const *ws, int nseg, float *const *gxs,
                            const int64_t *ldgxs, void *workspace, int64_t workspace_bytes,
                            int64_t N, int64_t Fi, int64_t Fo, const float *g_rowmax,
                            const float *w_rowmax, dc_stream_t stream);
int dc_tag_linear_bwd_dw_h2(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo,
                            const float *const *xs, const int64_t *ldxs, int nseg,
                            float *const *gws, int ngw, int64_t gw_cols, float *gbias,
                            int accumulate, void *partials, int64_t partials_bytes, int64_t N,
                            int64_t Fi, int64_t Fo, const float *g_rowmax, const float *x_rowmax,
                            dc_stream_t stream);
/* dc_tag_linear_bwd_dw_h2 (no mask, no bias) with the gradient operand g[i, :] - g2_coef[i] * g2[i, :] formed at load
 * time (g2 with g's leading dimension): the attention backward's dK = (dS' - eps o P)^T Q without materialising the
 * corrected dS.  128 x 256 tile kernel only: Fo % 128 == 0, Fi == 256 per segment, N % 32 == 0. */
int dc_tag_linear_bwd_dw_h2_corr(const float *g, int64_t ldg, const float *g2, const float *g2_coef,
                                 const float *const *xs, const int64_t *ldxs, int nseg, float *const *gws, int ngw,
                                 int64_t gw_cols, int accumulate, void *partials, int64_t partials_bytes, int64_t N,
                                 int64_t Fi, int64_t Fo, const float *g_rowmax, const float *x_rowmax,
                                 dc_stream_t stream);
/* Backward in the forward's shape (used with the h2 entries):  the input gradient of a TAGConv layer
 *   gx = sum_k (A^T)^k (gm W_k),  gm = g * relu'
 * equals  sum_k ((A^T)^k gm) W_k  (A acts on rows, W_k on columns), i.e. K transposed hops on gm
 * followed by ONE dense block with the K = (K+1)*Fo reduction of the forward: dc_tag_linear_fwd_h2
 * over the hop slab of gm with the transposed weights.  Helpers:
 *   dc_tag_mask_grad         gm[i,:] = g[i,:] * (out_for_mask[i,:] > 0) into (a column block of) the
 *                            slab, rowmax_a[i] = rowmax_b[i] = max |gm[i,:]| (rowmax_b may be NULL);
 *   dc_tag_transpose_weights wt[s] [Fi, Fo] = ws[s]^T, s < nseg, packed back to back. */
int dc_tag_mask_grad(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo, float *gm,
                     int64_t ldgm, int64_t N, int64_t F, float *rowmax_a, float *rowmax_b,
                     dc_stream_t stream);
int dc_tag_transpose_weights(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *wt,
                             dc_stream_t stream);
/* One launch for everything the h2 blocks of a layer need from its weights W_s [Fo, Fi], s < nseg:
 *   w_rowmax [Fo]          as dc_tag_weight_rowmax;
 *   w_image                (optional) the weights ALREADY scaled and split for the forward block: for
 *                          row o and every 16-wide stage of the concatenated reduction k = s*Fi + f one
 *                          64-byte record {h1[16], h2[16]} (fp16), W*2^e(o) = h1 + h2 with the row's scale
 *                          - Fo * nseg*Fi * 4 bytes, 16-byte aligned; consumed by dc_tag_linear_fwd_h2p,
 *                          which moves it global -> LDS by LDS-DMA instead of re-splitting the weights in
 *                          every row tile;
 *   wt_image, wt_rowmax    (optional, together) the same for the transposed weights (row f, reduction
 *                          k = s*Fo + o, wt_rowmax[f] = max_s,o |W_s[o,f]| [Fi]) for the forward-shaped
 *                          backward block. */
int dc_tag_weight_prep(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *w_rowmax,
                       void *w_image, void *wt_image, float *wt_rowmax, dc_stream_t stream);
/* dc_tag_weight_prep that ALSO clears zero[0:zero_n] (fp32) in the same launch: the row-maxima buffer the layer's
 * dc_hop_chain_f32 launch joins its maxima into (mode | 2) - a separate memset node costs ~5 us on the critical path
 * of every forward chain. */
int dc_tag_weight_prep_zero(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *w_rowmax,
                            void *w_image, void *wt_image, float *wt_rowmax, float *zero, int64_t zero_n,
                            dc_stream_t stream);
/* dc_tag_linear_fwd_h2 for ONE segment x [N, K] (ldx) with pre-split weights (w_image of
 * dc_tag_weight_prep; K = nseg*Fi of that call): out = act(x . W^T + b).  With a workspace of
 * dc_tag_linear_fwd_h2p_workspace_bytes (may be NULL / 0) a long reduction with too few output
 * tiles for the chip (no bias, no relu, ldo == Fo) is cut into ranges whose partials are summed in
 * range order by a second launch (deterministic). */
int64_t dc_tag_linear_fwd_h2p_workspace_bytes(int64_t N, int64_t K, int64_t Fo);
int dc_tag_linear_fwd_h2p(const float *x, int64_t ldx, const void *w_image, const float *bias, int relu,
                          float *out, int64_t ldo, int64_t N, int64_t K, int64_t Fo,
                          const float *x_rowmax, const float *w_rowmax, void *workspace,
                          int64_t workspace_bytes, dc_stream_t stream);
/* The attention backward's recompute of a block's softmax weights in ONE launch (models/model.py:15-17 as the
 * blocked form evaluates it): out[i, j] = exp(x[i,:] . W[j,:] - row_lse[i]) for j < ncols_valid, 0 for the padded
 * columns ncols_valid <= j < Fo - dc_tag_linear_fwd_h2p (no bias, no relu) with dc_attn_exp_rows applied in the
 * epilogue; bit-identical to the two launches. */
int dc_tag_linear_fwd_h2p_exp(const float *x, int64_t ldx, const void *w_image, float *out, int64_t ldo,
                              int64_t N, int64_t K, int64_t Fo, const float *x_rowmax, const float *w_rowmax,
                              const float *row_lse, int64_t ncols_valid, dc_stream_t stream);
/* dc_tag_linear_fwd_h2p (no bias, no relu, no split reduction) with the left operand x[i, :] - x2_coef[i] * x2[i, :]
 * formed at load time (x2 with x's leading dimension): the attention backward's dQ = (dS' - eps o P) K without
 * materialising the corrected dS.  128 x 256 tile kernel only (K % 32 == 0, 16-byte aligned rows). */
int dc_tag_linear_fwd_h2p_corr(const float *x, int64_t ldx, const float *x2, const float *x2_coef, const void *w_image,
                               float *out, int64_t ldo, int64_t N, int64_t K, int64_t Fo, const float *x_rowmax,
                               const float *w_rowmax, dc_stream_t stream);
/* ---- grouped launches: both encoder branches as ONE block-diagonal launch ----
 * The reference runs its two branches as 2 x 2 conv calls (models/model.py:69-78).  Over a merged node
 * space (dc_graph_build_parts) the second layers of both branches - same widths, different weights -
 * become single launches: the hops need nothing new (one adjacency), the dense blocks take GROUPS:
 * group g = rows [row_beg[g], row_beg[g] + rows[g]) of the merged matrices, with its own weights.
 * row_beg[g] and N_total are multiples of DC_GROUP_ALIGN; the rows between groups are padding the
 * CALLER keeps at zero in x and in the gradients (isolated nodes: the hops write zeros there), so
 * they add exact zeros to dW and their outputs are never read.  Per row the arithmetic is exactly
 * that of the ungrouped h2 entries on that group alone: outputs, dX and (with the per-group node
 * chunks of the ungrouped plan) dW are bit-identical to separate launches.
 *   dc_tag_grouped_weight_prep   dc_tag_weight_prep for every group in one launch; ws[g*nseg + s],
 *                                per-group output pointer arrays [ngroups]
 *   dc_tag_grouped_fwd_h2p       dc_tag_linear_fwd_h2p (K % 32 == 0; forward and, with the transposed
 *                                images over the gradient slab, dX)
 *   dc_tag_grouped_mask_grad     dc_tag_mask_grad with one upstream-gradient matrix per group
 *                                (g[k] = rows[k] x F, local row numbering); padding rows get zeros
 *   dc_tag_grouped_bwd_dw_h2     dc_tag_linear_bwd_dw_h2 on the pre-masked gradient (Fi == 256 per
 *                                segment, Fo % 128 == 0): gws[g*nseg + s] [Fo, Fi], gbias[g] [Fo] */
int dc_tag_grouped_weight_prep(const float *const *ws, int ngroups, int nseg, int64_t Fo, int64_t Fi,
                               float *const *w_rowmax, void *const *w_image, void *const *wt_image,
                               float *const *wt_rowmax, dc_stream_t stream);
int dc_tag_grouped_fwd_h2p(const float *x, int64_t ldx, int ngroups, const int64_t *row_beg,
                           const int64_t *rows, int64_t N_total, const void *const *w_images,
                           const float *const *biases, int relu, float *out, int64_t ldo, int64_t K,
                           int64_t Fo, const float *x_rowmax, const float *const *w_rowmaxes,
                           dc_stream_t stream);
int dc_tag_grouped_mask_grad(const float *const *g, const int64_t *ldg, int ngroups, const int64_t *row_beg,
                             const int64_t *rows, int64_t N_total, const float *out_for_mask, int64_t ldo,
                             float *gm, int64_t ldgm, int64_t F, float *rowmax_a, float *rowmax_b,
                             dc_stream_t stream);
int64_t dc_tag_grouped_bwd_dw_workspace_bytes(const int64_t *rows, int ngroups, int64_t Fi, int64_t Fo,
                                              int nseg);
int dc_tag_grouped_bwd_dw_h2(const float *g, int64_t ldg, const float *const *xs, const int64_t *ldxs,
                             int nseg, int ngroups, const int64_t *row_beg, const int64_t *rows,
                             int64_t N_total, float *const *gws, float *const *gbias, int accumulate,
                             void *partials, int64_t partials_bytes, int64_t Fi, int64_t Fo,
                             const float *g_rowmax, const float *x_rowmax, dc_stream_t stream);

/* Number of launches of the GENERIC dense kernels (any shape / alignment, bounds logic per load, scalar
 * loads when misaligned) since the library was loaded or the counter was last reset (reset != 0 returns the
 * count and clears it).  Every shape of the shipped configuration has a tuned kernel; a non-zero count after a
 * default-config step means an operand lost its alignment or a shape fell off the fast paths (round 2 found two
 * such silent fallbacks only through DC_DENSE_TRACE=1) - the test suite pins it at zero. */
int64_t dc_generic_dense_launches(int reset);

/* Launch log: which kernels does a step actually launch?  dc_kernel_trace(1) clears the log and starts counting every
 * kernel launch of the library by kernel name as written at its launch site (e.g. "k_fwd_h2w<true, false>"; template
 * parameters of the launching host function stay symbolic: "k_hop_chain_gcn<STEPS>"), dc_kernel_trace(0) stops.
 * dc_kernel_trace_dump writes "name count\n" lines into buf (NUL-terminated, truncated to cap bytes) and returns the
 * size the whole text needs.  Process-wide, thread-safe, off by default (one predictable branch per launch). */
void dc_kernel_trace(int on);
int64_t dc_kernel_trace_dump(char *buf, int64_t cap);

/* rowmax[i] = max |x[i, 0:F]| for a row-major [N, F] view with leading dimension ld. */
int dc_rowabsmax_f32(const float *x, int64_t ld, int64_t N, int64_t F, float *rowmax,
                     dc_stream_t stream);
/* w_rowmax[o] = max over the nseg weight blocks W_s [Fo, Fi] and f of |W_s[o, f]|. */
int dc_tag_weight_rowmax(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *w_rowmax,
                         dc_stream_t stream);

/* pos_of[perm[p]] = p  (inverse permutation over the E' = *ptr_last sorted edges) */
int dc_invert_perm(const int32_t *perm, const int32_t *ptr_last /* &ptr[N] */, int32_t *pos_of,
                   int64_t max_edges, dc_stream_t stream);
/* out[p] = a[b[p]] for p < *count_ptr (e.g. position-in-forward-order of every
 * backward-ordered edge: a = pos_of, b = perm of the key_row=0 set). */
int dc_compose_perm(const int32_t *a, const int32_t *b, int32_t *out, const int32_t *count_ptr,
                    int64_t cap, dc_stream_t stream);

/* ---- GCNConv / GATConv: the row-wise passes around the aggregation, fused (dc_gnn_epi.hip) ----
 * PyG gcn_conv.py / gat_conv.py: out = propagate(...) + bias; the encoder loop then applies relu
 * (models/model.py:71,77); GAT: alpha_src = (h * att_src).sum(-1), alpha_dst likewise.
 *   dc_spmm_f32_bias_act : y[i,:] = act(sum_p w[p] x[other[p],:] + bias)  (bias may be NULL, relu 0/1);
 *                          the sum as dc_spmm_f32, bias added to the finished sum: same values as the
 *                          three separate passes
 *   dc_mask_colsum_f32   : gm = g * (y_mask > 0) (y_mask NULL: gm = g; gm NULL: not written) and
 *                          colsum[c] (+)= sum_i gm[i,c] (the bias gradient), per-block partials combined
 *                          in block order - deterministic; workspace of dc_colsum_workspace_bytes(N,F,1)
 *   dc_gat_alpha_fwd     : a_src[i] = h[i,:] . att_src, a_dst[i] = h[i,:] . att_dst, one pass over h
 *   dc_gat_alpha_bwd     : gh[i,:] += ga_src[i] att_src + ga_dst[i] att_dst (gh holds the aggregation's
 *                          gradient), g_att_src (+)= sum_i ga_src[i] h[i,:], g_att_dst likewise; workspace
 *                          of dc_colsum_workspace_bytes(N,F,2)
 * F % 4 == 0 and 16-byte aligned rows; the column-sum passes need F/4 to divide 256. */
int dc_spmm_f32_bias_act(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                         int64_t ldx, const float *bias, int relu, float *y, int64_t ldy, int64_t N,
                         int64_t F, dc_stream_t stream);
int64_t dc_colsum_workspace_bytes(int64_t N, int64_t F, int nvec);
int dc_mask_colsum_f32(const float *g, int64_t ldg, const float *y_mask, int64_t ldy, float *gm,
                       int64_t ldgm, int64_t N, int64_t F, void *workspace, int64_t workspace_bytes,
                       float *colsum, int accumulate, dc_stream_t stream);
int dc_gat_alpha_fwd(const float *h, int64_t ldh, const float *att_src, const float *att_dst, float *a_src,
                     float *a_dst, int64_t N, int64_t F, dc_stream_t stream);
int dc_gat_alpha_bwd(const float *h, int64_t ldh, const float *ga_src, const float *ga_dst,
                     const float *att_src, const float *att_dst, float *gh, int64_t ldgh, int64_t N,
                     int64_t F, void *workspace, int64_t workspace_bytes, float *g_att_src,
                     float *g_att_dst, int accumulate, dc_stream_t stream);

/* ---- GATConv edge terms (nn/conv/gat_conv.py edge_update + utils/_softmax.py) -
 * Only reached when the reference is configured with backbone "GATConv"
 * (models/model.py:39).  Arrays are in key_row=1 order of a self_loops=1 set;
 * segments = incoming edges of node i.
 *   e_p     = leaky_relu(a_src[other[p]] + a_dst[i], slope)
 *   alpha_p = exp(e_p - max_seg e) / (sum_seg exp(e - max) + 1e-16)
 * The aggregation out_i = sum_p alpha_p h[other[p]] is dc_spmm_f32 with w = alpha.
 */
int dc_gat_edge_softmax_fwd(const int32_t *ptr, const int32_t *other, const float *a_src,
                            const float *a_dst, float slope, float *alpha, int64_t N,
                            dc_stream_t stream);
/* Backward: given galpha[E'] writes ge[p] = d loss / d (a_src[other[p]] + a_dst[i])
 * (softmax and leaky-relu chained) and g_a_dst[i] = sum of ge over segment i. */
int dc_gat_edge_softmax_bwd(const int32_t *ptr, const int32_t *other, const float *a_src,
                            const float *a_dst, float slope, const float *alpha,
                            const float *galpha, float *ge, float *g_a_dst, int64_t N,
                            dc_stream_t stream);
/* Sampled dense-dense product: d[p] = <g[i,0:F], h[other[p],0:F]> for p in segment i
 * (gradient of the aggregation w.r.t. alpha). */
int dc_sddmm_f32(const int32_t *ptr, const int32_t *other, const float *g, int64_t ldg,
                 const float *h, int64_t ldh, float *d, int64_t N, int64_t F,
                 dc_stream_t stream);
/* out[i] = sum over p in [ptr[i],ptr[i+1]) of v[map ? map[p] : p]  (p order) */
int dc_segment_sum_f32(const int32_t *ptr, const int32_t *map, const float *v, float *out,
                       int64_t N, dc_stream_t stream);
/* out[p] = v[idx[p]] for p < *count_ptr */
int dc_gather_f32(const float *v, const int32_t *idx, float *out, const int32_t *count_ptr,
                  int64_t cap, dc_stream_t stream);

/* ---- packing helpers of the narrow-layer path (F_in = 21 / 25) ----------------
 * A TAGConv layer whose K+1 column blocks are narrow runs its dense block over ONE K segment:
 * the hop slab [N, wpad] (wpad = (K+1)*F rounded up to 16).  pack_input: slab[:, 0:F] = x and
 * slab[:, width:wpad] = 0.  pack_weights: wcat[Fo, wpad] = [ws[0] | ... | ws[nw-1] | 0]. */
int dc_tag_pack_input(const float *x, int64_t ldx, float *slab, int64_t ld_slab, int64_t N,
                      int64_t F, int64_t width, int64_t wpad, dc_stream_t stream);
int dc_tag_pack_weights(const float *const *ws, int nw, float *wcat, int64_t Fo, int64_t fi,
                        int64_t wpad, dc_stream_t stream);

/* ---- row kernels of the blocked cross-attention -------------------
 * models/model.py:7-21: per head  softmax(head(x_soft) . head(x_rigid)^T, dim=-1) . x_rigid, unmasked,
 * no 1/sqrt(d).  The score matrix exists only for a block of soft rows s [rows, ld]; its three
 * GEMMs run on the h2 dense entries above and these kernels do the row-wise parts in place:
 *   softmax_rows: s[i,0:n] <- softmax(s[i,0:n]), s[i,n:npad] <- 0, lse[i] = log sum_j exp(s[i,j])
 *   exp_rows    : s[i,0:n] <- exp(s[i,0:n] - lse[i]), s[i,n:npad] <- 0   (backward recompute)
 *   ds_rows     : dp[i,j] <- p[i,j] * (dp[i,j] - delta[i]) for j < npad; rowmax[i] = max_j |.| */
/* Flash-style FORWARD of one attention head (models/model.py:13-21), d = dv = DC_ATTN_FLASH_D:
 *   o[i,:] = sum_j softmax_j(q[i,:] . k[j,:]) v[j,:],   lse[i] = log sum_j exp(q[i,:] . k[j,:]),  j < nr
 * in ONE launch - the scores of a 128-query tile never leave the compute unit (online softmax over 32-key tiles).
 * Operands as the blocked form hands them to dc_tag_linear_fwd_h2p: q fp32 [ns, ldq] with its row maxima
 * (dc_rowabsmax_f32); k_image = dc_tag_weight_prep of the keys [nr_padded, d], k_unscale from dc_attn_flash_prep;
 * vt_image / vt_rowmax = the transposed image of the values (V^T [dv, nr_padded]) AFTER dc_attn_flash_prep; nr_padded % 32 == 0,
 * rows nr..nr_padded of k / v are zero padding and are masked.  Same products, same order as the blocked form: score (i, j) is bit-identical to
 * dc_tag_linear_fwd_h2p's, so the blocked backward (dc_tag_linear_fwd_h2p_exp with this lse) recomputes exactly
 * the weights that were normalised here. */
#define DC_ATTN_FLASH_D 256
/* Operand preparation of dc_attn_flash_fwd, two small launches: (i) rewrites the transposed image of
 * dc_tag_weight_prep vt_image [dv, nr_padded] IN PLACE into the key order the kernel consumes (the four 4-key chunks of
 * every plane of every 16-key record in the order 0, 2, 1, 3: a lane of the transposed score tile holds the weights
 * of keys {4h..4h+3, 8+4h..11+4h}; the rewrite is its own inverse); (ii) k_unscale[j] = the exact power of two that
 * undoes the scaling of row j of the key image (from k_rowmax of dc_tag_weight_prep), which the kernel fetches with
 * scalar loads. */
int dc_attn_flash_prep(void *vt_image, int64_t dv, int64_t nr_padded, const float *k_rowmax, float *k_unscale,
                       dc_stream_t stream);
int dc_attn_flash_fwd(const float *q, int64_t ldq, const float *q_rowmax, const void *k_image,
                      const float *k_unscale, const void *vt_image, const float *vt_rowmax, int64_t ns,
                      int64_t nr, int64_t nr_padded, int64_t d, float *o, int64_t ldo, float *lse,
                      dc_stream_t stream);
/* The backward's score-sized operands for ALL query rows in one launch (replaces, per 2,048-row block, dc_tag_linear_fwd_h2p_exp
 * + dc_tag_linear_fwd_h2p + dc_attn_ds_rows):  p_out[i, j] = exp(q_i . k_j - lse[i]),  ds_out[i, j] = p_ij (dp_ij - delta_i)
 * with dp_ij = go_i . v_j and delta_i = sum_j p_ij dp_ij / sum_j p_ij formed from the SAME recomputed p and dp (two
 * sweeps over the keys per 128-query tile; nothing score-sized is read back), ds_rowmax[i] = max_j |ds_ij|; columns
 * nr..nr_padded come out as exact zeros.  q / go fp32 with their row maxima; k_image / v_image = dc_tag_weight_prep images
 * of the keys' and the values' ROWS [nr_padded, d], k_unscale / v_unscale from dc_attn_flash_prep (dv = 0 skips the
 * image rewrite); d = dv = DC_ATTN_FLASH_D, nr_padded % 32 == 0, p_out / ds_out [ns, ldp >= nr_padded].  Products as in
 * the blocked form (scores bit-identical to dc_tag_linear_fwd_h2p's).
 * SINGLE-SWEEP mode (delta_in / eps_out not NULL, both [ns]): ds_out = p (dp - delta_in) with the caller's delta
 * (rowsum(dO o O)) and eps_out[i] = sum_j ds_ij / sum_j p_ij, i.e. consistent delta = delta_in + eps: half the matrix
 * work; consumers whose result is sensitive to sum_j ds_ij != 0 take ds - eps_i p (dc_tag_linear_bwd_dw_h2_corr). */
int dc_attn_flash_ds(const float *q, int64_t ldq, const float *q_rowmax, const float *go, int64_t ldgo,
                     const float *go_rowmax, const void *k_image, const float *k_unscale, const void *v_image,
                     const float *v_unscale, const float *lse, int64_t ns, int64_t nr, int64_t nr_padded, int64_t d,
                     float *p_out, float *ds_out, int64_t ldp, float *ds_rowmax, const float *delta_in,
                     float *eps_out, dc_stream_t stream);
int dc_attn_softmax_rows(float *s, int64_t ld, int64_t rows, int64_t n, int64_t npad, float *lse,
                         dc_stream_t stream);
int dc_attn_exp_rows(float *s, int64_t ld, int64_t rows, int64_t n, int64_t npad, const float *lse,
                     dc_stream_t stream);
int dc_attn_ds_rows(const float *p, float *dp, int64_t ld, int64_t rows, int64_t npad,
                    const float *delta, float *rowmax, dc_stream_t stream);

/* ---- optimizer step of the path's training loop -----------------------------
 * torch.optim.Adam(lr) at its defaults (train.py:20: no weight decay, no amsgrad) over ONE
 * flat fp32 bucket of parameters / gradients / moments: a single elementwise pass.  `step` is a
 * zero-initialised device buffer of DC_ADAM_STEP_WORDS 32-bit words: step[0] (float) counts the
 * completed updates and is advanced on the device by the launch itself (the other words are its
 * internal tickets), so the call is replayable inside a hipGraph.  zero_grad != 0 also clears g (optimizer.zero_grad(), train.py:71). */
#define DC_ADAM_STEP_WORDS 34
int dc_adam_flat(float *p, float *g, float *m, float *v, int64_t n, float *step, float lr,
                 float beta1, float beta2, float eps, int zero_grad, dc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEFORMCONTACT_H */
