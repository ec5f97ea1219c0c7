/*
 * dagnn_hip.h - C ABI of the MI355X (gfx950) DAGNN message-passing library (libdagnn_hip.so).
 *
 * The reference (vthost/DAGNN) is pure Python on PyTorch/PyG and has no FFI of its own; the
 * drop-in boundary it exposes is the nn.Module `DAGNN.forward(G)`
 * (ogbg-code/model/dagnn.py:128-215; dvae/dagnn.py:99-184; dvae/dagnn_bn.py:98-177).  The
 * Python mirror of that module lives in dagnn_amd/{model,dvae}.py; every device-side step it
 * takes goes through the entry points declared here, each of which replaces the reference lines
 * cited next to it.  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a BORROWED DEVICE pointer (e.g. torch.Tensor.data_ptr()) unless marked
 *     "host"; the library never allocates, frees or synchronises, so calls are stream-ordered
 *     and hipGraph-capturable (one caveat: a replayed graph repeats the `epoch` it was captured with, so a graph
 *     that contains dagnn_frontier_run / dagnn_backward_run with granule buffers must also contain the memset that
 *     re-zeroes those buffers - see dagnn_frontier_cell.granules);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return value: 0 on success, DAGNN_E* (<0) on bad arguments, -(1000+hipError_t) when a HIP
 *     call fails; nothing throws across the boundary;
 *   - no global mutable state: k threads / k ranks may call concurrently on different devices
 *     (ogbg-code/tg/data_parallel.py:59-62 runs one Python thread per device);
 *   - floats are IEEE fp32; hidden size H must be a multiple of 4 (the host pads otherwise).
 */
#ifndef DAGNN_HIP_H
#define DAGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAGNN_OK 0
#define DAGNN_EINVAL (-22)     /* bad argument (null pointer, H % 4 != 0, negative size ...) */
#define DAGNN_ENOSPC (-28)     /* workspace too small */
#define DAGNN_EHIP(e) (-(1000 + (int)(e)))

#define DAGNN_MAX_DIRS 2
#define DAGNN_MAX_GROUPS 4

/* Library / ABI version string, e.g. "dagnn_hip 0.1 gfx950". Host pointer, static storage. */
const char* dagnn_version(void);

/* ------------------------------------------------------------------------------------------
 * Plan: the layer-sorted, per-graph CSR the recurrence walks.
 * Replaces, per forward call, the reference's frontier selection (dagnn.py:146-147), its
 * per-node scan of the whole edge_index (dagnn.py:151-157) and the output-node selection
 * (dagnn.py:119-126).  Consumes the layer ids that src/utils_dag.py:39-52 attaches offline.
 * ---------------------------------------------------------------------------------------- */

/* Host-side descriptor of a plan: a device workspace plus the sizes it was laid out for. */
typedef struct dagnn_plan {
    void* data;          /* device workspace, >= dagnn_plan_bytes(N, E, B, num_edge_feats) bytes */
    size_t bytes;
    int64_t N, E, B;     /* nodes, edges, graphs in the batch */
    int num_edge_feats;  /* floats of edge_attr per edge carried into the plan (0 = none) */
    int flags;           /* 0, or DAGNN_PLAN_GENERAL_BUILD */
} dagnn_plan;

/* Batches of up to 2048 nodes, 4096 edges and 512 graphs (the D-VAE batches of dvae/dagnn.py:99-175) have their plan and
 * their dataflow schedule built by ONE workgroup each instead of 7 + 6 launches - the same words either way.  This flag
 * keeps a plan on the general kernels whatever its size (tests compare the two). */
#define DAGNN_PLAN_GENERAL_BUILD 1
/* 1 when dagnn_plan_build takes the one-workgroup path for these sizes (dagnn_dataflow_schedule: E does not matter, pass 0) */
int dagnn_plan_is_small(int64_t N, int64_t E, int64_t B);

/* Bytes of device workspace `dagnn_plan_build` needs for N nodes, E edges, B graphs and
 * `num_edge_feats` floats of edge_attr per edge (0 if the model has no edge features). */
size_t dagnn_plan_bytes(int64_t N, int64_t E, int64_t B, int num_edge_feats);

/* Build the plan for both directions.
 *   edge_index [2,E] int64 row-major: row 0 = source, row 1 = target (PyG layout)
 *   layer_fwd  [N] int64: longest-path layer of each node        (G._bi_layer_idx0)
 *   layer_bwd  [N] int64: same on the reversed graph             (G._bi_layer_idx1)
 *   batch      [N] int64: graph id of each node, non-decreasing  (G.batch)
 *   edge_attr  [E,num_edge_feats] fp32 or NULL
 * Edges must be grouped by graph in the same order as nodes (what PyG collation produces).
 * `status` [4] int32 device words, written by the kernels: status[0] != 0 flags a contract
 * violation (bit 0: edges not grouped by graph, bit 1: edge crosses graphs, bit 2: batch not
 * sorted, bit 3: layer id >= nodes of its graph).  The caller clears status[0] before the call (the
 * kernels OR into it); a batch for which dagnn_plan_is_small() holds has it written, cleared or not.
 * Read it back only when debugging: the forward path itself never synchronises. */
int dagnn_plan_build(const dagnn_plan* plan /* host */,
                     const int64_t* edge_index, const int64_t* layer_fwd, const int64_t* layer_bwd,
                     const int64_t* batch, const float* edge_attr, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * AST node encoder (ogbg-code/utils.py:26-28, called at dagnn.py:139):
 *   out[v,:] = type_emb[x[v,0]] + attr_emb[x[v,1]] + depth_emb[min(depth[v], max_depth)]
 * and clamps depth[v] IN PLACE to max_depth, as the reference does (utils.py:27).
 *   x [N,2] int64, depth [N] int64, tables [*,H] fp32, out [N,ld_out] fp32 (ld_out >= H).
 * ---------------------------------------------------------------------------------------- */
int dagnn_encode_ast(const int64_t* x, int64_t* depth, const float* type_emb, const float* attr_emb,
                     const float* depth_emb, int max_depth, float* out, int ld_out,
                     int64_t N, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Everything in front of the recurrence as ONE fused pipeline of 7 launches (instead of 13 + the encoder's): the plan
 * (dagnn_plan_build: dagnn.py:146-157), the dataflow schedule (dagnn_dataflow_schedule, groups > 0) and - riding on the
 * plan's longest kernel, which is latency-bound - the HBM-bound row work of forward() that does not depend on the plan:
 * the node encoder (utils.py:26-28, dagnn.py:139), up to three table sets at once (the embedding itself and, for an
 * evaluation pass, the tables folded through W_ih of stacked layer 0 of every direction: gi0 = (T W_ih^T)[type] +
 * (A W_ih^T)[attr] + (D W_ih^T + b_ih)[depth]), and side effect 1 (dagnn.py:130-133: four [N] index arrays stacked).
 * Launches: pointers | per-graph sorts + rows | batch-level layers + work items + LPT assignment + workspace fill |
 * first slots + group-layer counts | row records (+ seal) + group prefixes | group-layer bases | schedule records.
 * Same plan / schedule words as the separate calls (tests compare them); small batches (dagnn_plan_is_small) and empty
 * ones take the separate calls internally.
 * ---------------------------------------------------------------------------------------- */
#define DAGNN_PREPARE_MAX_TABLES 3
typedef struct dagnn_prepare_rows {
    const int64_t* x;            /* [N,2] (type, attribute) indices; NULL: no encoder rows */
    int64_t* depth;              /* [N], clamped IN PLACE to max_depth (utils.py:27) */
    int max_depth;
    int num_tables;              /* 1..DAGNN_PREPARE_MAX_TABLES */
    struct {
        const float *type_emb, *attr_emb, *depth_emb;   /* [*, width] each */
        float* out;                                     /* [N, ld_out]: (type + attr) + depth rows */
        int width, ld_out;                              /* multiples of 4 */
    } table[DAGNN_PREPARE_MAX_TABLES];
    const int64_t* stack_src[4]; /* four [N] arrays ... */
    int64_t* stack_out;          /* ... copied to [4, N]; NULL: none */
} dagnn_prepare_rows;

int dagnn_prepare(const dagnn_plan* plan, const int64_t* edge_index, const int64_t* layer_fwd, const int64_t* layer_bwd,
                  const int64_t* batch, const float* edge_attr, int32_t* status,
                  void* schedule, size_t schedule_bytes, int groups, int cost_layer, int cost_row,   /* groups = 0: no schedule */
                  const dagnn_prepare_rows* rows /* or NULL */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input-side GRU GEMMs, batched over all nodes (the W_i* x + b_i* half of nn.GRUCell,
 * dagnn.py:181; independent of the recurrence, so done once per layer on the MFMA units):
 *   for g < num_groups:  C_g[M,Nc] = A_g[M,K] * W_g[Nc,K]^T + bias_g[Nc]
 * A row-major with leading dim lda, W row-major (torch weight layout) with leading dim ldw,
 * C row-major with leading dim ldc.  fp32 in, fp32 MFMA accumulate (bit-equal to an fmaf chain).
 * ---------------------------------------------------------------------------------------- */
typedef struct dagnn_gemm_group {
    const float* A;
    const float* W;
    const float* bias; /* may be NULL */
    float* C;
} dagnn_gemm_group;

int dagnn_gemm_nt_bias(const dagnn_gemm_group* groups /* host array */, int num_groups,
                       int64_t M, int Nc, int K, int lda, int ldw, int ldc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Recurrent weights are consumed k-major: Wt[k, c] = W_hh[c, k]  ([H, 3H] from torch's [3H, H]).
 * ---------------------------------------------------------------------------------------- */
int dagnn_pack_whh(const float* w_hh /* [3H,H] */, float* w_hh_t /* [H,3H] */, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * One stacked GRU layer of the recurrence, all topological layers, for the directions in
 * `dir_mask` (bit d).  One workgroup walks one (graph, direction): for every topological layer
 * t of that graph and every frontier node v
 *     a_v  = sum_e softmax_e( score[u_e] + gain . edge_attr_e ) * h[u_e]     (t > 0, else 0)
 *     h[v] = GRU gates( gi[v], W_hh a_v + b_hh, a_v )
 *     score[v] = w_key . h[v] (+ vid_bias[v mod vid_mod])
 * Replaces AttnConv.forward/message + PyG propagate/softmax/scatter (dagnn.py:362-373), the
 * hidden half of nn.GRUCell (dagnn.py:181) and the state write (dagnn.py:182); the query half
 * of attn_lin and its bias cancel inside the segment softmax.
 * The dvae NA variant's `vids` one-hot on the keys (dvae/dagnn.py:130-139) is the vid_bias term.
 *
 * Per direction d (arrays indexed by d; entries for directions not in dir_mask are ignored):
 *   gi[d]       [N,3H]  W_ih u + b_ih for every node (from dagnn_gemm_nt_bias)
 *   w_hh_t[d]   [H,3H]  packed by dagnn_pack_whh
 *   b_hh[d]     [3H]
 *   w_key[d]    [H]     key half of attn_lin.weight (last H entries; dagnn.py:359,370)
 *   edge_gain[d][num_edge_feats]  = W_e^T w_key (edge_encoder folded onto the key vector), or NULL
 *   vid_bias[d] [vid_mod] or NULL
 *   h[d]        [N,ld_h] out: this layer's hidden states (rows are written exactly once)
 *   score[d]    [N]     scratch
 * ---------------------------------------------------------------------------------------- */
typedef struct dagnn_layer_args {
    const float* gi[DAGNN_MAX_DIRS];
    const float* w_hh_t[DAGNN_MAX_DIRS];
    const float* b_hh[DAGNN_MAX_DIRS];
    const float* w_key[DAGNN_MAX_DIRS];
    const float* edge_gain[DAGNN_MAX_DIRS];
    const float* vid_bias[DAGNN_MAX_DIRS];
    float* h[DAGNN_MAX_DIRS];
    float* score[DAGNN_MAX_DIRS];
    int vid_mod;
    int ld_h;
    int static_score;   /* 1: score[d] is an INPUT (w_key . x_v for the `*_x` aggregators whose keys are the node
                         * inputs, dagnn.py:175-177) and is not rewritten; w_key may then be NULL */
    void* debug_timing; /* NULL, or 8 uint64 device words: per-phase 100 MHz ticks of the deepest work item */
} dagnn_layer_args;

int dagnn_recurrence_layer(const dagnn_plan* plan /* host */, const dagnn_layer_args* args /* host */, int dir_mask,
                           int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Lock-step schedule of the same recurrence (the default path): one launch per batch-level
 * topological layer covering every (direction, stacked layer) cell; stacked layer i runs one
 * launch behind layer i-1, so the whole loop nest of dagnn.py:144-182 becomes T + L - 1 dependent
 * launches.  Each workgroup owns a 32-unit slice of one cell's GRU weights for a block of <= 8
 * frontier rows (see dagnn_amd/csrc/frontier.hip).  Requires H % 64 == 0 (the host pads).
 *
 * Weights are consumed in slice/lane order: pack W [3H, K] (torch layout) with dagnn_pack_slices for
 * both slice widths the launches use (16 hidden units for thin launches, 32 for fat ones); K = H for
 * weight_hh and for weight_ih of stacked layers > 0.  H % 32 == 0, K % 64 == 0.
 *
 * State rows carry their attention scores: h_out is [N, ld_h] with ld_h >= H + H/16 (multiple of 4);
 * floats [H, H + H/16) of row v are the partial dots w_key[16q:16q+16] . h[v, 16q:16q+16], written
 * by the producing workgroups and summed in index order by the consumers (deterministic).
 * ---------------------------------------------------------------------------------------- */
int dagnn_pack_slices(const float* w /* [3H,K] */, float* out /* 3H*K floats */, int H, int K, int slice_units,
                      void* stream);

/* W [3H,K] -> B-fragment order of v_mfma_f32_16x16x4_f32 for 32-unit slices (six 16-column blocks per slice, 16 k per
 * 1-KiB wave load; 3H*K floats out; K % 16 == 0); used by the 64-row tiles of the fat launches (csrc/fat.hip). */
int dagnn_pack_mfma(const float* w /* [3H,K] */, float* out, int H, int K, void* stream);

/* Both slice layouts and the MFMA layout of several matrices in ONE launch (a training step re-packs every cell's
 * weights: 18 launches at L = 2 otherwise).  Outputs that are NULL are skipped. */
#define DAGNN_MAX_PACK_JOBS 16
typedef struct dagnn_pack_job {
    const float* w;      /* [3H,K] torch layout */
    float* out_slices16; /* 3H*K floats each, as dagnn_pack_slices(.., 16) / (.., 32) / dagnn_pack_mfma would write them */
    float* out_slices32;
    float* out_mfma;
    int32_t H, K;
} dagnn_pack_job;
int dagnn_pack_batch(const dagnn_pack_job* jobs /* host */, int num_jobs, void* stream);

typedef struct dagnn_frontier_cell {
    const float* w_hh_pk16; /* weight_hh packed for 16-unit slices */
    const float* w_hh_pk32; /* ... and for 32-unit slices */
    const float* w_ih_pk16; /* weight_ih (stacked layers > 0), else NULL */
    const float* w_ih_pk32;
    const float* w_hh_mfma; /* weight_hh in MFMA fragment order (dagnn_pack_mfma), or NULL: no MFMA tiles */
    const float* w_ih_mfma; /* weight_ih likewise (stacked layers > 0) */
    const float* b_hh;      /* [3H] */
    const float* b_ih;      /* [3H] (stacked layers > 0; layer 0 has it folded into gi0) */
    const float* w_key;     /* [H] key half of attn_lin.weight (NULL when static_score is given) */
    const float* static_score; /* NULL, or [N]: attention score of every node when the keys are the node inputs
                             * (`attn_x`, `self_attn_x`: w_key . x_v, constant over the recurrence) */
    const float* edge_gain; /* [num_edge_feats] or NULL */
    const float* vid_bias;  /* [vid_mod] or NULL */
    const float* gi0;       /* [N,3H] W_ih x + b_ih (stacked layer 0 only), from dagnn_gemm_nt_bias */
    float* h_out;           /* [N,ld_h] hidden states + partial scores (every row written exactly once) */
    void* granules;         /* NULL, or uint64 [N, H + H/16]: tagged copies {epoch, fp32 bits} of the same row,
                             * the hand-off format of the persistent tail kernel.  The buffer must have been
                             * zero-initialised once and only ever used with strictly increasing epochs. */
} dagnn_frontier_cell;

#define DAGNN_MAX_STACKED 8
typedef struct dagnn_frontier_args {
    dagnn_frontier_cell cell[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED];
    int num_stacked; /* L */
    int dir_mask;
    int H, ld_h, vid_mod;
    int num_cus;     /* compute units of the device (launch geometry heuristic), e.g. 256 */
    int rb4_max_wgs; /* streamed 32-unit slices: launches of up to this many 4-row-block workgroups use 4-row blocks,
                      * bigger ones 8-row blocks; 0 = default (1.5 per CU) */
    int mfma_min_rows;    /* launches with at least this many rows (all cells of both directions; a one-direction chain takes its
                           * share) run as 64-row MFMA tiles with the gather fused (csrc/fat.hip); 0 = never */
    void* agg_scratch;    /* NULL, or fp32 [agg_scratch_rows, H]: scratch rows for the aggregates of rows with more than two
                           * predecessors (the fat launches build them in their prologue; nothing else goes through memory) */
    int agg_scratch_rows; /* >= (largest number of rows of direction 0's cells in one launch) + (the same for direction 1): the two
                           * directions' launches may run concurrently (below) and keep their scratch rows apart */
    /* persistent tail: one dataflow launch for all layers after the fat head (needs every cell's
     * `granules`, epoch != 0 and H <= 256): */
    int tail_replicas;   /* workgroups per (cell, slice) in the tail kernel; 0 disables it */
    int tail_slice_units; /* 16 or 32 hidden units per tail workgroup (32 if anything else) */
    int tail_max_blocks; /* a layer may have up to 4 * tail_replicas * tail_max_blocks rows per cell in the tail */
    unsigned epoch;      /* tag of this forward pass in the granule buffers: nonzero, larger than any used before */
    void* tail_err;      /* device int32: set to 1 if a bounded wait in the tail kernel ever expires */
    void* debug_timing; /* NULL, or (T+L-1)*8 uint64 device words: 100 MHz stamps of workgroup 0 per launch */
    /* split mode (optional): with a second stream and the plan's per-layer split pointers (blsplit_d, read back with
     * the schedule) the persistent kernel walks the DEEP graphs (depth > thr_d) from layer 0 on `side_stream`,
     * concurrently with the per-layer launches, which then only cover the shallow graphs.  The call forks from and
     * joins back into `stream` with the caller's `fork_event` / `join_event` (capturable); results agree with the unsplit mode to rounding (a row may be
     * handled by a different kernel), each mode by itself is deterministic.  (CU-masked streams for the two
     * halves were measured and dropped: masked queues slowed every other launch of the process.) */
    void* side_stream;                              /* hipStream_t or NULL: runs the persistent kernel.  Without a persistent tail in
                                                     * the call (H > 256, or tail_replicas = 0) and with both directions, the same
                                                     * stream and events run direction 1's per-layer launches as a second chain next to
                                                     * direction 0's on `stream` (the directions share nothing; each chain's launches
                                                     * fill the CUs the other's leave idle while their last workgroups drain) */
    const int32_t* layer_split[DAGNN_MAX_DIRS];     /* HOST, num_layers[d] int32: first deep slot of every layer, or NULL */
    void* fork_event;                               /* hipEvent_t x 2, CALLER-OWNED (the library creates nothing): split */
    void* join_event;                               /* mode needs both, otherwise it is off */
} dagnn_frontier_args;

/* layer_ptr[d] (HOST, num_layers[d] + 1 int32): row offsets of the batch-level layers of direction
 * d, i.e. the first num_layers[d]+1 words of the plan's blptr_d array read back by the caller
 * (the one device->host read of the forward pass; the reference reads T back at dagnn.py:137). */
int dagnn_frontier_run(const dagnn_plan* plan /* host */, const dagnn_frontier_args* args /* host */,
                       const int32_t* const* layer_ptr /* host */, const int32_t* num_layers /* host [2] */,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Dataflow schedule of the same recurrence (the default path for H <= 256): ONE persistent launch for the whole loop
 * nest of dagnn.py:144-182, cut along graphs instead of layers (dagnn_amd/csrc/dataflow.hip).
 *
 *   1. dagnn_dataflow_groups: how many independent groups G the device hosts.  A workgroup SET has one workgroup
 *      (one CU) per (kernel cell, 32-unit slice); the kernel cells of a direction are its L GRU cells plus, for every
 *      stacked layer above the first, one PROJECTION cell (the input-side product W_ih u + b_ih, which leaves the
 *      dependent chain that way).  A set serves TWO groups (their blocks interleave: one group's dependent hop hides
 *      behind the other's work): G = 2 * floor(num_cus / (num_dirs * (2L - 1) * H/32)), capped at 64 and at B;
 *      0 = this shape is not supported (H > 256, H % 64 != 0, more than 16 kernel cells, or one set does not fit the
 *      device) - use dagnn_frontier_run.  Any 1 <= G <= 64 is valid for the calls below (ceil(G / 2) sets launch).
 *   2. dagnn_dataflow_schedule: deals the graphs of a plan to the G groups - longest-processing-time first on
 *      cost_layer * depth + cost_row * nodes, integer arithmetic, ties to the lowest group - and re-sorts the plan's
 *      64-byte row records by (group, topological layer, graph, node), every group-layer padded to whole blocks of 4
 *      records (padding records: node = -1).  The result depends on the plan and on (G, costs) only: build it once per
 *      batch (it also serves the backward pass).  Workspace: dagnn_dataflow_bytes(N, B, G), any contents; the record
 *      arrays are defined over the records the groups use (first record .. first record + 4 * blocks of every group).
 *   3. dagnn_dataflow_run: the launch.  Every (direction, stacked layer) cell needs `granules`: uint64 [N, gld]
 *      (gld >= H) tagged copies {epoch, fp32 bits} of its state rows - the hand-off format between workgroups.  The
 *      buffers must have been zero-initialised once and only ever used with strictly increasing `epoch`s (a replayed
 *      hipGraph must contain the memset).  `err` (device int32, zeroed by the caller) is set when a bounded wait
 *      expires (results are then invalid): bit 0 a granule poll, bit 1 an LDS flag; bit 2: the plan's status word
 *      (`plan_status`) was nonzero, nothing was computed - bits 8-15 then carry that status word; bit 3: `schedule` does not carry the header of a schedule
 *      built for `groups` groups (dagnn_dataflow_schedule), nothing was computed.
 *      State rows h_out [N, ld_h] receive the H states only; dagnn_score_parts adds the H/16 partial attention scores
 *      behind them (the format dagnn_backward_prepare reads) when a backward pass follows.
 * Weights: dagnn_pack_dataflow(W [3H,H] torch layout) -> 3*H*H floats in slice / lane order (the A operands of the
 * kernel's v_mfma_f32_4x4x1 products: lane = (unit quad, K slice of H/8, unit of the quad)).
 * ---------------------------------------------------------------------------------------- */
typedef struct dagnn_dataflow_cell {
    const float* w_hh;      /* weight_hh packed by dagnn_pack_dataflow */
    const float* w_ih;      /* weight_ih packed likewise (stacked layers > 0), else NULL */
    const float* b_hh;      /* [3H] */
    const float* b_ih;      /* [3H] (stacked layers > 0; layer 0 has it folded into gi0) */
    const float* w_key;     /* [H] key half of attn_lin.weight (ignored when static_score is given) */
    const float* static_score; /* NULL, or [N]: see dagnn_frontier_cell */
    const float* edge_gain; /* [num_edge_feats] or NULL */
    const float* vid_bias;  /* [vid_mod] or NULL */
    const float* gi0;       /* [N,3H] W_ih x + b_ih (stacked layer 0 only) */
    float* h_out;           /* [N,ld_h] */
    void* granules;         /* uint64 [N,gld] */
    void* proj_granules;    /* stacked layers > 0: 16-byte granules {tag, r, z, n} [N,pld] (pld >= H, 16-byte aligned), the hand-off
                             * buffer of the cell's input-side pre-activations W_ih u + b_ih, one granule per unit (same
                             * zero-init / epoch contract as `granules`); else NULL */
    float* gh_out;          /* NULL, or [N,3H]: the hidden-side pre-activations W_hh a + b_hh of every node as the gates saw them */
    float* gi_out;          /* NULL, or [N,3H] (stacked layers > 0): the input-side pre-activations as plain floats - what a
                             * training pass keeps for its reverse sweep instead of recomputing both with GEMMs */
    /* plain aggregators (AggConv `add` / `max`, ogbg-code/model/dagnn.py:232-251: messages h_j + edge_encoder(edge_attr_j)):
     * agg = DAGNN_DF_AGG_ATTN (0, the soft-max above), _ADD, _MAX, or _NONE (no message lands on these rows - the reference's one
     * shared AggConv in the reverse direction - the aggregate is zero).  agg_edge_w [H, num_edge_feats <= 2] / agg_edge_b [H]:
     * the edge encoder in torch layout, or NULL (no edge features).  H <= 256; w_key / edge_gain / static_score are ignored. */
    const float* agg_edge_w;
    const float* agg_edge_b;
    int agg;
} dagnn_dataflow_cell;
#define DAGNN_DF_AGG_ATTN 0
#define DAGNN_DF_AGG_ADD 1
#define DAGNN_DF_AGG_MAX 2
#define DAGNN_DF_AGG_NONE 3

typedef struct dagnn_dataflow_args {
    dagnn_dataflow_cell cell[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED];
    int num_stacked, dir_mask;
    int H, ld_h, gld, pld, vid_mod;
    int groups;            /* G the schedule was built for */
    unsigned epoch;
    const void* schedule;  /* device workspace written by dagnn_dataflow_schedule */
    void* err;             /* device int32 */
    void* debug_timing;    /* NULL, or uint64 device words (100 MHz stamps): [workgroups][2] start / end of every
                            * workgroup, then [blocks][8] phase stamps of workgroup `debug_wg` */
    unsigned spin_limit;   /* polls before a wait gives up and raises `err`; 0 = default (1 << 22, seconds) */
    int debug_wg;          /* workgroup whose blocks are stamped (debug_timing) */
    /* XCD-aware placement (optional, a speed hint that is verified at run time): with `num_cus` (CUs of the device, one
     * workgroup per CU) and `xcc_table` (uint64 [num_cus], zero-initialised once, same epoch contract as `granules`) the
     * workgroups of a recurrent cell and of the projection cell reading its rows are given ids that the observed dispatch
     * rule (workgroup b -> XCD b % 8) puts on one XCD.  Every workgroup publishes the XCD it really runs on; a cell whose
     * readers all share its XCD hands its rows over through that XCD's L2 (plain stores) instead of write-through ones -
     * results are identical either way.  0 / NULL: linear ids, write-through stores. */
    int num_cus;
    void* xcc_table;
    const void* plan_status; /* NULL, or the device status word of dagnn_plan_build: if it is nonzero (the batch violates
                            * the layout contract) the launch walks nothing and raises bit 2 of `err` */
    int xcd_first;          /* XCD-aware placement: the XCD the packing starts with (0..7).  Launches that run next to each other
                            * (micro-batches in flight on several streams, each sized for a part of the device) pass different
                            * values, otherwise all of them pack their workgroups onto the same first XCDs and take turns */
    int stat_rows;          /* training passes whose reverse pass is dagnn_bwd_dataflow_run: nonzero = every cell's `gh_out` is that
                            * cell's static-record buffer `stat` (dagnn_bwd_dataflow_static_bytes_h; zero-filled by the caller when
                            * H is neither 256 nor 320) and the kernel writes the state and the six gate-coefficient rows of each
                            * node's record instead of the pre-activations; `gi_out` must be NULL.  The reverse pass then only
                            * adds the external-gradient row (dagnn_bwd_dataflow_args.stat_rows_written) */
    int slices64;           /* nonzero and H = 256 / 320: the workgroup shape of csrc/dataflow_x.hip - 64 hidden units, 8 compute
                            * waves and one stream per workgroup, `groups` workgroup sets of (kernel cells x H / 64) workgroups
                            * (the caller guarantees they fit the device: groups <= 2 * floor(CUs / (cells x H / 32)), what
                            * dagnn_dataflow_groups returns).  Same schedule, same results */
} dagnn_dataflow_args;

int dagnn_dataflow_groups(int num_cus, int num_dirs, int num_stacked, int H, int64_t B);
size_t dagnn_dataflow_bytes(int64_t N, int64_t B, int groups);
int dagnn_dataflow_schedule(const dagnn_plan* plan /* host */, void* workspace, size_t workspace_bytes, int groups,
                            int cost_layer, int cost_row, const int32_t* plan_status /* device, or NULL */, void* stream);
int dagnn_dataflow_run(const dagnn_plan* plan /* host */, const dagnn_dataflow_args* args /* host */, void* stream);
/* H = 320 (hidden sizes 257..320, zero-padded by the caller: the reference trains at emb_dim = 300, scripts/ogb_tok.sh:17): the
 * same launch in an 8-wave workgroup shape (csrc/dataflow_w.hip).  Exactly two edge features, no static scores, no vertex-id
 * key biases (anything else: DAGNN_EINVAL - the caller keeps such models on the other paths).  dagnn_dataflow_run forwards
 * H > 256 here; dagnn_dataflow_groups / dagnn_pack_dataflow accept H = 320. */
int dagnn_dataflow_run_wide(const dagnn_plan* plan /* host */, const dagnn_dataflow_args* args /* host */, void* stream);
int dagnn_dataflow_run_x(const dagnn_plan* plan /* host */, const dagnn_dataflow_args* args /* host */, void* stream);   /* slices64 */
int dagnn_pack_dataflow(const float* w /* [3H,H] */, float* out /* 3H*H floats */, int H, void* stream);
/* the same order of the gate-wise transposed matrix W'[g H + j][u] = W[g H + u][j] (reverse sweep, dagnn_bwd_dataflow_run) */
int dagnn_pack_dataflow_transposed(const float* w /* [3H,H] */, float* out /* 3H*H floats */, int H, void* stream);
/* Both layouts of several [3H, H] matrices in ONE launch (a training step re-packs every cell's weights). */
typedef struct dagnn_df_pack_job {
    const float* w;       /* [3H, H] torch layout (mode 2: edge_encoder.weight [rows, cols]) */
    float* out;           /* 3H*H floats (mode 2: [cols]) */
    const float* aux;     /* mode 2: the key weights [rows]; else unused */
    int32_t transposed;   /* 0: as dagnn_pack_dataflow, 1: as dagnn_pack_dataflow_transposed, 2: edge gain out = w^T aux */
    int32_t rows, cols;   /* mode 2 only */
} dagnn_df_pack_job;
int dagnn_pack_dataflow_batch(const dagnn_df_pack_job* jobs /* host */, int njob, int H, void* stream);
int dagnn_score_parts(float* h /* [N,ld_h] */, int ld_h, int H, const float* w_key /* [H] */, int64_t N, void* stream);
/* ... of several cells (the same ld_h / H / N) in one launch: n <= DAGNN_MAX_PACK_JOBS host arrays of device pointers. */
int dagnn_score_parts_batch(float* const* h, const float* const* w_key, int n, int ld_h, int H, int64_t N, void* stream);
/* Introspection (tests, host-side mirror): byte offsets of the schedule workspace's arrays, 13 entries: [grp_of,
 * gdepth, gload, loff, gtab0, gtab1, lcnt0, lcnt1, glbase0, glbase1, grec0, grec1, total]. */
int dagnn_dataflow_layout(int64_t N, int64_t B, int groups, int64_t* offsets13 /* host */);

/* ------------------------------------------------------------------------------------------
 * Weight-stationary tile schedule of the same recurrence for WIDE hidden states (H = 512; dagnn_amd/csrc/tiles.hip): the
 * path of BASELINE.json's cfg 5 (batch 256, hidden 512, 5 stacked layers, bidirectional), where the GRU matrices (56.6 MB)
 * are too large to be streamed once per topological layer (dagnn_frontier_run) and the rows too many for the 4-row blocks
 * of dagnn_dataflow_run.  One persistent launch per CHUNK of stacked layers (all of them at once when 32 workgroups per cell
 * fit the device - up to 4 layers x 2 directions on 256 CUs; else chunk 0 = stacked layer 0, then as many layers at a time as fit): every workgroup keeps its 16-unit slice of one cell's
 * W_ih | W_hh (torch layouts, read once) in registers and walks the plan's batch-level layers in tiles of 16 rows
 * (v_mfma_f32_16x16x4_f32); rows are handed between workgroups by write-through stores + {epoch, tiles done} progress
 * counters.  Replaces the loop nest of dagnn.py:144-182 like the two schedules above; needs no device->host read.
 *   h_out [N, ld_h], ld_h >= H + H/16: states + the H/16 partial attention scores behind them (as dagnn_frontier_run);
 *   counters: uint64 [2 * DAGNN_MAX_STACKED * 8 * 32], zero-initialised once, only ever used with strictly increasing
 *             `epoch`s (the granule contract); err: device int32, zeroed by the caller - bit 0 / 1 a bounded wait expired,
 *             bit 2 (+ bits 8-15) the plan's status word was set and nothing was computed.
 * dagnn_tiles_launches: number of launches dagnn_tiles_run makes for this shape, 0 = shape not supported (H != 512, more
 * than 2 edge features, fewer than 32 * num_dirs CUs): use dagnn_frontier_run.  Static scores (the `*_x` aggregators)
 * are not supported either; the vertex-id key bias of the NA encoder (vid_bias / vid_mod) is.
 * ---------------------------------------------------------------------------------------- */
typedef struct dagnn_tiles_cell {
    const float* w_hh;      /* [3H,H] torch layout */
    const float* w_ih;      /* [3H,H] torch layout (stacked layers > 0), else NULL */
    const float* b_hh;      /* [3H] */
    const float* b_ih;      /* [3H] (stacked layers > 0; layer 0 has it folded into gi0) */
    const float* w_key;     /* [H] key half of attn_lin.weight */
    const float* edge_gain; /* [num_edge_feats] or NULL */
    const float* gi0;       /* [N,3H] W_ih x + b_ih (stacked layer 0 only), from dagnn_gemm_nt_bias */
    float* h_out;           /* [N,ld_h] */
    const float* vid_bias;  /* [vid_mod] or NULL: score bias by vertex id (NA variant), as in dagnn_frontier_cell */
} dagnn_tiles_cell;

typedef struct dagnn_tiles_args {
    dagnn_tiles_cell cell[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED];
    int num_stacked, dir_mask;
    int H, ld_h;
    int num_cus;             /* compute units of the device: every workgroup of a launch must be resident */
    unsigned epoch;
    void* counters;
    void* err;
    unsigned spin_limit;     /* polls before a wait gives up and raises `err`; 0 = default (1 << 22) */
    const void* plan_status; /* NULL, or the device status word of dagnn_plan_build */
    int first_layer[DAGNN_MAX_DIRS]; /* per direction: walk the batch-level layers from this one on (0 = all).  The layers
                              * before it must be complete in every h_out (e.g. by dagnn_frontier_run called with num_layers =
                              * first_layer): the wide first layers on per-layer launches, the long thin tail on this kernel */
    void* debug_timing;      /* NULL, or uint64 [launches][1024][32] device words: per-workgroup phase sums in 100 MHz ticks,
                              * written by a -DT_STAMPS build only (scripts/tiles_stamps.py) */
    int vid_mod;             /* 0, or the node count per graph of the NA variant (node id % vid_mod selects vid_bias) */
} dagnn_tiles_args;

int dagnn_tiles_launches(int num_cus, int num_dirs, int num_stacked, int H, int num_edge_feats);
int dagnn_tiles_run(const dagnn_plan* plan /* host */, const dagnn_tiles_args* args /* host */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Read-out over output nodes (dagnn.py:119-126,184-193 with out_pool='max', out_pool_all=0):
 *   out[g, col_off[d] + j] = max over { v in graph g : layer_{1-d}(v) == 0 } of h[d][v, j]
 * d = 0 pools the sinks, d = 1 the sources.  h[d] is [N,ld_h]; `width` columns are pooled
 * (pass a [N, L*H] concatenation or call once per layer with a column offset).
 * ---------------------------------------------------------------------------------------- */
int dagnn_readout_max(const dagnn_plan* plan /* host */, const float* h, int ld_h, int width, int dir,
                      float* out, int ld_out, int col_off, void* stream);

/* The same for several state buffers in ONE launch (the bidirectional L-layer read-out of dagnn.py:184-193 is 2 L calls
 * of the above: launch-bound).  jobs: host array, n <= 16. */
typedef struct dagnn_readout_job {
    const float* h;   /* [N, ld_h] */
    int ld_h, width, dir, col_off;
} dagnn_readout_job;
int dagnn_readout_max_batch(const dagnn_plan* plan /* host */, const dagnn_readout_job* jobs /* host */, int n,
                            float* out, int ld_out, void* stream);

/* The other read-outs of dagnn.py:194-202 (`global_max_pool` / `global_mean_pool` / `global_add_pool`; `P_ATTN`,
 * dagnn.py:114-117, is a softmax over a size-1 dimension and therefore add-pooling): pool `width` columns of
 * h [N,ld_h] per graph over scope 0 / 1 = the output nodes of direction 0 / 1 (as dagnn_readout_max) or scope 2 = all
 * nodes of the graph (`out_pool_all=1`).  Nodes are visited in id order (a fixed summation order); a graph without
 * nodes in scope reads 0, the mean divides by max(count, 1). */
enum { DAGNN_POOL_MAX = 0, DAGNN_POOL_ADD = 1, DAGNN_POOL_MEAN = 2 };
int dagnn_readout_pool(const dagnn_plan* plan /* host */, const float* h, int ld_h, int width, int scope, int mode,
                       float* out, int ld_out, int col_off, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward pass of the recurrence (training: what `loss.backward()`, ogbg-code/main_pyg.py:62, does to
 * dagnn.py:144-182).  Additive attention with keys from the hidden states (`attn_h`), GRU cells.
 * The forward pass keeps only the lock-step state buffers h[d][i] ([N,ld_h], scores behind the rows).
 *
 *   1. dagnn_backward_prepare: a[c] [N,H] (attention aggregates) and alpha[c] [E] (attention weights,
 *      indexed by ORIGINAL edge id) of every cell, one parallel launch;
 *   2. caller: gh[c] = a[c] W_hh^T + b_hh and gi[c] = u W_ih^T + b_ih (dagnn_gemm_nt_bias), g_ext[c] [N,H]
 *      = gradient reaching h[c] from outside the recurrence (read-out; dagnn_readout_max_backward);
 *   3. dagnn_backward_run: T + L - 1 reverse lock-step launches.  Adds W_ih^T dgi of stacked layer i into
 *      g_ext of layer i-1 and writes, per cell: da [N,H], dgi, dgh [N,3H] (gradients of the GRU
 *      pre-activations, gate blocks r,z,n), sigma [N] (sum over a node's out-edges of the attention-logit
 *      gradients) and edge_feat_grad [N,R] (the same sum weighted by the edge features);
 *   4. caller (plain library GEMMs / reductions): dW_ih = dgi^T u, db_ih = colsum(dgi), dW_hh = dgh^T a,
 *      db_hh = colsum(dgh), dx = sum_d dgi[d][0] W_ih, d w_key = h^T sigma + W_e colsum(edge_feat_grad),
 *      dW_e = w_key (x) colsum(edge_feat_grad).  The attention query weights and biases get exact zeros
 *      (they cancel inside the segment softmax).
 * Deterministic: gradients are pulled (no atomics), sums run in a fixed order.  H % 64 == 0, H <= 1024.
 * ---------------------------------------------------------------------------------------- */
typedef struct dagnn_backward_cell {
    const float* w_hh;      /* [3H,H] torch layout (padded to H) */
    const float* w_ih;      /* [3H,H] (stacked layers > 0), else NULL */
    const float* w_key;     /* [H] */
    const float* edge_gain; /* [num_edge_feats] or NULL */
    const float* vid_bias;  /* [vid_mod] or NULL: score bias by vertex id (NA variant), as in dagnn_frontier_cell */
    const float* static_score; /* NULL, or [N]: the nodes' attention scores when the keys are the inputs (`attn_x`,
                             * `self_attn_x`); pass an all-zero w_key then (the scores do not depend on the states) */
    const float* h;         /* [N,ld_h] forward states + partial scores */
    float* a;               /* [N,H]  written by prepare, read by run */
    float* alpha;           /* [E]    written by prepare, read by run */
    const float* gi;        /* [N,3H] */
    const float* gh;        /* [N,3H] */
    float* g_ext;           /* [N,H] in; stacked layers below the top also receive the upper layer's du */
    float* da;              /* [N,H] out */
    float* dgi;             /* [N,3H] out */
    float* dgh;             /* [N,3H] out */
    float* sigma;           /* [N] out */
    float* edge_feat_grad;  /* [N,num_edge_feats] out, or NULL without edge features */
    /* persistent sweep (optional; all NULL = one launch per layer throughout): */
    void* da_granules;      /* uint64 [N,H]: tagged copies {epoch, fp32 bits} of da rows (zero-initialised once,
                             * strictly increasing epochs - the same contract as dagnn_frontier_cell.granules) */
    void* du_granules;      /* uint64 [N,H]: the upper stacked layer's du for THIS cell's rows (stacked layers below
                             * the top) */
    const float* g_ext_static; /* [N,H] copy of g_ext taken before the sweep (stacked layers below the top): inside
                             * the persistent launch a row's gradient is g_ext_static + its du granules */
} dagnn_backward_cell;

typedef struct dagnn_backward_args {
    dagnn_backward_cell cell[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED];
    int num_stacked, dir_mask, H, ld_h;
    int vid_mod;    /* 0, or the node count per graph of the NA variant (node id % vid_mod selects vid_bias) */
    int num_cus;
    int thin_wgs;   /* launches of up to this many slice workgroups use the register-resident slice kernel, bigger ones
                     * the rows + MFMA-tile kernels; 0 = default (2 * num_cus) */
    /* persistent sweep over the thin head of the reverse order (needs the cells' granule buffers, H <= 256): */
    int tail_replicas;   /* workgroups per (cell, 16-unit slice); 0 disables it */
    int tail_max_blocks; /* a layer may have up to 4 * tail_replicas * tail_max_blocks rows per cell in it */
    unsigned epoch;      /* tag of this backward pass in the granule buffers: nonzero, larger than any used before */
    void* tail_err;      /* device int32: set to 1 if a bounded wait ever expires (results are then invalid) */
    /* split mode (optional): with a second stream and the plan's per-layer split pointers the sweep runs as two
     * independent chains - the shallow graphs' per-layer launches on `side_stream`, the deep graphs (persistent head,
     * then per-layer launches) on `stream`; forks from and joins back into `stream` with the caller's two events */
    void* side_stream;
    const int32_t* layer_split[DAGNN_MAX_DIRS];   /* HOST, num_layers[d] int32, or NULL */
    void* fork_event;                             /* hipEvent_t x 2, CALLER-OWNED; split mode needs both */
    void* join_event;
} dagnn_backward_args;

int dagnn_backward_prepare(const dagnn_plan* plan /* host */, const dagnn_backward_args* args /* host */, void* stream);
int dagnn_backward_run(const dagnn_plan* plan /* host */, const dagnn_backward_args* args /* host */,
                       const int32_t* const* layer_ptr /* host */, const int32_t* num_layers /* host */, void* stream);

/* ------------------------------------------------------------------------------------------
 * The same reverse sweep as ONE persistent dataflow launch (csrc/bwd_dataflow.hip; H <= 256): the mirror image of
 * dagnn_dataflow_run on the same schedule workspace, replacing dagnn_backward_run's T + L - 1 launches.
 *   1. dagnn_backward_prepare (a, alpha) and the two pre-activation GEMMs (gi, gh) as before (no gi / gh when the forward
 *      launch wrote the static rows itself: stat_rows / stat_rows_written);
 *   2. dagnn_bwd_dataflow_prepare: successor records in schedule order (`records`, dagnn_bwd_dataflow_record_bytes: 256 bytes
 *      per record - the node, its successor row, the first four successors with their edge features and the attention weight
 *      of every stacked layer, so `alpha` of step 1 must be final) and the per-(cell, node) static rows `stat` (dagnn_bwd_dataflow_static_bytes each): Gext, h and the GRU-backward
 *      coefficients - the gate algebra is linear in the incoming gradient, so it leaves the dependent chain;
 *   3. dagnn_bwd_dataflow_run: the launch.  Hand-off buffers (uint64 granules {epoch, fp32 bits}, zero-initialised once,
 *      strictly increasing epochs): da [N,gld], q [N], dgi [N,3 gld] (stacked layers > 0), du [N,gld] (stacked layers
 *      below the top: the du arriving at THIS cell's rows).  Outputs for the weight-gradient epilogue: dgi, dgh [N,3H],
 *      sigma [N], edge_feat_grad [N,R].  `err` as for dagnn_dataflow_run.
 * Weights: dagnn_pack_dataflow_transposed(W): the forward kernel's slice / lane order of the gate-wise transposed matrix
 * W'[g H + j][u] = W[g H + u][j] (H x H blocks of the torch layout transposed in place). */
typedef struct dagnn_bwd_dataflow_cell {
    const float* w_hh_t;    /* packed gate-wise transposed W_hh */
    const float* w_ih_t;    /* ... W_ih (stacked layers > 0), else NULL */
    const float* w_key;     /* [H] (zeros when the scores are static) */
    const float* alpha;     /* [E] (dagnn_backward_prepare); read by dagnn_bwd_dataflow_prepare (into the successor records) and run */
    const float* gi;        /* prepare: [N,3H] */
    const float* gh;        /* prepare: [N,3H] */
    const float* a;         /* prepare: [N,H] */
    const float* b_hh;      /* prepare: [3H] */
    const float* h;         /* prepare: [N,ld_h] forward states */
    const float* g_ext;     /* prepare: [N,ld_g] gradient reaching h from outside the recurrence */
    float* stat;            /* [N, 8 * 256] static rows: written by prepare (or, all but the g_ext row, by the forward launch:
                             * stat_rows_written), read by run */
    void* da_granules;      /* uint64 [N,gld] */
    void* q_granules;       /* uint64 [N] */
    void* dgi_granules;     /* uint64 [N,3 gld] (stacked layers > 0) */
    void* du_granules;      /* uint64 [N,gld] (stacked layers below the top) */
    float* dgi;             /* out [N,3H] */
    float* dgh;             /* out [N,3H] */
    float* sigma;           /* out [N] */
    float* edge_feat_grad;  /* out [N,R] or NULL */
} dagnn_bwd_dataflow_cell;

typedef struct dagnn_bwd_dataflow_args {
    dagnn_bwd_dataflow_cell cell[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED];
    int num_stacked, dir_mask, H, ld_h, ld_g, gld, groups;
    unsigned epoch, spin_limit;
    const void* schedule;     /* dagnn_dataflow_schedule workspace for `groups` groups */
    void* records;            /* dagnn_bwd_dataflow_record_bytes(N) */
    void* err;                /* device int32 */
    const void* plan_status;  /* or NULL */
    int num_cus;              /* XCD-aware placement, as in dagnn_dataflow_args (0 / NULL: off) */
    void* xcc_table;
    int xcd_first;            /* as in dagnn_dataflow_args */
    int stat_rows_written;    /* nonzero: the forward launch of this step wrote rows 1..7 of every `stat` record
                               * (dagnn_dataflow_args.stat_rows); prepare only scatters g_ext into row 0 and gi / gh / a / b_hh / h
                               * of the cells are not read */
} dagnn_bwd_dataflow_args;

size_t dagnn_bwd_dataflow_record_bytes(int64_t N);
size_t dagnn_bwd_dataflow_static_bytes(int64_t N);
size_t dagnn_bwd_dataflow_static_bytes_h(int64_t N, int H);   /* ... for a given H: 8 KB per (cell, node) up to H = 256, 10 KB at H = 320 */
int dagnn_bwd_dataflow_prepare(const dagnn_plan* plan /* host */, const dagnn_bwd_dataflow_args* args /* host */, void* stream);
int dagnn_bwd_dataflow_run(const dagnn_plan* plan /* host */, const dagnn_bwd_dataflow_args* args /* host */, void* stream);
/* H = 320: the same sweep with five column blocks per lane (csrc/bwd_dataflow_w.hip; 12 waves like H <= 256); dagnn_bwd_dataflow_run forwards H > 256 here */
int dagnn_bwd_dataflow_run_wide(const dagnn_plan* plan /* host */, const dagnn_bwd_dataflow_args* args /* host */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Parameter gradients of the GRU cells from the sweep's outputs (csrc/wgrad.hip): for every job
 *     d_weight [3H, in_dim] = dg[:, real rows]^T in,      d_bias [3H] = column sums of dg          (both overwritten)
 * with dg [N, ld_dg] = dgi or dgh (three gate blocks of Hp columns, Hp >= H: padding units are dropped) and `in` [N, ld_in]
 * the product's input (u for W_ih, the aggregate a for W_hh).  The reduction over the nodes is split `splits` ways
 * (dagnn_wgrad_splits) into partial tiles in `workspace` (dagnn_wgrad_workspace_bytes) that a second kernel adds in
 * split order: deterministic, no atomics.  Replaces what autograd accumulates per `nn.GRUCell` call (dagnn.py:181). */
typedef struct dagnn_wgrad_job {
    const float* dg;     /* [N, ld_dg] */
    const float* in;     /* [N, ld_in] */
    float* d_weight;     /* [3H, in_dim] */
    float* d_bias;       /* [3H] or NULL */
    int ld_dg, ld_in, in_dim;   /* in_dim even; ld_dg % 4 == 0, ld_in % 2 == 0, rows 8-byte aligned */
} dagnn_wgrad_job;
/* Weighted column sums for the rest of the epilogue: out[k] = sum_n weight[n] x[n][k] (weight NULL: plain column sums) -
 * the attention-key gradients sum_v sigma_v keys_v, the edge-feature sums, sum_v sigma_v.  Rows are summed in a fixed
 * chunk order (deterministic). */
typedef struct dagnn_colsum_job {
    const float* x;        /* [N, ld_x] */
    const float* weight;   /* [N] or NULL */
    float* out;            /* [cols] */
    int ld_x, cols;
} dagnn_colsum_job;
size_t dagnn_colsum_workspace_bytes(int njob, int max_cols);
int dagnn_colsum_run(const dagnn_colsum_job* jobs /* host */, int njob, int64_t N, void* workspace, size_t workspace_bytes,
                     void* stream);
size_t dagnn_wgrad_workspace_bytes(int njob, int Hp, int max_in_dim, int splits);
int dagnn_wgrad_splits(int num_cus, int njob, int Hp, int max_in_dim, int64_t N);
int dagnn_wgrad_run(const dagnn_wgrad_job* jobs /* host */, int njob, int64_t N, int Hp, int H, int splits, void* workspace,
                    size_t workspace_bytes, void* stream);

/* The small gradients of one cell's attention / edge encoder from the three column sums above, every cell in ONE launch
 * (replaces ~10 torch ops per cell behind `loss.backward()`, dagnn.py:362-373 under autograd):
 *   g_attn[j]      = 0 outside [dq, dq + kd);  g_attn[dq + k] = key_sum[k] + edge_w[k, :] . feat_sum + edge_b[k] * sigma_sum
 *   g_edge_w[k, r] = attn_w[dq + k] * feat_sum[r];   g_edge_b[k] = attn_w[dq + k] * sigma_sum        (edge_w != NULL only) */
#define DAGNN_ATTN_GRAD_MAX_JOBS 16
typedef struct dagnn_attn_grad_job {
    const float* key_sum;    /* [kd]  sum_v sigma_v keys_v */
    const float* feat_sum;   /* [R]   sum_v (edge-feature sums)_v, or NULL */
    const float* sigma_sum;  /* [1]   sum_v sigma_v, or NULL */
    const float* edge_w;     /* [kd, R] edge_encoder.weight, or NULL: no edge encoder */
    const float* edge_b;     /* [kd] */
    const float* attn_w;     /* [attn_len] attn_lin.weight row */
    float* g_attn;           /* [attn_len] */
    float* g_edge_w;         /* [kd, R] */
    float* g_edge_b;         /* [kd] */
    int32_t dq, kd, attn_len, R;
} dagnn_attn_grad_job;
int dagnn_attn_grads_run(const dagnn_attn_grad_job* jobs /* host */, int njob, void* stream);

/* grad_h[v, j] += grad_out[g, col_off + j] for the first output node v of graph g attaining the maximum
 * of column j (the single winner of scatter-max); grad_h must be initialised by the caller. */
int dagnn_readout_max_backward(const dagnn_plan* plan /* host */, const float* h, int ld_h, int width, int dir,
                               const float* grad_out, int ld_out, int col_off, float* grad_h, int ld_g, void* stream);

/* ... for several state buffers in one launch (distinct grad_h per job) */
#define DAGNN_MAX_READOUT_JOBS 24
typedef struct dagnn_readout_bwd_job {
    const float* h;      /* [N, ld_h] */
    float* grad_h;       /* [N, ld_g] */
    int32_t ld_h, ld_g, width, dir, col_off;
} dagnn_readout_bwd_job;
int dagnn_readout_max_backward_batch(const dagnn_plan* plan /* host */, const dagnn_readout_bwd_job* jobs /* host */, int n,
                                     const float* grad_out, int ld_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Constructor-string variants of the same loop (no BASELINE configuration selects
 * them): the aggregators `MultAttnConv` (dagnn.py:379-409), `GatedSumConv` (:254-276), `AggConv` add / max
 * (:232-251), the additive-attention convs when they meet `agg_x` (:159-169) or the Linear cell `recurr=0`
 * (:83-85,181).  Forward only; one lock-step pass of generic kernels: per step an aggregate launch (one wave per
 * frontier row), a cell launch (GRU or Linear over [input ; aggregate]) and - for the aggregators that project the
 * states - a launch of per-node linear maps over the rows just produced.  All operands are indexed by node id.
 *
 * Aggregate of frontier row v over its in-edges e = (j -> v) in original edge order, attr_e = the plan's edge
 * features (num_edge_feats = R floats; pointers marked "or NULL" drop their term):
 *   ATTN   logit_e = edge_vec0 . node0[j] + edge_mat0 . attr_e        (key score; the query half and the biases are
 *          out[v]  = sum_e softmax_e(logit) * vals[j]                   constant inside a softmax segment)
 *   MATTN  logit_e = node1[v] . (node0[j] + edge_mat0 attr_e + edge_vec0);  out[v] as for ATTN
 *          node1 = W_l q + b_l per node, node0 = W_r k + b_r per node, edge_mat0 = W_r W_e [aux_dim,R],
 *          edge_vec0 = W_r b_e
 *   GATED  out[v] = sum_e sigmoid(node0[j] + edge_mat0 attr_e + edge_vec0) * (node1[j] + edge_mat1 attr_e + edge_vec1)
 *          node0 = W_g h + b_g, node1 = W_m h (+ b_m) per node; edge_mat0 = W_g W_e, edge_vec0 = W_g b_e, ..1 for W_m
 *   ADD    out[v] = sum_e (vals[j] + edge_mat0 attr_e + edge_vec0)      edge_mat0 = W_e [val_dim,R], edge_vec0 = b_e
 *   MAX    out[v] = max_e (...)                                          (as ADD)
 *   GIVEN  out is an input: the aggregate of every node (zeros for the nodes of layer 0), computed beforehand
 *          (`agg_x`: the aggregator runs on the node inputs, dagnn.py:159-169, so dagnn_variant_aggregate does all
 *          layers in one launch; nothing then couples the layers, and the caller may pass the whole batch as ONE
 *          layer: layer_ptr = {0, N})
 * Segment softmax as PyG 1.6: exp(x - max) / (sum + 1e-16).  Columns val_dim..out_dim-1 of out[v] are zero-filled
 * (the reference pads the aggregate of the inputs up to the hidden size, dagnn.py:166-168).
 * ---------------------------------------------------------------------------------------- */
enum { DAGNN_AGG_ATTN = 0, DAGNN_AGG_MATTN = 1, DAGNN_AGG_GATED = 2, DAGNN_AGG_ADD = 3, DAGNN_AGG_MAX = 4,
       DAGNN_AGG_GIVEN = 5 };

typedef struct dagnn_variant_aggregator {
    int32_t mode;      /* DAGNN_AGG_* */
    int32_t lands;     /* 0: the messages do not land on the frontier and every row reads zeros - the reference builds ONE
                        * AggConv for both directions (dagnn.py:74-75), whose flow is source -> target in direction 1 too */
    int32_t val_dim;   /* width of a message (<= 1024) */
    int32_t aux_dim;   /* ATTN: key width; MATTN: width of the projected query / key */
    int32_t out_dim;   /* columns of out written per row (>= val_dim) */
    int32_t reserved;
    const float* vals; /* [N,ld_vals] rows summed (ATTN, MATTN, ADD, MAX) */
    int64_t ld_vals;
    const float* node0; /* [N,ld_node], see above */
    const float* node1;
    int64_t ld_node;
    const float* edge_mat0; /* row-major [dim,R], or NULL */
    const float* edge_vec0; /* [dim], or NULL (ATTN: the key weights, required) */
    const float* edge_mat1;
    const float* edge_vec1;
    float* out;        /* [N,ld_out] aggregate by node id */
    int64_t ld_out;
} dagnn_variant_aggregator;

/* Aggregate the rowrec slots [slot_begin, slot_end) of direction `dir` (any range of whole layers >= 1). */
int dagnn_variant_aggregate(const dagnn_plan* plan /* host */, const dagnn_variant_aggregator* agg /* host */, int dir,
                            int32_t slot_begin, int32_t slot_end, void* stream);

typedef struct dagnn_variant_map { /* out[v, 0:out_dim] = w_t^T h[v] + bias (+ vid_bias[v mod vid_mod]) for every row v the cell just produced */
    const float* w_t;  /* [H,out_dim]: the weight transposed (k-major) */
    const float* bias; /* [out_dim] or NULL */
    float* out;        /* [N,ld_out] */
    int64_t ld_out;
    int32_t out_dim;
    int32_t vid_mod;   /* > 0: the mapped vector is [state ; one-hot(v mod vid_mod)] (D-VAE NA, dvae/dagnn.py:124-137): the one-hot
                        * columns of the weight act as a per-vertex-id bias */
    const float* vid_bias; /* [vid_mod, out_dim] (the weight's one-hot columns, transposed) when vid_mod > 0 */
} dagnn_variant_map;

typedef struct dagnn_variant_cell {
    dagnn_variant_aggregator agg; /* agg.out: scratch [N,ld_out] with out_dim = H (GIVEN: the input aggregate) */
    int32_t recurrent; /* 1: GRUCell(input, aggregate) (gate order r,z,n); 0: Linear([input ; aggregate]) (dagnn.py:181) */
    int32_t in_dim;    /* width of the input: emb_dim for stacked layer 0, H above */
    const float* input; /* [N,ld_input]: node inputs (stacked layer 0) or the states of the cell below */
    int64_t ld_input;
    const float* w_in_t;  /* [in_dim, G*H] k-major (G = 3 gates, or 1): weight_ih^T, or weight[:, :in_dim]^T */
    const float* w_agg_t; /* [H, G*H]: weight_hh^T, or weight[:, in_dim:]^T */
    const float* b_in;    /* [G*H] bias_ih, or the Linear bias */
    const float* b_agg;   /* [3H] bias_hh, or NULL */
    float* h;          /* [N,ld_h] states, every row written exactly once */
    int64_t ld_h;
    dagnn_variant_map map[3];
    int32_t num_maps;
    int32_t reserved;
} dagnn_variant_cell;

typedef struct dagnn_variant_args {
    dagnn_variant_cell cell[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED];
    int num_stacked, dir_mask, H;
} dagnn_variant_args;

/* layer_ptr / num_layers as for dagnn_frontier_run.  Layer t of stacked layer i runs in step t + i; at layer 0 the
 * aggregate is zero (GRUCell(x, None), dagnn.py:172-174,181) unless it is GIVEN. */
int dagnn_variant_run(const dagnn_plan* plan /* host */, const dagnn_variant_args* args /* host */,
                      const int32_t* const* layer_ptr /* host */, const int32_t* num_layers /* host [2] */, void* stream);

/* Reverse sweep of the variants `gated_sum`, `mattn_h`, `add`, `max` with GRU or Linear cells (csrc/variants_bwd.hip): what
 * `loss.backward()` does to dagnn.py:144-182 with GatedSumConv (:254-276), MultAttnConv (:379-409) or AggConv add / max (:232-251).
 * Every buffer is indexed by node id and contiguous ([N, width]); weights are in their torch layouts.  The caller
 * provides the forward quantities (states h, aggregates a, pre-activations gi / gh, the per-node projections the forward
 * pass of dagnn_variant_run computed: node0 = [P | M] for gated_sum, Kr for mattn; node1 = Ql for mattn; mattn also the
 * attention weights alpha by original edge id - dagnn_variant_mattn_prepare computes a and alpha), `g` = the gradient
 * reaching h from outside (modified: the sweep accumulates into it), `g_in` = the gradient of the cell's input (the `g`
 * of the stacked layer below, or a zero-initialised dx buffer of ITS OWN per direction).  Outputs for the parallel
 * epilogue: dgi, dgh [N,3H], dnode0 (dP | dM, or dKr), dnode1 (dQl), and per-node edge-feature sums `esum` (gated:
 * [N, 2 R H], mattn: [N, R proj_dim], add / max: [N, (R + 1) H]; R <= 2; NULL without an edge encoder).  At most 8 cells. */
typedef struct dagnn_variant_bwd_cell {
    int32_t mode, lands, in_dim, proj_dim;
    int32_t recurrent;       /* 1: GRU cell; 0: the Linear cell of `recurr=0` (dagnn.py:83-85): h = W [u ; a] + b - then gi / gh / dgi /
                              * dgh are unused, w_ih = W[:, :in_dim] and w_hh = W[:, in_dim:] as contiguous [H, .] copies */
    int32_t reserved;        /* additive attention (DAGNN_AGG_ATTN): 1 = the keys are this cell's states (their gradient gets
                              * sigma w_k), 0 = the keys are the node inputs; node0 = keys [N, proj_dim], w_node = w_k [proj_dim],
                              * edge_mat0 = the folded edge gains [R]; outputs dnode0 = sigma [N], esum = sum_e ds_e attr_e [N, R] */
    const float* h;  const float* a;  const float* gi;  const float* gh;
    const float* node0;  const float* node1;  const float* alpha;
    const float* edge_mat0;  const float* edge_vec0;  const float* edge_mat1;  const float* edge_vec1;
    const float* w_node;     /* gated: [W_g ; W_m] [2H, H]; mattn: W_r [proj_dim, H] */
    const float* w_query;    /* mattn: W_l [proj_dim, in_dim] */
    const float* w_hh;       /* [3H, H] */
    const float* w_ih;       /* [3H, in_dim] */
    float* g;  float* g_in;  float* da;  float* dgi;  float* dgh;
    float* dnode0;  float* dnode1;  float* dlogit;  float* esum;
} dagnn_variant_bwd_cell;

typedef struct dagnn_variant_bwd_args {
    dagnn_variant_bwd_cell cell[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED];
    int num_stacked, dir_mask, H;
} dagnn_variant_bwd_args;

/* (also the additive-attention aggregators: mode DAGNN_AGG_ATTN) */
int dagnn_variant_mattn_prepare(const dagnn_plan* plan /* host */, const dagnn_variant_bwd_cell* cell /* host */, int dir, int H,
                                int32_t row_begin, int32_t row_end, void* stream);
int dagnn_variant_backward_run(const dagnn_plan* plan /* host */, const dagnn_variant_bwd_args* args /* host */,
                               const int32_t* const* layer_ptr /* host */, const int32_t* num_layers /* host [2] */, void* stream);
/* `agg_x` (dagnn.py:159-169: the aggregator reads the node inputs only): the cells' sweep runs with mode DAGNN_AGG_GIVEN
 * (nothing is pulled) over ONE pseudo-layer holding all rows, then the aggregator's reverse pass is one shot over all rows:
 * `cell` with h = x [N, width], a = the aggregator's output, da = the gradient of that output summed over the stacked
 * cells, g = g_in = the gradient of x (accumulated into). */
int dagnn_variant_aggregator_backward(const dagnn_plan* plan /* host */, const dagnn_variant_bwd_cell* cell /* host */, int dir,
                                      int width, int32_t row_begin, int32_t row_end, void* stream);

/* D-VAE read-out (dvae/dagnn.py:147-161, dvae/dagnn_bn.py:138-152): every graph has exactly
 * `stride` nodes; gather row g*stride + node_off of h [N,ld_h] into out[g, col_off : col_off+width]. */
int dagnn_gather_rows(const float* h, int ld_h, int width, int64_t num_graphs, int stride, int node_off,
                      float* out, int ld_out, int col_off, void* stream);
/* ... for up to 16 (matrix, node offset, column offset) jobs in one launch: every (direction, stacked layer) of the read-out */
typedef struct dagnn_gather_job { const float* h; int ld_h, width, node_off, col_off; } dagnn_gather_job;
int dagnn_gather_rows_batch(const dagnn_gather_job* jobs /* host */, int n, int64_t num_graphs, int stride, float* out, int ld_out,
                            void* stream);

/* Decoder-side single-vertex step of the D-VAE models: `_ipropagate_to(G, v, propagator, H)` (dvae/dagnn.py:187-239,
 * dvae/dagnn_bn.py:179-238; called as `_update_iv` from models_pyg.py:247-250), for all B graphs that have vertex v, in
 * ONE launch.  values [B,P,hs]: layer-0 states of v's predecessors, lists padded to P with zero rows; pred_vid [B,P]:
 * vertex id of each predecessor, -1 for padding.  The reference's soft-max runs over the padding too (a padded key
 * scores w_q.q + b; that common term cancels, so padded slots score 0 and real ones w_key.h_j + vid_bias[j]); the
 * aggregate is computed once and feeds every stacked GRU cell; with `H_given` [B,hs] it replaces the aggregate.
 * x [B,in0]: one-hot vertex types.  states [L,B,hs]: the new state of v per stacked layer (states[L-1] is the return
 * value of the reference's function). */
typedef struct dagnn_iprop_layer {
    const float* w_ih;   /* [3hs, in_dim] torch GRUCell layout */
    const float* w_hh;   /* [3hs, hs] */
    const float* b_ih;   /* [3hs] */
    const float* b_hh;   /* [3hs] */
    int in_dim;          /* in0 for layer 0, hs above */
} dagnn_iprop_layer;
int dagnn_iprop_step(const float* values, const int32_t* pred_vid, int64_t B, int P, int hs, const float* w_key,
                     const float* vid_bias /* [max_n] or NULL */, const float* H_given /* or NULL */, const float* x, int in0,
                     const dagnn_iprop_layer* layers /* host [L] */, int L, float* states, void* stream);

/* ------------------------------------------------------------------------------------------
 * Topological layering on the device: replaces `top_sort` / `add_order_info_01` (src/utils_dag.py:8-52) for a
 * whole collated batch.  layer_fwd[v] = longest-path distance of v from any source, layer_bwd[v] = the same on
 * the flipped edges; both int64 [N], i.e. `_bi_layer_idx0/1` (`_bi_layer_index0/1` is arange(N)).  `batch`
 * sorted, edges grouped by graph (the collation guarantees both).  A cyclic graph raises bit 16 of *status.
 * ---------------------------------------------------------------------------------------- */
int dagnn_topo_layers(const int64_t* edge_index /* [2,E] */, const int64_t* batch /* [N] */, int64_t N, int64_t E,
                      int64_t B, int64_t* layer_fwd, int64_t* layer_bwd, int32_t* status /* device int32 or NULL */,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * A whole encoder pass from ONE call (the evaluation pass of the D-VAE encoders, dvae/dagnn.py:99-161,
 * dvae/dagnn_bn.py:98-152: batches of 64 x 8 / 128 x 10 nodes are host-bound - the device needs ~95 us for what one call
 * at a time took the host 115-135 us to issue).  Issues, in this order, exactly the launches of dagnn_plan_build,
 * dagnn_gemm_nt_bias (stacked layer 0's input side, one group per direction: C = x W_ih^T + b_ih), dagnn_dataflow_schedule,
 * dagnn_dataflow_run (`df`: the cells' gi0 must point at the GEMM groups' outputs), dagnn_gather_rows_batch (the end /
 * start vertex of every graph into `hcat`) and - `w_out` given - the model's final Linear (out = hcat W_out^T + b_out).
 * Every buffer is the caller's; nothing is allocated or synchronised.
 * ---------------------------------------------------------------------------------------- */
typedef struct dagnn_encode_args {
    dagnn_plan plan;
    const int64_t* edge_index; const int64_t* layer_fwd; const int64_t* layer_bwd; const int64_t* batch;
    const float* edge_attr;          /* or NULL */
    int32_t* plan_status;            /* device int32[4] */
    dagnn_gemm_group gemm[2];        /* per direction: {x, W_ih [gemm_cols, in_dim], b_ih, gi0 [N, gemm_cols]} */
    int num_gemm, gemm_cols, in_dim, ld_x;
    void* schedule; size_t schedule_bytes; int cost_layer, cost_row;
    dagnn_dataflow_args df;
    dagnn_gather_job jobs[16]; int num_jobs, stride;
    float* hcat; int ld_hcat;        /* [B, ld_hcat] */
    const float* w_out; const float* b_out; float* out; int out_dim;   /* w_out [out_dim, ld_hcat] or NULL */
} dagnn_encode_args;
int dagnn_encode_forward(const dagnn_encode_args* args /* host */, void* stream);

/* out [M, K2] = A^T B with A [N, lda >= M rounded up to 4], B [N, ldb]: the reduction over N rows is SMALL, the output large -
 * the S vocabulary heads' weight gradient d logits^T x pooled vectors (dagnn.py:212-215 under main_pyg.py:62), where the
 * library GEMM's pick for 25 010 x 1 024 x 128 runs at a quarter of the matrix rate; colsum [M] (may be NULL) = column sums of
 * A (the heads' bias gradient).  fp32 MFMA, fixed summation order.  lda % 4 == 0, ldb and K2 even, A 16-byte aligned. */
int dagnn_tn_product(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t N, int M, int K2, float* out,
                     float* colsum, void* stream);

/* The TOK task's training loss and its gradient in one launch (csrc/loss.hip; replaces the caller-side loop of
 * ogbg-code/main_pyg.py:55-60: `loss += CrossEntropyLoss()(pred_list[i], y_arr[:, i])` over the S heads, `/ S`).
 * logits [B, ld >= S * V]: head s's V outputs of graph b at b * ld + s * V (how DAGNN lays its heads' outputs side by side);
 * y [B, S] int64 targets in [0, V) (no ignore_index: an out-of-range target makes the loss NaN); dlogits (same layout with its own row pitch, may be
 * NULL) receives d loss / d logits = (softmax - onehot) / (B S); row_loss [B * S] scratch; loss [1]; counter [1] device word,
 * zero before the first call (the kernel leaves it zero).  Sums in a fixed order: bitwise reproducible. */
int dagnn_seq_ce(const float* logits, int64_t ld, const int64_t* y, int B, int S, int V, float* dlogits, int64_t ld_dlogits,
                 float* row_loss, float* loss, unsigned* counter, void* stream);

/* The tail of the reference's training step - `clip_grad_norm_(model.parameters(), clip)` + `optim.Adam.step()`
 * (ogbg-code/main_pyg.py:63-65,179) - over a table of fp32 tensors (csrc/optim.hip).  dagnn_grad_norm: the 2-norm of up to
 * DAGNN_MAX_OPT_TENSORS gradients (`partial`: scratch of dagnn_opt_chunks() floats; `accumulate` != 0 adds the tensors' sum of
 * squares to *norm_sq from an earlier call - tables beyond the limit) as a device float, summed in a fixed order.
 * dagnn_clip_adam: Adam's update (no amsgrad, L2 weight decay, bias-corrected moments; `step` = 1, 2, ... of this update) with
 * the gradient scaled by min(1, max_norm / (*norm + 1e-6)) as it is read (max_norm <= 0: no clipping, norm may be NULL);
 * gradients are not modified.  Nothing synchronises; every pointer is a borrowed device pointer. */
#define DAGNN_MAX_OPT_TENSORS 48
typedef struct { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64_t numel; } dagnn_opt_tensor;
int64_t dagnn_opt_chunks(const int64_t* numel /* host */, int n);
int dagnn_grad_norm(const float* const* grads /* host array of device pointers */, const int64_t* numel /* host */, int n,
                    float* partial, int64_t partial_len, float* norm_sq /* device [1] */, int accumulate, float* norm /* device [1] */,
                    void* stream);
int dagnn_clip_adam(const dagnn_opt_tensor* tensors /* host */, int n, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step, float max_norm, const float* norm /* device */, void* stream);

/* Guard of the host side's derived-weight caches (dagnn_amd/core.py: ParamGuard; nothing in the reference corresponds - its
 * modules read their parameters on every call).  A fingerprint of up to DAGNN_MAX_FP_TENSORS fp32 / int32 tensors: 1024 words
 * spread evenly over each, weighted by position, summed mod 2^64.  mode 0 writes fp[t]; mode 1 compares with fp[t] and ORs
 * `bit` into *err (device word) for every tensor whose fingerprint moved - read back with the pass's other error words, no
 * synchronisation.  One launch, one workgroup per tensor. */
#define DAGNN_MAX_FP_TENSORS 96
int dagnn_param_fingerprint(const void* const* ptrs /* host array of device pointers */, const int64_t* numel /* host */, int n,
                            uint64_t* fp /* device [n] */, int mode, int* err /* device */, int bit, void* stream);

/* Test utility for the co-residency rule of the persistent kernels (engine.reserved_cus): occupies `num_wgs` workgroups of
 * `threads` threads for `ticks` of the 100 MHz constant clock (bounded: at most 2^31 ticks) on `stream` - a stand-in for the
 * kernels of a collective that runs next to a training pass.  Touches no memory besides `sink` (one float, may be NULL). */
int dagnn_debug_occupy(int num_wgs, int threads, int64_t ticks, float* sink, void* stream);

/* Introspection (tests, and the host-side read-back of the lock-step schedule): byte offsets of the
 * plan's arrays from `plan->data`, into a host array of 26 entries: [node_ptr, edge_ptr, depth0,
 * depth1, order0, order1, lstart0, lstart1, rowptr0, rowptr1, col0, col1, eattr0, eattr1, items,
 * total, blptr0, blptr1, rowrec0, rowrec1, slot0, slot1, eidx0, eidx1, blsplit0, blsplit1].  slot_d [N]: rowrec
 * slot of every node; eidx_d [E]: original edge id (column of edge_index) of every CSR slot; blsplit_d [N+2]: per
 * batch-level layer the first slot of the rows of the DEEP graphs of direction d (depth > thr_d, int32 header word
 * 5 + d of the plan; thr_d = 1 + the last layer with more than 14 rows) - inside a layer the shallow graphs' rows
 * come first.  blptr_d holds N+2 int32: the offsets of the
 * batch-level topological layers of direction d (entries 0..T_d) and T_d itself at index N+1. */
int dagnn_plan_layout(int64_t N, int64_t E, int64_t B, int num_edge_feats, int64_t* offsets26 /* host */);

#ifdef __cplusplus
}
#endif
#endif /* DAGNN_HIP_H */
