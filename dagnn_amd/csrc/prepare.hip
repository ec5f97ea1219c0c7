// prepare.hip - everything in front of the recurrence as one fused pipeline: plan + dataflow schedule + the row work of
// forward() that does not depend on the plan (dagnn_prepare, include/dagnn_hip.h).
//
// Reference path replaced: `DAGNN.forward` up to the layer loop - side effect 1 (ogbg-code/model/dagnn.py:130-133), the node
// encoder (dagnn.py:139, utils.py:26-28), frontier selection and edge scan of every layer (dagnn.py:146-157) - and the
// input half of the first GRUCell (dagnn.py:181) where the caller folds it into the encoder's tables.
//
// Why.  Round 5's trace of a headline forward (profiles/r05_kernel_stats.csv): 13 plan / schedule launches (157 us: every
// one a dependent step of a few microseconds on a handful of workgroups, ~5 us of launch floor each) + encoder + the input
// GEMM (170 us) in front of a 1.25 ms recurrence.  The bodies are unchanged (plan_dev.h, sched_dev.h); what changes is which
// of them share a launch - a launch runs every body whose inputs are ready, picked by workgroup id:
//   1  plan_ptr                                           node / edge offsets, contract checks
//   2  plan_graph (2B workgroups)  |  rows                per-graph sorts (latency-bound: 15 dependent passes of ONE workgroup
//                                                         per graph, the 657-node graph sets the time) next to the HBM-bound
//                                                         encoder rows / folded gi0 rows / index stack
//   3  plan_blptr (2)  |  plan_items + LPT assignment (1)  |  workspace fill  |  the rest of the rows
//   4  df_count (2B)  |  plan_lbase
//   5  df_prefix (2G)  |  plan_rowrec (+ seal)
//   6  df_lbase
//   7  df_records (+ the groups' first records)
// Measured: DESIGN.md section 4h.
#include "plan_dev.h"
#include "sched_dev.h"

#ifndef DAGNN_PREPARE_SHARES
#define DAGNN_PREPARE_SHARES 20, 80, 0, 0, 0, 0   // percent of the encoder rows that ride on launches 2..7 (with a schedule); measured: DESIGN.md 4h
#endif

namespace {

struct PrepRows {
    const int64_t* x;
    int64_t* depth;
    int max_depth, ntab;
    const float* type_emb[DAGNN_PREPARE_MAX_TABLES];
    const float* attr_emb[DAGNN_PREPARE_MAX_TABLES];
    const float* depth_emb[DAGNN_PREPARE_MAX_TABLES];
    float* out[DAGNN_PREPARE_MAX_TABLES];
    int width[DAGNN_PREPARE_MAX_TABLES], ld_out[DAGNN_PREPARE_MAX_TABLES];
    const int64_t* stack_src[4];
    int64_t* stack_out;
    int64_t N;
};

// the share of the rows that rides on one launch: nodes [v0, v1) by the workgroups from `first` on
struct RowsShare { int64_t v0, v1; int first; };

// Workgroup rb of nrb, nodes [v0, v1): the index stack as one flat coalesced copy, then one wave per node: indices read
// once, out_k[v,:] = (type_k[x0] + attr_k[x1]) + depth_k[min(depth, max_depth)] for every table set k (the association of
// utils.py:28; misc.hip's encode_ast_kernel is the single-table form).
__device__ __forceinline__ void rows_body(const PrepRows& J, const int64_t rb, const int64_t nrb, const int64_t v0, const int64_t v1,
                                          const bool with_stack) {
    const int64_t N = J.N;
    if (with_stack && J.stack_out)
        for (int64_t idx = rb * blockDim.x + threadIdx.x; idx < 4 * N; idx += nrb * blockDim.x) {
            const int j = (int)(idx / N);
            J.stack_out[idx] = J.stack_src[j][idx - j * N];
        }
    if (!J.x) return;
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int64_t v = v0 + rb * wpb + (threadIdx.x >> 6); v < v1; v += nrb * wpb) {
        const int64_t t = J.x[2 * v], a = J.x[2 * v + 1];
        int64_t dp = J.depth[v];
        if (dp > J.max_depth) { dp = J.max_depth; if (lane == 0) J.depth[v] = dp; }
        for (int k = 0; k < J.ntab; ++k) {
            const int W = J.width[k], W4 = W >> 2;
            const float4* pt = reinterpret_cast<const float4*>(J.type_emb[k] + t * W);
            const float4* pa = reinterpret_cast<const float4*>(J.attr_emb[k] + a * W);
            const float4* pd = reinterpret_cast<const float4*>(J.depth_emb[k] + dp * W);
            float4* po = reinterpret_cast<float4*>(J.out[k] + v * J.ld_out[k]);
            for (int c = lane; c < W4; c += 64) {
                const float4 u = pt[c], w = pa[c], z = pd[c];
                float4 r;
                r.x = (u.x + w.x) + z.x; r.y = (u.y + w.y) + z.y; r.z = (u.z + w.z) + z.z; r.w = (u.w + w.w) + z.w;
                // (non-temporal: the rows are read next by the recurrence, a whole launch later - they need not push the
                // tables out of the L2 on their way to memory; measured -5 us per forward)
                typedef float prep_v4f __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(prep_v4f{r.x, r.y, r.z, r.w}, reinterpret_cast<prep_v4f*>(po + c));
            }
        }
    }
}

__global__ void __launch_bounds__(256) prep_ptr_kernel(int32_t* plan, PlanLayout L, const int64_t* __restrict__ edge_index,
                                                        const int64_t* __restrict__ batch, int64_t N, int64_t E, int64_t B, int R, int32_t* status) {
    plan_ptr_body(plan, L, edge_index, batch, N, E, B, R, status, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void __launch_bounds__(256) prep_rows_kernel(PrepRows J) { rows_body(J, blockIdx.x, gridDim.x, 0, J.N, true); }

// launch 2: workgroups [0, 2B) = (graph, direction) sorts - dispatched first: they are the long pole -, the rest = rows
__global__ void __launch_bounds__(PB) prep_graph_rows_kernel(int32_t* plan, PlanLayout L, const int64_t* __restrict__ edge_index,
                                                              const int64_t* __restrict__ layer_fwd, const int64_t* __restrict__ layer_bwd,
                                                              const float* __restrict__ edge_attr, int R, int64_t N, int64_t E,
                                                              int32_t* status, int B, PrepRows J, RowsShare rs) {
    const int b = blockIdx.x;
    if (b < 2 * B) {
        plan_graph_body(plan, L, edge_index, layer_fwd, layer_bwd, edge_attr, R, N, E, status, b >> 1, b & 1);
        return;
    }
    rows_body(J, b - 2 * B, gridDim.x - 2 * B, rs.v0, rs.v1, true);
}

// Work items AND the schedule's assignment by one workgroup of 1024 threads, from ONE trip to memory (B <= 2048): the depths
// of both directions and the node counts go to LDS, every thread ranks its item (plan_items_body: depth descending, ties by
// index) and its graph (the direction-0 items in that order = graphs by depth0 descending, ties by graph: what
// df_assign_wave compacts out of the item list), wave 0 walks the chain.  The separate bodies go through memory three
// times on the way (items -> cursor -> items, then items -> depths / node counts, 64 at a time): 13 + 25 us at B = 128.
// buf: 12288 words.
__device__ __forceinline__ void prep_items_assign_body(int32_t* plan, const PlanLayout& L, int32_t* ws, const DfLayout& S, int N, int B,
                                                       int G, int c_layer, int c_row, int32_t* buf) {
    int32_t* keys = buf;            // [2B] depth of item i = 2 g + d
    int32_t* nn = buf + 4096;       // [B] nodes of graph g
    int32_t* s_g = buf + 6144, *s_d = buf + 8192, *s_n = buf + 10240;
    const int tid = threadIdx.x, n = 2 * B;
    for (int i = tid; i < n; i += blockDim.x) keys[i] = plan[L.depth[i & 1] + (i >> 1)];
    for (int g = tid; g < B; g += blockDim.x) nn[g] = plan[L.node_ptr + g + 1] - plan[L.node_ptr + g];
    if (G > 0) for (int64_t i = tid; i < S.gtab[0]; i += blockDim.x) ws[i] = 0;   // header + grp_of / gdepth / gload / loff (+ padding)
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const int ki = keys[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (keys[j] > ki) || (keys[j] == ki && j < i);
        plan[L.items + rank] = i;
    }
    if (G <= 0) return;
    for (int g = tid; g < B; g += blockDim.x) {
        const int kg = keys[2 * g];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (keys[2 * j] > kg) || (keys[2 * j] == kg && j < g);
        s_g[rank] = g; s_d[rank] = max(kg, keys[2 * g + 1]); s_n[rank] = nn[g];
    }
    __syncthreads();   // (also: the header's zeros are in memory before the chain writes into it)
    if (tid < 64) df_assign_chain(nullptr, nullptr, nullptr, nullptr, ws, S, B, G, c_layer, c_row, s_g, s_d, s_n, true, B, (long long)N);
}

// launch 3 (1024 threads): workgroups 0 / 1 the batch-level layers of a direction, workgroup 2 the work items and - they
// are its input, still in this workgroup's hands - the LPT assignment of the schedule, the rest the workspace's initial state
__global__ void __launch_bounds__(1024) prep_mid_kernel(int32_t* plan, PlanLayout L, int N, int B, const int32_t* __restrict__ status,
                                                         int32_t* ws, DfLayout S, int G, int c_layer, int c_row, int nfill,
                                                         PrepRows J, RowsShare rs) {
    constexpr int CAP = 4096;
    __shared__ int32_t buf[3 * CAP];   // the items' keys, then the assignment's three staging arrays
    const int b = blockIdx.x;
    if (b >= 3 + nfill) {   // the rest of the rows: this launch waits for ONE wave's 128-step chain (the assignment) otherwise
        rows_body(J, b - 3 - nfill, gridDim.x - 3 - nfill, rs.v0, rs.v1, false);
        return;
    }
    if (b < 2) { plan_blptr_body(plan, L, N, B, status, b); return; }
    if (b == 2) {
        if (status[0] & 7) return;   // contract violated (plan_ptr_body): the tables are garbage - do not walk them
        if (B <= 2048) {
            prep_items_assign_body(plan, L, ws, S, N, B, status[0] != 0 ? 0 : G, c_layer, c_row, buf);
            return;
        }
        plan_items_body(plan, L, B, status, buf);
        if (G <= 0 || status[0] != 0) return;   // (no schedule asked for; or the batch violates the plan contract)
        __syncthreads();                        // the items are in memory, the keys are done with
        df_assign_block<CAP>(plan, L, ws, S, B, G, c_layer, c_row, buf, buf + CAP, buf + 2 * CAP);
        return;
    }
    if (G <= 0 || status[0] != 0) return;
    df_fill_body(ws, S, (int64_t)(b - 3) * blockDim.x + threadIdx.x, (int64_t)nfill * blockDim.x);
}

// launch 4: workgroups [0, 2B) (with a schedule) the rows per (group, layer), the rest the first slot of every (graph, layer)
__global__ void __launch_bounds__(256) prep_lbase_count_kernel(int32_t* plan, PlanLayout L, int N, int B, const int32_t* __restrict__ status,
                                                                int32_t* ws, DfLayout S, int G, PrepRows J, RowsShare rs) {
    const int nc = G > 0 ? 2 * B : 0;
    const int b = blockIdx.x;
    if (b >= rs.first) { rows_body(J, b - rs.first, gridDim.x - rs.first, rs.v0, rs.v1, false); return; }
    if (b < nc) {
        if (status[0] != 0) return;
        df_count_body(plan, L, ws, S, b >> 1, b & 1);
        return;
    }
    plan_lbase_body(plan, L, N, B, status, (b - nc) >> 1, (b - nc) & 1);
}

// launch 5: workgroups [0, 2G) the groups' padded prefixes, the rest the row records - or, for a batch that violates the
// contract, the seal (an EMPTY plan: plan_seal_body; every status bit is final since launch 2)
__global__ void __launch_bounds__(256) prep_rowrec_prefix_kernel(int32_t* plan, PlanLayout L, const int64_t* __restrict__ batch,
                                                                  const int64_t* __restrict__ layer_fwd, const int64_t* __restrict__ layer_bwd,
                                                                  int N, int B, int R, const int32_t* __restrict__ status,
                                                                  int32_t* ws, DfLayout S, int G, PrepRows J, RowsShare rs) {
    const int np = G > 0 ? 2 * G : 0;
    const int b = blockIdx.x;
    if (b >= rs.first) { rows_body(J, b - rs.first, gridDim.x - rs.first, rs.v0, rs.v1, false); return; }
    if (b < np) {
        if (status[0] != 0) return;
        df_prefix_body(ws, S, b >> 1, b & 1);
        return;
    }
    if (status[0] != 0) {
        plan_seal_body(plan, L, N, B, (int64_t)(b - np) * 256 + threadIdx.x, (int64_t)(rs.first - np) * 256);
        return;
    }
    plan_rowrec_body(plan, L, batch, layer_fwd, layer_bwd, N, R, status, (b - np) >> 1, (b - np) & 1);
}

__global__ void __launch_bounds__(256) prep_df_lbase_kernel(const int32_t* __restrict__ plan, PlanLayout L, int32_t* ws, DfLayout S,
                                                             int B, int G, const int32_t* __restrict__ status, PrepRows J, RowsShare rs) {
    const int b = blockIdx.x;
    if (b >= rs.first) { rows_body(J, b - rs.first, gridDim.x - rs.first, rs.v0, rs.v1, false); return; }
    if (status[0] != 0) return;
    df_lbase_body(plan, L, ws, S, B, G, (b >> 1) * 4 + (threadIdx.x >> 6), (rs.first >> 1) * 4, b & 1);
}

// launch 7: the schedule's records; every workgroup derives the groups' first records itself (one wave scan over <= 64
// block counts), workgroup 0 of a direction also stores them (gtab[2k]: what the dataflow kernels read)
__global__ void __launch_bounds__(256) prep_records_kernel(const int32_t* __restrict__ plan, PlanLayout L, int32_t* ws, DfLayout S,
                                                            int N, int G, const int32_t* __restrict__ status, PrepRows J, RowsShare rs) {
    const int b = blockIdx.x;
    if (b >= rs.first) { rows_body(J, b - rs.first, gridDim.x - rs.first, rs.v0, rs.v1, false); return; }
    if (status[0] != 0) return;
    __shared__ int32_t s_base[DF_MAX_GROUPS];
    const int d = b & 1;
    if (threadIdx.x < 64) {
        const int base = df_base_wave(ws, S, G, d, threadIdx.x);
        if ((int)threadIdx.x < G) {
            s_base[threadIdx.x] = base;
            if (b < 2) ws[S.gtab[d] + 2 * threadIdx.x] = base;
        }
    }
    __syncthreads();
    df_records_body(plan, L, ws, S, N, b >> 1, d, s_base, 1);
}

}  // namespace

static int prep_rows_of(const dagnn_prepare_rows* rows, int64_t N, PrepRows* J) {
    *J = PrepRows{};
    J->N = N;
    if (!rows) return DAGNN_OK;
    if (rows->x) {
        if (!rows->depth || rows->num_tables < 1 || rows->num_tables > DAGNN_PREPARE_MAX_TABLES) return DAGNN_EINVAL;
        J->x = rows->x; J->depth = rows->depth; J->max_depth = rows->max_depth; J->ntab = rows->num_tables;
        for (int k = 0; k < rows->num_tables; ++k) {
            const auto& t = rows->table[k];
            if (!t.type_emb || !t.attr_emb || !t.depth_emb || !t.out || t.width <= 0 || (t.width & 3) || (t.ld_out & 3) || t.ld_out < t.width)
                return DAGNN_EINVAL;
            J->type_emb[k] = t.type_emb; J->attr_emb[k] = t.attr_emb; J->depth_emb[k] = t.depth_emb; J->out[k] = t.out;
            J->width[k] = t.width; J->ld_out[k] = t.ld_out;
        }
    }
    if (rows->stack_out) {
        for (int j = 0; j < 4; ++j) {
            if (!rows->stack_src[j]) return DAGNN_EINVAL;
            J->stack_src[j] = rows->stack_src[j];
        }
        J->stack_out = rows->stack_out;
    }
    return DAGNN_OK;
}

extern "C" int dagnn_prepare(const dagnn_plan* pl, const int64_t* edge_index, const int64_t* layer_fwd, const int64_t* layer_bwd,
                             const int64_t* batch, const float* edge_attr, int32_t* status, void* schedule, size_t schedule_bytes,
                             int groups, int cost_layer, int cost_row, const dagnn_prepare_rows* rows, void* stream_) {
    if (!pl || !pl->data || !status || groups < 0 || groups > DF_MAX_GROUPS || (groups > 0 && (!schedule || cost_layer < 0 || cost_row < 0)))
        return DAGNN_EINVAL;
    const int64_t N = pl->N, E = pl->E, B = pl->B;
    const int R = pl->num_edge_feats;
    if (N < 0 || E < 0 || B < 0 || R < 0) return DAGNN_EINVAL;
    if (N > 0 && (!layer_fwd || !layer_bwd || !batch)) return DAGNN_EINVAL;
    if (E > 0 && !edge_index) return DAGNN_EINVAL;
    if (R > 0 && E > 0 && !edge_attr) return DAGNN_EINVAL;
    if (N >= (int64_t(1) << 30) || E >= (int64_t(1) << 30)) return DAGNN_EINVAL;  // int32 plan
    PrepRows J;
    int rc = prep_rows_of(rows, N, &J);
    if (rc != DAGNN_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const bool has_rows = N > 0 && (J.x || J.stack_out);
    const int64_t rows_blocks = has_rows ? (N + 3) / 4 < 2048 ? (N + 3) / 4 : 2048 : 0;
    if (N == 0 || B == 0 || (!(pl->flags & DAGNN_PLAN_GENERAL_BUILD) && dagnn_plan_is_small(N, E, B))) {
        // empty and small batches: the separate calls (one workgroup each for a small batch, csrc/small.hip)
        rc = dagnn_plan_build(pl, edge_index, layer_fwd, layer_bwd, batch, edge_attr, status, stream_);
        if (rc != DAGNN_OK) return rc;
        if (has_rows) {
            hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)rows_blocks), dim3(256), 0, stream, J);
            DAGNN_CHECK_LAUNCH();
        }
        if (groups > 0) return dagnn_dataflow_schedule(pl, schedule, schedule_bytes, groups, cost_layer, cost_row, status, stream_);
        return DAGNN_OK;
    }
    const PlanLayout L = dagnn_plan_layout_words(N, E, B, R);
    if ((size_t)L.total * 4 > pl->bytes) return DAGNN_ENOSPC;
    DfLayout S = df_layout_words(N, B, groups > 0 ? groups : 1);
    if (groups > 0 && (size_t)S.total * 4 > schedule_bytes) return DAGNN_ENOSPC;
    int32_t* p = (int32_t*)pl->data;
    int32_t* ws = (int32_t*)schedule;
    int64_t work = N + 2 > E ? N + 2 : E;
    if (work < B + 1) work = B + 1;
    // 1
    hipLaunchKernelGGL(prep_ptr_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, p, L, edge_index, batch, N, E, B, R, status);
    DAGNN_CHECK_LAUNCH();
    // The rows ride on the launches that wait for ONE workgroup's dependent steps anyway (with a schedule; shares in percent
    // of the nodes, DAGNN_PREPARE_SHARES): launch 2 the largest graph's sorts (12 us), launch 3 one wave's 128-step chain
    // (25 us).  Launches 4-7 can carry a share too, but a few microseconds of launch floor each hide nothing: measured
    // 1.43 ms per forward with 9 % of the rows on each against 1.41 ms without
    static const int share_sched[6] = {DAGNN_PREPARE_SHARES};
    int64_t cut[7];
    cut[0] = 0;
    for (int k = 0; k < 6; ++k) {
        int64_t pct = (J.x && groups > 0) ? share_sched[k] : (k == 0 ? 100 : 0);
        cut[k + 1] = k == 5 ? N : cut[k] + N * pct / 100;
        if (cut[k + 1] > N) cut[k + 1] = N;
    }
    if (!J.x) for (int k = 1; k <= 6; ++k) cut[k] = N;
    auto rows_wgs = [&](int k, int64_t per_wg, int64_t cap) -> int64_t {   // workgroups of launch k + 2 for its share
        const int64_t n = J.x ? cut[k + 1] - cut[k] : 0;
        const int64_t w = (n + per_wg - 1) / per_wg;
        return w < cap ? w : cap;
    };
    const int nfill = groups > 0 ? 128 : 0;
    {   // 2 (also the index stack)
        int64_t rw = rows_wgs(0, 4, 2048);
        if (J.stack_out && rw < 64) rw = 64;
        const RowsShare rs = {cut[0], cut[1], (int)(2 * B)};
        hipLaunchKernelGGL(prep_graph_rows_kernel, dim3((unsigned)(2 * B + rw)), dim3(PB), 0, stream, p, L, edge_index, layer_fwd,
                           layer_bwd, edge_attr, R, N, E, status, (int)B, J, rs);
        DAGNN_CHECK_LAUNCH();
    }
    {   // 3
        const RowsShare rs = {cut[1], cut[2], 3 + nfill};
        hipLaunchKernelGGL(prep_mid_kernel, dim3((unsigned)(3 + nfill + rows_wgs(1, 16, 512))), dim3(1024), 0, stream, p, L, (int)N, (int)B,
                           status, ws, S, groups, cost_layer, cost_row, nfill, J, rs);
        DAGNN_CHECK_LAUNCH();
    }
    {   // 4
        const int64_t own = (groups > 0 ? 2 * B : 0) + 2 * ((N + 3) / 4);
        const RowsShare rs = {cut[2], cut[3], (int)own};
        hipLaunchKernelGGL(prep_lbase_count_kernel, dim3((unsigned)(own + rows_wgs(2, 4, 2048))), dim3(256), 0, stream, p, L, (int)N, (int)B,
                           status, ws, S, groups, J, rs);
        DAGNN_CHECK_LAUNCH();
    }
    const int64_t rb = (N + 255) / 256;
    {   // 5
        const int64_t own = (groups > 0 ? 2 * groups : 0) + 2 * rb;
        const RowsShare rs = {cut[3], cut[4], (int)own};
        hipLaunchKernelGGL(prep_rowrec_prefix_kernel, dim3((unsigned)(own + rows_wgs(3, 4, 2048))), dim3(256), 0, stream, p, L, batch,
                           layer_fwd, layer_bwd, (int)N, (int)B, R, status, ws, S, groups, J, rs);
        DAGNN_CHECK_LAUNCH();
    }
    if (groups > 0) {
        // 6, 7
        int64_t dlb = (N + groups + 3) / 4;
        if (dlb > 2048) dlb = 2048;
        const RowsShare r6 = {cut[4], cut[5], (int)(2 * dlb)};
        hipLaunchKernelGGL(prep_df_lbase_kernel, dim3((unsigned)(2 * dlb + rows_wgs(4, 4, 2048))), dim3(256), 0, stream, p, L, ws, S, (int)B,
                           groups, status, J, r6);
        DAGNN_CHECK_LAUNCH();
        const RowsShare r7 = {cut[5], cut[6], (int)(2 * rb)};
        hipLaunchKernelGGL(prep_records_kernel, dim3((unsigned)(2 * rb + rows_wgs(5, 4, 2048))), dim3(256), 0, stream, p, L, ws, S, (int)N,
                           groups, status, J, r7);
        DAGNN_CHECK_LAUNCH();
    }
    return DAGNN_OK;
}
