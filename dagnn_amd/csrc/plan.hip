// plan.hip - builds the layer-sorted per-graph CSR ("plan") the recurrence kernel walks.
//
// Replaces, per forward call, the reference's `layer = ids[layer_id == l_idx]`
// (ogbg-code/model/dagnn.py:146-147), its per-frontier-node scan of the whole edge_index
// (dagnn.py:151-157, O(N*E) per direction) and `_get_output_nodes` (dagnn.py:119-126).
//
// One workgroup per (graph, direction) - graphs are independent, a PyG batch stores each graph's
// nodes and edges contiguously - does two stable counting sorts in its own slice of the plan:
// nodes by layer (frontiers become contiguous ranges) and edges by the sorted position of the
// node they feed (rows of the CSR, original edge order kept inside a row so that the softmax /
// weighted sum adds in the same order as the reference's scan produces).  Everything is int32;
// the int64 inputs are narrowed on the way in.  The device code lives in plan_dev.h (csrc/prepare.hip runs the same bodies
// several per launch, next to the schedule's and the encoder's).
#include "plan_dev.h"

namespace {

__global__ void plan_ptr_kernel(int32_t* plan, PlanLayout L, const int64_t* __restrict__ edge_index,
                                const int64_t* __restrict__ batch, int64_t N, int64_t E, int64_t B, int R,
                                int32_t* status) {
    plan_ptr_body(plan, L, edge_index, batch, N, E, B, R, status, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void __launch_bounds__(PB) plan_graph_kernel(int32_t* plan, PlanLayout L,
                                                         const int64_t* __restrict__ edge_index,
                                                         const int64_t* __restrict__ layer_fwd,
                                                         const int64_t* __restrict__ layer_bwd,
                                                         const float* __restrict__ edge_attr, int R,
                                                         int64_t N, int64_t E, int32_t* status) {
    plan_graph_body(plan, L, edge_index, layer_fwd, layer_bwd, edge_attr, R, N, E, status, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(256) plan_seal_kernel(int32_t* plan, PlanLayout L, int64_t N, int64_t B,
                                                         const int32_t* __restrict__ status) {
    if (status[0] == 0) return;
    plan_seal_body(plan, L, N, B, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

__global__ void __launch_bounds__(1024) plan_items_kernel(int32_t* plan, PlanLayout L, int B, const int32_t* __restrict__ status) {
    __shared__ int32_t keys[4096];
    plan_items_body(plan, L, B, status, keys);
}

__global__ void __launch_bounds__(1024) plan_blptr_kernel(int32_t* plan, PlanLayout L, int N, int B, const int32_t* __restrict__ status) {
    plan_blptr_body(plan, L, N, B, status, blockIdx.x);
}

__global__ void __launch_bounds__(256) plan_lbase_kernel(int32_t* plan, PlanLayout L, int N, int B, const int32_t* __restrict__ status) {
    plan_lbase_body(plan, L, N, B, status, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(256) plan_rowrec_kernel(int32_t* plan, PlanLayout L, const int64_t* __restrict__ batch,
                                                           const int64_t* __restrict__ layer_fwd,
                                                           const int64_t* __restrict__ layer_bwd, int N, int R,
                                                           const int32_t* __restrict__ status) {
    plan_rowrec_body(plan, L, batch, layer_fwd, layer_bwd, N, R, status, blockIdx.x, blockIdx.y);
}

}  // namespace


extern "C" size_t dagnn_plan_bytes(int64_t N, int64_t E, int64_t B, int num_edge_feats) {
    if (N < 0 || E < 0 || B < 0 || num_edge_feats < 0) return 0;
    return (size_t)dagnn_plan_layout_words(N, E, B, num_edge_feats).total * sizeof(int32_t);
}

extern "C" int dagnn_plan_layout(int64_t N, int64_t E, int64_t B, int R, int64_t* o) {
    if (!o) return DAGNN_EINVAL;
    PlanLayout L = dagnn_plan_layout_words(N, E, B, R);
    int64_t w[26] = {L.node_ptr, L.edge_ptr, L.depth[0], L.depth[1], L.order[0], L.order[1], L.lstart[0],
                     L.lstart[1], L.rowptr[0], L.rowptr[1], L.col[0], L.col[1], L.eattr[0], L.eattr[1],
                     L.items, L.total, L.blptr[0], L.blptr[1], L.rowrec[0], L.rowrec[1],
                     L.pos[0], L.pos[1], L.eidx[0], L.eidx[1], L.blsplit[0], L.blsplit[1]};
    for (int i = 0; i < 26; ++i) o[i] = w[i] * 4;
    return DAGNN_OK;
}

extern "C" int dagnn_plan_build(const dagnn_plan* pl, const int64_t* edge_index, const int64_t* layer_fwd,
                                const int64_t* layer_bwd, const int64_t* batch, const float* edge_attr,
                                int32_t* status, void* stream_) {
    if (!pl || !pl->data) return DAGNN_EINVAL;
    void* plan = pl->data;
    const size_t plan_bytes = pl->bytes;
    const int64_t N = pl->N, E = pl->E, B = pl->B;
    const int R = pl->num_edge_feats;
    if (N < 0 || E < 0 || B < 0 || R < 0) return DAGNN_EINVAL;
    if (N > 0 && (!layer_fwd || !layer_bwd || !batch)) return DAGNN_EINVAL;
    if (E > 0 && !edge_index) return DAGNN_EINVAL;
    if (R > 0 && E > 0 && !edge_attr) return DAGNN_EINVAL;
    if (N >= (int64_t(1) << 30) || E >= (int64_t(1) << 30)) return DAGNN_EINVAL;  // int32 plan
    PlanLayout L = dagnn_plan_layout_words(N, E, B, R);
    if ((size_t)L.total * 4 > plan_bytes) return DAGNN_ENOSPC;
    hipStream_t stream = (hipStream_t)stream_;
    if (!(pl->flags & DAGNN_PLAN_GENERAL_BUILD) && dagnn_plan_is_small(N, E, B))   // one workgroup, everything in LDS (small.hip)
        return dagnn_plan_build_small(pl, edge_index, layer_fwd, layer_bwd, batch, edge_attr, status, stream);
    int32_t* p = (int32_t*)plan;
    int64_t work = N + 2 > E ? N + 2 : E;
    if (work < B + 1) work = B + 1;
    hipLaunchKernelGGL(plan_ptr_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, p, L,
                       edge_index, batch, N, E, B, R, status);
    DAGNN_CHECK_LAUNCH();
    if (B > 0) {
        hipLaunchKernelGGL(plan_graph_kernel, dim3((unsigned)B, 2), dim3(PB), 0, stream, p, L, edge_index,
                           layer_fwd, layer_bwd, edge_attr, R, N, E, status);
        DAGNN_CHECK_LAUNCH();
        hipLaunchKernelGGL(plan_items_kernel, dim3(1), dim3(1024), 0, stream, p, L, (int)B, status);
        DAGNN_CHECK_LAUNCH();
        hipLaunchKernelGGL(plan_blptr_kernel, dim3(2), dim3(1024), 0, stream, p, L, (int)N, (int)B, status);
        DAGNN_CHECK_LAUNCH();
        if (N > 0) {
            hipLaunchKernelGGL(plan_lbase_kernel, dim3((unsigned)((N + 3) / 4), 2), dim3(256), 0, stream, p, L,
                               (int)N, (int)B, status);
            DAGNN_CHECK_LAUNCH();
            hipLaunchKernelGGL(plan_rowrec_kernel, dim3((unsigned)((N + 255) / 256), 2), dim3(256), 0, stream, p, L,
                               batch, layer_fwd, layer_bwd, (int)N, R, status);
            DAGNN_CHECK_LAUNCH();
        }
    }
    if (status) {   // a batch that violates the contract leaves an EMPTY plan behind, whatever the kernels above made of it
        int64_t w = N + 2 > B ? N + 2 : B;
        hipLaunchKernelGGL(plan_seal_kernel, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, stream, p, L, N, B, status);
        DAGNN_CHECK_LAUNCH();
    }
    return DAGNN_OK;
}
