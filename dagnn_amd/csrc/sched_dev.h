// sched_dev.h - device bodies of the dataflow schedule (dagnn_dataflow_schedule: csrc/dataflow.hip runs one kernel each,
// csrc/prepare.hip several bodies per launch next to the plan's).  The schedule deals the graphs of a batch to G
// independent groups and re-sorts the plan's row records by (group, layer, graph): the order the persistent dataflow
// kernels (dataflow.hip, bwd_dataflow.hip) walk - their replacement for the reference's per-layer frontier selection
// (ogbg-code/model/dagnn.py:146-149).
#pragma once
#include "df_common.h"

namespace {

// Initial state of the workspace behind the assignment's tables: tables and counters = 0, records = -1 (thread `me` of
// `nfill`); the header and grp_of / gdepth / gload / loff are the assignment's (df_assign_block).
__device__ __forceinline__ void df_fill_body(int32_t* ws, const DfLayout& S, int64_t me, int64_t nfill) {
    int4* z = reinterpret_cast<int4*>(ws + S.gtab[0]);   // (every array of the layout starts on a multiple of 4 words)
    const int64_t nz = (S.grec[0] - S.gtab[0]) / 4, nf = (S.total - S.grec[0]) / 4;
    for (int64_t i = me; i < nz; i += nfill) z[i] = make_int4(0, 0, 0, 0);
    int4* f = reinterpret_cast<int4*>(ws + S.grec[0]);
    for (int64_t i = me; i < nf; i += nfill) f[i] = make_int4(-1, -1, -1, -1);
}

// ---- LPT assignment: graphs in order of decreasing depth (plan items), each to the group whose load it raises the
// least; load_k = c_layer * (depth of the first = deepest graph of k) + c_row * (nodes of k).  One wave (threadIdx.x <
// 64), lane = group.  The sequential part is B dependent steps: its operands (graph, depth, nodes, in schedule order) are
// staged in LDS first (s_g / s_d / s_n: CAP words each) - from global memory every step is three dependent round trips
// (measured 76 us at B = 128).
template <int CAP>
__device__ __forceinline__ void df_assign_block(const int32_t* __restrict__ plan, const PlanLayout& L, int32_t* ws, const DfLayout& S,
                                                int B, int G, int c_layer, int c_row, int32_t* s_g, int32_t* s_d, int32_t* s_n) {
    if (threadIdx.x >= 64) return;
    for (int64_t i = threadIdx.x; i < S.gtab[0]; i += 64) ws[i] = 0;   // header + grp_of / gdepth / gload / loff (+ padding)
    df_assign_wave<CAP>(plan + L.items, plan + L.depth[0], plan + L.depth[1], plan + L.node_ptr, ws, S, B, G, c_layer, c_row, s_g, s_d, s_n);
}

// rows per (group, layer): one workgroup per (graph g, direction d) adds its layer widths
__device__ __forceinline__ void df_count_body(const int32_t* __restrict__ plan, const PlanLayout& L, int32_t* ws, const DfLayout& S,
                                              const int g, const int d) {
    const int n0 = plan[L.node_ptr + g];
    const int k = ws[S.grp_of + g];
    // (the group's table has gdepth_k + 1 entries, gdepth_k = max over its graphs of max(depth0, depth1): the bound
    // below holds even if a batch's two layerings ever disagreed in depth)
    const int depth = min(plan[L.depth[d] + g], ws[S.gdepth + k]);
    const int32_t* ls = plan + L.lstart[d] + n0 + g;
    int32_t* cnt = ws + S.lcnt[d] + ws[S.loff + k];
    for (int t = threadIdx.x; t < depth; t += blockDim.x) atomicAdd(&cnt[t], ls[t + 1] - ls[t]);
}

// per (group k, direction d), one workgroup of 256 threads: counts -> exclusive prefix of the block-padded counts;
// gtab = {.., blocks}
__device__ __forceinline__ void df_prefix_body(int32_t* ws, const DfLayout& S, const int k, const int d) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int depth = ws[S.gdepth + k];
    int32_t* cnt = ws + S.lcnt[d] + ws[S.loff + k];
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 <= depth; c0 += 256) {
        const int t = c0 + tid;
        const int c = t < depth ? cnt[t] : 0;
        const int padded = (c + DF_RB - 1) / DF_RB * DF_RB;
        int x = padded;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wsum[tid >> 6] = x;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < (tid >> 6)) woff += wsum[w]; tot += wsum[w]; }
        const int carry = carry_s;
        if (t <= depth) cnt[t] = carry + woff + x - padded;
        __syncthreads();
        if (tid == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (tid == 0) ws[S.gtab[d] + 2 * k + 1] = carry_s / DF_RB;
}

// first record of every group (exclusive prefix over the groups' record counts); the 64 lanes of one wave, direction d.
// Returns lane k's value (lanes >= G: the total).
__device__ __forceinline__ int df_base_wave(const int32_t* ws, const DfLayout& S, int G, const int d, const int lane) {
    int x = lane < G ? ws[S.gtab[d] + 2 * lane + 1] * DF_RB : 0;
    const int own = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    return x - own;
}

// glbase[(g, t)] = padded prefix of (group, t) + rows of layer t in the group's graphs ordered before g.
// One wave per (group, layer) pair (wave w of nw), lanes over graphs.
__device__ __forceinline__ void df_lbase_body(const int32_t* __restrict__ plan, const PlanLayout& L, int32_t* ws, const DfLayout& S,
                                              int B, int G, const int w, const int nw, const int d) {
    const int lane = threadIdx.x & 63;
    const int total = ws[S.loff + G];
    const int32_t* __restrict__ ls = plan + L.lstart[d];
    const int32_t* __restrict__ node_ptr = plan + L.node_ptr;
    const int32_t* __restrict__ depth = plan + L.depth[d];
    for (int pair = w; pair < total; pair += nw) {
        int k = 0;
        while (k + 1 < G && pair >= ws[S.loff + k + 1]) ++k;
        const int t = pair - ws[S.loff + k];
        if (t >= ws[S.gdepth + k]) continue;   // the table has depth + 1 entries per group
        int carry = ws[S.lcnt[d] + pair];
        for (int g0 = 0; g0 < B; g0 += 64) {
            const int g = g0 + lane;
            int cnt = 0, base = 0;
            bool has = false;
            if (g < B && ws[S.grp_of + g] == k && t < depth[g]) {
                base = node_ptr[g] + g + t;
                cnt = ls[base + 1] - ls[base];
                has = true;
            }
            int x = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
            if (has) ws[S.glbase[d] + base] = carry + x - cnt;
            carry += __shfl(x, 63, 64);
        }
    }
}

// copy every row record to its place in the group order (padding records were preset to -1); thread p = 256 bx + tid of
// direction d.  `gbase`: first record of every group (ws gtab[d][2k], or an LDS copy of it).
__device__ __forceinline__ void df_records_body(const int32_t* __restrict__ plan, const PlanLayout& L, int32_t* ws, const DfLayout& S,
                                                int N, const int bx, const int d, const int32_t* gbase, const int gstride) {
    const int p = bx * 256 + threadIdx.x;   // per-graph sorted position
    if (p >= N) return;
    const int v = plan[L.order[d] + p];
    const int slot = plan[L.pos[d] + v];
    const int4* src = reinterpret_cast<const int4*>(plan + L.rowrec[d]) + 4 * (int64_t)slot;
    const int4 r0 = src[0];
    const int g = r0.w;
    const int n0 = plan[L.node_ptr + g];
    const int depth = plan[L.depth[d] + g];
    const int32_t* ls = plan + L.lstart[d] + n0 + g;   // depth + 1 absolute positions
    int lo = 0, hi = depth;                            // largest t with ls[t] <= p
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ls[mid] <= p) lo = mid; else hi = mid; }
    const int t = lo;
    const int k = ws[S.grp_of + g];
    const int rec = gbase[gstride * k] + ws[S.glbase[d] + n0 + g + t] + (p - ls[t]);
    int4* dst = reinterpret_cast<int4*>(ws + S.grec[d]) + 4 * (int64_t)rec;
    dst[0] = r0; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
}

}  // namespace
