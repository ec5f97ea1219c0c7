// Definitions shared by the two persistent dataflow kernels: dataflow.hip (forward recurrence) and bwd_dataflow.hip (its
// reverse-mode sweep).  Both walk the same schedule workspace (dagnn_dataflow_schedule) and use the same workgroup
// shape: 4 compute waves + 2 loader sets x 4 loader waves, 32 hidden units per slice, blocks of 4 rows.
#pragma once
#include "common.h"

namespace {

constexpr int DF_JS = 32;      // hidden units per slice
constexpr int DF_RB = 4;       // rows per block = loader waves
constexpr int DF_NCW = 4;      // compute waves
constexpr int DF_NLS = 2;          // streams per workgroup = loader sets: set s serves group NLS * pair + s - a block costs a loader wave one trip to
                               // memory plus ~1 us of scalar work, twice what the compute waves need for it
constexpr int DF_NSLOT = 4;             // LDS ring depth (blocks the loaders may run ahead)
// (Measured and removed: two chunks per trip for rows with > 4 in-edges - the second sweep's 32 registers spilled the
// loader at 3 waves per SIMD.)
#ifndef DF_NLW_V
#define DF_NLW_V (12 - DF_NCW)
#endif
constexpr int DF_NLW = DF_NLW_V;                 // loader waves per workgroup (12 waves = 3 per SIMD at <= 168 VGPRs)
constexpr int DF_WPS = DF_NLW / DF_NLS;     // ... per stream
constexpr int DF_RPW = DF_RB / DF_WPS;      // rows of a block per loader wave (one after the other)
// group served by stream `set` of workgroup set `pair` (-1: none)
__device__ __host__ __forceinline__ int df_group_of_stream(int pair, int set, int groups) {
    const int g = DF_NLS * pair + set;
    return g < groups ? g : -1;
}
constexpr int DF_THREADS = 64 * (DF_NCW + DF_NLW);
constexpr int DF_MAX_GROUPS = 64;
constexpr int DF_MAGIC = 0x44463031;   // "DF01"

// ---------------------------------------------------------------- schedule workspace (int32 words)
struct DfLayout {
    int64_t grp_of;        // [B]     group of every graph
    int64_t gdepth;        // [G]     deepest graph of every group (layers)
    int64_t gload;         // [G]     LPT load of every group (cost units; diagnostics)
    int64_t loff;          // [G+1]   offset of group k's per-layer table in lcnt (both directions: depths are equal)
    int64_t gtab[2];       // [2G]    per group {first record, number of blocks}
    int64_t lcnt[2];       // [N+G+1] per (group, layer): rows, then the padded exclusive prefix (records)
    int64_t glbase[2];     // [N+B]   per (graph, layer) (indexed like lstart): first record inside its group
    int64_t grec[2];       // [16 * (4N + 4)] records in (group, layer, graph, node) order, group-layers padded to
                           //         whole blocks; padding records have node = -1
    int64_t total;
};

__host__ __device__ inline DfLayout df_layout_words(int64_t N, int64_t B, int G) {
    DfLayout L;
    int64_t o = 16;
    auto take = [&](int64_t n) { int64_t r = o; o = dagnn_align4(o + n); return r; };
    L.grp_of = take(B);
    L.gdepth = take(G);
    L.gload = take(G);
    L.loff = take(G + 1);
    for (int d = 0; d < 2; ++d) L.gtab[d] = take(2 * (int64_t)G);
    for (int d = 0; d < 2; ++d) L.lcnt[d] = take(N + G + 1);
    for (int d = 0; d < 2; ++d) L.glbase[d] = take(N + B);
    for (int d = 0; d < 2; ++d) L.grec[d] = take(16 * (4 * N + 4));
    L.total = o;
    return L;
}


// ---------------------------------------------------------------- LPT assignment (one wave, lane = group)
// minimum over the 64 lanes as a wave-uniform value: inclusive row scans (row_shr 1, 2, 4, 8), then the last lane of a
// row into the next row (row_bcast15) and of the first half into the second (row_bcast31); a lane without a source
// keeps its own value (old operand = the value, bound_ctrl off)
__device__ __forceinline__ unsigned df_wave_umin(unsigned v) {
#define DF_DPP_MIN(ctrl, rmask) \
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    DF_DPP_MIN(0x111, 0xf); DF_DPP_MIN(0x112, 0xf); DF_DPP_MIN(0x114, 0xf); DF_DPP_MIN(0x118, 0xf);
    DF_DPP_MIN(0x142, 0xa); DF_DPP_MIN(0x143, 0xc);
#undef DF_DPP_MIN
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Graphs in order of decreasing depth (plan items), each to the group whose load it raises the least; load_k = c_layer *
// (depth of the first = deepest graph of k) + c_row * (nodes of k); writes grp_of / gdepth / gload / loff and the header
// words of the schedule workspace `ws`.
// The chain itself: `count` graphs staged in schedule order in s_g / s_d / s_n (graph, max depth of the two directions, nodes)
// when `staged`, else read through `items` (B > CAP).  The 64 lanes of ONE wave; `n_total` = nodes of the batch.
__device__ __forceinline__ void df_assign_chain(const int32_t* items, const int32_t* depth0, const int32_t* depth1,
                                                const int32_t* node_ptr, int32_t* ws, const DfLayout& S,
                                                int B, int G, int c_layer, int c_row, const int32_t* s_g, const int32_t* s_d, const int32_t* s_n,
                                                const bool staged, const int count, const long long n_total) {
    const int lane = threadIdx.x;
    long long load = 0;
    int depth = 0;
    bool empty = true;
    const int steps = staged ? count : 2 * B;
    // fast form of the B-step chain when (cost << 6 | group) fits 32 bits: ONE wave minimum per graph gives the least
    // load and, through the low bits, the lowest group that has it (38 -> 17 us at B = 128)
    const long long bound = (long long)c_row * n_total + (long long)c_layer * (staged && count > 0 ? s_d[0] : 0);
    // every graph has the same node count (the D-VAE batches: dvae/dagnn.py:150-158 hard-codes that stride): no B-step chain -
    // the depth-sorted graphs are dealt round-robin (graph j of the order -> group j mod G), all lanes at once
    bool uniform = staged && count > 0;
    if (uniform) {
        int lo = 0x7fffffff, hi = 0;
        for (int j = lane; j < count; j += 64) { lo = min(lo, s_n[j]); hi = max(hi, s_n[j]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
        uniform = lo == hi;
    }
    if (uniform) {
        const int ng = s_n[0];
        for (int j = lane; j < count; j += 64) ws[S.grp_of + s_g[j]] = j % G;
        if (lane < G && lane < count) {
            depth = s_d[lane];
            empty = false;
            load = (long long)c_layer * depth + (long long)c_row * ng * ((count - lane + G - 1) / G);
        }
    } else if (staged && bound < (1ll << 25)) {
        unsigned load32 = 0;
#pragma unroll 4
        for (int j = 0; j < count; ++j) {
            const int g = s_g[j], dg = s_d[j], ng = s_n[j];
            const unsigned cand = load32 + (unsigned)(c_row * ng) + (empty ? (unsigned)(c_layer * dg) : 0u);
            const unsigned best = df_wave_umin(lane < G ? (cand << 6) | (unsigned)lane : 0xffffffffu);
            const int k = (int)(best & 63u);
            if (lane == k) {
                load32 = cand;
                if (empty) { depth = dg; empty = false; }
            }
            if (lane == 0) ws[S.grp_of + g] = k;
        }
        load = load32;
    } else
    for (int j = 0; j < steps; ++j) {
        int g, dg, ng;
        if (staged) {
            g = s_g[j]; dg = s_d[j]; ng = s_n[j];
        } else {
            const int it = items[j];
            if (it & 1) continue;
            g = it >> 1;
            dg = max(depth0[g], depth1[g]);
            ng = node_ptr[g + 1] - node_ptr[g];
        }
        long long cand = load + (long long)c_row * ng + (empty ? (long long)c_layer * dg : 0);
        if (lane >= G) cand = 0x7fffffffffffffffLL;
        // wave minimum of the 64-bit candidates, high words first (two DPP scans instead of six 64-bit shuffles through
        // the LDS crossbar on the B-step dependent chain: 57 -> 20 us at B = 128)
        const unsigned hi = (unsigned)((unsigned long long)cand >> 32), lo = (unsigned)cand;
        const unsigned best_hi = df_wave_umin(hi);
        const unsigned best_lo = df_wave_umin(hi == best_hi ? lo : 0xffffffffu);
        const unsigned long long m = __ballot(hi == best_hi && lo == best_lo);
        const int k = __ffsll((long long)m) - 1;
        if (lane == k) {
            load = cand;
            if (empty) { depth = dg; empty = false; }
        }
        if (lane == 0) ws[S.grp_of + g] = k;
    }
    if (lane < G) { ws[S.gdepth + lane] = depth; ws[S.gload + lane] = (int)(load > 0x7fffffff ? 0x7fffffff : load); }
    // loff = exclusive prefix of (depth_k + 1)
    int x = lane < G ? depth + 1 : 0;
    const int own = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane < G) ws[S.loff + lane] = x - own;
    if (lane == G - 1) ws[S.loff + G] = x;
    if (lane == 0) { ws[0] = G; ws[1] = DF_MAGIC; ws[2] = DF_RB; }
}

// Staging + chain, called by the 64 lanes of ONE wave (threadIdx.x < 64).  s_g / s_d / s_n: LDS staging of CAP words each.
// The plan's tables (items [2B], depth of either direction [B], node_ptr [B + 1]) come as pointers: global memory for
// df_assign_kernel (dataflow.hip), LDS copies for the one-workgroup build of small batches (small.hip).
template <int CAP>
__device__ __forceinline__ void df_assign_wave(const int32_t* items, const int32_t* depth0, const int32_t* depth1,
                                               const int32_t* node_ptr, int32_t* ws, const DfLayout& S,
                                               int B, int G, int c_layer, int c_row, int32_t* s_g, int32_t* s_d, int32_t* s_n) {
    const bool staged = B <= CAP;
    const int lane = threadIdx.x;
    // compact the direction-0 entries in order (wave-level prefix over 64 entries at a time)
    int count = 0;
    if (staged) {
        for (int j0 = 0; j0 < 2 * B; j0 += 64) {
            const int j = j0 + lane;
            const int it = j < 2 * B ? items[j] : 1;
            const bool keep = !(it & 1);
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                const int g = it >> 1;
                s_g[pos] = g;
                s_d[pos] = max(depth0[g], depth1[g]);
                s_n[pos] = node_ptr[g + 1] - node_ptr[g];
            }
            count += __popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    df_assign_chain(items, depth0, depth1, node_ptr, ws, S, B, G, c_layer, c_row, s_g, s_d, s_n, staged, count, (long long)node_ptr[B]);
}

// The reverse sweep's static record of a (cell, node) (csrc/bwd_dataflow.hip): eight rows - external gradient, state, the
// five gate coefficients and c_q - of 256 floats in lane order (lane l holds columns {l, 64 + l, 128 + l, 192 + l} at
// [4 l, 4 l + 4)); H = 320 adds part B behind them: eight 64-float rows of the columns 256 + l.  The forward kernel's training
// epilogue (dataflow.hip) writes rows 1..7, the reverse pass's preparation row 0.
constexpr int DF_NSTAT = 8;
constexpr int DF_STAT_SP = 256;
__host__ __device__ constexpr int df_stat_floats(int H) { return DF_NSTAT * (H > 256 ? 320 : 256); }
enum { DF_ST_GEXT = 0, DF_ST_H = 1, DF_ST_CR = 2, DF_ST_CZ = 3, DF_ST_CNR = 4, DF_ST_CN = 5, DF_ST_Z = 6, DF_ST_CQ = 7 };

}  // namespace
