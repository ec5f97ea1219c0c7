// small.hip - plan and dataflow schedule of a SMALL batch, each by ONE workgroup with every table in LDS.
//
// Reference path replaced: the same as plan.hip (frontier selection dagnn.py:146-147, the per-node edge scan :151-157,
// `_get_output_nodes` :119-126 - and dvae/dagnn.py:112-125 for the D-VAE batches this path is made for).
//
// Why.  plan.hip + dataflow.hip's schedule kernels are 13 launches; every one is a few dependent round trips to memory
// plus ~4 us of launch, and on a 64-graph D-VAE batch (512 nodes) they were 96 us of a 0.22 ms forward pass
// (profiles/README.md, cfg 1 / cfg 4).  Up to PS_N nodes / PS_E edges / PS_B graphs the whole build fits one CU's LDS:
// one workgroup of 1024 threads walks the same phases with workgroup barriers in place of launches, and writes the
// SAME words (tests: host mirror == general build == this build, word for word).
// The stable placements (nodes by layer, edges by row) count the earlier items of the same graph with the same key by
// a plain loop over LDS: graphs of such batches are tens of nodes; the loop is bounded by the graph, the cliff of a
// degenerate batch (one 2048-node graph) is ~0.1 ms.
#include "df_common.h"

namespace {

constexpr int PS_T = 1024;   // threads
constexpr int PS_N = 2048;   // nodes
constexpr int PS_E = 4096;   // edges
constexpr int PS_B = 512;    // graphs
constexpr int PS_F = PS_N + PS_B + 1;   // flat per-graph tables (n_g + 1 words per graph)
constexpr int PS_PER = 3;    // elements per thread of a workgroup scan (3 * 1024 >= PS_F)
static_assert(PS_PER * PS_T >= PS_F && PS_PER * PS_T >= PS_N + 2, "scan width");
typedef unsigned short u16;

// phase stamps (100 MHz) of thread 0, only in a build with -DPS_STAMPS (scripts/small_stamps.py): the plan kernel's go to the
// plan's cursor scratch, the schedule kernel's to the last 64 words of the workspace
#ifdef PS_STAMPS
#define PS_STAMP(buf, k) do { if (threadIdx.x == 0) (buf)[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PS_STAMP(buf, k) do { } while (0)
#endif

// inclusive sum scan over the 64 lanes on DPP (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast15 / row_bcast31):
// six VALU operations - the same scan on __shfl_up is six round trips through the LDS crossbar, and these kernels are
// chains of such scans
__device__ __forceinline__ int ps_wave_scan(int x) {
#define PS_DPP_ADD(ctrl, rmask) x += __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, true)
    PS_DPP_ADD(0x111, 0xf); PS_DPP_ADD(0x112, 0xf); PS_DPP_ADD(0x114, 0xf); PS_DPP_ADD(0x118, 0xf);
    PS_DPP_ADD(0x142, 0xa); PS_DPP_ADD(0x143, 0xc);
#undef PS_DPP_ADD
    return x;
}
__device__ __forceinline__ int ps_wave_total(int scanned) { return __builtin_amdgcn_readlane(scanned, 63); }

// number of elements of keys[lo, hi) equal to `key` (u16 keys in LDS, 8-byte aligned array): four keys per LDS read
__device__ __forceinline__ int ps_count_equal(const unsigned short* keys, int lo, int hi, unsigned key) {
    int cnt = 0;
    for (int q = lo & ~3; q < hi; q += 4) {
        const uint2 w = *reinterpret_cast<const uint2*>(keys + q);
        cnt += (q >= lo && (w.x & 0xffffu) == key) + (q + 1 >= lo && q + 1 < hi && (w.x >> 16) == key) +
               (q + 2 >= lo && q + 2 < hi && (w.y & 0xffffu) == key) + (q + 3 >= lo && q + 3 < hi && (w.y >> 16) == key);
    }
    return cnt;
}

// Workgroup barrier for data exchanged through LDS ONLY: waits for this wave's LDS operations, not for its global stores
// (__syncthreads() also drains those - a store round trip per barrier, and these kernels are a dozen barriers long).
__device__ __forceinline__ void ps_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive scans of a0[0..n0) and a1[0..n1) in place (LDS), both at once; every thread of the workgroup calls.
__device__ __forceinline__ void ps_scan2(int32_t* a0, int n0, int32_t* a1, int n1, int32_t (*wsum)[PS_T / 64]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = tid * PS_PER;
    int v0[PS_PER], v1[PS_PER];
#pragma unroll
    for (int k = 0; k < PS_PER; ++k) {
        v0[k] = i0 + k < n0 ? a0[i0 + k] : 0;
        v1[k] = i0 + k < n1 ? a1[i0 + k] : 0;
        if (k) { v0[k] += v0[k - 1]; v1[k] += v1[k - 1]; }
    }
    const int x0 = ps_wave_scan(v0[PS_PER - 1]), x1 = ps_wave_scan(v1[PS_PER - 1]);
    if (lane == 63) { wsum[0][wave] = x0; wsum[1][wave] = x1; }
    ps_barrier();
    int b0 = x0 - v0[PS_PER - 1], b1 = x1 - v1[PS_PER - 1];
    for (int w = 0; w < wave; ++w) { b0 += wsum[0][w]; b1 += wsum[1][w]; }
#pragma unroll
    for (int k = 0; k < PS_PER; ++k) {
        if (i0 + k < n0) a0[i0 + k] = b0 + v0[k];
        if (i0 + k < n1) a1[i0 + k] = b1 + v1[k];
    }
    ps_barrier();
}

// ------------------------------------------------------------------------------------------------ the plan
__global__ void __launch_bounds__(PS_T) plan_small_kernel(int32_t* plan, PlanLayout L, const int64_t* __restrict__ edge_index,
                                                           const int64_t* __restrict__ layer_fwd,
                                                           const int64_t* __restrict__ layer_bwd,
                                                           const int64_t* __restrict__ batch,
                                                           const float* __restrict__ edge_attr, int N, int E, int B, int R,
                                                           int32_t* status) {
    __shared__ u16 s_gof[PS_N];            // graph of every node
    __shared__ __attribute__((aligned(16))) u16 s_layer[2][PS_N];       // clamped layer of every node
    __shared__ int32_t s_nptr[PS_B + 1], s_eptr[PS_B + 1];
    __shared__ int32_t s_depth[2][PS_B];
    __shared__ int32_t s_ls[2][PS_F];      // lstart, flat: graph g's depth_g + 1 entries at node_ptr[g] + g
    __shared__ int32_t s_rp[2][PS_F];      // rowptr, flat: graph g's n_g + 1 entries at node_ptr[g] + g
    __shared__ u16 s_pos[2][PS_N];         // sorted position of every node
    __shared__ u16 s_order[2][PS_N];       // node at every sorted position
    __shared__ int32_t s_bl[2][PS_N + 2];  // batch-level layer offsets
    __shared__ __attribute__((aligned(16))) u16 s_edge[2][PS_E];        // [0] sources, [1] targets; dead after the edge placement: ...
    __shared__ u16 s_col[2][PS_E], s_eid[2][PS_E];   // predecessor / original edge per CSR slot
    __shared__ __attribute__((aligned(16))) int32_t s_key[2 * PS_B + 4];
    __shared__ int32_t s_wsum[2][PS_T / 64];
    __shared__ int32_t s_bad, s_T[2], s_thr[2];
    u16 (*s_lb)[PS_F] = reinterpret_cast<u16 (*)[PS_F]>(&s_edge[0][0]);   // ... lbase lives there afterwards
    static_assert(sizeof(u16) * 2 * PS_F <= sizeof(u16) * 2 * PS_E, "lbase aliases the edge lists");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = N + B;   // words of a flat table
    const int64_t* const layer_of[2] = {layer_fwd, layer_bwd};
    [[maybe_unused]] unsigned long long* stamp = reinterpret_cast<unsigned long long*>(plan + L.cursor[0]);
    PS_STAMP(stamp, 0);

    // ---- phase A: every global load of the kernel is issued here, in one trip to memory (batch vector, edge lists, layer
    // ids; clamped indices: the loads are unconditional); header, zeroed tables
    constexpr int NPER = PS_N / PS_T, EPER = PS_E / PS_T, LPER = 2 * PS_N / PS_T;
    int64_t r_b[NPER], r_bp[NPER], r_s[EPER], r_t[EPER], r_l[LPER];
#pragma unroll
    for (int q = 0; q < NPER; ++q) {
        const int i = min(tid + q * PS_T, N - 1);
        r_b[q] = batch[i];
        r_bp[q] = batch[max(i - 1, 0)];
    }
#pragma unroll
    for (int q = 0; q < EPER; ++q) {
        const int e = min(tid + q * PS_T, max(E - 1, 0));
        r_s[q] = E > 0 ? edge_index[e] : 0;
        r_t[q] = E > 0 ? edge_index[(int64_t)E + e] : 0;
    }
#pragma unroll
    for (int q = 0; q < LPER; ++q) {
        const int it = min(tid + q * PS_T, 2 * N - 1);
        const int d = it >= N;
        r_l[q] = layer_of[d][it - d * N];
    }
    if (tid == 0) {
        plan[PH_N] = N; plan[PH_E] = E; plan[PH_B] = B; plan[PH_R] = R; plan[PH_MAGIC] = DAGNN_PLAN_MAGIC;
        s_bad = 0;   // (the status word is WRITTEN at the end, not accumulated: the caller need not clear it for this build)
        s_T[0] = s_T[1] = 0; s_thr[0] = s_thr[1] = 0;
    }
    for (int i = tid; i < 2 * PS_F; i += PS_T) { (&s_ls[0][0])[i] = 0; (&s_rp[0][0])[i] = 0; }
    for (int i = tid; i < 2 * PS_B; i += PS_T) (&s_depth[0][0])[i] = 0;
    for (int i = tid; i < 2 * (PS_N + 2); i += PS_T) (&s_bl[0][0])[i] = 0;
    for (int i = tid; i <= B; i += PS_T) { s_nptr[i] = 0; s_eptr[i] = 0; }
    int bad = 0;
#pragma unroll
    for (int q = 0; q < NPER; ++q) {
        const int i = tid + q * PS_T;
        if (i < N) {
            const int64_t b = r_b[q];
            if (i > 0 && b < r_bp[q]) bad |= 4;
            if (b < 0 || b >= B) bad |= 4;
            s_gof[i] = (u16)(b < 0 ? 0 : (b >= B ? B - 1 : b));
        }
    }
    bool e_ok[EPER];
#pragma unroll
    for (int q = 0; q < EPER; ++q) {
        const int e = tid + q * PS_T;
        e_ok[q] = r_s[q] >= 0 && r_s[q] < N && r_t[q] >= 0 && r_t[q] < N;
        if (e < E) {
            if (!e_ok[q]) bad |= 2;
            s_edge[0][e] = (u16)(e_ok[q] ? r_s[q] : 0);
            s_edge[1][e] = (u16)(e_ok[q] ? r_t[q] : 0);
        }
    }
    ps_barrier();
    PS_STAMP(stamp, 1);
    // contract checks on the staged copies (plan_ptr_kernel; graph ids are in range or bit 2 is up already)
#pragma unroll
    for (int q = 0; q < EPER; ++q) {
        const int e = tid + q * PS_T;
        if (e < E && e_ok[q]) {
            const int bs = s_gof[s_edge[0][e]];
            if (bs != (int)s_gof[s_edge[1][e]]) bad |= 2;
            if (e > 0 && (int)s_gof[s_edge[0][e - 1]] > bs) bad |= 1;
        }
    }
    if (bad && status) atomicOr(&s_bad, bad);
    // node_ptr[g] = first node of a graph >= g, edge_ptr[g] = first edge whose source is in a graph >= g: node i (edge e)
    // is that first one for every g in (graph of its predecessor, its own graph]
    for (int i = tid; i <= N; i += PS_T) {
        const int prev = i > 0 ? (int)s_gof[i - 1] : -1, cur = i < N ? (int)s_gof[i] : B;
        for (int g = prev + 1; g <= cur; ++g) s_nptr[g] = i;
    }
    for (int e = tid; e <= E; e += PS_T) {
        const int prev = e > 0 ? (int)s_gof[s_edge[0][e - 1]] : -1, cur = e < E ? (int)s_gof[s_edge[0][e]] : B;
        for (int g = prev + 1; g <= cur; ++g) s_eptr[g] = e;
    }
    ps_barrier();
    for (int i = tid; i <= B; i += PS_T) { plan[L.node_ptr + i] = s_nptr[i]; plan[L.edge_ptr + i] = s_eptr[i]; }
    const bool broken = (s_bad & 7) != 0;   // contract violated: the tables would be garbage - nothing below walks them

    PS_STAMP(stamp, 2);
    if (!broken) {
        // ---- phase B: layers, depths, histograms (plan_graph_kernel, first half)
        // (a D-VAE batch has ten layers: the batch-level counters are bumped once per wave and distinct layer, not once
        // per node - same-address LDS atomics of a wave are served one lane after the other)
#pragma unroll
        for (int q = 0; q < LPER; ++q) {
            const int it = tid + q * PS_T;
            const bool on = it < 2 * N;
            int key = -1;
            if (on) {
                const int d = it >= N, v = it - d * N;
                const int g = s_gof[v], n0 = s_nptr[g], n = s_nptr[g + 1] - n0;
                int64_t l = r_l[q];
                if (l < 0 || l >= n) { if (status) atomicOr(&s_bad, 8); l = l < 0 ? 0 : n - 1; }
                s_layer[d][v] = (u16)l;
                atomicMax(&s_depth[d][g], (int)l + 1);
                atomicAdd(&s_ls[d][n0 + g + (int)l + 1], 1);
                key = d * (PS_N + 2) + (int)l + 1;   // flat index into s_bl
            }
            if (q * PS_T < 2 * N) {   // (uniform: the ballots below need every lane of the waves that take part)
                unsigned long long todo = __ballot(on);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int k = __builtin_amdgcn_readlane(key, leader);
                    const unsigned long long m = __ballot(key == k);
                    if (lane == leader) {
                        atomicAdd(&(&s_bl[0][0])[k], __popcll(m));
                        const int dd = k >= PS_N + 2;
                        atomicMax(&s_T[dd], k - dd * (PS_N + 2));   // (l + 1 of this key)
                    }
                    todo &= ~m;
                }
            }
        }
        ps_barrier();
        for (int i = tid; i < 2 * B; i += PS_T) {
            const int d = i >= B, g = i - d * B;
            plan[L.depth[d] + g] = s_depth[d][g];
        }
        PS_STAMP(stamp, 3);
        ps_scan2(s_ls[0], F, s_ls[1], F, s_wsum);
        PS_STAMP(stamp, 4);
        // lstart: entries 0 .. depth_g of every graph
        for (int i = tid; i < 2 * B; i += PS_T) {
            const int d = i >= B, g = i - d * B, j = s_nptr[g] + g;
            plan[L.lstart[d] + j] = s_ls[d][j];
        }
        for (int it = tid; it < 2 * N; it += PS_T) {
            const int d = it >= N, v = it - d * N;
            const int g = s_gof[v], n0 = s_nptr[g], i = v - n0 + 1;
            if (i <= s_depth[d][g]) plan[L.lstart[d] + n0 + g + i] = s_ls[d][n0 + g + i];
        }
        PS_STAMP(stamp, 5);
        // ---- phase C: nodes in (graph, layer, id) order
        for (int it = tid; it < 2 * N; it += PS_T) {
            const int d = it >= N, v = it - d * N;
            const int g = s_gof[v], n0 = s_nptr[g];
            const u16 l = s_layer[d][v];
            const int p = s_ls[d][n0 + g + l] + ps_count_equal(s_layer[d], n0, v, l);
            s_pos[d][v] = (u16)p;
            s_order[d][p] = (u16)v;
            plan[L.order[d] + p] = v;
        }
        ps_barrier();
        PS_STAMP(stamp, 6);
        // ---- phase D: rows of the CSR (d = 0: an edge feeds its target, d = 1: its source)
        for (int it = tid; it < 2 * E; it += PS_T) {
            const int d = it >= E, e = it - d * E;
            const int f = s_edge[1 - d][e];
            atomicAdd(&s_rp[d][(int)s_pos[d][f] + (int)s_gof[f] + 1], 1);
        }
        ps_barrier();
        PS_STAMP(stamp, 7);
        ps_scan2(s_rp[0], F, s_rp[1], F, s_wsum);
        PS_STAMP(stamp, 8);
        for (int it = tid; it < 2 * F; it += PS_T) {
            const int d = it >= F, j = it - d * F;
            plan[L.rowptr[d] + j] = s_rp[d][j];
        }
        PS_STAMP(stamp, 9);
        // ---- phase E: edges in (row, original order) order
        for (int it = tid; it < 2 * E; it += PS_T) {
            const int d = it >= E, e = it - d * E;
            const u16 f = s_edge[1 - d][e];
            const int g = s_gof[f];
            const int slot = s_rp[d][(int)s_pos[d][f] + g] + ps_count_equal(s_edge[1 - d], s_eptr[g], e, f);
            const int o = s_edge[d][e];
            s_col[d][slot] = (u16)o;
            s_eid[d][slot] = (u16)e;
            plan[L.col[d] + slot] = o;
            plan[L.eidx[d] + slot] = e;
            float* eattr = reinterpret_cast<float*>(plan + L.eattr[d]);
            for (int r = 0; r < R; ++r) eattr[(int64_t)slot * R + r] = edge_attr[(int64_t)e * R + r];
        }
        PS_STAMP(stamp, 10);
        // ---- phase F: work items (g * 2 + d) by depth, deepest first, ties by index (plan_items_kernel)
        const int nitems = 2 * B;
        for (int i = tid; i < nitems + 4; i += PS_T) s_key[i] = i < nitems ? s_depth[i & 1][i >> 1] : -1;   // (-1: ranks behind every item)
        ps_barrier();   // (also: the edge lists are dead from here on, s_col / s_eid complete)
        {
            int split = 1;   // threads per item: a power of two, <= 64, split * nitems <= PS_T
            while (split < 64 && 2 * split * nitems <= PS_T) split <<= 1;
            const int item = tid / split, part = tid % split;
            const int len = ((nitems + split - 1) / split + 3) & ~3;   // four keys per LDS read
            int rank = 0;
            if (item < nitems) {
                const int ki = s_key[item];
                const int j1 = min(nitems, (part + 1) * len);
                for (int j = part * len; j < j1; j += 4) {
                    const int4 kj = *reinterpret_cast<const int4*>(s_key + j);
                    rank += ((kj.x > ki) || (kj.x == ki && j < item)) + ((kj.y > ki) || (kj.y == ki && j + 1 < item)) +
                            ((kj.z > ki) || (kj.z == ki && j + 2 < item)) + ((kj.w > ki) || (kj.w == ki && j + 3 < item));
                }
            }
            for (int o = 1; o < split; o <<= 1) rank += __shfl_xor(rank, o, 64);
            if (item < nitems && part == 0) plan[L.items + rank] = item;
        }
        PS_STAMP(stamp, 11);
        // ---- phase G: batch-level layers (plan_blptr_kernel); every word of blptr, the tail of blsplit
        const int T0 = s_T[0], T1 = s_T[1];
        ps_scan2(s_bl[0], T0 + 1, s_bl[1], T1 + 1, s_wsum);
        for (int it = tid; it < 2 * (N + 2); it += PS_T) {
            const int d = it >= N + 2, t = it - d * (N + 2);
            const int T = d ? T1 : T0;
            plan[L.blptr[d] + t] = t <= T ? s_bl[d][t] : (t == N + 1 ? T : 0);
            if (t >= T) plan[L.blsplit[d] + t] = 0;
            if (t < T && s_bl[d][t + 1] - s_bl[d][t] > DAGNN_PLAN_THIN_ROWS) atomicMax(&s_thr[d], t + 1);
        }
        ps_barrier();
        if (tid < 2) plan[PH_THR0 + tid] = s_thr[tid];
        PS_STAMP(stamp, 12);
        // ---- phase H: first slot of every (graph, layer): shallow graphs first, then the deep ones (plan_lbase_kernel)
        for (int idx = wave; idx < T0 + T1; idx += PS_T / 64) {
            const int d = idx >= T0, t = idx - d * T0;
            const int thr = s_thr[d];
            int carry = s_bl[d][t];
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1 && lane == 0) plan[L.blsplit[d] + t] = carry;
                for (int g0 = 0; g0 < B; g0 += 64) {
                    const int g = g0 + lane;
                    int cnt = 0, base = 0;
                    bool has = false;
                    if (g < B && t < s_depth[d][g] && (s_depth[d][g] > thr) == (pass == 1)) {
                        base = s_nptr[g] + g + t;
                        cnt = s_ls[d][base + 1] - s_ls[d][base];
                        has = true;
                    }
                    const int x = ps_wave_scan(cnt);
                    if (has) { s_lb[d][base] = (u16)(carry + x - cnt); plan[L.lbase[d] + base] = carry + x - cnt; }
                    carry += ps_wave_total(x);
                }
            }
        }
        ps_barrier();
        PS_STAMP(stamp, 13);
        // ---- phase I: 64-byte row records in slot order (plan_rowrec_kernel)
        for (int it = tid; it < 2 * N; it += PS_T) {
            const int d = it >= N, p = it - d * N;
            const int v = s_order[d][p], g = s_gof[v];
            const int base = s_nptr[g] + g + s_layer[d][v];
            const int slot = (int)s_lb[d][base] + (p - s_ls[d][base]);
            const int eb = s_rp[d][p + g], ee = s_rp[d][p + g + 1];
            int w[16];
            w[0] = v; w[1] = eb; w[2] = ee; w[3] = g;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = eb + q < ee;
                w[4 + q] = ok ? (int)s_col[d][eb + q] : 0;
                const int32_t* ea = reinterpret_cast<const int32_t*>(edge_attr) + (ok ? (int64_t)s_eid[d][eb + q] * R : 0);
                w[8 + 2 * q] = (ok && R >= 1) ? ea[0] : 0;
                w[9 + 2 * q] = (ok && R >= 2) ? ea[1] : 0;
            }
            plan[L.pos[d] + v] = slot;
            int4* out = reinterpret_cast<int4*>(plan + L.rowrec[d] + 16 * (int64_t)slot);
            out[0] = make_int4(w[0], w[1], w[2], w[3]);
            out[1] = make_int4(w[4], w[5], w[6], w[7]);
            out[2] = make_int4(w[8], w[9], w[10], w[11]);
            out[3] = make_int4(w[12], w[13], w[14], w[15]);
        }
    }
    // ---- seal (plan_seal_kernel): a batch that violates the contract leaves an EMPTY plan behind
    ps_barrier();
    PS_STAMP(stamp, 14);
    if (status && s_bad != 0) {
        __syncthreads();   // (the zeros below must land after the values the phases above stored to the same words)
        for (int i = tid; i < B; i += PS_T) { plan[L.depth[0] + i] = 0; plan[L.depth[1] + i] = 0; }
        for (int i = tid; i < N + 2; i += PS_T) {
            plan[L.blptr[0] + i] = 0; plan[L.blptr[1] + i] = 0; plan[L.blsplit[0] + i] = 0; plan[L.blsplit[1] + i] = 0;
        }
    }
    if (tid == 0 && status) status[0] = s_bad;
}

// ------------------------------------------------------------------------------------------------ the dataflow schedule
// dagnn_dataflow_schedule of a small batch (dataflow.hip: df_assign / count / prefix / base / lbase / records kernels).
// Writes the tables in full and the records every group USES (first record .. first record + 4 * blocks, padding
// records node = -1); the unused tail of the record arrays keeps whatever the workspace held - nothing reads it.
__global__ void __launch_bounds__(PS_T) schedule_small_kernel(const int32_t* __restrict__ plan, PlanLayout L, int32_t* ws, DfLayout S,
                                                               int N, int B, int G, int c_layer, int c_row,
                                                               const int32_t* __restrict__ status) {
    if (status && status[0] != 0) return;   // the batch violates the plan contract: nothing here can be trusted
    __shared__ int32_t s_nptr[PS_B + 1];
    __shared__ int32_t s_dep[2][PS_B];
    __shared__ int32_t s_ls[2][PS_F];      // the plan's lstart (flat)
    __shared__ u16 s_lb[2][PS_F];          // the plan's lbase (flat): first rowrec slot of every (graph, layer)
    __shared__ int32_t s_grp[PS_B];
    __shared__ __attribute__((aligned(16))) u16 s_grp16[PS_B];
    __shared__ int32_t s_gcnt[DF_MAX_GROUPS], s_goff[DF_MAX_GROUPS + 1];
    __shared__ u16 s_glist[PS_B];          // graphs in (group, graph) order
    __shared__ int32_t s_gd[DF_MAX_GROUPS], s_loff[DF_MAX_GROUPS + 1];
    __shared__ int32_t s_gtab[2][2 * DF_MAX_GROUPS];
    __shared__ int32_t s_lcnt[2][PS_N + DF_MAX_GROUPS + 1];   // rows per (group, layer), then their padded exclusive prefix
    __shared__ u16 s_cnt[2][PS_N + DF_MAX_GROUPS + 1];        // ... the rows, kept
    __shared__ unsigned char s_kof[PS_N + DF_MAX_GROUPS + 1]; // group of every (group, layer) pair
    __shared__ int32_t s_glb[2][PS_F];
    __shared__ u16 s_gof[PS_N];            // graph of every sorted position (= of every node: graphs are contiguous)
    __shared__ u16 s_gslot[2][PS_N];       // graph of every rowrec slot
    __shared__ u16 s_rec[2][PS_N];         // schedule record of every rowrec slot
    __shared__ int32_t s_items[2 * PS_B];
    __shared__ int32_t s_g[PS_B], s_d[PS_B], s_n[PS_B];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = N + B;
    [[maybe_unused]] unsigned long long* stamp = reinterpret_cast<unsigned long long*>(ws + S.total - 64);
    PS_STAMP(stamp, 0);
    // ---- tables and header of the workspace = 0; stage the plan's per-graph tables (one trip to memory)
    {
        int4* z = reinterpret_cast<int4*>(ws);
        const int nz = (int)(S.grec[0] / 4);
        for (int i = tid; i < nz; i += PS_T) z[i] = make_int4(0, 0, 0, 0);
    }
    for (int i = tid; i <= B; i += PS_T) s_nptr[i] = plan[L.node_ptr + i];
    for (int i = tid; i < 2 * B; i += PS_T) { const int d = i >= B, g = i - d * B; s_dep[d][g] = plan[L.depth[d] + g]; s_items[i] = plan[L.items + i]; }
    for (int it = tid; it < 2 * F; it += PS_T) {
        const int d = it >= F, j = it - d * F;
        s_ls[d][j] = plan[L.lstart[d] + j]; s_lb[d][j] = (u16)plan[L.lbase[d] + j]; s_glb[d][j] = 0;
    }
    for (int it = tid; it < 2 * N; it += PS_T) { const int d = it >= N, r = it - d * N; s_gslot[d][r] = (u16)plan[L.rowrec[d] + 16 * (int64_t)r + 3]; }
    for (int i = tid; i < 2 * (PS_N + DF_MAX_GROUPS + 1); i += PS_T) (&s_lcnt[0][0])[i] = 0;
    if (tid < DF_MAX_GROUPS) s_gcnt[tid] = 0;
    __syncthreads();   // (the zeros must be in memory before the values below go to the same words)
    PS_STAMP(stamp, 1);
    // ---- the LPT assignment (one wave), straight into the workspace
    if (wave == 0) df_assign_wave<PS_B>(s_items, s_dep[0], s_dep[1], s_nptr, ws, S, B, G, c_layer, c_row, s_g, s_d, s_n);
    else   // meanwhile: graph of every position (last g with node_ptr[g] <= p; empty graphs share their successor's offset)
        for (int p = tid - 64; p < N; p += PS_T - 64) {
            int lo = 0, hi = B;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_nptr[mid] <= p) lo = mid; else hi = mid; }
            s_gof[p] = (u16)lo;
        }
    __syncthreads();   // (the assignment is read back from memory)
    PS_STAMP(stamp, 2);
    for (int i = tid; i < B; i += PS_T) { const int k = ws[S.grp_of + i]; s_grp[i] = k; s_grp16[i] = (u16)k; atomicAdd(&s_gcnt[k], 1); }
    for (int i = tid; i < G; i += PS_T) s_gd[i] = ws[S.gdepth + i];
    for (int i = tid; i <= G; i += PS_T) s_loff[i] = ws[S.loff + i];
    ps_barrier();
    PS_STAMP(stamp, 3);
    // ---- rows per (group, layer): position n0 + t of a graph stands for its layer t (a graph has at least as many nodes
    // as layers); group of every (group, layer) pair; graphs in (group, graph) order
    for (int it = tid; it < 2 * N; it += PS_T) {
        const int d = it >= N, p = it - d * N;
        const int g = s_gof[p], n0 = s_nptr[g], t = p - n0, k = s_grp[g];
        if (t < min(s_dep[d][g], s_gd[k])) atomicAdd(&s_lcnt[d][s_loff[k] + t], s_ls[d][n0 + g + t + 1] - s_ls[d][n0 + g + t]);
    }
    for (int k = wave; k < G; k += PS_T / 64)
        for (int t = lane; t <= s_gd[k]; t += 64) s_kof[s_loff[k] + t] = (unsigned char)k;
    if (wave == 0) {
        const int own = lane < G ? s_gcnt[lane] : 0;
        const int x = ps_wave_scan(own);
        if (lane < G) s_goff[lane] = x - own;
        if (lane == 63) s_goff[G] = x;
    }
    ps_barrier();
    for (int g = tid; g < B; g += PS_T) s_glist[s_goff[s_grp[g]] + ps_count_equal(s_grp16, 0, g, s_grp16[g])] = (u16)g;
    PS_STAMP(stamp, 4);
    // ---- per (group, direction): exclusive prefix of the block-padded counts; blocks of the group
    for (int i = wave; i < 2 * G; i += PS_T / 64) {
        const int d = i >= G, k = i - d * G;
        const int depth = s_gd[k];
        int32_t* cnt = s_lcnt[d] + s_loff[k];
        u16* raw = s_cnt[d] + s_loff[k];
        int carry = 0;
        for (int c0 = 0; c0 <= depth; c0 += 64) {
            const int t = c0 + lane;
            const int c = t < depth ? cnt[t] : 0;
            const int padded = (c + DF_RB - 1) / DF_RB * DF_RB;
            const int x = ps_wave_scan(padded);
            if (t <= depth) { cnt[t] = carry + x - padded; raw[t] = (u16)c; }
            carry += ps_wave_total(x);
        }
        if (lane == 0) s_gtab[d][2 * k + 1] = carry / DF_RB;
    }
    ps_barrier();
    PS_STAMP(stamp, 5);
    // ---- first record of every group
    if (wave < 2) {
        const int d = wave;
        const int own = lane < G ? s_gtab[d][2 * lane + 1] * DF_RB : 0;
        const int x = ps_wave_scan(own);
        if (lane < G) s_gtab[d][2 * lane] = x - own;
    }
    ps_barrier();
    PS_STAMP(stamp, 6);
    // ---- glbase, and the padding records (node = -1) behind the rows of every (group, layer)
    const int total = s_loff[G];
    if (B > 16 * G) {   // many graphs per group: a wave per (direction, group, layer), lanes over ALL graphs
        for (int i = wave; i < 2 * total; i += PS_T / 64) {
            const int d = i >= total, pair = i - d * total;
            const int k = s_kof[pair];
            const int t = pair - s_loff[k];
            if (t >= s_gd[k]) continue;   // the table has depth + 1 entries per group
            int carry = s_lcnt[d][pair];
            for (int g0 = 0; g0 < B; g0 += 64) {
                const int g = g0 + lane;
                int cnt = 0, base = 0;
                bool has = false;
                if (g < B && s_grp[g] == k && t < s_dep[d][g]) {
                    base = s_nptr[g] + g + t;
                    cnt = s_ls[d][base + 1] - s_ls[d][base];
                    has = true;
                }
                const int x = ps_wave_scan(cnt);
                if (has) s_glb[d][base] = carry + x - cnt;
                carry += ps_wave_total(x);
            }
        }
    } else {            // a thread per (direction, group, layer) walks the group's graphs
        for (int it = tid; it < 2 * total; it += PS_T) {
            const int d = it >= total, pair = it - d * total;
            const int k = s_kof[pair];
            const int t = pair - s_loff[k];
            if (t >= s_gd[k]) continue;
            int run = s_lcnt[d][pair];
            for (int q = s_goff[k]; q < s_goff[k + 1]; ++q) {
                const int g = s_glist[q];
                if (t < s_dep[d][g]) {
                    const int base = s_nptr[g] + g + t;
                    s_glb[d][base] = run;
                    run += s_ls[d][base + 1] - s_ls[d][base];
                }
            }
        }
    }
    for (int it = tid; it < 2 * total; it += PS_T) {
        const int d = it >= total, pair = it - d * total;
        const int k = s_kof[pair];
        if (pair - s_loff[k] >= s_gd[k]) continue;
        const int first = s_gtab[d][2 * k] + s_lcnt[d][pair] + s_cnt[d][pair], end = s_gtab[d][2 * k] + s_lcnt[d][pair + 1];
        for (int r = first; r < end; ++r) {   // (at most three)
            int4* f = reinterpret_cast<int4*>(ws + S.grec[d]) + 4 * (int64_t)r;
            f[0] = make_int4(-1, -1, -1, -1); f[1] = f[0]; f[2] = f[0]; f[3] = f[0];
        }
    }
    ps_barrier();
    PS_STAMP(stamp, 7);
    // ---- tables out
    for (int it = tid; it < 4 * G; it += PS_T) { const int d = it >= 2 * G, j = it - d * 2 * G; ws[S.gtab[d] + j] = s_gtab[d][j]; }
    for (int it = tid; it < 2 * total; it += PS_T) { const int d = it >= total, j = it - d * total; ws[S.lcnt[d] + j] = s_lcnt[d][j]; }
    for (int it = tid; it < 2 * F; it += PS_T) { const int d = it >= F, j = it - d * F; ws[S.glbase[d] + j] = s_glb[d][j]; }
    PS_STAMP(stamp, 8);
    // ---- every row record to its place in the group order: the record of every rowrec slot (its layer: the last one whose
    // first slot is not behind it), then four lanes per record copy its 64 bytes
    for (int it = tid; it < 2 * N; it += PS_T) {
        const int d = it >= N, r = it - d * N;
        const int g = s_gslot[d][r], j = s_nptr[g] + g;
        int lo = 0, hi = s_dep[d][g];
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)s_lb[d][j + mid] <= r) lo = mid; else hi = mid; }
        s_rec[d][r] = (u16)(s_gtab[d][2 * s_grp[g]] + s_glb[d][j + lo] + (r - (int)s_lb[d][j + lo]));
    }
    ps_barrier();
    {
        constexpr int PER = 2 * PS_N * 4 / PS_T / 2;   // two halves of PER loads each
        for (int half = 0; half < 2; ++half) {
            int4 v[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int it = min(tid + (half * PER + q) * PS_T, 8 * N - 1);   // (clamped: the loads are unconditional)
                const int d = it >= 4 * N, j = it - d * 4 * N;
                v[q] = reinterpret_cast<const int4*>(plan + L.rowrec[d])[j];
            }
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int it = tid + (half * PER + q) * PS_T;
                if (it < 8 * N) {
                    const int d = it >= 4 * N, j = it - d * 4 * N;
                    reinterpret_cast<int4*>(ws + S.grec[d])[4 * (int64_t)s_rec[d][j >> 2] + (j & 3)] = v[q];
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    PS_STAMP(stamp, 9);
}

}  // namespace

// 1 when this batch takes the one-workgroup builds (host-side decision: sizes only)
extern "C" int dagnn_plan_is_small(int64_t N, int64_t E, int64_t B) {
    return N >= 1 && B >= 1 && N <= PS_N && E <= PS_E && B <= PS_B;
}

int dagnn_plan_build_small(const dagnn_plan* pl, const int64_t* edge_index, const int64_t* layer_fwd, const int64_t* layer_bwd,
                           const int64_t* batch, const float* edge_attr, int32_t* status, hipStream_t stream) {
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipLaunchKernelGGL(plan_small_kernel, dim3(1), dim3(PS_T), 0, stream, (int32_t*)pl->data, L, edge_index, layer_fwd, layer_bwd,
                       batch, edge_attr, (int)pl->N, (int)pl->E, (int)pl->B, pl->num_edge_feats, status);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

int dagnn_dataflow_schedule_small(const dagnn_plan* pl, int32_t* ws, int groups, int cost_layer, int cost_row,
                                  const int32_t* status, hipStream_t stream) {
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    const DfLayout S = df_layout_words(pl->N, pl->B, groups);
    hipLaunchKernelGGL(schedule_small_kernel, dim3(1), dim3(PS_T), 0, stream, (const int32_t*)pl->data, L, ws, S, (int)pl->N,
                       (int)pl->B, groups, cost_layer, cost_row, status);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
