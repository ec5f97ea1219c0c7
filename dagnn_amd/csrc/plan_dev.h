// plan_dev.h - device bodies of the plan build (csrc/plan.hip: one kernel each; csrc/prepare.hip: several bodies per
// launch, picked by workgroup id).  A body takes its workgroup coordinates as arguments and uses threadIdx / blockDim only.
//
// Replaces, per forward call, the reference's `layer = ids[layer_id == l_idx]` (ogbg-code/model/dagnn.py:146-147), its
// per-frontier-node scan of the whole edge_index (dagnn.py:151-157, O(N*E) per direction) and `_get_output_nodes`
// (dagnn.py:119-126).
#pragma once
#include "common.h"

namespace {

constexpr int PB = 256;  // threads per plan workgroup
constexpr int PLAN_NMAX = 2048;  // graphs up to this many nodes are planned in LDS
constexpr int PLAN_FAST_N = 4 * PB, PLAN_FAST_E = 8 * PB;   // ... and those up to this many nodes and edges (<= 2 edge features) on the FAST form of plan_graph_impl
constexpr int PLAN_GRAPH_LDS = 4 * (PLAN_FAST_N + 1) + 8 * (PLAN_FAST_N + 1);   // words: its four arrays + the key masks (> 4 (PLAN_NMAX + 1))
static_assert(PLAN_GRAPH_LDS >= 4 * (PLAN_NMAX + 1) && (4 * (PLAN_FAST_N + 1)) % 4 == 0, "LDS layout of plan_graph_body");

// node_ptr / edge_ptr + contract checks; thread i of max(N + 2, E, B + 1).
// node_ptr[k] = first node whose graph id is >= k, edge_ptr[k] = first edge whose SOURCE node belongs to a graph >= k: the
// element at which the (sorted) graph id steps from gp to g writes the entries gp + 1 .. g, the last element the entries
// behind its own graph - two dependent loads per thread (a binary search per graph took 14 + 28 of them: 15 us).  On a
// batch that violates the contract (unsorted ids: status bits 0 / 2) some entries keep what the buffer held; nothing walks
// the tables then (plan_seal_body).
__device__ __forceinline__ void plan_ptr_body(int32_t* plan, const PlanLayout& L, const int64_t* __restrict__ edge_index,
                                              const int64_t* __restrict__ batch, int64_t N, int64_t E, int64_t B, int R,
                                              int32_t* status, int64_t i) {
    if (i == 0) {
        plan[PH_N] = (int32_t)N; plan[PH_E] = (int32_t)E; plan[PH_B] = (int32_t)B; plan[PH_R] = R;
        plan[PH_MAGIC] = DAGNN_PLAN_MAGIC;
    }
    int bad = 0;
    if (i < N) {
        const int64_t g = batch[i], gp = i > 0 ? batch[i - 1] : -1;
        if (g < 0 || g >= B) bad |= 4;
        if (g < gp) bad |= 4;
        const int64_t hi = g < B ? g : B;                           // (clamped: an id out of range writes nothing out of bounds)
        for (int64_t k = gp < -1 ? 0 : gp + 1; k <= hi; ++k) plan[L.node_ptr + k] = (int32_t)i;
        if (i == N - 1) for (int64_t k = (g < -1 ? -1 : g) + 1; k <= B; ++k) plan[L.node_ptr + k] = (int32_t)N;
    } else if (N == 0 && i <= B) plan[L.node_ptr + i] = 0;
    if (i < E) {
        const int64_t s = edge_index[i], t = edge_index[E + i];
        int64_t g = -1, gp = -1;
        if (s < 0 || s >= N || t < 0 || t >= N) bad |= 2;
        else {
            g = batch[s];
            if (g != batch[t]) bad |= 2;
        }
        if (i > 0) { const int64_t sp = edge_index[i - 1]; gp = (sp >= 0 && sp < N) ? batch[sp] : -1; }
        if (g >= 0 && gp > g) bad |= 1;
        if (g >= 0) {
            const int64_t hi = g < B ? g : B;
            for (int64_t k = gp < -1 ? 0 : gp + 1; k <= hi; ++k) plan[L.edge_ptr + k] = (int32_t)i;
        }
        if (i == E - 1) for (int64_t k = (g < -1 ? -1 : g) + 1; k <= B; ++k) plan[L.edge_ptr + k] = (int32_t)E;
    } else if (E == 0 && i <= B) plan[L.edge_ptr + i] = 0;
    if (i < N + 2) { plan[L.blptr[0] + i] = 0; plan[L.blptr[1] + i] = 0; plan[L.blsplit[0] + i] = 0; plan[L.blsplit[1] + i] = 0; }
    if (bad && status) atomicOr(status, bad);
}

// Block-wide inclusive scan of a[0..n) in place (global memory owned by this workgroup), plus `base`.
__device__ void block_scan_inplace(int32_t* a, int n, int base, int32_t* lds /* PB+1 */) {
    const int tid = threadIdx.x;
    int carry = base;
    for (int c0 = 0; c0 < n; c0 += PB) {
        int i = c0 + tid;
        int v = (i < n) ? a[i] : 0;
        // wave inclusive scan
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if ((tid & 63) >= o) x += y; }
        if ((tid & 63) == 63) lds[tid >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (tid >> 6); ++w) woff += lds[w];
        int tot = 0;
        for (int w = 0; w < PB / 64; ++w) tot += lds[w];
        if (i < n) a[i] = carry + woff + x;
        carry += tot;
        __syncthreads();
    }
}

// Stable placement of one chunk of PB elements: thread `tid` holds `key` (a cursor index, or -1) and
// gets slot = cur[key] + (number of earlier threads of the chunk with the same key); cur[] advances
// by the chunk's count per key.  Inside a wave the rank comes from ballots over the wave's distinct
// keys (AMD has no match-any instruction: one ballot per distinct key, a few dozen at most); the four
// waves take turns on the cursors so that the order across waves is the thread order.
__device__ __forceinline__ int chunk_place(int key, int32_t* cur) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank = 0, count = 0;
    bool last = false;
    // one trip per distinct key of the wave, everything wave-uniform but the three results: the leader's key by v_readlane
    // (lane index in an SGPR - `__shfl` with a lane it cannot prove uniform is a ds_bpermute, ~100 cycles on the loop's
    // dependent chain: 5-6 us per chunk of mostly distinct keys, 40 of plan_graph's 49 us on the headline batch), members
    // by one compare, the rank by v_mbcnt
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int k = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long m = __ballot(key == k);
        if (key == k) {
            rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            count = __popcll(m);
            last = lane == 63 - __builtin_clzll(m);  // highest lane holding this key
        }
        todo &= ~m;
    }
    int slot = -1;
    for (int w = 0; w < PB / 64; ++w) {
        if (wave == w && key >= 0) {
            const int base = cur[key];  // every lane of the wave reads before the single writer below stores
            slot = base + rank;
            if (last) cur[key] = base + count;
        }
        __syncthreads();
    }
    return slot;
}

// One workgroup (PB threads) per (graph g, direction d) - graphs are independent, a PyG batch stores each graph's
// nodes and edges contiguously - does two stable counting sorts in its own slice of the plan:
// nodes by layer (frontiers become contiguous ranges) and edges by the sorted position of the
// node they feed (rows of the CSR, original edge order kept inside a row so that the softmax /
// weighted sum adds in the same order as the reference's scan produces).  Everything is int32;
// the int64 inputs are narrowed on the way in.
// FAST (graphs of <= 4 PB nodes and <= 8 PB edges, <= 2 edge features: every AST of ogbg-code2 but a handful): the workgroup
// reads ALL its inputs in one go - four layer ids and eight (feeding node, other node, features) per thread, held in registers -
// and its ~15 dependent passes then touch LDS only; the general form re-reads them pass by pass (a round trip to memory in
// five of the passes, two more per chunk of edges: 49 us for the 657-node graph of the headline batch, 2/3 of it waiting).
// The same placement for a FAST graph (keys < 4 PB), without the trip per distinct key: every lane ORs its bit into the
// 256-bit mask of its key (one 64-bit word per wave: four waves, no conflict, no order), a barrier later the four words of
// a key ARE the ballot of the whole chunk - rank = set bits below this thread, the highest set bit advances the cursor and
// clears the words.  Three barriers and three LDS round trips per chunk whatever the keys (the loop above: ~100 ns per
// distinct key, and an AST's 64 consecutive nodes / edges have ~64 distinct keys: 4-6 us per chunk, 31 of plan_graph's 37 us
// on the headline batch's largest graph).  `masks`: [keys][4] words, zero on entry and on exit.
__device__ __forceinline__ int chunk_place_masks(int key, int32_t* cur, unsigned long long* masks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (key >= 0) atomicOr(&masks[key * 4 + wave], 1ull << lane);
    __syncthreads();
    int slot = -1, count = 0;
    bool last = false;
    if (key >= 0) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(masks + key * 4), b = *reinterpret_cast<const ulonglong2*>(masks + key * 4 + 2);
        const unsigned long long m[4] = {a.x, a.y, b.x, b.y};
        int before = 0, hw = 0;
        unsigned long long mine = 0ull, top = 0ull;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = __popcll(m[w]);
            if (w < wave) before += c;
            if (w == wave) mine = m[w];
            if (m[w]) { hw = w; top = m[w]; }
            count += c;
        }
        const int rank = before + __popcll(mine & ((1ull << lane) - 1ull));
        last = wave == hw && lane == 63 - __builtin_clzll(top);
        slot = cur[key] + rank;
    }
    __syncthreads();   // every thread has read its cursor and its masks
    if (last) {
        cur[key] += count;
        *reinterpret_cast<ulonglong2*>(masks + key * 4) = make_ulonglong2(0ull, 0ull);
        *reinterpret_cast<ulonglong2*>(masks + key * 4 + 2) = make_ulonglong2(0ull, 0ull);
    }
    __syncthreads();
    return slot;
}

template <bool FAST, bool SMALL>
__device__ __forceinline__ void plan_graph_impl(int32_t* plan, const PlanLayout& L, const int64_t* __restrict__ edge_index,
                                                const int64_t* __restrict__ layer, const float* __restrict__ edge_attr,
                                                int R, int64_t E, int32_t* status, const int g, const int d,
                                                const int n0, const int n1, const int e0, const int e1,
                                                int32_t* lds, int32_t* s_depth, int32_t* small_ws) {
    constexpr int NC = PLAN_FAST_N / PB, NCE = PLAN_FAST_E / PB;   // chunks of PB nodes / edges a FAST graph has at most
    const int tid = threadIdx.x;
    const int n = n1 - n0;
    // SMALL is a template parameter, not a run-time choice: a pointer SELECTED between LDS and global memory is a generic
    // pointer, and every access through it a flat_load / flat_store / flat_atomic at a vector-memory round trip each - the four
    // cursor turns of chunk_place alone were 5 us per chunk, 40 of this kernel's 49 us on the headline batch
    constexpr bool small = SMALL;
    int32_t* ls_g = plan + L.lstart[d] + n0 + g;   // n+1 words (final home)
    int32_t* rp_g = plan + L.rowptr[d] + n0 + g;   // n+1 words (final home)
    int32_t *ls, *rp, *cur, *pos;
    constexpr int SN = FAST ? PLAN_FAST_N : PLAN_NMAX;   // nodes the LDS arrays are laid out for
    unsigned long long* masks = nullptr;
    if constexpr (SMALL) {
        ls = small_ws; rp = small_ws + (SN + 1); cur = small_ws + 2 * (SN + 1);
        pos = small_ws + 3 * (SN + 1) - n0;   // indexed by node id
        if constexpr (FAST) {
            masks = reinterpret_cast<unsigned long long*>(small_ws + 4 * (SN + 1));   // (a multiple of 16 bytes in)
            for (int i = tid; i < 4 * (n + 1); i += PB) masks[i] = 0ull;           // (the first barrier below is long before their first use)
        }
    } else {
        ls = ls_g; rp = rp_g; cur = plan + L.cursor[d] + n0 + g;  // n+1 words
        pos = plan + L.pos[d];                                    // indexed by node id
    }
    int32_t* order = plan + L.order[d];
    int32_t* col = plan + L.col[d];
    int32_t* eidx = plan + L.eidx[d];
    float* eattr = reinterpret_cast<float*>(plan + L.eattr[d]);
    // the node an edge feeds is its target (d=0) or its source (d=1)
    const int64_t* feed = d == 0 ? edge_index + E : edge_index;
    const int64_t* other = d == 0 ? edge_index : edge_index + E;
    const int nchunk_n = FAST ? NC : (n + PB - 1) / PB, nchunk_e = FAST ? NCE : (e1 - e0 + PB - 1) / PB;

    // ---- FAST: everything this workgroup will ever read, in one trip
    int lreg[NC], freg[NCE], oreg[NCE];   // (narrowed on the way in: values outside int32 are out of range anyway)
    float areg[NCE][2];
    if constexpr (FAST) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int v = n0 + c * PB + tid;
            const int64_t l = v < n1 ? layer[v] : 0;
            lreg[c] = l < 0 ? -1 : l > 0x7fffffff ? 0x7fffffff : (int)l;
        }
#pragma unroll
        for (int c = 0; c < NCE; ++c) {
            const int e = e0 + c * PB + tid;
            const int64_t f = e < e1 ? feed[e] : -1, o = e < e1 ? other[e] : -1;
            freg[c] = (f < 0 || f > 0x7fffffff) ? -1 : (int)f;
            oreg[c] = (o < 0 || o > 0x7fffffff) ? -1 : (int)o;
            areg[c][0] = (e < e1 && R >= 1) ? edge_attr[(int64_t)e * R] : 0.f;
            areg[c][1] = (e < e1 && R >= 2) ? edge_attr[(int64_t)e * R + 1] : 0.f;
        }
    }
    auto layer_of = [&](int c, int v) -> int64_t { if constexpr (FAST) return lreg[c]; else return layer[v]; };
    auto feed_of = [&](int c, int e) -> int64_t { if constexpr (FAST) return freg[c]; else return feed[e]; };

    // ---- depth of this graph in this direction
    int mx = -1, bad = 0;
#pragma unroll
    for (int c = 0; c < nchunk_n; ++c) {
        const int v = n0 + c * PB + tid;
        if (v < n1) {
            int64_t l = layer_of(c, v);
            if (l < 0 || l >= n) { bad = 8; l = l < 0 ? 0 : n - 1; }
            mx = max(mx, (int)l);
        }
    }
    mx = wave_max_i(mx);
    if (tid == 0) *s_depth = -1;
    __syncthreads();
    if ((tid & 63) == 0) atomicMax(s_depth, mx);
    if (bad && status) atomicOr(status, bad);
    for (int i = tid; i <= n; i += PB) { ls[i] = 0; rp[i] = 0; }
    __syncthreads();
    const int depth = *s_depth + 1;  // 0 for an empty graph
    if (tid == 0) plan[L.depth[d] + g] = depth;

    // ---- histogram of layers -> lstart (absolute positions into order[])
#pragma unroll
    for (int c = 0; c < nchunk_n; ++c) {
        const int v = n0 + c * PB + tid;
        if (v < n1) {
            int l = (int)min((int64_t)max(layer_of(c, v), (int64_t)0), (int64_t)(n - 1));
            atomicAdd(&ls[l + 1], 1);
        }
    }
    __syncthreads();
    block_scan_inplace(ls, depth + 1, n0, lds);
    __syncthreads();
    for (int i = tid; i < depth; i += PB) {
        cur[i] = ls[i];
        atomicAdd(&plan[L.blptr[d] + i + 1], ls[i + 1] - ls[i]);  // rows of batch-level layer i
    }
    if (small) for (int i = tid; i <= depth; i += PB) ls_g[i] = ls[i];
    __syncthreads();

    // ---- stable placement of nodes: order[] sorted by (layer, node id)
#pragma unroll
    for (int c = 0; c < nchunk_n; ++c) {
        if (c * PB >= n) break;   // (uniform)
        const int v = n0 + c * PB + tid;
        int key = -1;
        if (v < n1) key = (int)min((int64_t)max(layer_of(c, v), (int64_t)0), (int64_t)(n - 1));
        int slot;
        if constexpr (FAST) slot = chunk_place_masks(key, cur, masks); else slot = chunk_place(key, cur);
        if (key >= 0) {
            order[slot] = v;
            pos[v] = slot;
        }
    }
    __syncthreads();  // pos[] is read across waves below: an unwritten LDS word is an arbitrary index into rp[]

    // ---- rows of the CSR
#pragma unroll
    for (int c = 0; c < nchunk_e; ++c) {
        const int e = e0 + c * PB + tid;
        if (e < e1) {
            const int64_t f = feed_of(c, e);
            if (f >= n0 && f < n1) atomicAdd(&rp[pos[f] - n0 + 1], 1);
        }
    }
    __syncthreads();
    block_scan_inplace(rp, n + 1, e0, lds);
    __syncthreads();
    for (int i = tid; i < n; i += PB) cur[i] = rp[i];
    if (small) for (int i = tid; i <= n; i += PB) rp_g[i] = rp[i];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < nchunk_e; ++c) {
        if (c * PB >= e1 - e0) break;   // (uniform)
        const int e = e0 + c * PB + tid;
        int key = -1;
        int64_t f = -1;
        if (e < e1) { f = feed_of(c, e); if (f >= n0 && f < n1) key = pos[f] - n0; }
        int slot;
        if constexpr (FAST) slot = chunk_place_masks(key, cur, masks); else slot = chunk_place(key, cur);
        if (key >= 0) {
            int64_t o;
            if constexpr (FAST) o = oreg[c]; else o = other[e];
            col[slot] = (int)((o >= n0 && o < n1) ? o : f);
            eidx[slot] = e;  // original edge id: per-edge quantities of the backward pass are stored by it
            if constexpr (FAST) {
                if (R >= 1) eattr[(int64_t)slot * R] = areg[c][0];
                if (R >= 2) eattr[(int64_t)slot * R + 1] = areg[c][1];
            } else {
                for (int r = 0; r < R; ++r) eattr[(int64_t)slot * R + r] = edge_attr[(int64_t)e * R + r];
            }
        }
    }}

__device__ __forceinline__ void plan_graph_body(int32_t* plan, const PlanLayout& L, const int64_t* __restrict__ edge_index,
                                                const int64_t* __restrict__ layer_fwd, const int64_t* __restrict__ layer_bwd,
                                                const float* __restrict__ edge_attr, int R, int64_t N, int64_t E,
                                                int32_t* status, const int g, const int d) {
    if (status && status[0] & 7) return;   // contract violated (plan_ptr_body): the tables are garbage - do not walk them (plan_seal_body)
    __shared__ int32_t lds[PB + 8];
    __shared__ int32_t s_depth;
    // graphs of up to PLAN_NMAX nodes (all of ogbg-code2's typical ASTs) keep their counters, cursors
    // and positions in LDS: the kernel is a chain of ~15 dependent passes, and every pass through
    // global memory costs a round trip
    __shared__ __attribute__((aligned(16))) int32_t small_ws[PLAN_GRAPH_LDS];
    const int n0 = plan[L.node_ptr + g], n1 = plan[L.node_ptr + g + 1];
    const int e0 = plan[L.edge_ptr + g], e1 = plan[L.edge_ptr + g + 1];
    const int64_t* layer = d == 0 ? layer_fwd : layer_bwd;
    if (n1 - n0 <= PLAN_FAST_N && e1 - e0 <= PLAN_FAST_E && R <= 2)
        plan_graph_impl<true, true>(plan, L, edge_index, layer, edge_attr, R, E, status, g, d, n0, n1, e0, e1, lds, &s_depth, small_ws);
    else if (n1 - n0 <= PLAN_NMAX)
        plan_graph_impl<false, true>(plan, L, edge_index, layer, edge_attr, R, E, status, g, d, n0, n1, e0, e1, lds, &s_depth, small_ws);
    else
        plan_graph_impl<false, false>(plan, L, edge_index, layer, edge_attr, R, E, status, g, d, n0, n1, e0, e1, lds, &s_depth, small_ws);
}

// Last step of the build.  Once the status word is set (unsorted `batch`, edges across graphs, layers out of range)
// the tables above hold whatever the violated assumptions produced; every consumer - read-outs, per-layer launches,
// the dataflow schedule - takes its loop bounds from the depths and the layer offsets, so zero those: the batch reads
// as one without layers, nothing walks the garbage, and the host raises on the status word at its next poll.
// Thread i of `stride` (grid-stride over max(N + 2, B)).
__device__ __forceinline__ void plan_seal_body(int32_t* plan, const PlanLayout& L, int64_t N, int64_t B, int64_t i, int64_t stride) {
    const int64_t w = N + 2 > B ? N + 2 : B;
    for (; i < w; i += stride) {
        if (i < B) { plan[L.depth[0] + i] = 0; plan[L.depth[1] + i] = 0; }
        if (i < N + 2) { plan[L.blptr[0] + i] = 0; plan[L.blptr[1] + i] = 0; plan[L.blsplit[0] + i] = 0; plan[L.blsplit[1] + i] = 0; }
    }
}

// Work items (g*2+d) sorted by depth, deepest first, so that the hardware's in-order workgroup
// dispatch starts the longest dependency chains first (LPT scheduling).  One workgroup; keys: 4096 words of LDS.
__device__ __forceinline__ void plan_items_body(int32_t* plan, const PlanLayout& L, int B, const int32_t* __restrict__ status, int32_t* keys) {
    if (status && status[0] & 7) return;   // contract violated (plan_ptr_body): the tables are garbage - do not walk them (plan_seal_body)
    const int n = 2 * B;
    for (int c0 = 0; c0 < n; c0 += 4096) {  // candidates staged through LDS, 4096 at a time
        __syncthreads();
        for (int j = threadIdx.x; j < 4096 && c0 + j < n; j += blockDim.x)
            keys[j] = plan[L.depth[(c0 + j) & 1] + ((c0 + j) >> 1)];
        __syncthreads();
        const int m = min(4096, n - c0);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int ki = plan[L.depth[i & 1] + (i >> 1)];
            int rank = 0;
            for (int j = 0; j < m; ++j) rank += (keys[j] > ki) || (keys[j] == ki && c0 + j < i);
            if (c0 == 0) plan[L.cursor[0] + i] = rank; else plan[L.cursor[0] + i] += rank;  // scratch: B <= N/1
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) plan[L.items + plan[L.cursor[0] + i]] = i;
}

// Batch-level layers (the lock-step schedule): blptr[d][t] = first rowrec slot of layer t over the
// whole batch; T_d is stored at blptr[d][N+1].  One workgroup of 1024 threads per direction.
__device__ __forceinline__ void plan_blptr_body(int32_t* plan, const PlanLayout& L, int N, int B, const int32_t* __restrict__ status, const int d) {
    if (status && status[0] & 7) return;   // contract violated (plan_ptr_body): the tables are garbage - do not walk them (plan_seal_body)
    __shared__ int32_t lds[1024 / 64 + 1];
    __shared__ int32_t s_T;
    const int tid = threadIdx.x;
    int mx = 0;
    for (int g = tid; g < B; g += 1024) mx = max(mx, plan[L.depth[d] + g]);
    mx = wave_max_i(mx);
    if (tid == 0) s_T = 0;
    __syncthreads();
    if ((tid & 63) == 0) atomicMax(&s_T, mx);
    __syncthreads();
    const int T = s_T;
    int32_t* a = plan + L.blptr[d];
    int carry = 0;
    for (int c0 = 0; c0 <= T; c0 += 1024) {  // inclusive scan of a[0..T]
        const int i = c0 + tid;
        const int v = (i <= T) ? a[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if ((tid & 63) >= o) x += y; }
        if ((tid & 63) == 63) lds[tid >> 6] = x;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < (tid >> 6)) woff += lds[w]; tot += lds[w]; }
        if (i <= T) a[i] = carry + woff + x;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) a[N + 1] = T;
    // thr_d = 1 + last layer with more than DAGNN_PLAN_THIN_ROWS rows (the scan above is complete: width = a[t+1] - a[t])
    __shared__ int32_t s_thr;
    if (tid == 0) s_thr = 0;
    __syncthreads();
    int thr = 0;
    for (int t = tid; t < T; t += 1024)
        if (a[t + 1] - a[t] > DAGNN_PLAN_THIN_ROWS) thr = t + 1;
    thr = wave_max_i(thr);
    if ((tid & 63) == 0) atomicMax(&s_thr, thr);
    __syncthreads();
    if (tid == 0) plan[PH_THR0 + d] = s_thr;
}

// lbase[g][t] = blptr[t] + rows of layer t in graphs ordered before g - shallow graphs first, then deep ones, each
// group in graph order (deterministic slot assignment); blsplit[t] = first slot of the deep group.
// One WAVE per batch-level layer (workgroup bx of 256 threads: layers 4 bx .. 4 bx + 3): lane l takes graphs l, l+64, ...;
// an exclusive wave scan over the per-graph row counts gives every graph its first slot (two memory round trips per 64 graphs).
__device__ __forceinline__ void plan_lbase_body(int32_t* plan, const PlanLayout& L, int N, int B, const int32_t* __restrict__ status,
                                                const int bx, const int d) {
    if (status && status[0] & 7) return;   // contract violated (plan_ptr_body): the tables are garbage - do not walk them (plan_seal_body)
    const int lane = threadIdx.x & 63;
    const int t = bx * 4 + (threadIdx.x >> 6);
    const int T = plan[L.blptr[d] + N + 1];
    if (t >= T) return;
    const int32_t* __restrict__ ls = plan + L.lstart[d];
    const int32_t* __restrict__ node_ptr = plan + L.node_ptr;
    const int32_t* __restrict__ depth = plan + L.depth[d];
    int32_t* __restrict__ lb = plan + L.lbase[d];
    const int thr = plan[PH_THR0 + d];
    int carry = plan[L.blptr[d] + t];
    for (int pass = 0; pass < 2; ++pass) {   // shallow graphs first, then the deep ones (depth > thr)
        if (pass == 1 && lane == 0) plan[L.blsplit[d] + t] = carry;
        for (int g0 = 0; g0 < B; g0 += 64) {
            const int g = g0 + lane;
            int cnt = 0, base = 0;
            bool has = false;
            if (g < B && t < depth[g] && (depth[g] > thr) == (pass == 1)) {
                base = node_ptr[g] + g + t;
                cnt = ls[base + 1] - ls[base];
                has = true;
            }
            int x = cnt;  // inclusive wave scan
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
            if (has) lb[base] = carry + x - cnt;
            carry += __shfl(x, 63, 64);
        }
    }
}

// rowrec[slot] (64 B) = {node, edge begin, edge end, graph, pred[0..3], edge feats of the first four
// in-edges (2 floats each)} for every node, slots ordered by batch-level layer.  Rows with <= 4
// in-edges (every node of an AST in the forward direction) need no second indirection in the
// lock-step kernel: its dependent chain is record -> predecessor rows.  Thread p = 256 bx + tid of direction d.
__device__ __forceinline__ void plan_rowrec_body(int32_t* plan, const PlanLayout& L, const int64_t* __restrict__ batch,
                                                 const int64_t* __restrict__ layer_fwd, const int64_t* __restrict__ layer_bwd,
                                                 int N, int R, const int32_t* __restrict__ status, const int bx, const int d) {
    if (status && status[0] & 7) return;   // contract violated (plan_ptr_body): the tables are garbage - do not walk them (plan_seal_body)
    const int p = bx * 256 + threadIdx.x;  // per-graph sorted position
    if (p >= N) return;
    const int v = plan[L.order[d] + p];
    const int g = (int)batch[v];
    const int n0 = plan[L.node_ptr + g];
    const int n = plan[L.node_ptr + g + 1] - n0;
    int t = (int)(d == 0 ? layer_fwd[v] : layer_bwd[v]);
    t = min(max(t, 0), n - 1);
    const int base = n0 + g + t;
    const int slot = plan[L.lbase[d] + base] + (p - plan[L.lstart[d] + base]);
    const int32_t* rp = plan + L.rowptr[d] + n0 + g + (p - n0);
    const int eb = rp[0], ee = rp[1];
    const int32_t* col = plan + L.col[d];
    const int32_t* eattr = plan + L.eattr[d];  // float bits
    int w[16];
    w[0] = v; w[1] = eb; w[2] = ee; w[3] = g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool ok = eb + q < ee;
        w[4 + q] = ok ? col[eb + q] : 0;
        w[8 + 2 * q] = (ok && R >= 1) ? eattr[(int64_t)(eb + q) * R] : 0;
        w[9 + 2 * q] = (ok && R >= 2) ? eattr[(int64_t)(eb + q) * R + 1] : 0;
    }
    plan[L.pos[d] + v] = slot;  // final meaning of pos[]: batch-level slot of every node (backward: node -> record)
    int4* out = reinterpret_cast<int4*>(plan + L.rowrec[d] + 16 * (int64_t)slot);
    out[0] = make_int4(w[0], w[1], w[2], w[3]);
    out[1] = make_int4(w[4], w[5], w[6], w[7]);
    out[2] = make_int4(w[8], w[9], w[10], w[11]);
    out[3] = make_int4(w[12], w[13], w[14], w[15]);
}

}  // namespace
