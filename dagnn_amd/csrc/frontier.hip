// frontier.hip - lock-step schedule of the recurrence: ONE launch per batch-level topological
// layer, every (direction, stacked layer) cell in the same launch.
//
// Reference path replaced: the three nested loops of ogbg-code/model/dagnn.py:144-182
//   for d in dirs: for l_idx in range(T): [edge scan :151-157] for i, cell in cells_d:
//       ps_h = AttnConv(...)[layer]  (:175-179, message :366-373);  inp = GRUCell(inp, ps_h) (:181)
//       h[d][i][layer] += inp (:182)
// re-ordered legally: launch s processes layer t = s - i of stacked layer i in both directions
// (cell (d,i) at layer t needs h[d][i] of its predecessors - layers < t, finished in earlier
// launches - and h[d][i-1] of the same node, finished in launch s-1).  T + L - 1 dependent
// launches replace D*L*T dependent micro-steps.
//
// gfx950 design.  This path is bound by its dependent chain (T = 374 layers deep at a median
// frontier of 4-7 nodes), so every launch is built to be SHORT:
//  * its own chain is two memory round trips: 64-byte row record (scalar load; node id, edge
//    range and the first four predecessors + their edge features inline) -> predecessor rows.
//    The per-slice partial attention scores travel in 16 trailing floats of each state row, so
//    the same round trip brings them;
//  * the GRU weights never sit on that chain: a workgroup owns a slice of JS hidden units
//    (3 gates x JS weight columns, K = H) of one cell for a block of <= RB frontier rows, and
//    issues the loads of its whole slice - pre-packed in exactly the order its lanes consume it,
//    every load instruction one contiguous 1 KiB - before it starts the aggregate.  A CU moves
//    64 B/clk, so a slice is sized by what one CU can pull in ~1 us: thin launches use JS=16
//    slices spread over more CUs, fat ones JS=32;
//  * K is split over the 16 lanes of a DPP row, so the K reduction is 4 v_add_dpp per value
//    instead of LDS-crossbar shuffles.
//   A  one wave per row: softmax over the in-edges, alpha-weighted float4 gather -> LDS; for
//      stacked layers > 0 also the node's own lower-layer row;
//   B  fp32 FMA GEMV on the slice, hidden- and input-side, RB rows register-blocked;
//   C  RB x JS threads: gates, h' store, partial score store.
// fp32 VALU FMA: at <= 8 rows per weight pass the fp32 MFMA has the same per-row rate.
#include "common.h"

#define DAGNN_MAX_CELLS 16

namespace {

constexpr int FT = 384;     // threads per workgroup (6 waves)
constexpr int PU = 16;      // hidden units per stored score part

struct Cell {
    const float4* whh;   // packed hidden-side slices
    const float4* wih;   // packed input-side slices, or null (stacked layer 0: gi0 instead)
    const float4* whh_m; // the same matrices in MFMA fragment order (fat launches), or null
    const float4* wih_m;
    const float* bhh;    // [3H]
    const float* bih;    // [3H] (only with wih)
    const float* wkey;   // [H], or null when the scores are static
    const float* sscore; // [N] static attention score of every node (keys taken from the inputs x), or null
    const float* gain;   // [R] or null
    const float* vid;    // [vid_mod] or null
    const float* gi0;    // [N,3H] precomputed input side (stacked layer 0) or null
    const float* h_in;   // [N,ld_h] lower stacked layer (with wih) or null
    float* h_out;        // [N,ld_h]: H state floats + H/16 partial scores per row
    const float* a_pre;  // fat launches: aggregates of this launch's rows [row_end - row_base, H], written by
                         // aggregate_rows_kernel one launch earlier; null = aggregate inside the block
    unsigned long long* g_out;        // [N,gld] {epoch tag, fp32 bits} granules of h_out rows + parts, or null
    const unsigned long long* g_in;   // granules of h_in, or null
    int dir;             // direction (selects the plan arrays)
    int row_base;        // first rowrec slot of the layer processed in this launch
    int row_end;         // one past the last
    int has_pred;        // layer > 0
};

struct StepArgs {
    Cell cell[DAGNN_MAX_CELLS];
    int blk_start[DAGNN_MAX_CELLS + 1];  // row-block prefix sums over the active cells
    int ncell, H, ld_h, R, vid_mod, step;
    unsigned epoch;           // tag of this forward pass in the granule copies (never 0)
    unsigned long long* dbg;  // optional [steps][8] wall_clock64 stamps of workgroup 0
};

// LDS index of element k of an operand row: 4 floats of pad per K-lane segment so the 16
// segments a DPP row reads concurrently (ds_read_b128) fall on disjoint banks.
__device__ __forceinline__ int apad(int k, int kpt) { return k + 4 * (k / kpt); }

__device__ __forceinline__ float dpp_row_sum16(float v) {
    // inclusive scan over the 16 lanes of a DPP row (row_shr 1,2,4,8; out-of-row lanes read 0):
    // lane 15 of every row ends with the row total, always in the same order -> deterministic
#define DAGNN_DPP_ADD(ctrl) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    DAGNN_DPP_ADD(0x111); DAGNN_DPP_ADD(0x112); DAGNN_DPP_ADD(0x114); DAGNN_DPP_ADD(0x118);
#undef DAGNN_DPP_ADD
    return v;
}

__device__ __forceinline__ void fma4(float4& acc, float al, const float4& v) {
    acc.x = fmaf(al, v.x, acc.x); acc.y = fmaf(al, v.y, acc.y); acc.z = fmaf(al, v.z, acc.z); acc.w = fmaf(al, v.w, acc.w);
}

// sum of the H/16 partial scores stored behind a state row, in index order (deterministic).
__device__ __forceinline__ float score_of(const float* __restrict__ hrow_tail, int nparts) {
    float s = 0.f;
    for (int q = 0; q < nparts; q += 4) {
        const float4 p = *reinterpret_cast<const float4*>(hrow_tail + q);
        s += p.x; if (q + 1 < nparts) s += p.y; if (q + 2 < nparts) s += p.z; if (q + 3 < nparts) s += p.w;
    }
    return s;
}

// One wave: float4 chunk `lane` of granule row `grow` (H <= 256: one chunk per lane), waiting for it.
__device__ __forceinline__ float4 gran_row_chunk(const gran_t* grow, int lane, int H4, const GranCtx& G) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
        if (lane < H4) {
            const gran_t x0 = gran_ld(grow + 4 * lane), x1 = gran_ld(grow + 4 * lane + 1),
                         x2 = gran_ld(grow + 4 * lane + 2), x3 = gran_ld(grow + 4 * lane + 3);
            ok = (unsigned)(x0 >> 32) == G.epoch && (unsigned)(x1 >> 32) == G.epoch &&
                 (unsigned)(x2 >> 32) == G.epoch && (unsigned)(x3 >> 32) == G.epoch;
            v = make_float4(__uint_as_float((unsigned)x0), __uint_as_float((unsigned)x1),
                            __uint_as_float((unsigned)x2), __uint_as_float((unsigned)x3));
        }
        if (__all(ok) || !gran_retry(spins, G)) break;
    }
    return v;
}

// One wave: a_row[:] = sum_e alpha_e * h[pred_e, :] with alpha = softmax_e(score[pred_e] + gain . feat_e)
// (PyG: exp(x - max) / (sum + 1e-16)).  rec1 = first four predecessors, rec2/rec3 = their edge features.
// GRAN: predecessor rows and scores are read (and waited for) through their granule copies.
template <bool GRAN>
__device__ __forceinline__ void aggregate(const Cell& C, const int32_t* __restrict__ col,
                                          const float* __restrict__ eattr, int eb, int ee, int4 rec1, int4 rec2,
                                          int4 rec3, int H, int ld_h, int R, int vid_mod, int kpt, float* a_row,
                                          int lane, const GranCtx& G) {
    const int H4 = H >> 2;
    const int nparts = H / PU;
    const int gld = H + nparts;
    const float* hsrc = C.h_out;  // predecessors' states of THIS stacked layer (earlier launches)
    const gran_t* gsrc = C.g_out;
    const int deg = ee - eb;
    if (deg <= 4 && R <= 2) {
        // ---- inline path: predecessor ids and edge features came with the row record
        const int pj[4] = {rec1.x, rec1.y, rec1.z, rec1.w};
        const float f0[4] = {__int_as_float(rec2.x), __int_as_float(rec2.z), __int_as_float(rec3.x), __int_as_float(rec3.z)};
        const float f1[4] = {__int_as_float(rec2.y), __int_as_float(rec2.w), __int_as_float(rec3.y), __int_as_float(rec3.w)};
        float al[4] = {1.f, 0.f, 0.f, 0.f};
        float4 row0[4];
        float sc[4] = {0.f, 0.f, 0.f, 0.f};
        if (GRAN) {
            // rows and score parts of all <= 4 predecessors in ONE polling loop (one round trip when ready)
            float pv[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned spins = 0;
            const gran_t ready = (gran_t)G.epoch << 32;   // stands in for granules that are not read
            const bool want_parts = deg > 1 && !C.sscore && lane < nparts;
            for (;;) {
                // every load of the iteration is issued before the first tag is looked at: the loads are atomics,
                // which the compiler keeps in program order - a compare between two groups would serialise them
                gran_t x[4][4], xp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const gran_t* grow = gsrc + (int64_t)pj[e] * gld;
                    const bool on = e < deg && lane < H4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[e][q] = on ? gran_ld(grow + 4 * lane + q) : ready;
                    xp[e] = (e < deg && want_parts) ? gran_ld(grow + H + lane) : ready;
                }
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(x[e][q] >> 32) == G.epoch;
                    ok = ok && (unsigned)(xp[e] >> 32) == G.epoch;
                    row0[e] = make_float4(__uint_as_float((unsigned)x[e][0]), __uint_as_float((unsigned)x[e][1]),
                                          __uint_as_float((unsigned)x[e][2]), __uint_as_float((unsigned)x[e][3]));
                    pv[e] = __uint_as_float((unsigned)xp[e]);
                }
                if (__all(ok) || !gran_retry(spins, G)) break;
            }
            if (deg > 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e >= deg) continue;
                    if (C.sscore) sc[e] = C.sscore[pj[e]];
                    else  // the <= 16 parts sit in lanes 0..15 (0 beyond nparts): one DPP row scan, fixed order
                        sc[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                    __builtin_bit_cast(int, dpp_row_sum16(pv[e])), 15));
                }
            }
        } else {
            // first 64 float4 columns of every predecessor row: issued before the scores are touched so
            // that rows and scores share one memory round trip
#pragma unroll
            for (int e = 0; e < 4; ++e)
                row0[e] = (e < deg && lane < H4) ? reinterpret_cast<const float4*>(hsrc + (int64_t)pj[e] * ld_h)[lane]
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            if (deg > 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < deg) sc[e] = C.sscore ? C.sscore[pj[e]] : score_of(hsrc + (int64_t)pj[e] * ld_h + H, nparts);
            }
        }
        if (deg > 1) {
            float lg[4], mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lg[e] = -INFINITY;
                if (e < deg) {
                    float s = sc[e];
                    if (C.vid) s += C.vid[pj[e] % vid_mod];
                    if (R >= 1) s = fmaf(C.gain[0], f0[e], s);
                    if (R >= 2) s = fmaf(C.gain[1], f1[e], s);
                    lg[e] = s;
                    mx = fmaxf(mx, s);
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { al[e] = e < deg ? expf(lg[e] - mx) : 0.f; sum += al[e]; }
            const float denom = sum + 1e-16f;
#pragma unroll
            for (int e = 0; e < 4; ++e) al[e] = al[e] / denom;
        }
        if (lane < H4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) fma4(acc, al[e], row0[e]);  // al[e] == 0 and row0[e] == 0 beyond deg
            *reinterpret_cast<float4*>(a_row + apad(4 * lane, kpt)) = acc;
        }
        if (!GRAN) {
            for (int c = lane + 64; c < H4; c += 64) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < deg) fma4(acc, al[e], reinterpret_cast<const float4*>(hsrc + (int64_t)pj[e] * ld_h)[c]);
                *reinterpret_cast<float4*>(a_row + apad(4 * c, kpt)) = acc;
            }
        }
        return;
    }
    // ---- general path (fan-in > 4): lanes own edges
    auto logit = [&](int e, int cj) {
        float s = 0.f;
        if (C.sscore) {
            s = C.sscore[cj];
        } else if (GRAN) {  // this lane's predecessor: its H/16 part granules (H <= 256: at most 16), all loads in
                            // flight together, re-polled as a group, summed in index order
            const gran_t* gp = gsrc + (int64_t)cj * gld + H;
            unsigned spins = 0;
            for (;;) {
                gran_t x[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = q < nparts ? gran_ld(gp + q) : ((gran_t)G.epoch << 32);
                bool ok = true;
                s = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    ok = ok && (unsigned)(x[q] >> 32) == G.epoch;
                    if (q < nparts) s += __uint_as_float((unsigned)x[q]);
                }
                if (ok) break;   // per-lane wait: producers never wait on us
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { __hip_atomic_store(G.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        } else {
            s = score_of(hsrc + (int64_t)cj * ld_h + H, nparts);
        }
        if (C.vid) s += C.vid[cj % vid_mod];
        for (int r = 0; r < R; ++r) s = fmaf(C.gain[r], eattr[(int64_t)e * R + r], s);
        return s;
    };
    float mx = -INFINITY, sum = 0.f, lg0 = -INFINITY;
    int col0 = 0;
    const bool one_pass = deg <= 64;  // every lane owns at most one edge: its logit stays in a register
    if (one_pass) {
        if (lane < deg) { col0 = col[eb + lane]; lg0 = logit(eb + lane, col0); }
        mx = wave_max(lg0);
        sum = wave_sum(lane < deg ? expf(lg0 - mx) : 0.f);
    } else {
        for (int e = eb + lane; e < ee; e += 64) mx = fmaxf(mx, logit(e, col[e]));
        mx = wave_max(mx);
        for (int e = eb + lane; e < ee; e += 64) sum += expf(logit(e, col[e]) - mx);
        sum = wave_sum(sum);
    }
    const float denom = sum + 1e-16f;
    for (int c0 = 0; c0 < H4; c0 += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = c0 + lane;
        for (int base = eb; base < ee; base += 64) {
            const int e = base + lane;
            float my_alpha = 0.f;
            int my_col = 0;
            if (e < ee) {
                if (one_pass) { my_col = col0; my_alpha = expf(lg0 - mx) / denom; }
                else { my_col = col[e]; my_alpha = expf(logit(e, my_col) - mx) / denom; }
            }
            const int cnt = min(64, ee - base);
            int i = 0;
            if (!GRAN) {
                for (; i + 4 <= cnt; i += 4) {  // four row loads in flight per lane
                    float4 v[4]; float a4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a4[u] = __shfl(my_alpha, i + u, 64);
                        const int cj = __shfl(my_col, i + u, 64);
                        v[u] = c < H4 ? reinterpret_cast<const float4*>(hsrc + (int64_t)cj * ld_h)[c] : make_float4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) fma4(acc, a4[u], v[u]);
                }
            }
            if (GRAN) {
                for (; i + 4 <= cnt; i += 4) {  // four granule rows polled together: one round trip when they are ready
                    float a4[4]; const gran_t* gr[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a4[u] = __shfl(my_alpha, i + u, 64);
                        gr[u] = gsrc + (int64_t)__shfl(my_col, i + u, 64) * gld;
                    }
                    float4 v[4];
                    unsigned spins = 0;
                    for (;;) {
                        gran_t x[4][4];   // all 16 loads first, then the tags (see the inline path)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                x[u][q] = lane < H4 ? gran_ld(gr[u] + 4 * lane + q) : ((gran_t)G.epoch << 32);
                        bool ok = true;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(x[u][q] >> 32) == G.epoch;
                            v[u] = make_float4(__uint_as_float((unsigned)x[u][0]), __uint_as_float((unsigned)x[u][1]),
                                               __uint_as_float((unsigned)x[u][2]), __uint_as_float((unsigned)x[u][3]));
                        }
                        if (__all(ok) || !gran_retry(spins, G)) break;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) fma4(acc, a4[u], v[u]);
                }
            }
            for (; i < cnt; ++i) {
                const float a1 = __shfl(my_alpha, i, 64);
                const int cj = __shfl(my_col, i, 64);
                if (GRAN) fma4(acc, a1, gran_row_chunk(gsrc + (int64_t)cj * gld, lane, H4, G));
                else if (c < H4) fma4(acc, a1, reinterpret_cast<const float4*>(hsrc + (int64_t)cj * ld_h)[c]);
            }
        }
        if (c < H4) *reinterpret_cast<float4*>(a_row + apad(4 * c, kpt)) = acc;
    }
}

// acc[r] += W-slice x operand row r for the first NRW rows of the block (NRW <= RBT is a compile-time
// bound so that blocks with few live rows do not pay for the padding rows).
template <int RBT, int KW, int NRW>
__device__ __forceinline__ void fma_rows(float4 (&acc)[RBT], const float4 (&w)[KW], const float* op, int op_ld,
                                         int koff, int n) {
#pragma unroll
    for (int q = 0; q < KW / 4; ++q) {
        if (4 * q < n) {
#pragma unroll
            for (int r = 0; r < NRW; ++r) {
                const float4 a = *reinterpret_cast<const float4*>(op + r * op_ld + koff + 4 * q);
                fma4(acc[r], a.x, w[4 * q]); fma4(acc[r], a.y, w[4 * q + 1]);
                fma4(acc[r], a.z, w[4 * q + 2]); fma4(acc[r], a.w, w[4 * q + 3]);
            }
        }
    }
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// -----------------------------------------------------------------------------------------------
// One row block (<= RBT frontier rows of one cell) x one slice of JS hidden units.
//   JS   hidden units per slice (3*JS weight columns = 3*JS/4 float4 column groups, 4 per wave)
//   KW   k values per lane held in registers at a time.  KW = 16 keeps a whole K = 256 slice in
//        registers (issued ahead of the dependent chain in the launch-per-layer kernel, resident
//        across steps in the persistent tail kernel); KW = 4 streams it in chunks so that two
//        workgroups fit a CU and hide each other's latency (fat launches).
//   RESIDENT  the first weight chunk is already in wh/wi (persistent kernel); rows are published
//        with write-through (sc1) stores for the other workgroups of the same launch.
// -----------------------------------------------------------------------------------------------
// Threads per workgroup: 6 waves own the weight columns of a 32-unit slice; 8-row blocks get 8 waves so that
// phase A is one row per wave (a second row per wave is a second dependent memory chain).
template <int RBT> struct WgShape { static constexpr int threads = RBT > 6 ? 512 : FT; };

template <int JS, int RBT, int KW, bool RESIDENT>
__device__ __forceinline__ void process_block(const int32_t* __restrict__ plan, int64_t rowrec_off, int64_t col_off,
                                              int64_t eattr_off, const Cell& C, bool has_pred, int slot0, int nr,
                                              int sl, int H, int ld_h, int Rfeat, int vid_mod, float* smem,
                                              float4 (&wh)[KW], float4 (&wi)[KW], unsigned long long* stamp,
                                              const GranCtx& G) {
    constexpr int NCW = 3 * JS / 16;   // waves that own weight columns (4 column groups each)
    constexpr int SW = 3 * JS;         // slice width in columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool has_in = C.wih != nullptr;
    const int kpt = H >> 4;                 // k values per K-lane (H % 64 == 0)
    const int op_ld = H + 4 * 16;           // padded operand row
    float* a_s = smem;                      // [RBT][op_ld]  aggregates
    float* u_s = a_s + RBT * op_ld;         // [RBT][op_ld]  own lower-layer rows (has_in)
    float* gh_s = u_s + RBT * op_ld;        // [RBT][SW]     hidden-side pre-activations of the slice
    float* gi_s = gh_s + RBT * SW;          // [RBT][SW]     input-side
    int* v_s = reinterpret_cast<int*>(gi_s + RBT * SW);  // [RBT] node ids

    // first row record of this wave: a scalar load (own counter), issued ahead of the weight loads
    const int4* __restrict__ recs = reinterpret_cast<const int4*>(plan + rowrec_off);
    int4 rec_first = make_int4(0, 0, 0, 0);
    if (wave < nr) rec_first = recs[4 * (int64_t)(slot0 + wave)];

    // ---- weights of this slice: issued first, consumed after the aggregate (phase B)
    const int ksl = lane & 15;              // K-lane within the DPP row
    const int nchunk = (kpt + KW - 1) / KW;
    const int64_t wstride = (int64_t)NCW * 64;  // float4 per kk plane of one slice
    const bool owns_cols = wave < NCW;
    const float4* whh = C.whh + (int64_t)sl * kpt * wstride + tid;
    const float4* wih = has_in ? C.wih + (int64_t)sl * kpt * wstride + tid : nullptr;
    if (!RESIDENT) {
        const int n0k = min(KW, kpt);
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            wh[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
            wi[kk] = wh[kk];
            if (owns_cols && kk < n0k) {
                if (has_pred) wh[kk] = whh[kk * wstride];
                if (has_in) wi[kk] = wih[kk * wstride];
            }
        }
    }
    if (stamp) stamp[1] = wall_clock64();

    // ---- phase A: one wave per row (rows wave, wave+6)
    const int32_t* col = plan + col_off;
    const float* eattr = reinterpret_cast<const float*>(plan + eattr_off);
    const int R = C.gain ? Rfeat : 0;
    const int H4 = H >> 2;
    for (int r = wave; r < RBT; r += WgShape<RBT>::threads / 64) {
        float* a_row = a_s + r * op_ld;
        float* u_row = u_s + r * op_ld;
        if (r < nr) {
            const int4* rp = recs + 4 * (int64_t)(slot0 + r);  // wave-uniform address: scalar loads
            const int4 rec0 = r == wave ? rec_first : rp[0];
            if (lane == 0) v_s[r] = rec0.x;
            if (has_in) {
                if (RESIDENT) {  // produced one step ago by another workgroup of this launch: wait on its granules
                    const float4 uv = gran_row_chunk(C.g_in + (int64_t)rec0.x * (H + H / PU), lane, H4, G);
                    if (lane < H4) *reinterpret_cast<float4*>(u_row + apad(4 * lane, kpt)) = uv;
                } else {
                    const float4* ur = reinterpret_cast<const float4*>(C.h_in + (int64_t)rec0.x * ld_h);
                    for (int cc = lane; cc < H4; cc += 64)
                        *reinterpret_cast<float4*>(u_row + apad(4 * cc, kpt)) = ur[cc];
                }
            }
            if (!RESIDENT && C.a_pre != nullptr) {
                // fat launch: the aggregate was computed once per row by aggregate_rows_kernel
                const float4* ap = reinterpret_cast<const float4*>(C.a_pre + (int64_t)(slot0 + r - C.row_base) * H);
                for (int cc = lane; cc < H4; cc += 64) *reinterpret_cast<float4*>(a_row + apad(4 * cc, kpt)) = ap[cc];
            } else if (has_pred && rec0.z > rec0.y) {
                aggregate<RESIDENT>(C, col, eattr, rec0.y, rec0.z, rp[1], rp[2], rp[3], H, ld_h, R, vid_mod, kpt, a_row,
                                    lane, G);
            } else {
                for (int cc = lane; cc < H4; cc += 64)
                    *reinterpret_cast<float4*>(a_row + apad(4 * cc, kpt)) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {  // padding rows: finite zeros so the shared FMA pass stays NaN-free
            for (int cc = lane; cc < H4; cc += 64) {
                *reinterpret_cast<float4*>(a_row + apad(4 * cc, kpt)) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has_in) *reinterpret_cast<float4*>(u_row + apad(4 * cc, kpt)) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __syncthreads();
    if (stamp) stamp[2] = wall_clock64();

    // gate inputs that do not depend on the GEMV: issue now, consume in phase C
    const int gr_ = tid / JS, gj_ = tid - gr_ * JS;   // gate thread -> (row, unit of the slice)
    const bool gate_thread = tid < RBT * JS && gr_ < nr;
    const int gv = gate_thread ? v_s[gr_] : 0;
    const int j = sl * JS + gj_;
    float pre_r = 0.f, pre_z = 0.f, pre_n = 0.f, bh_r = 0.f, bh_z = 0.f, bh_n = 0.f, wk = 0.f;
    if (gate_thread) {
        if (has_in) { pre_r = C.bih[j]; pre_z = C.bih[H + j]; pre_n = C.bih[2 * H + j]; }
        else { const float* g0 = C.gi0 + (int64_t)gv * 3 * H; pre_r = g0[j]; pre_z = g0[H + j]; pre_n = g0[2 * H + j]; }
        bh_r = C.bhh[j]; bh_z = C.bhh[H + j]; bh_n = C.bhh[2 * H + j];
        wk = C.wkey ? C.wkey[j] : 0.f;
    }

    // ---- phase B: slice GEMV, K over the 16 lanes of each DPP row
    if (owns_cols) {
        float4 acc_h[RBT], acc_i[RBT];
#pragma unroll
        for (int r = 0; r < RBT; ++r) { acc_h[r] = make_float4(0.f, 0.f, 0.f, 0.f); acc_i[r] = acc_h[r]; }
        const int kbase = ksl * kpt + 4 * ksl;  // == apad(ksl * kpt, kpt)
        // wave-uniform: blocks with <= 2 live rows (the thin tail); only in the latency-bound shapes - the
        // second code copy costs the streamed shapes registers and instruction-cache room
        const bool few = RBT == 4 && KW == 16 && nr <= 2;
        const bool one = RESIDENT && few && nr == 1;   // the persistent tail spreads thin layers: one row per block is the common case
        for (int ch = 0; ch < nchunk; ++ch) {
            const int n = min(KW, kpt - ch * KW);
            if (ch > 0) {  // never taken when RESIDENT (the host only uses it for kpt <= KW)
#pragma unroll
                for (int kk = 0; kk < KW; ++kk) {
                    if (kk < n) {
                        if (has_pred) wh[kk] = whh[(int64_t)(ch * KW + kk) * wstride];
                        if (has_in) wi[kk] = wih[(int64_t)(ch * KW + kk) * wstride];
                    }
                }
            }
            if (one) {
                if (has_pred) fma_rows<RBT, KW, 1>(acc_h, wh, a_s, op_ld, kbase + ch * KW, n);
                if (has_in) fma_rows<RBT, KW, 1>(acc_i, wi, u_s, op_ld, kbase + ch * KW, n);
            } else if (few) {
                if (has_pred) fma_rows<RBT, KW, 2>(acc_h, wh, a_s, op_ld, kbase + ch * KW, n);
                if (has_in) fma_rows<RBT, KW, 2>(acc_i, wi, u_s, op_ld, kbase + ch * KW, n);
            } else {
                if (has_pred) fma_rows<RBT, KW, RBT>(acc_h, wh, a_s, op_ld, kbase + ch * KW, n);
                if (has_in) fma_rows<RBT, KW, RBT>(acc_i, wi, u_s, op_ld, kbase + ch * KW, n);
            }
        }
        if (stamp) stamp[3] = wall_clock64();
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
            if ((few && r >= 2) || (one && r >= 1)) continue;
            if (has_pred) { acc_h[r].x = dpp_row_sum16(acc_h[r].x); acc_h[r].y = dpp_row_sum16(acc_h[r].y);
                            acc_h[r].z = dpp_row_sum16(acc_h[r].z); acc_h[r].w = dpp_row_sum16(acc_h[r].w); }
            if (has_in) { acc_i[r].x = dpp_row_sum16(acc_i[r].x); acc_i[r].y = dpp_row_sum16(acc_i[r].y);
                          acc_i[r].z = dpp_row_sum16(acc_i[r].z); acc_i[r].w = dpp_row_sum16(acc_i[r].w); }
        }
        if (ksl == 15) {  // last lane of each DPP row holds the totals of column group 4*wave + row
            const int cg = 4 * wave + (lane >> 4);
#pragma unroll
            for (int r = 0; r < RBT; ++r) {
                *reinterpret_cast<float4*>(gh_s + r * SW + 4 * cg) = acc_h[r];
                *reinterpret_cast<float4*>(gi_s + r * SW + 4 * cg) = acc_i[r];
            }
        }
    }
    __syncthreads();
    if (stamp) stamp[4] = wall_clock64();

    // ---- phase C: gates for RBT rows x JS units
    if (tid < RBT * JS) {
        float sp = 0.f, hv = 0.f;
        if (gate_thread) {
            const float gr = pre_r + (has_in ? gi_s[gr_ * SW + gj_] : 0.f);
            const float gz = pre_z + (has_in ? gi_s[gr_ * SW + JS + gj_] : 0.f);
            const float gn = pre_n + (has_in ? gi_s[gr_ * SW + 2 * JS + gj_] : 0.f);
            const float hr = gh_s[gr_ * SW + gj_] + bh_r;
            const float hz = gh_s[gr_ * SW + JS + gj_] + bh_z;
            const float hn = gh_s[gr_ * SW + 2 * JS + gj_] + bh_n;
            const float a = a_s[gr_ * op_ld + apad(j, kpt)];
            const float rg = sigm(gr + hr);
            const float zg = sigm(gz + hz);
            const float ng = tanhf(fmaf(rg, hn, gn));
            hv = fmaf(zg, a - ng, ng);  // n + z * (a - n)
            sp = wk * hv;
        }
        // partial score of every 16-unit group (16 = one DPP row): summed by the consumers
        sp = dpp_row_sum16(sp);
        if (gate_thread) {
            float* po = C.h_out + (int64_t)gv * ld_h;
            po[j] = hv;
            if ((tid & 15) == 15) po[H + (j >> 4)] = sp;
            if (C.g_out) {  // granule copy: the hand-off format of the persistent tail kernel
                gran_t* pg = C.g_out + (int64_t)gv * (H + H / PU);
                if (RESIDENT) {  // consumers run in this very launch: one write-through 8-byte store each
                    __hip_atomic_store(pg + j, gran_pack(G.epoch, hv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((tid & 15) == 15)
                        __hip_atomic_store(pg + H + (j >> 4), gran_pack(G.epoch, sp), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                } else {         // consumers run in a later launch: the launch boundary publishes
                    pg[j] = gran_pack(G.epoch, hv);
                    if ((tid & 15) == 15) pg[H + (j >> 4)] = gran_pack(G.epoch, sp);
                }
            }
        }
    }
    if (stamp) stamp[5] = wall_clock64();
}

// ---- fat launches, stage 1: the aggregate of every frontier row ONCE (one wave per row, 4 rows per
// workgroup: a low-register, high-occupancy gather kernel) instead of once per weight slice.
__global__ void __launch_bounds__(256) aggregate_rows_kernel(const int32_t* __restrict__ plan, PlanLayout L, StepArgs S) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x * 4 + wave;          // row index over all cells of this launch
    int c = 0;
    while (c + 1 < S.ncell && row >= S.blk_start[c + 1]) ++c;   // blk_start = ROW prefix sums here
    if (row >= S.blk_start[S.ncell]) return;
    const Cell& C = S.cell[c];
    const int local = row - S.blk_start[c];
    const int d = C.dir, H = S.H, H4 = H >> 2;
    const int4* rp = reinterpret_cast<const int4*>(plan + L.rowrec[d]) + 4 * (int64_t)(C.row_base + local);
    const int4 rec0 = rp[0];
    float* out = const_cast<float*>(C.a_pre) + (int64_t)local * H;
    GranCtx G;
    G.epoch = S.epoch; G.err = nullptr;
    if (C.has_pred && rec0.z > rec0.y) {
        // kpt = H (no padding): apad(k, H) == k for k < H, so the rows land contiguously
        aggregate<false>(C, plan + L.col[d], reinterpret_cast<const float*>(plan + L.eattr[d]), rec0.y, rec0.z, rp[1],
                         rp[2], rp[3], H, S.ld_h, C.gain ? S.R : 0, S.vid_mod, H, out, lane, G);
    } else {
        for (int cc = lane; cc < H4; cc += 64) reinterpret_cast<float4*>(out)[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}


// ---- fat launches, stage 2 on the matrix cores: 32 frontier rows x one 32-unit slice per workgroup.
// The widest topological layers hold thousands of rows; with 8-row blocks every block re-reads its
// 98-196 KB weight slice (528 MB of L2->CU traffic for one 2 700-row layer).  Here the aggregates
// (stage 1, aggregate_rows_kernel) and the nodes' lower-layer rows of 32 rows are staged k-major in
// LDS and the slice GEMMs [32 x K] x [K x 96] run as v_mfma_f32_32x32x2_f32 chains (exact fp32, the
// fmaf order is k ascending), one (matrix, gate) chain per wave: 4x fewer weight bytes per row and 4x
// fewer workgroups.  The B fragments are pre-packed in lane order (dagnn_pack_mfma), 16 B per lane.
typedef float mf32x16 __attribute__((ext_vector_type(16)));
constexpr int MT = 32;          // rows per tile
constexpr int MLD = MT + 1;     // k-major LDS pitch (conflict-free lane == row reads)
constexpr int MKC = 256;        // K chunk staged in LDS at a time

__global__ void __launch_bounds__(512, 2) frontier_mfma_kernel(const int32_t* __restrict__ plan, PlanLayout L, StepArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = S.H, ld_h = S.ld_h, NS = H / 32;
    const int sl = blockIdx.x % NS;
    const int gb = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gb >= S.blk_start[c + 1]) ++c;   // blk_start = 32-row tile prefix sums
    const Cell& C = S.cell[c];
    const int slot0 = C.row_base + (gb - S.blk_start[c]) * MT;
    const int nr = min(MT, C.row_end - slot0);
    const int d = C.dir;
    const bool has_in = C.wih_m != nullptr;
    const bool has_pred = C.has_pred != 0;
    unsigned long long* stamp = (S.dbg != nullptr && blockIdx.x == 0 && tid == 0) ? S.dbg + 8 * (int64_t)S.step : nullptr;
    if (stamp) { stamp[0] = wall_clock64(); stamp[6] = gridDim.x; }

    const int KC = min(H, MKC);           // K is staged through LDS in chunks of <= MKC
    float* a_t = smem;                    // [KC][MLD]  aggregates, k-major
    float* u_t = a_t + KC * MLD;          // [KC][MLD]  own lower-layer rows, k-major; later the GEMM outputs
    float* g_s = u_t;                     // [2][MT][96] + [MT][64] after the MFMA phase (u_t is dead by then)
    int* v_s = reinterpret_cast<int*>(u_t + max(KC * MLD, 2 * MT * 96 + MT * 64));  // [MT] node ids

    const int4* __restrict__ recs = reinterpret_cast<const int4*>(plan + L.rowrec[d]);
    if (tid < MT) v_s[tid] = tid < nr ? recs[4 * (int64_t)(slot0 + tid)].x : 0;
    __syncthreads();

    // operands of the gate epilogue that do not depend on the products: issued now, consumed after the chains
    // {input-side pre-activations or biases (r, z, n), hidden-side biases (r, z, n), aggregate, key weight}
    float pre[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int id = p * 512 + tid;
        const int r = id >> 5, j = sl * 32 + (id & 31);
#pragma unroll
        for (int q = 0; q < 8; ++q) pre[p][q] = 0.f;
        if (r < nr) {
            if (has_in) { pre[p][0] = C.bih[j]; pre[p][1] = C.bih[H + j]; pre[p][2] = C.bih[2 * H + j]; }
            else {
                const float* g0 = C.gi0 + (int64_t)v_s[r] * 3 * H;
                pre[p][0] = g0[j]; pre[p][1] = g0[H + j]; pre[p][2] = g0[2 * H + j];
            }
            pre[p][3] = C.bhh[j]; pre[p][4] = C.bhh[H + j]; pre[p][5] = C.bhh[2 * H + j];
            pre[p][6] = C.a_pre[(int64_t)(slot0 + r - C.row_base) * H + j];
            pre[p][7] = C.wkey ? C.wkey[j] : 0.f;
        }
    }

    mf32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // chain c = matrix * 3 + gate.  With both products (6 chains on 4 SIMDs, two waves per SIMD) waves 0-3 run
    // chains 0-3 over the whole K and waves 4|5, 6|7 the two K halves of chains 4, 5: 1.5 chains per SIMD
    // instead of 2 on two of them; the second halves land in their own LDS tile and are added in the epilogue.
    const int cid = has_in ? (wave < 4 ? wave : 4 + ((wave - 4) >> 1)) : wave;
    const int khalf = (has_in && wave >= 4) ? ((wave - 4) & 1) : -1;   // -1: whole K
    const int mat = cid / 3, gate = cid - mat * 3;
    const bool chain = cid < 6 && (mat == 0 ? has_pred : has_in);
    const float* op = mat == 0 ? a_t : u_t;
    const float4* wp = chain ? (mat == 0 ? C.whh_m : C.wih_m) + ((int64_t)(sl * 3 + gate) * (H / 8)) * 64 + lane : nullptr;
    const int arow = lane & 31, ak = lane >> 5;

    for (int k0 = 0; k0 < H; k0 += KC) {
        const int kc = min(KC, H - k0);
        if (k0 > 0) __syncthreads();   // the previous chunk's MFMAs are done with a_t / u_t
        // this wave's k range of the chunk and its first group of B fragments: in flight during the staging
        const int k8n = kc >> 3, k8b = khalf < 0 ? 0 : khalf * (k8n >> 1), k8e = khalf < 0 ? k8n : k8b + (k8n >> 1);
        float4 wn[4];
        if (chain) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wn[q] = wp[(int64_t)((k0 >> 3) + k8b + q) * 64];
        }
        // ---- stage the operand chunk k-major: wave w copies rows w, w+8, w+16, w+24 (coalesced float4 row
        // reads); the loads of all four rows are issued before the first LDS store - one round trip, not four
        for (int cc = lane; cc < (kc >> 2); cc += 64) {
            float4 av[MT / 8], uv[MT / 8];
#pragma unroll
            for (int q = 0; q < MT / 8; ++q) {
                const int r = wave + 8 * q;
                av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                uv[q] = av[q];
                if (r < nr) {
                    av[q] = reinterpret_cast<const float4*>(C.a_pre + (int64_t)(slot0 + r - C.row_base) * H + k0)[cc];
                    if (has_in) uv[q] = reinterpret_cast<const float4*>(C.h_in + (int64_t)v_s[r] * ld_h + k0)[cc];
                }
            }
#pragma unroll
            for (int q = 0; q < MT / 8; ++q) {
                const int r = wave + 8 * q;
                a_t[(4 * cc + 0) * MLD + r] = av[q].x; a_t[(4 * cc + 1) * MLD + r] = av[q].y;
                a_t[(4 * cc + 2) * MLD + r] = av[q].z; a_t[(4 * cc + 3) * MLD + r] = av[q].w;
                if (has_in) {
                    u_t[(4 * cc + 0) * MLD + r] = uv[q].x; u_t[(4 * cc + 1) * MLD + r] = uv[q].y;
                    u_t[(4 * cc + 2) * MLD + r] = uv[q].z; u_t[(4 * cc + 3) * MLD + r] = uv[q].w;
                }
            }
        }
        __syncthreads();
        if (stamp) stamp[1] = wall_clock64();
        // ---- MFMA chains over this K chunk
        if (chain) {
            for (int k8 = k8b; k8 < k8e; k8 += 4) {   // 4 x 16 B of B fragments per group, the next group in flight
                float4 w4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w4[q] = wn[q];
                if (k8 + 4 < k8e) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) wn[q] = wp[(int64_t)((k0 >> 3) + k8 + 4 + q) * 64];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kb = 8 * (k8 + q) + ak;   // this lane's k (within the chunk) for the first MFMA of the fragment
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(op[(kb + 0) * MLD + arow], w4[q].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(op[(kb + 2) * MLD + arow], w4[q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(op[(kb + 4) * MLD + arow], w4[q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(op[(kb + 6) * MLD + arow], w4[q].w, acc, 0, 0, 0);
                }
            }
        }
    }
    if (stamp) stamp[2] = wall_clock64();
    __syncthreads();   // every chain has read u_t: it can now hold the outputs
    if (stamp) stamp[3] = wall_clock64();
    if (cid < 6) {
        // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
        const bool second = khalf == 1;   // second K half of chain 4 / 5: its own [MT][64] tile behind g_s
        float* out = second ? g_s + 2 * MT * 96 + (gate - 1) * 32 + (lane & 31) : g_s + mat * (MT * 96) + gate * 32 + (lane & 31);
        const int pitch = second ? 64 : 96;
#pragma unroll
        for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * pitch] = acc[e];
    }
    __syncthreads();

    if (stamp) stamp[4] = wall_clock64();
    // ---- gates: 32 rows x 32 units, two elements per thread; 16 consecutive lanes = 16 units of a row
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int id = p * 512 + tid;
        const int r = id >> 5, jj = id & 31;
        const bool live = r < nr;
        float sp = 0.f, hv = 0.f;
        const int gv = live ? v_s[r] : 0;
        const int j = sl * 32 + jj;
        if (live) {
            float gr = pre[p][0], gz = pre[p][1], gn = pre[p][2];
            if (has_in) {
                const float* gi = g_s + MT * 96 + r * 96;
                const float* g2 = g_s + 2 * MT * 96 + r * 64;
                gr += gi[jj]; gz += gi[32 + jj] + g2[jj]; gn += gi[64 + jj] + g2[32 + jj];
            }
            const float* gh = g_s + r * 96;
            const float hr = gh[jj] + pre[p][3], hz = gh[32 + jj] + pre[p][4], hn = gh[64 + jj] + pre[p][5];
            const float a = pre[p][6];
            const float rg = sigm(gr + hr);
            const float zg = sigm(gz + hz);
            const float ng = tanhf(fmaf(rg, hn, gn));
            hv = fmaf(zg, a - ng, ng);
            sp = pre[p][7] * hv;
        }
        sp = dpp_row_sum16(sp);
        if (live) {
            float* po = C.h_out + (int64_t)gv * ld_h;
            po[j] = hv;
            if ((tid & 15) == 15) po[H + (j >> 4)] = sp;
            if (C.g_out) {
                gran_t* pg = C.g_out + (int64_t)gv * (H + H / PU);
                pg[j] = gran_pack(S.epoch, hv);
                if ((tid & 15) == 15) pg[H + (j >> 4)] = gran_pack(S.epoch, sp);
            }
        }
    }
    if (stamp) stamp[5] = wall_clock64();
}

// Pack W [3H, K] (torch layout) into MFMA B-fragment order for 32-unit slices:
// out[((sl * 3 + g) * (K/8) + k8) * 64 + lane] (float4): element q = W[g*H + sl*32 + (lane & 31)][8*k8 + 2*q + (lane >> 5)],
// i.e. the B operand of the q-th of four consecutive v_mfma_f32_32x32x2_f32 (k pair 2*(4*k8+q)).
__device__ __forceinline__ void pack_mfma_range(const float* __restrict__ W, float4* __restrict__ out, int H, int K,
                                                int64_t total, int64_t first, int64_t stride) {
    const int k8n = K >> 3;
    for (int64_t idx = first; idx < total; idx += stride) {
        const int lane = (int)(idx & 63);
        int64_t rest = idx >> 6;
        const int k8 = (int)(rest % k8n); rest /= k8n;
        const int g = (int)(rest % 3);
        const int sl = (int)(rest / 3);
        const int64_t row = (int64_t)g * H + sl * 32 + (lane & 31);
        const int k = 8 * k8 + (lane >> 5);
        out[idx] = make_float4(W[row * K + k], W[row * K + k + 2], W[row * K + k + 4], W[row * K + k + 6]);
    }
}

__global__ void __launch_bounds__(256) pack_mfma_kernel(const float* __restrict__ W, float4* __restrict__ out, int H,
                                                         int K, int64_t total) {
    pack_mfma_range(W, out, H, K, total, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// ---- launch-per-layer kernel: one launch = one batch-level topological layer, all cells
template <int JS, int RBT, int KW, int MINW>
__global__ void __launch_bounds__(WgShape<RBT>::threads, MINW) frontier_step_kernel(const int32_t* __restrict__ plan, PlanLayout L, StepArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool prof = S.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long* stamp = prof ? S.dbg + 8 * (int64_t)S.step : nullptr;
    if (prof) stamp[0] = wall_clock64();
    const int NS = S.H / JS;
    const int sl = blockIdx.x % NS;
    const int gb = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gb >= S.blk_start[c + 1]) ++c;
    const Cell& C = S.cell[c];
    const int slot0 = C.row_base + (gb - S.blk_start[c]) * RBT;
    const int nr = min(RBT, C.row_end - slot0);
    const int d = C.dir;
    float4 wh[KW], wi[KW];
    GranCtx G;
    G.epoch = S.epoch; G.err = nullptr;
    process_block<JS, RBT, KW, false>(plan, L.rowrec[d], L.col[d], L.eattr[d], C, C.has_pred != 0, slot0, nr, sl, S.H,
                                      S.ld_h, S.R, S.vid_mod, smem, wh, wi, stamp, G);
    if (prof) stamp[6] = gridDim.x;
}

// ---- persistent tail kernel: ONE launch walks all remaining layers, as a dataflow.
// The tail of the schedule is hundreds of dependent layers with a handful of rows each; a launch
// boundary (~3 us) + kernarg fetch + weight reload + cross-die row fetch per layer is most of their
// cost, and a grid barrier would cost as much (measured: 5.3 us per layer for 64 workgroups).
// Here every workgroup owns (cell, slice, replica) for the whole tail, keeps its weight slice in
// registers, and walks its row blocks in schedule order WITHOUT any barrier: a row block starts as
// soon as the granules (tagged 8-byte copies, see above) of the rows it reads carry this pass's
// epoch.  Dependencies only point to earlier layers / the lower stacked layer, producers never wait
// on consumers and the grid is far below the CU count (all workgroups resident), so it cannot
// deadlock; spins are bounded anyway and report through err_flag.  Nothing depends on placement.
struct TailArgs {
    Cell cell[DAGNN_MAX_CELLS];
    int stacked_idx[DAGNN_MAX_CELLS];  // i of each cell (layer processed at step s is s - i)
    int ncell, nrep, H, ld_h, R, vid_mod;
    int s_begin, s_end;                // steps [s_begin, s_end)
    int use_split;                     // rows of a layer start at blsplit[t] (the deep graphs only) instead of blptr[t]
    unsigned epoch;
    int* err_flag;
    unsigned long long* dbg;
};

template <int JS, int RBT, int KW>
__global__ void __launch_bounds__(FT, 1) frontier_tail_kernel(const int32_t* __restrict__ plan, PlanLayout L, TailArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NCW = 3 * JS / 16;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NS = S.H / JS;
    const int sl = blockIdx.x % NS;
    const int rep = (blockIdx.x / NS) % S.nrep;
    const int c = blockIdx.x / (NS * S.nrep);
    const Cell& C = S.cell[c];
    const int d = C.dir, si = S.stacked_idx[c];
    const int N = plan[PH_N];
    const int32_t* __restrict__ blptr = plan + L.blptr[d];
    const int T = blptr[N + 1];
    const bool has_in = C.wih != nullptr;
    const int kpt = S.H >> 4;
    const bool prof = S.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    GranCtx G;
    G.epoch = S.epoch; G.err = S.err_flag;

    // resident weights: the whole slice (kpt <= KW, checked by the host)
    float4 wh[KW], wi[KW];
    {
        const int64_t wstride = (int64_t)NCW * 64;
        const float4* whh = C.whh + (int64_t)sl * kpt * wstride + tid;
        const float4* wih = has_in ? C.wih + (int64_t)sl * kpt * wstride + tid : nullptr;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            wh[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
            wi[kk] = wh[kk];
            if (wave < NCW && kk < kpt) {
                wh[kk] = whh[kk * wstride];
                if (has_in) wi[kk] = wih[kk * wstride];
            }
        }
    }
    for (int s = S.s_begin; s < S.s_end; ++s) {
        unsigned long long* stamp = prof ? S.dbg + 8 * (int64_t)s : nullptr;
        if (prof) { stamp[0] = wall_clock64(); stamp[6] = gridDim.x; }
        const int t = s - si;
        if (t < 0 || t >= T) continue;
        const int r0 = S.use_split ? plan[L.blsplit[d] + t] : blptr[t], r1 = blptr[t + 1];
        // rows of a thin layer are spread over the replicas (blocks of ceil(rows / nrep) <= RBT rows): a block's
        // latency grows with its live rows, and the layer is as slow as its slowest replica
        const int rbs = min(max((r1 - r0 + S.nrep - 1) / S.nrep, 1), RBT);
        const int nblk = (r1 - r0 + rbs - 1) / rbs;
        for (int rb = rep; rb < nblk; rb += S.nrep) {
            const int slot0 = r0 + rb * rbs;
            process_block<JS, RBT, KW, true>(plan, L.rowrec[d], L.col[d], L.eattr[d], C, t > 0, slot0,
                                             min(rbs, r1 - slot0), sl, S.H, S.ld_h, S.R, S.vid_mod, smem, wh, wi,
                                             stamp, G);
            __syncthreads();  // LDS is reused by the next block
        }
    }
}

// Pack W [3H, K] (torch GRUCell layout: row g*H + j, K contiguous) into slice / lane order for
// slices of JS units: out[((sl * kpt + kk) * NCW + w) * 64 + lane] (float4) = the 4 columns of column
// group cg = 4w + (lane >> 4) of slice sl at k = (lane & 15) * kpt + kk; local column lc = 4cg + q
// -> gate lc / JS, unit sl*JS + lc % JS.
__device__ __forceinline__ void pack_slices_range(const float* __restrict__ W, float4* __restrict__ out, int H, int K,
                                                  int JS, int64_t total, int64_t first, int64_t stride) {
    const int kpt = K >> 4;
    const int NCW = 3 * JS / 16;
    for (int64_t idx = first; idx < total; idx += stride) {
        const int lane = (int)(idx & 63);
        int64_t rest = idx >> 6;
        const int w = (int)(rest % NCW); rest /= NCW;
        const int kk = (int)(rest % kpt);
        const int sl = (int)(rest / kpt);
        const int cg = 4 * w + (lane >> 4);
        const int k = (lane & 15) * kpt + kk;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lc = 4 * cg + q;
            const int row = (lc / JS) * H + sl * JS + (lc % JS);
            v[q] = W[(int64_t)row * K + k];
        }
        out[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ void __launch_bounds__(256) pack_slices_kernel(const float* __restrict__ W, float4* __restrict__ out, int H,
                                                           int K, int JS, int64_t total) {
    pack_slices_range(W, out, H, K, JS, total, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                      (int64_t)gridDim.x * blockDim.x);
}

// All three layouts of up to DAGNN_MAX_PACK_JOBS matrices in ONE launch (training re-packs every cell every step:
// 18 small launches otherwise).  blockIdx.y = layout (16-unit slices, 32-unit slices, MFMA fragments), blockIdx.z = job.
struct PackJobs { dagnn_pack_job j[DAGNN_MAX_PACK_JOBS]; };
__global__ void __launch_bounds__(256) pack_batch_kernel(PackJobs P) {
    const dagnn_pack_job& J = P.j[blockIdx.z];
    const int64_t total = (int64_t)3 * J.H * J.K / 4;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if (blockIdx.y == 0) { if (J.out_slices16) pack_slices_range(J.w, reinterpret_cast<float4*>(J.out_slices16), J.H, J.K, 16, total, first, stride); }
    else if (blockIdx.y == 1) { if (J.out_slices32) pack_slices_range(J.w, reinterpret_cast<float4*>(J.out_slices32), J.H, J.K, 32, total, first, stride); }
    else if (J.out_mfma) pack_mfma_range(J.w, reinterpret_cast<float4*>(J.out_mfma), J.H, J.K, total, first, stride);
}

template <int JS, int RBT, int KW, int MINW>
hipError_t launch_step(int blocks, int H, hipStream_t st, const int32_t* plan, const PlanLayout& L, const StepArgs& S) {
    const int op_ld = H + 64;
    const size_t lds = (size_t)(2 * RBT * op_ld + 2 * RBT * 3 * JS) * sizeof(float) + RBT * sizeof(int);
    hipLaunchKernelGGL((frontier_step_kernel<JS, RBT, KW, MINW>), dim3((unsigned)(blocks * (H / JS))),
                       dim3(WgShape<RBT>::threads), lds, st, plan, L, S);
    return hipGetLastError();
}

}  // namespace

extern "C" int dagnn_pack_slices(const float* w, float* out, int H, int K, int slice_units, void* stream) {
    if (!w || !out || H <= 0 || K <= 0 || (slice_units != 16 && slice_units != 32) || (H % 32) || (K % 64))
        return DAGNN_EINVAL;
    const int64_t total = (int64_t)3 * H * K / 4;  // float4 elements
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_slices_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(out), H, K, slice_units, total);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_pack_batch(const dagnn_pack_job* jobs, int num_jobs, void* stream) {
    if (!jobs || num_jobs < 0 || num_jobs > DAGNN_MAX_PACK_JOBS) return DAGNN_EINVAL;
    if (num_jobs == 0) return DAGNN_OK;
    PackJobs P;
    int64_t most = 0;
    for (int i = 0; i < num_jobs; ++i) {
        const dagnn_pack_job& J = jobs[i];
        if (!J.w || J.H <= 0 || J.K <= 0 || (J.H % 32) || (J.K % 64)) return DAGNN_EINVAL;
        P.j[i] = J;
        const int64_t total = (int64_t)3 * J.H * J.K / 4;
        most = total > most ? total : most;
    }
    int64_t blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)blocks, 3, (unsigned)num_jobs), dim3(256), 0, (hipStream_t)stream, P);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

static void fill_cell(Cell& K, const dagnn_frontier_args* a, const dagnn_plan* pl, int d, int i, int js) {
    const dagnn_frontier_cell& c = a->cell[d][i];
    K.whh = (const float4*)(js == 16 ? c.w_hh_pk16 : c.w_hh_pk32);
    K.wih = i > 0 ? (const float4*)(js == 16 ? c.w_ih_pk16 : c.w_ih_pk32) : nullptr;
    K.whh_m = (const float4*)c.w_hh_mfma;
    K.wih_m = i > 0 ? (const float4*)c.w_ih_mfma : nullptr;
    K.bhh = c.b_hh; K.bih = c.b_ih; K.wkey = c.w_key; K.sscore = c.static_score;
    K.gain = pl->num_edge_feats > 0 ? c.edge_gain : nullptr;
    K.vid = a->vid_mod > 0 ? c.vid_bias : nullptr;
    K.gi0 = i == 0 ? c.gi0 : nullptr;
    K.h_in = i > 0 ? a->cell[d][i - 1].h_out : nullptr;
    K.h_out = c.h_out;
    K.g_out = (gran_t*)c.granules;
    K.g_in = i > 0 ? (const gran_t*)a->cell[d][i - 1].granules : nullptr;
    K.a_pre = nullptr;
    K.dir = d; K.row_base = 0; K.row_end = 0; K.has_pred = 0;
}

extern "C" int dagnn_pack_mfma(const float* w, float* out, int H, int K, void* stream) {
    if (!w || !out || H <= 0 || K <= 0 || (H % 32) || (K % 8)) return DAGNN_EINVAL;
    const int64_t total = (int64_t)3 * H * K / 4;  // float4 elements
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(out), H, K, total);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_frontier_run(const dagnn_plan* pl, const dagnn_frontier_args* a, const int32_t* const* layer_ptr,
                                  const int32_t* num_layers, void* stream) {
    if (!pl || !pl->data || !a || !layer_ptr || !num_layers) return DAGNN_EINVAL;
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3;
    if (H <= 0 || (H % 64) || Ls <= 0 || !dir_mask || a->ld_h < H + H / PU || (a->ld_h & 3) || a->num_cus <= 0)
        return DAGNN_EINVAL;
    int ndir = 0, dirs[2];
    for (int d = 0; d < 2; ++d) if ((dir_mask >> d) & 1) dirs[ndir++] = d;
    if (ndir * Ls > DAGNN_MAX_CELLS || Ls > DAGNN_MAX_STACKED) return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    int Tmax = 0;
    for (int q = 0; q < ndir; ++q) {
        const int d = dirs[q];
        if (!layer_ptr[d] || num_layers[d] < 0) return DAGNN_EINVAL;
        if (num_layers[d] > Tmax) Tmax = num_layers[d];
        for (int i = 0; i < Ls; ++i) {
            const dagnn_frontier_cell& c = a->cell[d][i];
            if (!c.w_hh_pk16 || !c.w_hh_pk32 || !c.b_hh || (!c.w_key && !c.static_score) || !c.h_out) return DAGNN_EINVAL;
            if (i == 0 ? !c.gi0 : (!c.w_ih_pk16 || !c.w_ih_pk32 || !c.b_ih)) return DAGNN_EINVAL;
        }
    }
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipStream_t st = (hipStream_t)stream;
    const int32_t* plan = (const int32_t*)pl->data;
    const int nsteps = Tmax + Ls - 1;

    // split mode (side stream + per-layer split pointers): the persistent kernel walks the DEEP graphs from layer 0
    // on the side stream while the launches below handle the shallow graphs' rows [ptr[t], split[t]) - the two
    // sets of graphs share nothing, so the deepest chains no longer wait behind the fat layers
    bool split = a->side_stream != nullptr && a->debug_timing == nullptr && a->fork_event != nullptr && a->join_event != nullptr;
    for (int q = 0; q < ndir && split; ++q) split = a->layer_split[dirs[q]] != nullptr;
    auto rows_all = [&](int d, int i, int s) {
        const int t = s - i;
        return (t < 0 || t >= num_layers[d]) ? 0 : layer_ptr[d][t + 1] - layer_ptr[d][t];
    };

    // ---- where the persistent tail starts: the first step after which no cell ever has more rows
    // than its replicas cover in `tail_max_blocks` blocks of 8.  Needs the whole slice in registers
    // (H <= 256), line-aligned state rows (ld_h % 32 == 0) and a sync workspace.
    int s_tail = nsteps;
    const int tail_js = a->tail_slice_units == 16 ? 16 : 32, tail_rb = 4;
    int nrep = a->tail_replicas > 0 ? a->tail_replicas : 0;
    const int tail_wgs = ndir * Ls * (H / tail_js) * (nrep > 0 ? nrep : 1);
    bool tail_ok = nrep > 0 && a->tail_err && a->epoch != 0 && H <= 256 && tail_wgs <= a->num_cus / 2;
    for (int q = 0; q < ndir && tail_ok; ++q)
        for (int i = 0; i < Ls; ++i) tail_ok = tail_ok && a->cell[dirs[q]][i].granules != nullptr;
    if (tail_ok) {
        const int cap = tail_rb * nrep * (a->tail_max_blocks > 0 ? a->tail_max_blocks : 1);
        s_tail = 0;
        for (int s = nsteps - 1; s >= 0; --s) {
            int mx = 0;
            for (int q = 0; q < ndir; ++q)
                for (int i = 0; i < Ls; ++i) mx = mx > rows_all(dirs[q], i, s) ? mx : rows_all(dirs[q], i, s);
            if (mx > cap) { s_tail = s + 1; break; }
        }
        if (nsteps - s_tail < 8) s_tail = nsteps;  // not worth a second kernel
    }
    split = split && tail_ok && s_tail < nsteps;
    auto row_lo = [&](int d, int t) { return layer_ptr[d][t]; };
    auto row_hi = [&](int d, int t) { return split ? a->layer_split[d][t] : layer_ptr[d][t + 1]; };
    auto rows_of = [&](int d, int i, int s) {
        const int t = s - i;
        return (t < 0 || t >= num_layers[d]) ? 0 : row_hi(d, t) - row_lo(d, t);
    };
    const int s_eager = split ? nsteps : s_tail;   // split mode: every step that still has shallow rows

    // ---- the persistent dataflow kernel (launched first in split mode, after the per-layer launches otherwise)
    auto launch_tail = [&](hipStream_t ts, int s_begin) -> int {
        TailArgs T;
        int nc = 0;
        for (int q = 0; q < ndir; ++q)
            for (int i = 0; i < Ls; ++i) {
                fill_cell(T.cell[nc], a, pl, dirs[q], i, tail_js);
                T.stacked_idx[nc++] = i;
            }
        T.ncell = nc; T.nrep = nrep; T.H = H; T.ld_h = a->ld_h; T.R = pl->num_edge_feats;
        T.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
        T.s_begin = s_begin; T.s_end = nsteps;
        T.use_split = split ? 1 : 0;
        T.epoch = a->epoch;
        T.err_flag = (int*)a->tail_err;
        T.dbg = (unsigned long long*)a->debug_timing;
        const int op_ld = H + 64;
        const size_t lds = (size_t)(2 * tail_rb * op_ld + 2 * tail_rb * 3 * tail_js) * sizeof(float) + tail_rb * sizeof(int);
        if (tail_js == 16)
            hipLaunchKernelGGL((frontier_tail_kernel<16, 4, 16>), dim3((unsigned)tail_wgs), dim3(FT), lds, ts, plan, L, T);
        else
            hipLaunchKernelGGL((frontier_tail_kernel<32, 4, 16>), dim3((unsigned)tail_wgs), dim3(FT), lds, ts, plan, L, T);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? DAGNN_OK : DAGNN_EHIP(e);
    };
    DagnnForkJoin fj;   // joins and releases its events on every return path
    bool forked = false;
    if (split) {   // no shallow rows at all (small batches: every graph is "deep"): nothing to overlap, no fork
        int64_t shallow = 0, deep = 0;
        for (int q = 0; q < ndir; ++q)
            for (int t = 0; t < num_layers[dirs[q]]; ++t) {
                shallow += a->layer_split[dirs[q]][t] - layer_ptr[dirs[q]][t];
                deep += layer_ptr[dirs[q]][t + 1] - a->layer_split[dirs[q]][t];
            }
        forked = shallow > 0 && deep > 0;
        if (!forked && deep > 0) {   // only deep graphs: the persistent kernel alone, on the caller's stream
            const int rc = launch_tail(st, 0);
            if (rc != DAGNN_OK) return rc;
        }                            // only shallow graphs: the per-layer launches alone
    }
    if (forked) {
        hipStream_t side = (hipStream_t)a->side_stream;
        const hipError_t ef = fj.begin(st, side, a->fork_event, a->join_event);
        if (ef != hipSuccess) return DAGNN_EHIP(ef);
        const int rc = launch_tail(side, 0);
        fj.mark();
        if (rc != DAGNN_OK) return rc;
    }

    StepArgs S;
    S.H = H; S.ld_h = a->ld_h; S.R = pl->num_edge_feats; S.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
    S.dbg = (unsigned long long*)a->debug_timing;
    S.epoch = a->epoch;
    for (int s = 0; s < s_eager; ++s) {
        // geometry of this launch: thin launches use 16-unit slices (and 4-row blocks when that
        // still fits one round of workgroups), fat ones 32-unit slices, 8-row blocks, 2 per CU
        int rows_total = 0, blocks8 = 0, blocks4 = 0;
        for (int q = 0; q < ndir; ++q)
            for (int i = 0; i < Ls; ++i) {
                const int n = rows_of(dirs[q], i, s);
                rows_total += n; blocks8 += (n + 7) / 8; blocks4 += (n + 3) / 4;
            }
        if (rows_total == 0) continue;
        // Launch shape {slice units, rows per block}: thin launches prefetch a whole 16-unit slice per
        // workgroup (1 workgroup per CU) and must fit ONE round of CUs; otherwise 32-unit slices with
        // streamed weights: 4-row blocks (96 VGPRs: three 6-wave workgroups per CU) while they fit one
        // round of slots, else 8-row blocks (125 VGPRs, two per CU).
        int js, rb;
        if (blocks4 * (H / 16) <= a->num_cus) { js = 16; rb = 4; }
        else if (blocks4 * (H / 32) <= (a->rb4_max_wgs > 0 ? a->rb4_max_wgs : 3 * a->num_cus / 2)) { js = 32; rb = 4; }
        else if (blocks8 * (H / 16) <= a->num_cus) { js = 16; rb = 8; }
        else { js = 32; rb = 8; }
        int nc = 0, blocks = 0;
        S.blk_start[0] = 0;
        for (int q = 0; q < ndir; ++q) {
            const int d = dirs[q];
            for (int i = 0; i < Ls; ++i) {
                const int n = rows_of(d, i, s);
                if (n <= 0) continue;
                Cell& K = S.cell[nc];
                fill_cell(K, a, pl, d, i, js);
                K.row_base = row_lo(d, s - i); K.row_end = K.row_base + n; K.has_pred = (s - i) > 0;
                if (split) K.g_out = nullptr;   // nothing of the shallow graphs is read through granules
                blocks += (n + rb - 1) / rb;
                S.blk_start[++nc] = blocks;
            }
        }
        S.ncell = nc;
        S.step = s;
        hipError_t e;
        // the fattest launches: aggregate every row once (stage 1), then 32-row MFMA tiles (stage 2)
        bool mfma_ok = a->agg_scratch != nullptr && rows_total <= a->agg_scratch_rows && (H <= MKC || H % MKC == 0) &&
                       a->mfma_min_rows > 0 && rows_total >= a->mfma_min_rows;
        for (int k = 0; k < nc && mfma_ok; ++k)
            mfma_ok = S.cell[k].whh_m != nullptr && (S.cell[k].wih == nullptr || S.cell[k].wih_m != nullptr);
        if (mfma_ok) {
            StepArgs A = S;
            int off = 0, tiles = 0;
            for (int k = 0; k < nc; ++k) {
                const int n = S.cell[k].row_end - S.cell[k].row_base;
                A.cell[k].a_pre = (const float*)a->agg_scratch + (int64_t)off * H;
                S.cell[k].a_pre = A.cell[k].a_pre;
                A.blk_start[k] = off;
                S.blk_start[k] = tiles;
                off += n;
                tiles += (n + MT - 1) / MT;
            }
            A.blk_start[nc] = off;
            S.blk_start[nc] = tiles;
            hipLaunchKernelGGL(aggregate_rows_kernel, dim3((unsigned)((off + 3) / 4)), dim3(256), 0, st, plan, L, A);
            e = hipGetLastError();
            if (e != hipSuccess) return DAGNN_EHIP(e);
            const int kcl = H < MKC ? H : MKC;
            const int outw = 2 * MT * 96 + MT * 64;
            const size_t lds = (size_t)(kcl * MLD + (kcl * MLD > outw ? kcl * MLD : outw)) * sizeof(float) +
                               MT * sizeof(int);
            hipLaunchKernelGGL(frontier_mfma_kernel, dim3((unsigned)(tiles * (H / 32))), dim3(512), lds, st, plan, L, S);
            e = hipGetLastError();
            if (e != hipSuccess) return DAGNN_EHIP(e);
            continue;
        }
        if (js == 32) e = rb == 8 ? launch_step<32, 8, 4, 4>(blocks, H, st, plan, L, S)
                                  : launch_step<32, 4, 4, 3>(blocks, H, st, plan, L, S);
        else if (rb == 4) e = launch_step<16, 4, 16, 1>(blocks, H, st, plan, L, S);
        else e = launch_step<16, 8, 16, 1>(blocks, H, st, plan, L, S);
        if (e != hipSuccess) return DAGNN_EHIP(e);
    }
    if (forked) {   // join (fj's destructor): the caller's stream continues only when the deep graphs are finished too
    } else if (!split && s_tail < nsteps) {
        const int rc = launch_tail(st, s_tail);
        if (rc != DAGNN_OK) return rc;
    }
    return DAGNN_OK;
}
