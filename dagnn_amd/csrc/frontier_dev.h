// frontier_dev.h - device-side types and helpers shared by the lock-step kernels (frontier.hip: per-layer launches
// and the persistent tail; fat.hip: the 64-row MFMA tiles of the fat launches).  Internal linkage: every translation
// unit gets its own copy.
#pragma once
#include "common.h"

#define DAGNN_MAX_CELLS 16

namespace dagnn_lockstep {   // (named: the type crosses translation units through dagnn_fat_launch)
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
    const float* a_pre;  // fat launches (csrc/fat.hip): scratch rows [row_end - row_base, H] for the aggregates of the rows
                         // with more than four predecessors; null in the per-layer kernels
    unsigned long long* g_out;        // [N,gld] {epoch tag, fp32 bits} granules of h_out rows + parts, or null
    const unsigned long long* g_in;   // granules of h_in, or null
    int dir;             // direction (selects the plan arrays)
    int row_base;        // first rowrec slot of the layer processed in this launch
    int row_end;         // one past the last
    int has_pred;        // layer > 0
};
}  // namespace dagnn_lockstep
using dagnn_lockstep::Cell;

namespace {

constexpr int FT = 384;     // threads per workgroup (6 waves)
constexpr int PU = 16;      // hidden units per stored score part

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

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace

// csrc/fat.hip: one fat launch of the lock-step schedule (64-row MFMA tiles); `cells` as dagnn_frontier_run fills them.
int dagnn_fat_launch(const int32_t* plan, const PlanLayout& L, const Cell* cells, int ncell, int H, int ld_h, int R, int vid_mod,
                     unsigned epoch, float* scratch, int num_cus, hipStream_t st);
