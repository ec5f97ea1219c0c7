// backward.hip - reverse-mode sweep of the recurrence (training path).
//
// Reference path replaced: what `loss.backward()` (ogbg-code/main_pyg.py:62) does to the loops of
// ogbg-code/model/dagnn.py:144-182 through torch autograd - one autograd node per (direction,
// topological layer, stacked layer) micro-step, each keeping [N,H]-sized intermediates.  Here the
// forward pass keeps only the states h[d][i]; everything else is recomputed in parallel and the
// dependent chain is walked once, in reverse lock-step:
//
//   prepare (parallel over all nodes, no dependency - every h is known):
//       a_v = sum_e alpha_e h_p,  alpha = segment softmax of  w_k.h_p + g.feat_e    (dagnn.py:366-373)
//       -> a[c] [N,H], alpha[c] [E] (stored by ORIGINAL edge id so both directions' CSRs find it);
//       gh = a W_hh^T + b_hh and, for stacked layers > 0, gi = h_lower W_ih^T + b_ih are two batched
//       MFMA GEMMs on the host side (dagnn_gemm_nt_bias).
//   sweep (dagnn_backward_run): launch s handles layer t = T-1-(s-(L-1-i)) of stacked layer i.  For a
//   node v of that layer every successor w (layers > t) is finished, so v PULLS its gradient - no atomics,
//   fixed summation order, bitwise reproducible:
//       G_v  = Gext_v + sum_{e=(v->w)} [ alpha_e da_w + ds_e w_k ],   ds_e = alpha_e da_w.(h_v - a_w)
//              (softmax backward: sum_e' alpha_e' (da_w.h_p') = da_w.a_w, so no per-edge state is kept)
//       GRU backward (gates recomputed from gi, gh):  dgi, dgh  (3H each)
//       da_v = z (.) G_v + W_hh^T dgh          <- the only matrix product on the dependent chain
//       du_v = W_ih^T dgi -> Gext of stacked layer i-1 (consumed by the next launch)
//   The successor list of v in direction d is the predecessor record of v in direction 1-d (same
//   edges, flipped), which the plan already holds.
//   Outputs for the parallel epilogue on the host side (library GEMMs): dgi, dgh [N,3H] (weight and
//   bias gradients are dgi^T u, dgh^T a and column sums), sigma_v = sum_e ds_e [N] and
//   sum_e ds_e feat_e [N,R] (attention-key and edge-encoder gradients).
//
// Work split of a launch: workgroup = (cell, block of RB rows, slice of 16 hidden units), 8 waves.  Wave w
// pulls row w, all threads do the gate algebra of the full rows (cheap, recomputed per slice), then every
// lane multiplies its k-group of the [3H x 16] weight slices (torch layout, no packed copy; for H <= 256 the
// slice sits in registers, loaded before anything else) against LDS broadcasts of the gradients, and the 32
// partial sums meet in LDS in k-group order.
#include "common.h"

#define DAGNN_BWD_MAX_CELLS 16

namespace {

constexpr int BT = 256;   // threads per workgroup (prepare kernel)
constexpr int ST = 512;   // threads per workgroup of the sweep kernel (8 waves)
constexpr int SW = ST / 64;
constexpr int KREG = 24;  // k values per lane and matrix held in registers (3H/32 for H = 256)
constexpr int BJS = 16;   // hidden units per slice (a wave = 16 units x 4 k-groups)
constexpr int BPU = 16;   // hidden units per stored score part (frontier.hip PU)

struct BCell {
    const float* whh;    // [3H,H] torch layout (gate blocks r,z,n)
    const float* wih;    // [3H,H] or null (stacked layer 0)
    const float* wkey;   // [H]
    const float* gain;   // [R] or null
    const float* vid;    // [vid_mod] per-vertex-id score bias of the NA variant (dvae/dagnn.py:130-134) or null
    const float* sscore; // [N] static attention score of every node (keys taken from the inputs x), or null
    const float* h;      // [N,ld_h] states + partial scores of this cell (forward output)
    const float* a;      // [N,H]
    float* a_w;          // same buffer, written by the prepare kernel
    float* alpha;        // [E] by original edge id
    const float* gi;     // [N,3H]
    const float* gh;     // [N,3H]
    const float* gext;   // [N,H] gradient reaching h from outside the cell (read-out, upper stacked layer)
    float* gext_lo;      // [N,H] the same buffer of stacked layer i-1 (du is added), or null
    float* da;           // [N,H]
    float* dgi;          // [N,3H]
    float* dgh;          // [N,3H]
    float* sig;          // [N]
    float* mrel;         // [N,R] or null
    // persistent sweep only: tagged 8-byte copies (see common.h) of the rows other workgroups of the SAME launch wait for
    gran_t* da_g;        // [N,H] granules of da
    const gran_t* du_in; // [N,H] granules of the upper stacked layer's du for this cell's rows, or null (top layer)
    gran_t* du_out;      // [N,H] granules of this cell's du (== du_in of stacked layer i-1), or null
    const float* gext0;  // [N,H] gext as it was before the sweep (du arrives through du_in), or null (top layer)
    int dir, row_base, row_end;
    int stacked, T;      // persistent sweep: stacked layer index i, number of topological layers of this direction
};

struct BArgs {
    BCell cell[DAGNN_BWD_MAX_CELLS];
    int blk_start[DAGNN_BWD_MAX_CELLS + 1];
    int ncell, H, ld_h, R, vid_mod;
};

__device__ __forceinline__ float bsigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float bscore(const float* __restrict__ tail, int nparts) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += tail[q];   // index order, as the forward consumer sums them
    return s;
}

// ---- prepare: one wave per (cell, row): a_v and alpha_e of every in-edge -------------------------
__global__ void __launch_bounds__(BT) bwd_prepare_kernel(const int32_t* __restrict__ plan, PlanLayout L, BArgs S, int N) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= N) return;
    const BCell& C = S.cell[blockIdx.y];
    const int d = C.dir, H = S.H, H4 = H >> 2, R = C.gain ? S.R : 0, ld_h = S.ld_h, nparts = H / BPU;
    const int4 rec = reinterpret_cast<const int4*>(plan + L.rowrec[d])[4 * (int64_t)slot];
    const int v = rec.x, eb = rec.y, ee = rec.z;
    const int32_t* col = plan + L.col[d];
    const int32_t* eidx = plan + L.eidx[d];
    const float* eattr = reinterpret_cast<const float*>(plan + L.eattr[d]);
    float* arow = C.a_w + (int64_t)v * H;
    if (ee <= eb) {
        for (int c = lane; c < H4; c += 64) reinterpret_cast<float4*>(arow)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    auto logit = [&](int e) {
        float s = C.sscore ? C.sscore[col[e]] : bscore(C.h + (int64_t)col[e] * ld_h + H, nparts);
        if (C.vid) s += C.vid[col[e] % S.vid_mod];
        for (int r = 0; r < R; ++r) s = fmaf(C.gain[r], eattr[(int64_t)e * R + r], s);
        return s;
    };
    float mx = -INFINITY, sum = 0.f;
    for (int e = eb + lane; e < ee; e += 64) mx = fmaxf(mx, logit(e));
    mx = wave_max(mx);
    for (int e = eb + lane; e < ee; e += 64) sum += expf(logit(e) - mx);
    sum = wave_sum(sum);
    const float denom = sum + 1e-16f;   // PyG softmax: exp(x - max) / (sum + 1e-16)
    for (int c0 = 0; c0 < H4; c0 += 64) {
        const int c = c0 + lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = eb; base < ee; base += 64) {
            const int e = base + lane;
            float my_al = 0.f;
            int my_col = 0;
            if (e < ee) {
                my_col = col[e];
                my_al = expf(logit(e) - mx) / denom;
                if (c0 == 0) C.alpha[eidx[e]] = my_al;
            }
            const int cnt = min(64, ee - base);
            for (int i = 0; i < cnt; ++i) {
                const float a1 = __shfl(my_al, i, 64);
                const int cj = __shfl(my_col, i, 64);
                if (c < H4) {
                    const float4 x = reinterpret_cast<const float4*>(C.h + (int64_t)cj * ld_h)[c];
                    acc.x = fmaf(a1, x.x, acc.x); acc.y = fmaf(a1, x.y, acc.y);
                    acc.z = fmaf(a1, x.z, acc.z); acc.w = fmaf(a1, x.w, acc.w);
                }
            }
        }
        if (c < H4) reinterpret_cast<float4*>(arow)[c] = acc;
    }
}

// ---- successor records: brec[d][slot] (64 B) = {node, first/last CSR slot of its row in direction 1-d,
// 0, first four successors, their original edge ids, 4 spare words}.  The dependent chain of a sweep
// launch becomes record -> successor rows (the forward kernels' rowrec trick, built once per batch).
__global__ void __launch_bounds__(256) bwd_succrec_kernel(int32_t* plan, PlanLayout L, int N) {
    const int d = blockIdx.y, od = 1 - d;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= N) return;
    const int v = plan[L.rowrec[d] + 16 * (int64_t)slot];
    const int4* orec = reinterpret_cast<const int4*>(plan + L.rowrec[od]) + 4 * (int64_t)plan[L.pos[od] + v];
    const int4 r0 = orec[0], r1 = orec[1];
    const int eb = r0.y, ee = r0.z;
    const int32_t* eidx = plan + L.eidx[od];
    int4* out = reinterpret_cast<int4*>(plan + L.brec[d] + 16 * (int64_t)slot);
    out[0] = make_int4(v, eb, ee, 0);
    out[1] = r1;
    out[2] = make_int4(eb < ee ? eidx[eb] : 0, eb + 1 < ee ? eidx[eb + 1] : 0, eb + 2 < ee ? eidx[eb + 2] : 0,
                       eb + 3 < ee ? eidx[eb + 3] : 0);
    out[3] = make_int4(0, 0, 0, 0);
}

// One wave: grow[0..H) (LDS) = Gext_v + sum over the successors of the node of record `rec` (see file header);
// `publish`: also store sigma_v and the edge-feature sums.
// GRAN (persistent sweep, H <= 256): successor rows da_w and the upper layer's du_v were written by other
// workgroups of this very launch - they are read, and waited for, through their granule copies.
// One wave: float4 chunk `lane` of a granule row, waiting for it.
__device__ __forceinline__ float4 bgran_chunk(const gran_t* grow, int lane, int H4, const GranCtx& G) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned spins = 0;
    for (;;) {
        gran_t x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = lane < H4 ? gran_ld(grow + 4 * lane + q) : ((gran_t)G.epoch << 32);
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(x[q] >> 32) == G.epoch;
        v = make_float4(__uint_as_float((unsigned)x[0]), __uint_as_float((unsigned)x[1]),
                        __uint_as_float((unsigned)x[2]), __uint_as_float((unsigned)x[3]));
        if (__all(ok) || !gran_retry(spins, G)) break;
    }
    return v;
}

template <bool GRAN>
__device__ __forceinline__ void pull_row(const int32_t* __restrict__ plan, const PlanLayout& L, const BCell& C,
                                         const int4* __restrict__ rec, int R, int H, int ld_h, float* grow, int lane,
                                         bool publish, const GranCtx& G) {
    const int H4 = H >> 2, od = 1 - C.dir;
    const int4 b0 = rec[0];
    const int v = b0.x, eb = b0.y, ee = b0.z, deg = ee - eb;
    const float* eattr = reinterpret_cast<const float*>(plan + L.eattr[od]);
    const float* hv = C.h + (int64_t)v * ld_h;
    float sig = 0.f, m0 = 0.f, m1 = 0.f;
    if (deg <= 4 && H4 <= 64) {
        // inline path: successor ids and edge ids came with the record; all rows in one round trip
        const int4 b1 = rec[1], b2 = rec[2];
        auto pick = [](const int4& b, int e) { return e == 0 ? b.x : e == 1 ? b.y : e == 2 ? b.z : b.w; };
        const bool on = lane < H4;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g = zero, y = zero, wk = zero;
        const bool via_du = GRAN && C.du_in != nullptr;
        if (on) {
            g = reinterpret_cast<const float4*>((via_du ? C.gext0 : C.gext) + (int64_t)v * H)[lane];
            y = reinterpret_cast<const float4*>(hv)[lane];
            wk = reinterpret_cast<const float4*>(C.wkey)[lane];
        }
        float4 x[4], z[4];
        float al[4], f0[4], f1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[e] = zero; z[e] = zero; al[e] = 0.f; f0[e] = 0.f; f1[e] = 0.f;
            if (e < deg) {
                al[e] = C.alpha[pick(b2, e)];
                if (on) {
                    if (!GRAN) x[e] = reinterpret_cast<const float4*>(C.da + (int64_t)pick(b1, e) * H)[lane];
                    z[e] = reinterpret_cast<const float4*>(C.a + (int64_t)pick(b1, e) * H)[lane];
                }
                if (R >= 1) f0[e] = eattr[(int64_t)(eb + e) * R];
                if (R >= 2) f1[e] = eattr[(int64_t)(eb + e) * R + 1];
            }
        }
        if (GRAN) {
            // one polling loop for the du row and all (<= 4) successor rows; every load of an iteration is issued
            // before the first tag is looked at (atomic loads keep program order: a compare between two groups
            // would serialise them)
            const gran_t ready = (gran_t)G.epoch << 32;
            unsigned spins = 0;
            float4 du = zero;
            for (;;) {
                gran_t xu[4], xs[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) xu[q] = (via_du && on) ? gran_ld(C.du_in + (int64_t)v * H + 4 * lane + q) : ready;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        xs[e][q] = (e < deg && on) ? gran_ld(C.da_g + (int64_t)pick(b1, e) * H + 4 * lane + q) : ready;
                bool ok = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(xu[q] >> 32) == G.epoch;
                du = make_float4(__uint_as_float((unsigned)xu[0]), __uint_as_float((unsigned)xu[1]),
                                 __uint_as_float((unsigned)xu[2]), __uint_as_float((unsigned)xu[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(xs[e][q] >> 32) == G.epoch;
                    x[e] = make_float4(__uint_as_float((unsigned)xs[e][0]), __uint_as_float((unsigned)xs[e][1]),
                                       __uint_as_float((unsigned)xs[e][2]), __uint_as_float((unsigned)xs[e][3]));
                }
                if (__all(ok) || !gran_retry(spins, G)) break;
            }
            if (via_du) { g.x += du.x; g.y += du.y; g.z += du.z; g.w += du.w; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= deg) continue;
            float dot = x[e].x * (y.x - z[e].x);
            dot = fmaf(x[e].y, y.y - z[e].y, dot); dot = fmaf(x[e].z, y.z - z[e].z, dot);
            dot = fmaf(x[e].w, y.w - z[e].w, dot);
            const float ds = al[e] * wave_sum(dot);
            sig += ds;
            m0 = fmaf(ds, f0[e], m0); m1 = fmaf(ds, f1[e], m1);
            g.x = fmaf(al[e], x[e].x, g.x); g.y = fmaf(al[e], x[e].y, g.y);
            g.z = fmaf(al[e], x[e].z, g.z); g.w = fmaf(al[e], x[e].w, g.w);
        }
        g.x = fmaf(sig, wk.x, g.x); g.y = fmaf(sig, wk.y, g.y); g.z = fmaf(sig, wk.z, g.z); g.w = fmaf(sig, wk.w, g.w);
        if (on) reinterpret_cast<float4*>(grow)[lane] = g;
    } else if (H4 <= 64) {
        // fan-out > 4, one chunk per lane: successors in groups of four, every load of a group in flight together
        const int32_t* col = plan + L.col[od];
        const int32_t* eidx = plan + L.eidx[od];
        const bool on = lane < H4;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g = zero, y = zero, wk = zero;
        const bool via_du = GRAN && C.du_in != nullptr;
        if (on) {
            g = reinterpret_cast<const float4*>((via_du ? C.gext0 : C.gext) + (int64_t)v * H)[lane];
            y = reinterpret_cast<const float4*>(hv)[lane];
            wk = reinterpret_cast<const float4*>(C.wkey)[lane];
        }
        if (via_du) {
            const float4 du = bgran_chunk(C.du_in + (int64_t)v * H, lane, H4, G);
            g.x += du.x; g.y += du.y; g.z += du.z; g.w += du.w;
        }
        for (int base = eb; base < ee; base += 4) {
            const int n = min(4, ee - base);
            int w[4];
            float al[4];
            float4 x[4], z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = u < n ? col[base + u] : 0;
                al[u] = u < n ? C.alpha[eidx[base + u]] : 0.f;
                x[u] = zero; z[u] = zero;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < n && on) {
                    if (!GRAN) x[u] = reinterpret_cast<const float4*>(C.da + (int64_t)w[u] * H)[lane];
                    z[u] = reinterpret_cast<const float4*>(C.a + (int64_t)w[u] * H)[lane];
                }
            }
            if (GRAN) {
                const gran_t ready = (gran_t)G.epoch << 32;
                unsigned spins = 0;
                for (;;) {
                    gran_t xs[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            xs[u][q] = (u < n && on) ? gran_ld(C.da_g + (int64_t)w[u] * H + 4 * lane + q) : ready;
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(xs[u][q] >> 32) == G.epoch;
                        x[u] = make_float4(__uint_as_float((unsigned)xs[u][0]), __uint_as_float((unsigned)xs[u][1]),
                                           __uint_as_float((unsigned)xs[u][2]), __uint_as_float((unsigned)xs[u][3]));
                    }
                    if (__all(ok) || !gran_retry(spins, G)) break;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= n) continue;
                float dot = x[u].x * (y.x - z[u].x);
                dot = fmaf(x[u].y, y.y - z[u].y, dot); dot = fmaf(x[u].z, y.z - z[u].z, dot);
                dot = fmaf(x[u].w, y.w - z[u].w, dot);
                const float ds = al[u] * wave_sum(dot);
                sig += ds;
                if (R >= 1) m0 = fmaf(ds, eattr[(int64_t)(base + u) * R], m0);
                if (R >= 2) m1 = fmaf(ds, eattr[(int64_t)(base + u) * R + 1], m1);
                g.x = fmaf(al[u], x[u].x, g.x); g.y = fmaf(al[u], x[u].y, g.y);
                g.z = fmaf(al[u], x[u].z, g.z); g.w = fmaf(al[u], x[u].w, g.w);
            }
        }
        g.x = fmaf(sig, wk.x, g.x); g.y = fmaf(sig, wk.y, g.y); g.z = fmaf(sig, wk.z, g.z); g.w = fmaf(sig, wk.w, g.w);
        if (on) reinterpret_cast<float4*>(grow)[lane] = g;
    } else {
        // wide rows (H > 256; never in the persistent sweep): successors one at a time, G accumulated in LDS
        const int32_t* col = plan + L.col[od];
        const int32_t* eidx = plan + L.eidx[od];
        for (int cc = lane; cc < H4; cc += 64)
            reinterpret_cast<float4*>(grow)[cc] = reinterpret_cast<const float4*>(C.gext + (int64_t)v * H)[cc];
        for (int e = eb; e < ee; ++e) {
            const int w = col[e];
            const float al = C.alpha[eidx[e]];
            const float* daw = C.da + (int64_t)w * H;
            const float* aw = C.a + (int64_t)w * H;
            float dot = 0.f;
            for (int cc = lane; cc < H4; cc += 64) {
                const float4 x = reinterpret_cast<const float4*>(daw)[cc];
                const float4 y = reinterpret_cast<const float4*>(hv)[cc];
                const float4 z = reinterpret_cast<const float4*>(aw)[cc];
                dot = fmaf(x.x, y.x - z.x, dot); dot = fmaf(x.y, y.y - z.y, dot);
                dot = fmaf(x.z, y.z - z.z, dot); dot = fmaf(x.w, y.w - z.w, dot);
                float4 g = reinterpret_cast<float4*>(grow)[cc];
                g.x = fmaf(al, x.x, g.x); g.y = fmaf(al, x.y, g.y); g.z = fmaf(al, x.z, g.z); g.w = fmaf(al, x.w, g.w);
                reinterpret_cast<float4*>(grow)[cc] = g;
            }
            const float ds = al * wave_sum(dot);
            sig += ds;
            if (R >= 1) m0 = fmaf(ds, eattr[(int64_t)e * R], m0);
            if (R >= 2) m1 = fmaf(ds, eattr[(int64_t)e * R + 1], m1);
        }
        for (int cc = lane; cc < H4; cc += 64) {
            const float4 k = reinterpret_cast<const float4*>(C.wkey)[cc];
            float4 g = reinterpret_cast<float4*>(grow)[cc];
            g.x = fmaf(sig, k.x, g.x); g.y = fmaf(sig, k.y, g.y); g.z = fmaf(sig, k.z, g.z); g.w = fmaf(sig, k.w, g.w);
            reinterpret_cast<float4*>(grow)[cc] = g;
        }
    }
    if (publish && lane == 0) {
        C.sig[v] = sig;
        if (R >= 1) C.mrel[(int64_t)v * R] = m0;
        if (R >= 2) C.mrel[(int64_t)v * R + 1] = m1;
    }
}

// ---- one reverse lock-step launch -----------------------------------------------------------------
// Workgroup = (cell, block of RB rows, slice of 16 hidden units), 8 waves.  Lane = (unit, k-group): the 32
// k-groups of a workgroup each own 3H/32 rows of the [3H x 16] weight slices.
// LDS (floats): g_s[RB][H] | dgh_t[3H][RB] | dgi_t[3H][RB] | red[2][32][RB][16]
// PRE (H <= 256): every lane issues the loads of its whole k-range of both weight slices into registers
// first, then the operands of the gate algebra (they depend only on the node ids of the records), and only
// then walks record -> successor rows; the products are pure LDS-broadcast + FMA.  The dependent chain of a
// thin launch - which is what bounds the sweep - is two memory round trips plus three barriers.
// One row block of one cell x one 16-unit slice.  PERSIST: called from the persistent sweep - the weight slices are
// already in wr / wr2, rows other workgroups wait for are also published as granules.
template <int RB, bool PRE, bool PERSIST>
__device__ __forceinline__ void bwd_block(const int32_t* __restrict__ plan, const PlanLayout& L, const BCell& C, int H,
                                          int ld_h, int Rfeat, int row0, int nrows, int slice, float* lds,
                                          float (&wr)[(PRE && !PERSIST) ? KREG : 1],
                                          float (&wr2)[(PRE && !PERSIST) ? KREG : 1], const float* w_lds,
                                          const GranCtx& G) {
    const int H3 = 3 * H;
    float* g_s = lds;
    float* dgh_t = g_s + RB * H;
    float* dgi_t = dgh_t + H3 * RB;
    float* red = dgi_t + H3 * RB;
    constexpr int KG = ST / BJS;               // k-groups per workgroup (32)
    constexpr int NI = PRE ? RB * 256 / ST : 1;  // gate-algebra elements per thread (H <= 256)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = C.dir, R = C.mrel ? Rfeat : 0;
    const int ul = lane & (BJS - 1);
    const int unit = slice * BJS + ul;
    const int kg = wave * (64 / BJS) + (lane / BJS);
    const int KQ = H3 / KG, k0 = kg * KQ;
    const int4* brec = reinterpret_cast<const int4*>(plan + L.brec[d]) + 4 * (int64_t)row0;

    float pf[NI][7];
    if (PRE) {
        if (!PERSIST) {
            const float* wp = C.whh + (int64_t)k0 * H + unit;
#pragma unroll
            for (int k = 0; k < KREG; ++k) wr[k] = k < KQ ? wp[(int64_t)k * H] : 0.f;
            if (C.wih) {
                const float* wq = C.wih + (int64_t)k0 * H + unit;
#pragma unroll
                for (int k = 0; k < KREG; ++k) wr2[k] = k < KQ ? wq[(int64_t)k * H] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + j * ST, r = idx / H, u = idx - r * H;
#pragma unroll
            for (int q = 0; q < 7; ++q) pf[j][q] = 0.f;
            if (r < nrows) {
                const int v = brec[4 * r].x;
                const float* gi = C.gi + (int64_t)v * H3;
                const float* gh = C.gh + (int64_t)v * H3;
                pf[j][0] = gi[u]; pf[j][1] = gi[H + u]; pf[j][2] = gi[2 * H + u];
                pf[j][3] = gh[u]; pf[j][4] = gh[H + u]; pf[j][5] = gh[2 * H + u];
                pf[j][6] = C.a[(int64_t)v * H + u];
            }
        }
    }

    // ---- 1. pull: G_v = Gext_v + sum over successors, one wave per row
    for (int r = wave; r < nrows; r += SW)
        pull_row<PERSIST>(plan, L, C, brec + 4 * r, R, H, ld_h, g_s + r * H, lane, slice == 0, G);
    __syncthreads();

    // ---- 2. GRU backward of the full rows (gates recomputed); operands of the products go to LDS k-major
    auto gate_algebra = [&](int idx, float gir, float giz, float gin, float ghr, float ghz, float ghn, float av) {
        const int r = idx / H, u = idx - r * H;
        float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, zg = 0.f;
        if (r < nrows) {
            const float rr = bsigm(gir + ghr);
            const float zz = bsigm(giz + ghz);
            const float nn = tanhf(gin + rr * ghn);
            const float Gv = g_s[idx];
            dn = Gv * (1.0f - zz) * (1.0f - nn * nn);         // d pre-activation of n
            dz = Gv * (av - nn) * zz * (1.0f - zz);           // d pre-activation of z
            dr = dn * ghn * rr * (1.0f - rr);                 // d pre-activation of r
            dnr = dn * rr;                                    // hidden-side n input sits behind r
            zg = Gv * zz;                                     // direct path h' = n + z (a - n)
            if (u / BJS == slice) {
                const int v = brec[4 * r].x;
                float* og = C.dgi + (int64_t)v * H3;
                float* oh = C.dgh + (int64_t)v * H3;
                og[u] = dr; og[H + u] = dz; og[2 * H + u] = dn;
                oh[u] = dr; oh[H + u] = dz; oh[2 * H + u] = dnr;
            }
        }
        g_s[idx] = zg;
        dgh_t[u * RB + r] = dr; dgh_t[(H + u) * RB + r] = dz; dgh_t[(2 * H + u) * RB + r] = dnr;
        dgi_t[u * RB + r] = dr; dgi_t[(H + u) * RB + r] = dz; dgi_t[(2 * H + u) * RB + r] = dn;
    };
    if (PRE) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + j * ST;
            if (idx < RB * H) gate_algebra(idx, pf[j][0], pf[j][1], pf[j][2], pf[j][3], pf[j][4], pf[j][5], pf[j][6]);
        }
    } else {
        for (int idx = tid; idx < RB * H; idx += ST) {
            const int r = idx / H, u = idx - r * H;
            float q[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r < nrows) {
                const int v = brec[4 * r].x;
                const float* gi = C.gi + (int64_t)v * H3;
                const float* gh = C.gh + (int64_t)v * H3;
                q[0] = gi[u]; q[1] = gi[H + u]; q[2] = gi[2 * H + u];
                q[3] = gh[u]; q[4] = gh[H + u]; q[5] = gh[2 * H + u];
                q[6] = C.a[(int64_t)v * H + u];
            }
            gate_algebra(idx, q[0], q[1], q[2], q[3], q[4], q[5], q[6]);
        }
    }
    __syncthreads();

    // ---- 3. da[slice] = W_hh^T dgh, du[slice] = W_ih^T dgi: k-group q owns k in [q*3H/32, (q+1)*3H/32)
    float acc[RB], acc2[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    auto fma_rows = [&](float (&a)[RB], float wv, const float* dp) {
#pragma unroll
        for (int r4 = 0; r4 < RB; r4 += 4) {
            const float4 d4 = *reinterpret_cast<const float4*>(dp + r4);
            a[r4] = fmaf(wv, d4.x, a[r4]); a[r4 + 1] = fmaf(wv, d4.y, a[r4 + 1]);
            a[r4 + 2] = fmaf(wv, d4.z, a[r4 + 2]); a[r4 + 3] = fmaf(wv, d4.w, a[r4 + 3]);
        }
    };
    if (PERSIST) {   // resident slices in LDS: [3H][16] per matrix (registers are needed by the polling loops)
        const float* wl = w_lds + k0 * BJS + ul;
#pragma unroll 8
        for (int k = 0; k < KQ; ++k) fma_rows(acc, wl[k * BJS], dgh_t + (k0 + k) * RB);
        if (C.wih) {
            const float* wl2 = wl + H3 * BJS;
#pragma unroll 8
            for (int k = 0; k < KQ; ++k) fma_rows(acc2, wl2[k * BJS], dgi_t + (k0 + k) * RB);
        }
    } else if (PRE) {
#pragma unroll
        for (int k = 0; k < KREG; ++k)
            if (k < KQ) fma_rows(acc, wr[k], dgh_t + (k0 + k) * RB);
        if (C.wih) {
#pragma unroll
            for (int k = 0; k < KREG; ++k)
                if (k < KQ) fma_rows(acc2, wr2[k], dgi_t + (k0 + k) * RB);
        }
    } else {
        const float* wp = C.whh + (int64_t)k0 * H + unit;
#pragma unroll 16
        for (int k = 0; k < KQ; ++k) fma_rows(acc, wp[(int64_t)k * H], dgh_t + (k0 + k) * RB);
        if (C.wih) {
            const float* wq = C.wih + (int64_t)k0 * H + unit;
#pragma unroll 16
            for (int k = 0; k < KQ; ++k) fma_rows(acc2, wq[(int64_t)k * H], dgi_t + (k0 + k) * RB);
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        red[(kg * RB + r) * BJS + ul] = acc[r];
        red[((KG + kg) * RB + r) * BJS + ul] = acc2[r];
    }
    __syncthreads();

    // ---- 4. partial sums in k-group order, stores
    for (int idx = tid; idx < 2 * nrows * BJS; idx += ST) {
        const int m = idx / (nrows * BJS), rem = idx - m * nrows * BJS;
        const int r = rem / BJS, l = rem - r * BJS, u = slice * BJS + l;
        if (m == 1 && !C.gext_lo) continue;
        const int v = brec[4 * r].x;
        float s = 0.f;
#pragma unroll 8
        for (int w = 0; w < KG; ++w) s += red[((m * KG + w) * RB + r) * BJS + l];
        if (m == 0) {
            const float val = g_s[r * H + u] + s;
            C.da[(int64_t)v * H + u] = val;
            if (PERSIST) __hip_atomic_store(C.da_g + (int64_t)v * H + u, gran_pack(G.epoch, val), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
        } else {
            C.gext_lo[(int64_t)v * H + u] += s;   // this workgroup is the only writer of these 16 floats
            if (PERSIST) __hip_atomic_store(C.du_out + (int64_t)v * H + u, gran_pack(G.epoch, s), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int RB, bool PRE>
__global__ void __launch_bounds__(ST) bwd_step_kernel(const int32_t* __restrict__ plan, PlanLayout L, BArgs S) {
    extern __shared__ float lds[];
    const int NS = S.H / BJS;
    const int slice = blockIdx.x % NS, gblk = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gblk >= S.blk_start[c + 1]) ++c;
    const BCell& C = S.cell[c];
    const int row0 = C.row_base + (gblk - S.blk_start[c]) * RB;
    float wr[PRE ? KREG : 1], wr2[PRE ? KREG : 1];
    GranCtx G;
    G.epoch = 0; G.err = nullptr;
    bwd_block<RB, PRE, false>(plan, L, C, S.H, S.ld_h, S.R, row0, min(RB, C.row_end - row0), slice, lds, wr, wr2, nullptr, G);
}

// ---- persistent sweep over the thin head of the reverse order (the deepest layers come first): ONE launch walks
// steps [s_begin, s_end).  Workgroup = (cell, 16-unit slice, replica) with its two [3H x 16] weight slices resident
// in registers; the rows of a layer are spread over the replicas.  A row block starts as soon as the rows it pulls
// are there: da rows and the upper stacked layer's du rows travel as {epoch, value} granules (common.h) - no
// barrier between layers, bounded spins, nothing placement-dependent; the grid stays far below the CU count so
// every workgroup is resident.  Same arithmetic, same order as the per-layer launches: bitwise the same result.
struct BTailArgs {
    BArgs S;
    int nrep, s_begin, s_end, num_stacked;
    int use_split;   // rows of a layer start at blsplit[t] (the deep graphs only) instead of blptr[t]
    unsigned epoch;
    int* err;
};

__global__ void __launch_bounds__(ST) bwd_tail_kernel(const int32_t* __restrict__ plan, PlanLayout L, BTailArgs A) {
    extern __shared__ float lds[];
    constexpr int RB = 4;
    const BArgs& S = A.S;
    const int H = S.H, H3 = 3 * H, NS = H / BJS;
    const int slice = blockIdx.x % NS;
    const int rep = (blockIdx.x / NS) % A.nrep;
    const BCell& C = S.cell[blockIdx.x / (NS * A.nrep)];
    const int tid = threadIdx.x;
    GranCtx G;
    G.epoch = A.epoch; G.err = A.err;
    // resident weight slices: [3H][16] of W_hh, then of W_ih, behind the operand buffers
    float* w_lds = lds + (RB * H + 2 * H3 * RB + 2 * ST * RB);
    for (int i = tid; i < H3 * BJS; i += ST) {
        const int k = i / BJS, u = i - k * BJS;
        w_lds[i] = C.whh[(int64_t)k * H + slice * BJS + u];
        w_lds[H3 * BJS + i] = C.wih ? C.wih[(int64_t)k * H + slice * BJS + u] : 0.f;
    }
    __syncthreads();
    float wr[1], wr2[1];
    const int32_t* __restrict__ blptr = plan + L.blptr[C.dir];
    for (int s = A.s_begin; s < A.s_end; ++s) {
        const int t = C.T - 1 - (s - (A.num_stacked - 1 - C.stacked));   // the top stacked layer leads
        if (t < 0 || t >= C.T) continue;
        const int r0 = A.use_split ? plan[L.blsplit[C.dir] + t] : blptr[t], r1 = blptr[t + 1];
        const int rbs = min(max((r1 - r0 + A.nrep - 1) / A.nrep, 1), RB);
        for (int slot0 = r0 + rep * rbs; slot0 < r1; slot0 += A.nrep * rbs) {
            bwd_block<RB, true, true>(plan, L, C, H, S.ld_h, S.R, slot0, min(rbs, r1 - slot0), slice, lds, wr, wr2, w_lds, G);
            __syncthreads();   // LDS is reused by the next block
        }
    }
}

// ---- fat launches, stage 1: one wave per row of the launch (all cells): pull + GRU backward of the full row,
// once (the slice workgroups of the thin kernel each redo it).  Writes dgi, dgh, sigma / edge-feature sums
// and da = z (.) G; stage 2 adds the matrix products.  blk_start = ROW prefix sums here.
__global__ void __launch_bounds__(256) bwd_rows_kernel(const int32_t* __restrict__ plan, PlanLayout L, BArgs S) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x * 4 + wave;
    if (row >= S.blk_start[S.ncell]) return;
    int c = 0;
    while (c + 1 < S.ncell && row >= S.blk_start[c + 1]) ++c;
    const BCell& C = S.cell[c];
    const int H = S.H, H3 = 3 * H, H4 = H >> 2, R = C.mrel ? S.R : 0;
    const int4* rec = reinterpret_cast<const int4*>(plan + L.brec[C.dir]) + 4 * (int64_t)(C.row_base + row - S.blk_start[c]);
    float* grow = lds + wave * H;
    const int v = rec[0].x;
    GranCtx G;
    G.epoch = 0; G.err = nullptr;
    pull_row<false>(plan, L, C, rec, R, H, S.ld_h, grow, lane, true, G);
    const float4* gi = reinterpret_cast<const float4*>(C.gi + (int64_t)v * H3);
    const float4* gh = reinterpret_cast<const float4*>(C.gh + (int64_t)v * H3);
    float4* og = reinterpret_cast<float4*>(C.dgi + (int64_t)v * H3);
    float4* oh = reinterpret_cast<float4*>(C.dgh + (int64_t)v * H3);
    for (int cc = lane; cc < H4; cc += 64) {
        const float4 ir = gi[cc], iz = gi[H4 + cc], in = gi[2 * H4 + cc];
        const float4 hr = gh[cc], hz = gh[H4 + cc], hn = gh[2 * H4 + cc];
        const float4 av = reinterpret_cast<const float4*>(C.a + (int64_t)v * H)[cc];
        const float4 G = reinterpret_cast<const float4*>(grow)[cc];   // written by this wave's own lanes (same lane, same cc)
        float4 dr, dz, dn, dnr, zg;
#define DAGNN_GATE_BWD(f)                                                                    \
        {                                                                                    \
            const float rr = bsigm(ir.f + hr.f), zz = bsigm(iz.f + hz.f);                    \
            const float nn = tanhf(in.f + rr * hn.f);                                        \
            dn.f = G.f * (1.0f - zz) * (1.0f - nn * nn);                                     \
            dz.f = G.f * (av.f - nn) * zz * (1.0f - zz);                                     \
            dr.f = dn.f * hn.f * rr * (1.0f - rr);                                           \
            dnr.f = dn.f * rr;                                                               \
            zg.f = G.f * zz;                                                                 \
        }
        DAGNN_GATE_BWD(x) DAGNN_GATE_BWD(y) DAGNN_GATE_BWD(z) DAGNN_GATE_BWD(w)
#undef DAGNN_GATE_BWD
        og[cc] = dr; og[H4 + cc] = dz; og[2 * H4 + cc] = dn;
        oh[cc] = dr; oh[H4 + cc] = dz; oh[2 * H4 + cc] = dnr;
        reinterpret_cast<float4*>(C.da + (int64_t)v * H)[cc] = zg;
    }
}

// ---- fat launches, stage 2 on the matrix cores: 32 rows x one 32-unit slice per workgroup, 8 waves.
//   da[rows, slice] += dgh[rows, 3H] W_hh[3H, slice],   Gext_lower[rows, slice] += dgi[rows, 3H] W_ih[3H, slice]
// as v_mfma_f32_32x32x2_f32 chains (exact fp32).  K = 3H is walked gate block by gate block in chunks of <= 256:
// the chunk of the 32 gradient rows is staged k-major in LDS (dgi == dgh for the r and z blocks, so one copy
// serves both products), every wave owns 1/8 of the chunk's k range for both products and reads its B
// fragments straight from the torch-layout weights (128 B per half-wave), and the eight partial tiles meet in
// LDS in wave order.  Weight bytes per row: 1/4 of the 8-row blocks; pull and gate algebra are not redone.
typedef float bmf32x16 __attribute__((ext_vector_type(16)));
constexpr int BMT = 32;        // rows per tile
constexpr int BMLD = BMT + 1;  // k-major LDS pitch
constexpr int BKC = 256;       // K chunk

__global__ void __launch_bounds__(512, 2) bwd_mfma_kernel(const int32_t* __restrict__ plan, PlanLayout L, BArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = S.H, H3 = 3 * H, NS = H / 32;
    const int sl = blockIdx.x % NS, gb = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gb >= S.blk_start[c + 1]) ++c;   // blk_start = 32-row tile prefix sums
    const BCell& C = S.cell[c];
    const int slot0 = C.row_base + (gb - S.blk_start[c]) * BMT;
    const int nr = min(BMT, C.row_end - slot0);
    const bool has_in = C.wih != nullptr;
    const int KC = min(H, BKC);
    float* d_h = smem;                 // [KC][BMLD] chunk of dgh, k-major
    float* d_i = d_h + KC * BMLD;      // [KC][BMLD] chunk of dgi (n block only)
    float* red = smem;                 // [2][8][32][BMLD] partial tiles after the chains
    int* v_s = reinterpret_cast<int*>(smem + max(2 * KC * BMLD, 2 * 8 * BMT * BMLD));

    const int4* __restrict__ recs = reinterpret_cast<const int4*>(plan + L.brec[C.dir]);
    if (tid < BMT) v_s[tid] = tid < nr ? recs[4 * (int64_t)(slot0 + tid)].x : 0;
    __syncthreads();

    bmf32x16 acc_h, acc_i;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc_h[e] = 0.f; acc_i[e] = 0.f; }
    const int arow = lane & 31, ak = lane >> 5;
    const int kw = KC / 8, kb = wave * kw, nm = kw / 2;   // this wave's k range inside a chunk, MFMAs per product
    const int colw = sl * 32 + arow;

    for (int g = 0; g < 3; ++g) {
        for (int k0 = 0; k0 < H; k0 += KC) {
            const int kg = g * H + k0;   // global k of the chunk
            // B fragments of this wave for the chunk: issued before the staging so they are in flight during it
            float bh[16], bi[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                bh[j] = 0.f; bi[j] = 0.f;
                if (j < nm) {
                    const int64_t off = (int64_t)(kg + kb + 2 * j + ak) * H + colw;
                    bh[j] = C.whh[off];
                    if (has_in) bi[j] = C.wih[off];
                }
            }
            __syncthreads();   // the previous chunk's MFMAs are done with d_h / d_i
            const bool two = has_in && g == 2;
            for (int cc = lane; cc < (KC >> 2); cc += 64) {   // all four rows of this wave in flight together
                float4 x[BMT / 8], y[BMT / 8];
#pragma unroll
                for (int q = 0; q < BMT / 8; ++q) {
                    const int r = wave + 8 * q;
                    x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    y[q] = x[q];
                    if (r < nr) {
                        x[q] = reinterpret_cast<const float4*>(C.dgh + (int64_t)v_s[r] * H3 + kg)[cc];
                        if (two) y[q] = reinterpret_cast<const float4*>(C.dgi + (int64_t)v_s[r] * H3 + kg)[cc];
                    }
                }
#pragma unroll
                for (int q = 0; q < BMT / 8; ++q) {
                    const int r = wave + 8 * q;
                    d_h[(4 * cc + 0) * BMLD + r] = x[q].x; d_h[(4 * cc + 1) * BMLD + r] = x[q].y;
                    d_h[(4 * cc + 2) * BMLD + r] = x[q].z; d_h[(4 * cc + 3) * BMLD + r] = x[q].w;
                    if (two) {
                        d_i[(4 * cc + 0) * BMLD + r] = y[q].x; d_i[(4 * cc + 1) * BMLD + r] = y[q].y;
                        d_i[(4 * cc + 2) * BMLD + r] = y[q].z; d_i[(4 * cc + 3) * BMLD + r] = y[q].w;
                    }
                }
            }
            __syncthreads();
            const float* opi = two ? d_i : d_h;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < nm) {
                    const int kk = (kb + 2 * j + ak) * BMLD + arow;
                    acc_h = __builtin_amdgcn_mfma_f32_32x32x2f32(d_h[kk], bh[j], acc_h, 0, 0, 0);
                    if (has_in) acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(opi[kk], bi[j], acc_i, 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();   // every chain has read the staging buffers: they can now hold the partial tiles
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        red[((0 * 8 + wave) * BMT + row) * BMLD + arow] = acc_h[e];
        if (has_in) red[((1 * 8 + wave) * BMT + row) * BMLD + arow] = acc_i[e];
    }
    __syncthreads();
    for (int id = tid; id < BMT * 32; id += 512) {
        const int r = id >> 5, col = id & 31;
        if (r >= nr) continue;
        const int v = v_s[r];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s0 += red[((0 * 8 + w) * BMT + r) * BMLD + col];
        C.da[(int64_t)v * H + sl * 32 + col] += s0;   // da holds z (.) G from stage 1; one writer per element
        if (has_in) {
#pragma unroll
            for (int w = 0; w < 8; ++w) s1 += red[((1 * 8 + w) * BMT + r) * BMLD + col];
            C.gext_lo[(int64_t)v * H + sl * 32 + col] += s1;
        }
    }
}

inline size_t mfma_lds_bytes(int H) {
    const int KC = H < BKC ? H : BKC;
    const int a = 2 * KC * BMLD, b = 2 * 8 * BMT * BMLD;
    return (size_t)((a > b ? a : b) + BMT) * sizeof(float);
}

// Gradient of the max read-out (dagnn.py:184-193): the gradient of out[g, col_off + j] goes to the FIRST
// output node of graph g attaining the maximum of column j (scatter_max semantics: one winner).
__global__ void __launch_bounds__(256) readout_max_bwd_kernel(const int32_t* __restrict__ plan, PlanLayout L, int dir,
                                                               const float* __restrict__ h, int ld_h, int width,
                                                               const float* __restrict__ gout, int ld_out, int col_off,
                                                               float* __restrict__ gh, int ld_g) {
    const int g = blockIdx.x;
    const int od = 1 - dir;
    const int n0 = plan[L.node_ptr + g];
    const int depth = plan[L.depth[od] + g];
    const int32_t* ls = plan + L.lstart[od] + n0 + g;
    const int32_t* order = plan + L.order[od];
    const int p0 = depth > 0 ? ls[0] : 0, p1 = depth > 0 ? ls[1] : 0;
    if (p1 <= p0) return;
    for (int j = threadIdx.x; j < width; j += blockDim.x) {
        int best = order[p0];
        float m = h[(int64_t)best * ld_h + j];
        for (int p = p0 + 1; p < p1; ++p) {
            const int v = order[p];
            const float x = h[(int64_t)v * ld_h + j];
            if (x > m) { m = x; best = v; }
        }
        gh[(int64_t)best * ld_g + j] += gout[(int64_t)g * ld_out + col_off + j];
    }
}

// the same for several (state buffer, direction, column block) jobs in one launch (blockIdx.y = job)
struct RbJobs { dagnn_readout_bwd_job j[DAGNN_MAX_READOUT_JOBS]; };
__global__ void __launch_bounds__(256) readout_max_bwd_batch_kernel(const int32_t* __restrict__ plan, PlanLayout L, RbJobs J,
                                                                     const float* __restrict__ gout, int ld_out) {
    const dagnn_readout_bwd_job& K = J.j[blockIdx.y];
    const int g = blockIdx.x;
    const int od = 1 - K.dir;
    const int n0 = plan[L.node_ptr + g];
    const int depth = plan[L.depth[od] + g];
    const int32_t* ls = plan + L.lstart[od] + n0 + g;
    const int32_t* order = plan + L.order[od];
    const int p0 = depth > 0 ? ls[0] : 0, p1 = depth > 0 ? ls[1] : 0;
    if (p1 <= p0) return;
    for (int j = threadIdx.x; j < K.width; j += blockDim.x) {
        int best = order[p0];
        float m = K.h[(int64_t)best * K.ld_h + j];
        for (int p = p0 + 1; p < p1; ++p) {
            const int v = order[p];
            const float x = K.h[(int64_t)v * K.ld_h + j];
            if (x > m) { m = x; best = v; }
        }
        K.grad_h[(int64_t)best * K.ld_g + j] += gout[(int64_t)g * ld_out + K.col_off + j];
    }
}

inline bool H3_fits_registers(int H) { return 3 * H / (ST / BJS) <= KREG; }

template <int RB>
size_t step_lds_bytes(int H) { return (size_t)(RB * H + 2 * 3 * H * RB + 2 * ST * RB) * sizeof(float); }

void fill_cells(BArgs& S, const dagnn_backward_args* a, const int* dirs, int ndir) {
    S.ncell = 0;
    for (int q = 0; q < ndir; ++q)
        for (int i = 0; i < a->num_stacked; ++i) {
            const dagnn_backward_cell& c = a->cell[dirs[q]][i];
            BCell& K = S.cell[S.ncell++];
            K.whh = (const float*)c.w_hh; K.wih = i > 0 ? (const float*)c.w_ih : nullptr;
            K.wkey = (const float*)c.w_key; K.gain = (const float*)c.edge_gain;
            K.vid = a->vid_mod > 0 ? (const float*)c.vid_bias : nullptr;
            K.sscore = (const float*)c.static_score;
            K.h = (const float*)c.h; K.a = (const float*)c.a; K.a_w = (float*)c.a; K.alpha = (float*)c.alpha;
            K.gi = (const float*)c.gi; K.gh = (const float*)c.gh; K.gext = (const float*)c.g_ext;
            K.gext_lo = i > 0 ? (float*)a->cell[dirs[q]][i - 1].g_ext : nullptr;
            K.da = (float*)c.da; K.dgi = (float*)c.dgi; K.dgh = (float*)c.dgh; K.sig = (float*)c.sigma;
            K.mrel = (float*)c.edge_feat_grad;
            K.da_g = (gran_t*)c.da_granules;
            K.du_in = i + 1 < a->num_stacked ? (const gran_t*)c.du_granules : nullptr;
            K.du_out = i > 0 ? (gran_t*)a->cell[dirs[q]][i - 1].du_granules : nullptr;
            K.gext0 = i + 1 < a->num_stacked ? (const float*)c.g_ext_static : nullptr;
            K.stacked = i; K.T = 0;
            K.dir = dirs[q]; K.row_base = 0; K.row_end = 0;
        }
}

}  // namespace

extern "C" int dagnn_backward_prepare(const dagnn_plan* pl, const dagnn_backward_args* a, void* stream) {
    if (!pl || !pl->data || !a) return DAGNN_EINVAL;
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3;
    if (H <= 0 || (H % 64) || Ls <= 0 || !dir_mask || a->ld_h < H + H / BPU) return DAGNN_EINVAL;
    int ndir = 0, dirs[2];
    for (int d = 0; d < 2; ++d) if ((dir_mask >> d) & 1) dirs[ndir++] = d;
    if (ndir * Ls > DAGNN_BWD_MAX_CELLS) return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    for (int q = 0; q < ndir; ++q)
        for (int i = 0; i < Ls; ++i) {
            const dagnn_backward_cell& c = a->cell[dirs[q]][i];
            if (!c.h || !c.a || !c.alpha || (pl->num_edge_feats > 0 && !c.edge_gain)) return DAGNN_EINVAL;
        }
    BArgs S;
    S.H = H; S.ld_h = a->ld_h; S.R = pl->num_edge_feats; S.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
    fill_cells(S, a, dirs, ndir);
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipLaunchKernelGGL(bwd_succrec_kernel, dim3((unsigned)((pl->N + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream,
                       (int32_t*)pl->data, L, (int)pl->N);
    DAGNN_CHECK_LAUNCH();
    dim3 grid((unsigned)((pl->N + 3) / 4), (unsigned)S.ncell);
    hipLaunchKernelGGL(bwd_prepare_kernel, grid, dim3(BT), 0, (hipStream_t)stream, (const int32_t*)pl->data, L, S,
                       (int)pl->N);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_backward_run(const dagnn_plan* pl, const dagnn_backward_args* a, const int32_t* const* layer_ptr,
                                  const int32_t* num_layers, void* stream) {
    if (!pl || !pl->data || !a || !layer_ptr || !num_layers) return DAGNN_EINVAL;
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3;
    if (H <= 0 || (H % 64) || H > 1024 || Ls <= 0 || !dir_mask || a->ld_h < H + H / BPU || a->num_cus <= 0)
        return DAGNN_EINVAL;
    int ndir = 0, dirs[2];
    for (int d = 0; d < 2; ++d) if ((dir_mask >> d) & 1) dirs[ndir++] = d;
    if (ndir * Ls > DAGNN_BWD_MAX_CELLS) return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    int Tmax = 0;
    for (int q = 0; q < ndir; ++q) {
        const int d = dirs[q];
        if (!layer_ptr[d] || num_layers[d] < 0) return DAGNN_EINVAL;
        if (num_layers[d] > Tmax) Tmax = num_layers[d];
        for (int i = 0; i < Ls; ++i) {
            const dagnn_backward_cell& c = a->cell[d][i];
            if (!c.w_hh || !c.w_key || !c.h || !c.a || !c.alpha || !c.gi || !c.gh || !c.g_ext || !c.da || !c.dgi ||
                !c.dgh || !c.sigma || (i > 0 && !c.w_ih) || (pl->num_edge_feats > 0 && !c.edge_feat_grad))
                return DAGNN_EINVAL;
        }
    }
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipStream_t st = (hipStream_t)stream;
    const int32_t* plan = (const int32_t*)pl->data;
    BArgs S;
    S.H = H; S.ld_h = a->ld_h; S.R = pl->num_edge_feats; S.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
    fill_cells(S, a, dirs, ndir);
    const int NS = H / BJS;
    const bool pre = H3_fits_registers(H);
    const bool mfma = H <= BKC || H % BKC == 0;   // fat launches: rows kernel + 32-row MFMA tiles
    const int rb_fat = H <= 512 ? 8 : 4;          // ... or bigger row blocks of the slice kernel when H does not fit
    // dynamic LDS beyond 64 KB must be granted per kernel; idempotent, so no library-global state is kept
    const void* k4 = pre ? reinterpret_cast<const void*>(bwd_step_kernel<4, true>) : reinterpret_cast<const void*>(bwd_step_kernel<4, false>);
    const void* k8 = pre ? reinterpret_cast<const void*>(bwd_step_kernel<8, true>) : reinterpret_cast<const void*>(bwd_step_kernel<8, false>);
    if (hipFuncSetAttribute(k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds_bytes<4>(H)) != hipSuccess ||
        hipFuncSetAttribute(k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds_bytes<8>(H <= 512 ? H : 512)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)mfma_lds_bytes(H)) != hipSuccess)
        return DAGNN_EHIP(hipGetLastError());
    const int nsteps = Tmax + Ls - 1;
    // Row range of layer t that a chain covers: everything, or - split mode - the shallow graphs' rows [ptr, split) /
    // the deep graphs' rows [split, ptr').  The two sets of graphs share nothing, so in split mode the sweep runs as
    // two independent chains: the shallow graphs' per-layer launches on the side stream, and on the caller's stream
    // the deep graphs - persistent kernel over the thin head of the reverse order, then per-layer launches.
    bool split = a->side_stream != nullptr && a->fork_event != nullptr && a->join_event != nullptr;
    for (int q = 0; q < ndir && split; ++q) split = a->layer_split[dirs[q]] != nullptr;
    enum { ALL = 0, SHALLOW = 1, DEEP = 2 };
    auto row_lo = [&](int part, int d, int t) { return part == DEEP ? a->layer_split[d][t] : layer_ptr[d][t]; };
    auto row_hi = [&](int part, int d, int t) { return part == SHALLOW ? a->layer_split[d][t] : layer_ptr[d][t + 1]; };
    if (split) {   // nothing to overlap unless both kinds of graphs exist
        int64_t shallow = 0, deep = 0;
        for (int q = 0; q < ndir; ++q)
            for (int t = 0; t < num_layers[dirs[q]]; ++t) {
                shallow += row_hi(SHALLOW, dirs[q], t) - row_lo(SHALLOW, dirs[q], t);
                deep += row_hi(DEEP, dirs[q], t) - row_lo(DEEP, dirs[q], t);
            }
        split = shallow > 0 && deep > 0;
    }

    // per-layer launches of steps [s_from, nsteps) for one part of the rows, on stream `ss`
    auto run_steps = [&](hipStream_t ss, int s_from, int part) -> int {
        BArgs P = S;
        for (int s = s_from; s < nsteps; ++s) {
            // stacked layer i handles layer t = T_d - 1 - (s - (Ls-1-i)): the top layer leads, every lower one is a launch behind
            auto layout = [&](int unit) {   // row ranges of the active cells; blk_start in blocks of `unit` rows
                int k = 0, tot = 0;
                for (int q = 0; q < ndir; ++q)
                    for (int i = 0; i < Ls; ++i, ++k) {
                        const int d = dirs[q];
                        const int t = num_layers[d] - 1 - (s - (Ls - 1 - i));
                        const bool on = t >= 0 && t < num_layers[d];
                        P.cell[k].row_base = on ? row_lo(part, d, t) : 0;
                        P.cell[k].row_end = on ? row_hi(part, d, t) : 0;
                        P.blk_start[k] = tot;
                        tot += (P.cell[k].row_end - P.cell[k].row_base + unit - 1) / unit;
                    }
                P.blk_start[k] = tot;
                return tot;
            };
            const int rows = layout(1);
            if (rows == 0) continue;
            const int blocks4 = layout(4);
            const bool thin = blocks4 * NS <= (a->thin_wgs > 0 ? a->thin_wgs : 2 * a->num_cus);
            if (thin || (!mfma && rb_fat == 4)) {
                const dim3 grid((unsigned)(blocks4 * NS));
                if (pre) hipLaunchKernelGGL((bwd_step_kernel<4, true>), grid, dim3(ST), step_lds_bytes<4>(H), ss, plan, L, P);
                else hipLaunchKernelGGL((bwd_step_kernel<4, false>), grid, dim3(ST), step_lds_bytes<4>(H), ss, plan, L, P);
            } else if (!mfma) {
                const dim3 grid((unsigned)(layout(8) * NS));
                if (pre) hipLaunchKernelGGL((bwd_step_kernel<8, true>), grid, dim3(ST), step_lds_bytes<8>(H), ss, plan, L, P);
                else hipLaunchKernelGGL((bwd_step_kernel<8, false>), grid, dim3(ST), step_lds_bytes<8>(H), ss, plan, L, P);
            } else {
                layout(1);
                hipLaunchKernelGGL(bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 4 * H * sizeof(float), ss, plan, L, P);
                DAGNN_CHECK_LAUNCH();
                const int tiles = layout(BMT);
                hipLaunchKernelGGL(bwd_mfma_kernel, dim3((unsigned)(tiles * (H / 32))), dim3(512), mfma_lds_bytes(H), ss, plan, L, P);
            }
            DAGNN_CHECK_LAUNCH();
        }
        return DAGNN_OK;
    };

    DagnnForkJoin fj;   // joins and releases its events on every return path
    if (split) {   // fork: the shallow graphs' chain on the side stream
        hipStream_t side = (hipStream_t)a->side_stream;
        const hipError_t ef = fj.begin(st, side, a->fork_event, a->join_event);
        if (ef != hipSuccess) return DAGNN_EHIP(ef);
        const int rc = run_steps(side, 0, SHALLOW);
        fj.mark();
        if (rc != DAGNN_OK) return rc;
    }
    const int part = split ? DEEP : ALL;

    // ---- the thin head of the reverse order (deepest layers first) in ONE persistent launch: the prefix of steps in
    // which no cell has more rows than its replicas cover in `tail_max_blocks` 4-row blocks
    int s_first = 0;
    {
        const int nrep = a->tail_replicas;
        bool ok = pre && nrep > 0 && a->epoch != 0 && a->tail_err && S.ncell * NS * nrep <= a->num_cus / 2;
        for (int k = 0; k < S.ncell && ok; ++k)
            ok = S.cell[k].da_g && (S.cell[k].stacked + 1 == Ls || (S.cell[k].du_in && S.cell[k].gext0));
        if (ok) {
            const int cap = 4 * nrep * (a->tail_max_blocks > 0 ? a->tail_max_blocks : 1);
            int s_end = 0;
            for (; s_end < nsteps; ++s_end) {
                int mx = 0;
                for (int q = 0; q < ndir; ++q)
                    for (int i = 0; i < Ls; ++i) {
                        const int d = dirs[q], t = num_layers[d] - 1 - (s_end - (Ls - 1 - i));
                        if (t >= 0 && t < num_layers[d]) mx = mx > row_hi(part, d, t) - row_lo(part, d, t) ? mx : row_hi(part, d, t) - row_lo(part, d, t);
                    }
                if (mx > cap) break;
            }
            if (s_end >= 8) {   // worth a persistent launch
                BTailArgs A;
                A.S = S;
                for (int k = 0; k < S.ncell; ++k) A.S.cell[k].T = num_layers[A.S.cell[k].dir];
                A.nrep = nrep; A.s_begin = 0; A.s_end = s_end; A.num_stacked = Ls;
                A.use_split = split ? 1 : 0;
                A.epoch = a->epoch; A.err = (int*)a->tail_err;
                const size_t tail_lds = step_lds_bytes<4>(H) + (size_t)2 * 3 * H * BJS * sizeof(float);
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tail_lds) != hipSuccess)
                    return DAGNN_EHIP(hipGetLastError());
                hipLaunchKernelGGL(bwd_tail_kernel, dim3((unsigned)(S.ncell * NS * nrep)), dim3(ST), tail_lds, st, plan, L, A);
                DAGNN_CHECK_LAUNCH();
                s_first = s_end;
            }
        }
    }
    return run_steps(st, s_first, part);   // join (fj's destructor): the caller's stream continues only when the shallow graphs are finished too
}

extern "C" int dagnn_readout_max_backward_batch(const dagnn_plan* pl, const dagnn_readout_bwd_job* jobs, int n, const float* grad_out,
                                                int ld_out, void* stream) {
    if (!pl || !pl->data || !jobs || !grad_out || n <= 0 || n > DAGNN_MAX_READOUT_JOBS) return DAGNN_EINVAL;
    RbJobs J;
    for (int q = 0; q < n; ++q) {
        if (!jobs[q].h || !jobs[q].grad_h || jobs[q].width <= 0 || (jobs[q].dir != 0 && jobs[q].dir != 1)) return DAGNN_EINVAL;
        for (int p = 0; p < q; ++p)   // two jobs adding into the same rows would race
            if (jobs[p].grad_h == jobs[q].grad_h) return DAGNN_EINVAL;
        J.j[q] = jobs[q];
    }
    if (pl->B == 0) return DAGNN_OK;
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipLaunchKernelGGL(readout_max_bwd_batch_kernel, dim3((unsigned)pl->B, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)pl->data, L, J, grad_out, ld_out);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_readout_max_backward(const dagnn_plan* pl, const float* h, int ld_h, int width, int dir,
                                          const float* grad_out, int ld_out, int col_off, float* grad_h, int ld_g,
                                          void* stream) {
    if (!pl || !pl->data || !h || !grad_out || !grad_h || width <= 0 || (dir != 0 && dir != 1)) return DAGNN_EINVAL;
    if (pl->B == 0) return DAGNN_OK;
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipLaunchKernelGGL(readout_max_bwd_kernel, dim3((unsigned)pl->B), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)pl->data, L, dir, h, ld_h, width, grad_out, ld_out, col_off, grad_h, ld_g);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
