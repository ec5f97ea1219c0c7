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
// Work split of a launch: workgroup = (cell, block of RB rows, slice of 64 hidden units).  Wave w pulls
// row w (+4, ...), all threads do the gate algebra of the full rows (cheap, recomputed per slice), then
// wave w multiplies k-range w of the [3H x 64] weight slice - lanes own output units, so every weight
// load is one contiguous 256 B read of the torch-layout matrix (no packed copy), the gradients are LDS
// broadcasts - and the four partial sums meet in LDS in wave order.
#include "common.h"

#define DAGNN_BWD_MAX_CELLS 16

namespace {

constexpr int BT = 256;   // threads per workgroup
constexpr int BJS = 64;   // hidden units per slice
constexpr int BPU = 16;   // hidden units per stored score part (frontier.hip PU)

struct BCell {
    const float* whh;    // [3H,H] torch layout (gate blocks r,z,n)
    const float* wih;    // [3H,H] or null (stacked layer 0)
    const float* wkey;   // [H]
    const float* gain;   // [R] or null
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
    int dir, row_base, row_end;
};

struct BArgs {
    BCell cell[DAGNN_BWD_MAX_CELLS];
    int blk_start[DAGNN_BWD_MAX_CELLS + 1];
    int ncell, H, ld_h, R;
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
        float s = bscore(C.h + (int64_t)col[e] * ld_h + H, nparts);
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

// ---- one reverse lock-step launch -----------------------------------------------------------------
// LDS (floats): g_s[RB][H] | dgh_t[3H][RB] | dgi_t[3H][RB] | red[2][4][RB][64] | node[RB] (ints)
template <int RB>
__global__ void __launch_bounds__(BT) bwd_step_kernel(const int32_t* __restrict__ plan, PlanLayout L, BArgs S) {
    extern __shared__ float lds[];
    const int H = S.H, H3 = 3 * H, H4 = H >> 2, ld_h = S.ld_h;
    float* g_s = lds;
    float* dgh_t = g_s + RB * H;
    float* dgi_t = dgh_t + H3 * RB;
    float* red = dgi_t + H3 * RB;
    int* node_s = reinterpret_cast<int*>(red + 2 * 4 * RB * BJS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NS = H / BJS;
    const int slice = blockIdx.x % NS, gblk = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gblk >= S.blk_start[c + 1]) ++c;
    const BCell& C = S.cell[c];
    const int d = C.dir, od = 1 - d, R = C.mrel ? S.R : 0;
    const int row0 = C.row_base + (gblk - S.blk_start[c]) * RB;
    const int nrows = min(RB, C.row_end - row0);

    // ---- 1. pull: G_v = Gext_v + sum over successors, one wave per row
    for (int r = wave; r < nrows; r += 4) {
        const int v = plan[L.rowrec[d] + 16 * (int64_t)(row0 + r)];
        if (lane == 0) node_s[r] = v;
        const int4 srec = reinterpret_cast<const int4*>(plan + L.rowrec[od])[4 * (int64_t)plan[L.pos[od] + v]];
        const int eb = srec.y, ee = srec.z;
        const int32_t* col = plan + L.col[od];
        const int32_t* eidx = plan + L.eidx[od];
        const float* eattr = reinterpret_cast<const float*>(plan + L.eattr[od]);
        const float* hv = C.h + (int64_t)v * ld_h;
        float* grow = g_s + r * H;
        for (int cc = lane; cc < H4; cc += 64)
            reinterpret_cast<float4*>(grow)[cc] = reinterpret_cast<const float4*>(C.gext + (int64_t)v * H)[cc];
        float sig = 0.f, m0 = 0.f, m1 = 0.f;
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
        if (slice == 0 && lane == 0) {
            C.sig[v] = sig;
            if (R >= 1) C.mrel[(int64_t)v * R] = m0;
            if (R >= 2) C.mrel[(int64_t)v * R + 1] = m1;
        }
    }
    __syncthreads();

    // ---- 2. GRU backward of the full rows (gates recomputed); operands of the products go to LDS k-major
    for (int idx = tid; idx < RB * H; idx += BT) {
        const int r = idx / H, u = idx - r * H;
        float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, zg = 0.f;
        if (r < nrows) {
            const int v = node_s[r];
            const float* gi = C.gi + (int64_t)v * H3;
            const float* gh = C.gh + (int64_t)v * H3;
            const float ghn = gh[2 * H + u];
            const float rr = bsigm(gi[u] + gh[u]);
            const float zz = bsigm(gi[H + u] + gh[H + u]);
            const float nn = tanhf(gi[2 * H + u] + rr * ghn);
            const float G = g_s[idx];
            const float av = C.a[(int64_t)v * H + u];
            dn = G * (1.0f - zz) * (1.0f - nn * nn);          // d pre-activation of n
            dz = G * (av - nn) * zz * (1.0f - zz);            // d pre-activation of z
            dr = dn * ghn * rr * (1.0f - rr);                 // d pre-activation of r
            dnr = dn * rr;                                    // hidden-side n input sits behind r
            zg = G * zz;                                      // direct path h' = n + z (a - n)
            if (u / BJS == slice) {
                float* og = C.dgi + (int64_t)v * H3;
                float* oh = C.dgh + (int64_t)v * H3;
                og[u] = dr; og[H + u] = dz; og[2 * H + u] = dn;
                oh[u] = dr; oh[H + u] = dz; oh[2 * H + u] = dnr;
            }
        }
        g_s[idx] = zg;
        dgh_t[u * RB + r] = dr; dgh_t[(H + u) * RB + r] = dz; dgh_t[(2 * H + u) * RB + r] = dnr;
        dgi_t[u * RB + r] = dr; dgi_t[(H + u) * RB + r] = dz; dgi_t[(2 * H + u) * RB + r] = dn;
    }
    __syncthreads();

    // ---- 3. da[slice] = W_hh^T dgh, du[slice] = W_ih^T dgi: wave w owns k in [w*3H/4, (w+1)*3H/4)
    const int unit = slice * BJS + lane;
    const int KQ = H3 / 4, k0 = wave * KQ;
    float acc[RB], acc2[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    {
        const float* wp = C.whh + (int64_t)k0 * H + unit;
#pragma unroll 8
        for (int k = 0; k < KQ; ++k) {
            const float wv = wp[(int64_t)k * H];
            const float* dp = dgh_t + (k0 + k) * RB;
#pragma unroll
            for (int r4 = 0; r4 < RB; r4 += 4) {
                const float4 d4 = *reinterpret_cast<const float4*>(dp + r4);
                acc[r4] = fmaf(wv, d4.x, acc[r4]); acc[r4 + 1] = fmaf(wv, d4.y, acc[r4 + 1]);
                acc[r4 + 2] = fmaf(wv, d4.z, acc[r4 + 2]); acc[r4 + 3] = fmaf(wv, d4.w, acc[r4 + 3]);
            }
        }
    }
    if (C.wih) {
        const float* wp = C.wih + (int64_t)k0 * H + unit;
#pragma unroll 8
        for (int k = 0; k < KQ; ++k) {
            const float wv = wp[(int64_t)k * H];
            const float* dp = dgi_t + (k0 + k) * RB;
#pragma unroll
            for (int r4 = 0; r4 < RB; r4 += 4) {
                const float4 d4 = *reinterpret_cast<const float4*>(dp + r4);
                acc2[r4] = fmaf(wv, d4.x, acc2[r4]); acc2[r4 + 1] = fmaf(wv, d4.y, acc2[r4 + 1]);
                acc2[r4 + 2] = fmaf(wv, d4.z, acc2[r4 + 2]); acc2[r4 + 3] = fmaf(wv, d4.w, acc2[r4 + 3]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        red[(wave * RB + r) * BJS + lane] = acc[r];
        red[((4 + wave) * RB + r) * BJS + lane] = acc2[r];
    }
    __syncthreads();

    // ---- 4. partial sums in wave order, stores
    for (int idx = tid; idx < nrows * BJS; idx += BT) {
        const int r = idx / BJS, l = idx - r * BJS, u = slice * BJS + l;
        const int v = node_s[r];
        float s = red[(0 * RB + r) * BJS + l];
        s += red[(1 * RB + r) * BJS + l]; s += red[(2 * RB + r) * BJS + l]; s += red[(3 * RB + r) * BJS + l];
        C.da[(int64_t)v * H + u] = g_s[r * H + u] + s;
        if (C.gext_lo) {
            float t = red[((4 + 0) * RB + r) * BJS + l];
            t += red[((4 + 1) * RB + r) * BJS + l]; t += red[((4 + 2) * RB + r) * BJS + l];
            t += red[((4 + 3) * RB + r) * BJS + l];
            C.gext_lo[(int64_t)v * H + u] += t;   // this workgroup is the only writer of these 64 floats
        }
    }
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

template <int RB>
size_t step_lds_bytes(int H) { return (size_t)(RB * H + 2 * 3 * H * RB + 2 * 4 * RB * BJS + RB) * sizeof(float); }

void fill_cells(BArgs& S, const dagnn_backward_args* a, const int* dirs, int ndir) {
    S.ncell = 0;
    for (int q = 0; q < ndir; ++q)
        for (int i = 0; i < a->num_stacked; ++i) {
            const dagnn_backward_cell& c = a->cell[dirs[q]][i];
            BCell& K = S.cell[S.ncell++];
            K.whh = (const float*)c.w_hh; K.wih = i > 0 ? (const float*)c.w_ih : nullptr;
            K.wkey = (const float*)c.w_key; K.gain = (const float*)c.edge_gain;
            K.h = (const float*)c.h; K.a = (const float*)c.a; K.a_w = (float*)c.a; K.alpha = (float*)c.alpha;
            K.gi = (const float*)c.gi; K.gh = (const float*)c.gh; K.gext = (const float*)c.g_ext;
            K.gext_lo = i > 0 ? (float*)a->cell[dirs[q]][i - 1].g_ext : nullptr;
            K.da = (float*)c.da; K.dgi = (float*)c.dgi; K.dgh = (float*)c.dgh; K.sig = (float*)c.sigma;
            K.mrel = (float*)c.edge_feat_grad;
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
    S.H = H; S.ld_h = a->ld_h; S.R = pl->num_edge_feats;
    fill_cells(S, a, dirs, ndir);
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
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
    S.H = H; S.ld_h = a->ld_h; S.R = pl->num_edge_feats;
    fill_cells(S, a, dirs, ndir);
    const int NS = H / BJS;
    const int rb_fat = H <= 512 ? 8 : 4;
    // dynamic LDS beyond 64 KB must be granted per kernel; idempotent, so no library-global state is kept
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_step_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)step_lds_bytes<4>(H)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_step_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)step_lds_bytes<8>(H <= 512 ? H : 512)) != hipSuccess)
        return DAGNN_EHIP(hipGetLastError());
    const int nsteps = Tmax + Ls - 1;
    for (int s = 0; s < nsteps; ++s) {
        // stacked layer i handles layer t = T_d - 1 - (s - (Ls-1-i)): the top layer leads, every lower one is a launch behind
        int total4 = 0, total_fat = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const int rb = pass == 0 ? 4 : rb_fat;
            int k = 0, tot = 0;
            for (int q = 0; q < ndir; ++q)
                for (int i = 0; i < Ls; ++i, ++k) {
                    const int d = dirs[q];
                    const int t = num_layers[d] - 1 - (s - (Ls - 1 - i));
                    const bool on = t >= 0 && t < num_layers[d];
                    S.cell[k].row_base = on ? layer_ptr[d][t] : 0;
                    S.cell[k].row_end = on ? layer_ptr[d][t + 1] : 0;
                    S.blk_start[k] = tot;
                    tot += (S.cell[k].row_end - S.cell[k].row_base + rb - 1) / rb;
                }
            S.blk_start[k] = tot;
            if (pass == 0) {
                total4 = tot;
                if (rb_fat == 4 || total4 * NS <= 2 * a->num_cus) break;   // thin launch: keep 4-row blocks
            } else {
                total_fat = tot;
            }
        }
        const int blocks = total_fat ? total_fat : total4;
        if (blocks == 0) continue;
        if (total_fat && rb_fat == 8)
            hipLaunchKernelGGL(bwd_step_kernel<8>, dim3((unsigned)(blocks * NS)), dim3(BT), step_lds_bytes<8>(H), st, plan, L, S);
        else
            hipLaunchKernelGGL(bwd_step_kernel<4>, dim3((unsigned)(blocks * NS)), dim3(BT), step_lds_bytes<4>(H), st, plan, L, S);
        DAGNN_CHECK_LAUNCH();
    }
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
