// recurrence.hip - the layer-by-layer message-passing recurrence of one stacked GRU layer.
//
// Reference path replaced (per direction d, topological layer t, frontier node v):
//   AttnConv.forward/message + PyG propagate/softmax/scatter_add   ogbg-code/model/dagnn.py:362-373
//   hidden half of nn.GRUCell (W_hh a + b_hh, gates)                dagnn.py:181
//   state write h[d][i][layer] += h'                                dagnn.py:182
//   (dvae/dagnn.py:124-145 and dvae/dagnn_bn.py:123-136 are the same loop)
//
// Design (gfx950): graphs in a batch are independent, so ONE WORKGROUP WALKS ONE (graph,
// direction) through all of its topological layers with workgroup barriers only - no grid
// barrier, no kernel boundary per layer (the batch-level chain is ~1500 dependent micro-steps;
// a boundary costs ~1.5 us, a grid barrier ~4 us).  Work items are dispatched deepest-first.
// Per frontier chunk of <= 8 rows:
//   A  one wave per row: segment softmax over the row's in-edges (scores are one scalar per
//      source node, the query term cancels), then a coalesced float4 gather of the predecessors'
//      hidden rows weighted by alpha -> a[r,:] in LDS;
//   B  all waves: gh[r,:] = Wt^T a[r,:]  (Wt k-major [H,3H], float4 per lane = 4 gate columns,
//      K split over KSL lanes of the same wave and reduced by xor-shuffles; weights are read
//      once per chunk and register-blocked over the R rows);
//   C  one wave per row: GRU gates, h' store (row written exactly once), score = w_key . h'.
// fp32 throughout; VALU fma for the GEMV (M <= 8 rows: the fp32 MFMA has the same per-row rate).
#include "common.h"

namespace {

constexpr int RMAX = 8;      // rows per chunk (register blocking of phase B)
constexpr int MAX_WAVES = 12;  // 3 waves per SIMD: 168-VGPR budget for the register-blocked GEMV

struct RecArgs {
    const float* gi[2];
    const float* wt[2];
    const float* bhh[2];
    const float* wkey[2];
    const float* gain[2];
    const float* vid[2];
    float* h[2];
    float* score[2];
    int vid_mod, ld_h, H, R, dir_mask, kper, rmax, static_score;
    unsigned long long* dbg;  // optional [8] phase timing of work item 0 (wall_clock64 ticks, 100 MHz)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// padded index of element k of an aggregate row in LDS: 4 floats of pad between K slices so the
// KSL slices a wave reads concurrently (ds_read_b128) fall on different banks.
__device__ __forceinline__ int a_idx(int k, int kper) { return k + 4 * (k / kper); }

// ------------------------------------------------------------------------------------ phase B
template <int KSL, int R>
__device__ __forceinline__ void gemv_tiles(const float* __restrict__ wt, const float* a_s, float* gh_s, int H,
                                           int kper, int a_ld, int wave, int nwaves, int lane) {
    constexpr int CGW = 64 / KSL;
    const int H3 = 3 * H;
    const int CG = H3 >> 2;
    const int ntiles = (CG + CGW - 1) / CGW;
    const int ksl = lane / CGW, cgl = lane - ksl * CGW;
    const int kbeg = ksl * kper;
    const int kend = min(H, kbeg + kper);
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int cg = tile * CGW + cgl;
        const bool valid = cg < CG;
        float4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            const float* wp = wt + (int64_t)kbeg * H3 + 4 * cg;
            const float* ap = a_s + a_idx(kbeg, kper);
#pragma unroll 2
            for (int k = kbeg; k < kend; k += 4) {
                const float4 w0 = *reinterpret_cast<const float4*>(wp);
                const float4 w1 = *reinterpret_cast<const float4*>(wp + H3);
                const float4 w2 = *reinterpret_cast<const float4*>(wp + 2 * H3);
                const float4 w3 = *reinterpret_cast<const float4*>(wp + 3 * H3);
                wp += 4 * (int64_t)H3;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float4 a = *reinterpret_cast<const float4*>(ap + r * a_ld);
                    acc[r].x = fmaf(w0.x, a.x, acc[r].x); acc[r].y = fmaf(w0.y, a.x, acc[r].y);
                    acc[r].z = fmaf(w0.z, a.x, acc[r].z); acc[r].w = fmaf(w0.w, a.x, acc[r].w);
                    acc[r].x = fmaf(w1.x, a.y, acc[r].x); acc[r].y = fmaf(w1.y, a.y, acc[r].y);
                    acc[r].z = fmaf(w1.z, a.y, acc[r].z); acc[r].w = fmaf(w1.w, a.y, acc[r].w);
                    acc[r].x = fmaf(w2.x, a.z, acc[r].x); acc[r].y = fmaf(w2.y, a.z, acc[r].y);
                    acc[r].z = fmaf(w2.z, a.z, acc[r].z); acc[r].w = fmaf(w2.w, a.z, acc[r].w);
                    acc[r].x = fmaf(w3.x, a.w, acc[r].x); acc[r].y = fmaf(w3.y, a.w, acc[r].y);
                    acc[r].z = fmaf(w3.z, a.w, acc[r].z); acc[r].w = fmaf(w3.w, a.w, acc[r].w);
                }
                ap += 4;
            }
        }
        // reduce the KSL partial sums (lanes cgl, cgl+CGW, ...) - fixed order, deterministic
#pragma unroll
        for (int off = CGW; off < 64; off <<= 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r].x += __shfl_xor(acc[r].x, off, 64);
                acc[r].y += __shfl_xor(acc[r].y, off, 64);
                acc[r].z += __shfl_xor(acc[r].z, off, 64);
                acc[r].w += __shfl_xor(acc[r].w, off, 64);
            }
        }
        if (valid && ksl == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(gh_s + r * H3 + 4 * cg) = acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------ phase A
// One wave: a[:] = sum_e alpha_e * h[col_e, :],  alpha = softmax_e(score[col_e] + gain . eattr_e)
// with PyG's `exp(x - max) / (sum + 1e-16)`.
__device__ __forceinline__ void aggregate_row(const int32_t* __restrict__ col, const float* __restrict__ eattr,
                                              int e_beg, int e_end, const float* hbuf, int ld_h, const float* score,
                                              const float* __restrict__ gain, int R, float* a_row, int H, int kper,
                                              int lane) {
    const int H4 = H >> 2;
    float mx = -INFINITY;
    for (int e = e_beg + lane; e < e_end; e += 64) {
        float lg = score[col[e]];
        for (int r = 0; r < R; ++r) lg = fmaf(gain[r], eattr[(int64_t)e * R + r], lg);
        mx = fmaxf(mx, lg);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int e = e_beg + lane; e < e_end; e += 64) {
        float lg = score[col[e]];
        for (int r = 0; r < R; ++r) lg = fmaf(gain[r], eattr[(int64_t)e * R + r], lg);
        sum += expf(lg - mx);
    }
    sum = wave_sum(sum);
    const float denom = sum + 1e-16f;

    for (int c0 = 0; c0 < H4; c0 += 128) {  // 2 float4 accumulators per lane per pass
        float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
        const int ca = c0 + lane, cb = c0 + 64 + lane;
        for (int base = e_beg; base < e_end; base += 64) {
            const int e = base + lane;
            float my_alpha = 0.f;
            int my_col = 0;
            if (e < e_end) {
                my_col = col[e];
                float lg = score[my_col];
                for (int r = 0; r < R; ++r) lg = fmaf(gain[r], eattr[(int64_t)e * R + r], lg);
                my_alpha = expf(lg - mx) / denom;
            }
            const int cnt = min(64, e_end - base);
            for (int i = 0; i < cnt; ++i) {
                const float al = __shfl(my_alpha, i, 64);
                const int cj = __shfl(my_col, i, 64);
                const float4* hr = reinterpret_cast<const float4*>(hbuf + (int64_t)cj * ld_h);
                if (ca < H4) {
                    const float4 v = hr[ca];
                    acc0.x = fmaf(al, v.x, acc0.x); acc0.y = fmaf(al, v.y, acc0.y);
                    acc0.z = fmaf(al, v.z, acc0.z); acc0.w = fmaf(al, v.w, acc0.w);
                }
                if (cb < H4) {
                    const float4 v = hr[cb];
                    acc1.x = fmaf(al, v.x, acc1.x); acc1.y = fmaf(al, v.y, acc1.y);
                    acc1.z = fmaf(al, v.z, acc1.z); acc1.w = fmaf(al, v.w, acc1.w);
                }
            }
        }
        if (ca < H4) *reinterpret_cast<float4*>(a_row + a_idx(4 * ca, kper)) = acc0;
        if (cb < H4) *reinterpret_cast<float4*>(a_row + a_idx(4 * cb, kper)) = acc1;
    }
}

// ------------------------------------------------------------------------------------ phase C
// One wave: GRU gates for node v (torch gate order r, z, n; h' = n + z * (a - n)), state write,
// attention score of the new state.
__device__ __forceinline__ void gates_row(int v, const float* __restrict__ gi, const float* gh_row,
                                          const float* __restrict__ bhh, const float* a_row, bool has_pred,
                                          float* hbuf, int ld_h, float* score, const float* __restrict__ wkey,
                                          const float* __restrict__ vid, int vid_mod, int H, int kper, int lane) {
    const float* g = gi + (int64_t)v * 3 * H;
    float sp = 0.f;
    for (int j = lane; j < H; j += 64) {
        float hr = bhh[j], hz = bhh[H + j], hn = bhh[2 * H + j], a = 0.f;
        if (has_pred) {
            hr += gh_row[j]; hz += gh_row[H + j]; hn += gh_row[2 * H + j];
            a = a_row[a_idx(j, kper)];
        }
        const float r = sigmoidf_(g[j] + hr);
        const float z = sigmoidf_(g[H + j] + hz);
        const float n = tanhf(fmaf(r, hn, g[2 * H + j]));
        const float hv = fmaf(z, a - n, n);
        hbuf[(int64_t)v * ld_h + j] = hv;
        if (wkey) sp = fmaf(wkey[j], hv, sp);
    }
    if (wkey) {  // null: the scores are static (keys from the inputs) and were filled in by the caller
        sp = wave_sum(sp);
        if (lane == 0) score[v] = vid ? sp + vid[v % vid_mod] : sp;
    }
}

template <int KSL>
__global__ void __launch_bounds__(MAX_WAVES * 64) recurrence_kernel(const int32_t* __restrict__ plan, PlanLayout L,
                                                                     RecArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int item = plan[L.items + blockIdx.x];
    const int g = item >> 1, d = item & 1;
    if (!((P.dir_mask >> d) & 1)) return;
    const int n0 = plan[L.node_ptr + g];
    const int depth = plan[L.depth[d] + g];
    const int32_t* ls = plan + L.lstart[d] + n0 + g;
    const int32_t* rp = plan + L.rowptr[d] + n0 + g;
    const int32_t* order = plan + L.order[d];
    const int32_t* col = plan + L.col[d];
    const float* eattr = reinterpret_cast<const float*>(plan + L.eattr[d]);

    const int H = P.H, H3 = 3 * H, kper = P.kper, rmax = P.rmax;
    const int a_ld = H + 4 * KSL;
    float* a_s = smem;                 // [rmax][a_ld]
    float* gh_s = smem + rmax * a_ld;  // [rmax][3H]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const float* __restrict__ gi = P.gi[d];
    const float* __restrict__ wt = P.wt[d];
    const float* __restrict__ bhh = P.bhh[d];
    const float* __restrict__ wkey = P.static_score ? nullptr : P.wkey[d];
    const float* __restrict__ gain = P.gain[d];
    const float* __restrict__ vid = P.vid[d];
    const int R = gain ? P.R : 0;
    float* hbuf = P.h[d];      // read (predecessor rows) and written (frontier rows): no restrict
    float* score = P.score[d];

    const bool prof = P.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    unsigned long long tA = 0, tB = 0, tC = 0, nchunk = 0, t_start = 0, tq = 0;
    if (prof) t_start = wall_clock64();
    for (int t = 0; t < depth; ++t) {
        const int p0 = ls[t], p1 = ls[t + 1];
        if (t == 0) {
            // sources: no predecessors, a = 0 and gh = b_hh (ps_h=None at dagnn.py:172-173)
            for (int p = p0 + wave; p < p1; p += nwaves)
                gates_row(order[p], gi, nullptr, bhh, nullptr, false, hbuf, P.ld_h, score, wkey, vid, P.vid_mod, H,
                          kper, lane);
            __syncthreads();
            continue;
        }
        for (int c0 = p0; c0 < p1; c0 += rmax) {
            const int nr = min(rmax, p1 - c0);
            if (prof) tq = wall_clock64();
            // ---- A: aggregate predecessors (one wave per row)
            for (int r = wave; r < nr; r += nwaves) {
                const int p = c0 + r;
                aggregate_row(col, eattr, rp[p - n0], rp[p - n0 + 1], hbuf, P.ld_h, score, gain, R,
                              a_s + r * a_ld, H, kper, lane);
            }
            __syncthreads();
            if (prof) { unsigned long long n = wall_clock64(); tA += n - tq; tq = n; }
            // ---- B: gh = W_hh a for the chunk's rows
            if (nr == 1) gemv_tiles<KSL, 1>(wt, a_s, gh_s, H, kper, a_ld, wave, nwaves, lane);
            else if (nr == 2) gemv_tiles<KSL, 2>(wt, a_s, gh_s, H, kper, a_ld, wave, nwaves, lane);
            else if (nr <= 4) gemv_tiles<KSL, 4>(wt, a_s, gh_s, H, kper, a_ld, wave, nwaves, lane);
            else gemv_tiles<KSL, 8>(wt, a_s, gh_s, H, kper, a_ld, wave, nwaves, lane);
            __syncthreads();
            if (prof) { unsigned long long n = wall_clock64(); tB += n - tq; tq = n; }
            // ---- C: gates + state write + score (one wave per row)
            for (int r = wave; r < nr; r += nwaves)
                gates_row(order[c0 + r], gi, gh_s + r * H3, bhh, a_s + r * a_ld, true, hbuf, P.ld_h, score, wkey,
                          vid, P.vid_mod, H, kper, lane);
            __syncthreads();
            if (prof) { unsigned long long n = wall_clock64(); tC += n - tq; ++nchunk; }
        }
    }
    if (prof) {
        P.dbg[0] = tA; P.dbg[1] = tB; P.dbg[2] = tC; P.dbg[3] = nchunk;
        P.dbg[4] = wall_clock64() - t_start; P.dbg[5] = (unsigned long long)depth;
    }
}

}  // namespace

extern "C" int dagnn_recurrence_layer(const dagnn_plan* pl, const dagnn_layer_args* a, int dir_mask, int H,
                                      void* stream) {
    if (!pl || !pl->data || !a || H <= 0 || (H & 3) || H > 3072 || !(dir_mask & 3)) return DAGNN_EINVAL;
    if (a->ld_h < H || (a->ld_h & 3)) return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    RecArgs P;
    for (int d = 0; d < 2; ++d) {
        P.gi[d] = a->gi[d]; P.wt[d] = a->w_hh_t[d]; P.bhh[d] = a->b_hh[d]; P.wkey[d] = a->w_key[d];
        P.gain[d] = pl->num_edge_feats > 0 ? a->edge_gain[d] : nullptr;
        P.vid[d] = a->vid_mod > 0 ? a->vid_bias[d] : nullptr;
        P.h[d] = a->h[d]; P.score[d] = a->score[d];
        if ((dir_mask >> d) & 1)
            if (!P.gi[d] || !P.wt[d] || !P.bhh[d] || (!P.wkey[d] && !a->static_score) || !P.h[d] || !P.score[d])
                return DAGNN_EINVAL;
    }
    P.dbg = (unsigned long long*)a->debug_timing;
    P.static_score = a->static_score ? 1 : 0;
    P.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
    P.ld_h = a->ld_h; P.H = H; P.R = pl->num_edge_feats; P.dir_mask = dir_mask & 3;
    // K split: the largest KSL whose tile count fits the 12 waves of a workgroup
    const int CG = 3 * H / 4;
    int ksl = 16;
    while (ksl > 1 && (CG + (64 / ksl) - 1) / (64 / ksl) > MAX_WAVES) ksl >>= 1;
    const int ntiles = (CG + (64 / ksl) - 1) / (64 / ksl);
    int nwaves = ntiles < 8 ? 8 : (ntiles > MAX_WAVES ? MAX_WAVES : ntiles);
    P.kper = (((H + ksl - 1) / ksl) + 3) & ~3;
    P.rmax = H <= 384 ? RMAX : (H <= 768 ? 4 : (H <= 1536 ? 2 : 1));  // keeps LDS <= 52 KB
    const size_t lds = (size_t)P.rmax * ((H + 4 * ksl) + 3 * H) * sizeof(float);
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    dim3 grid((unsigned)(2 * pl->B)), block((unsigned)(nwaves * 64));
    const int32_t* plan = (const int32_t*)pl->data;
    hipStream_t s = (hipStream_t)stream;
    switch (ksl) {
        case 16: hipLaunchKernelGGL(recurrence_kernel<16>, grid, block, lds, s, plan, L, P); break;
        case 8: hipLaunchKernelGGL(recurrence_kernel<8>, grid, block, lds, s, plan, L, P); break;
        case 4: hipLaunchKernelGGL(recurrence_kernel<4>, grid, block, lds, s, plan, L, P); break;
        case 2: hipLaunchKernelGGL(recurrence_kernel<2>, grid, block, lds, s, plan, L, P); break;
        default: hipLaunchKernelGGL(recurrence_kernel<1>, grid, block, lds, s, plan, L, P); break;
    }
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
