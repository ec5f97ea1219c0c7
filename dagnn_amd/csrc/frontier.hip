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
// gfx950 design.  The dependent chain is what bounds this path (a launch boundary costs ~1.5 us,
// an HBM/L2 round trip ~0.4-0.9 us), so each launch keeps its own chain to three dependent loads:
//   rowrec[slot] -> {col, edge feats, score parts} -> predecessor rows.
// The GRU weights never sit on that chain: a workgroup owns a SLICE of 32 hidden units
// (3 gates x 32 = 96 weight columns, K = H rows) of one cell for a block of <= 8 frontier rows,
// and issues the loads of its whole slice - pre-packed in exactly the order its lanes consume it,
// so every load instruction is one contiguous 1 KiB - BEFORE it starts the aggregate; they land in
// registers while the chain is in flight.  H/32 slice-workgroups share a row block: the weight
// matrix is read once per row block by the whole GPU instead of once per graph-step by one CU
// (the per-graph persistent kernel in recurrence.hip streams 786 KB per step from L2, 6 us).
//   A  one wave per row: segment softmax over in-edges (score = sum of the H/32 per-slice partial
//      dots the producers stored, + edge-feature gain), alpha-weighted float4 gather of the
//      predecessors' rows -> LDS; for stacked layers > 0 also the node's own lower-layer row;
//   B  6 waves x (16 K-lanes x 4 column groups): fp32 FMA GEMV on the slice for hidden- and
//      input-side, K reduced by xor-shuffles;
//   C  256 threads = 8 rows x 32 units: gates, h' store (128 B per row), score partial store.
// fp32 VALU FMA: at <= 8 rows per weight pass the fp32 MFMA has the same per-row rate.
#include "common.h"

#define DAGNN_MAX_CELLS 16

namespace {

constexpr int FT = 384;   // threads per workgroup (6 waves)
constexpr int JS = 32;    // hidden units per slice
constexpr int RB = 8;     // frontier rows per block
constexpr int KCH = 16;   // k values per lane per register chunk (K chunk of 256)

struct Cell {
    const float4* whh;   // packed hidden-side slices
    const float4* wih;   // packed input-side slices, or null (stacked layer 0: gi0 instead)
    const float* bhh;    // [3H]
    const float* bih;    // [3H] (only with wih)
    const float* wkey;   // [H]
    const float* gain;   // [R] or null
    const float* vid;    // [vid_mod] or null
    const float* gi0;    // [N,3H] precomputed input side (stacked layer 0) or null
    const float* h_in;   // [N,ld_h] lower stacked layer (with wih) or null
    float* h_out;        // [N,ld_h]
    float* spart;        // [N,NS] per-slice score partials of h_out
    int dir;             // direction (selects the plan arrays)
    int row_base;        // first rowrec slot of the layer processed in this launch
    int row_end;         // one past the last
    int has_pred;        // layer > 0
};

struct StepArgs {
    Cell cell[DAGNN_MAX_CELLS];
    int blk_start[DAGNN_MAX_CELLS + 1];  // row-block prefix sums over the active cells
    int ncell, H, ld_h, NS, R, vid_mod;
};

// LDS index of element k of an 8-row operand: one float of pad per K-lane segment so the 16
// segments a wave reads concurrently start on different banks.
__device__ __forceinline__ int apad(int k, int kpt) { return k + 4 * (k / kpt); }

// One wave: aggregate the predecessors of one frontier row into a_row (LDS, padded layout).
__device__ __forceinline__ void aggregate(const Cell& C, const int32_t* __restrict__ col,
                                          const float* __restrict__ eattr, int eb, int ee, int H, int ld_h, int NS,
                                          int R, int vid_mod, int kpt, float* a_row, int lane) {
    const int H4 = H >> 2;
    const float* hsrc = C.h_out;  // predecessors' states of THIS stacked layer (earlier launches)
    const int deg = ee - eb;
    if (deg == 1) {  // softmax over one edge: alpha = exp(0) / (exp(0) + 1e-16) == 1.0f exactly
        const float4* hr = reinterpret_cast<const float4*>(hsrc + (int64_t)col[eb] * ld_h);
        for (int c = lane; c < H4; c += 64) *reinterpret_cast<float4*>(a_row + apad(4 * c, kpt)) = hr[c];
        return;
    }
    auto logit = [&](int e, int cj) {
        const float* sp = C.spart + (int64_t)cj * NS;
        float s = 0.f;
        for (int q = 0; q < NS; ++q) s += sp[q];  // fixed order: deterministic
        if (C.vid) s += C.vid[cj % vid_mod];
        for (int r = 0; r < R; ++r) s = fmaf(C.gain[r], eattr[(int64_t)e * R + r], s);
        return s;
    };
    float mx = -INFINITY, sum = 0.f, my_lg = 0.f;
    int my_col = 0;
    if (deg <= 64) {
        if (lane < deg) { my_col = col[eb + lane]; my_lg = logit(eb + lane, my_col); mx = my_lg; }
        mx = wave_max(mx);
        const float ex = lane < deg ? expf(my_lg - mx) : 0.f;
        sum = wave_sum(ex);
        const float denom = sum + 1e-16f;
        const float my_alpha = ex / denom;
        for (int c0 = 0; c0 < H4; c0 += 128) {
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            const int ca = c0 + lane, cb = c0 + 64 + lane;
            for (int i = 0; i < deg; ++i) {
                const float al = __shfl(my_alpha, i, 64);
                const int cj = __shfl(my_col, i, 64);
                const float4* hr = reinterpret_cast<const float4*>(hsrc + (int64_t)cj * ld_h);
                if (ca < H4) { const float4 v = hr[ca]; acc0.x = fmaf(al, v.x, acc0.x); acc0.y = fmaf(al, v.y, acc0.y);
                               acc0.z = fmaf(al, v.z, acc0.z); acc0.w = fmaf(al, v.w, acc0.w); }
                if (cb < H4) { const float4 v = hr[cb]; acc1.x = fmaf(al, v.x, acc1.x); acc1.y = fmaf(al, v.y, acc1.y);
                               acc1.z = fmaf(al, v.z, acc1.z); acc1.w = fmaf(al, v.w, acc1.w); }
            }
            if (ca < H4) *reinterpret_cast<float4*>(a_row + apad(4 * ca, kpt)) = acc0;
            if (cb < H4) *reinterpret_cast<float4*>(a_row + apad(4 * cb, kpt)) = acc1;
        }
        return;
    }
    // heavy rows (fan-in > 64): three passes over the edge list
    for (int e = eb + lane; e < ee; e += 64) mx = fmaxf(mx, logit(e, col[e]));
    mx = wave_max(mx);
    for (int e = eb + lane; e < ee; e += 64) sum += expf(logit(e, col[e]) - mx);
    sum = wave_sum(sum);
    const float denom = sum + 1e-16f;
    for (int c0 = 0; c0 < H4; c0 += 128) {
        float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
        const int ca = c0 + lane, cb = c0 + 64 + lane;
        for (int base = eb; base < ee; base += 64) {
            const int e = base + lane;
            float my_alpha = 0.f;
            if (e < ee) { my_col = col[e]; my_alpha = expf(logit(e, my_col) - mx) / denom; }
            const int cnt = min(64, ee - base);
            for (int i = 0; i < cnt; ++i) {
                const float al = __shfl(my_alpha, i, 64);
                const int cj = __shfl(my_col, i, 64);
                const float4* hr = reinterpret_cast<const float4*>(hsrc + (int64_t)cj * ld_h);
                if (ca < H4) { const float4 v = hr[ca]; acc0.x = fmaf(al, v.x, acc0.x); acc0.y = fmaf(al, v.y, acc0.y);
                               acc0.z = fmaf(al, v.z, acc0.z); acc0.w = fmaf(al, v.w, acc0.w); }
                if (cb < H4) { const float4 v = hr[cb]; acc1.x = fmaf(al, v.x, acc1.x); acc1.y = fmaf(al, v.y, acc1.y);
                               acc1.z = fmaf(al, v.z, acc1.z); acc1.w = fmaf(al, v.w, acc1.w); }
            }
        }
        if (ca < H4) *reinterpret_cast<float4*>(a_row + apad(4 * ca, kpt)) = acc0;
        if (cb < H4) *reinterpret_cast<float4*>(a_row + apad(4 * cb, kpt)) = acc1;
    }
}

__device__ __forceinline__ void fma_rows(float4 (&acc)[RB], const float4 (&w)[KCH], const float* op, int op_ld,
                                         int koff, int n) {
#pragma unroll
    for (int q = 0; q < KCH / 4; ++q) {
        if (4 * q < n) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float4 a = *reinterpret_cast<const float4*>(op + r * op_ld + koff + 4 * q);
                const float4 w0 = w[4 * q], w1 = w[4 * q + 1], w2 = w[4 * q + 2], w3 = w[4 * q + 3];
                acc[r].x = fmaf(w0.x, a.x, acc[r].x); acc[r].y = fmaf(w0.y, a.x, acc[r].y);
                acc[r].z = fmaf(w0.z, a.x, acc[r].z); acc[r].w = fmaf(w0.w, a.x, acc[r].w);
                acc[r].x = fmaf(w1.x, a.y, acc[r].x); acc[r].y = fmaf(w1.y, a.y, acc[r].y);
                acc[r].z = fmaf(w1.z, a.y, acc[r].z); acc[r].w = fmaf(w1.w, a.y, acc[r].w);
                acc[r].x = fmaf(w2.x, a.z, acc[r].x); acc[r].y = fmaf(w2.y, a.z, acc[r].y);
                acc[r].z = fmaf(w2.z, a.z, acc[r].z); acc[r].w = fmaf(w2.w, a.z, acc[r].w);
                acc[r].x = fmaf(w3.x, a.w, acc[r].x); acc[r].y = fmaf(w3.y, a.w, acc[r].y);
                acc[r].z = fmaf(w3.z, a.w, acc[r].z); acc[r].w = fmaf(w3.w, a.w, acc[r].w);
            }
        }
    }
}

__device__ __forceinline__ void reduce_k_lanes(float4 (&acc)[RB]) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            acc[r].x += __shfl_xor(acc[r].x, off, 64); acc[r].y += __shfl_xor(acc[r].y, off, 64);
            acc[r].z += __shfl_xor(acc[r].z, off, 64); acc[r].w += __shfl_xor(acc[r].w, off, 64);
        }
    }
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(FT) frontier_step_kernel(const int32_t* __restrict__ plan, PlanLayout L, StepArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NS = S.NS, H = S.H, ld_h = S.ld_h;
    const int sl = blockIdx.x % NS;
    const int gb = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gb >= S.blk_start[c + 1]) ++c;
    const Cell& C = S.cell[c];
    const int slot0 = C.row_base + (gb - S.blk_start[c]) * RB;
    const int nr = min(RB, C.row_end - slot0);
    const int d = C.dir;
    const bool has_in = C.wih != nullptr;

    const int kpt = H >> 4;                 // k values per K-lane (H % 64 == 0)
    const int op_ld = H + 4 * 16;           // padded operand row
    float* a_s = smem;                      // [RB][op_ld]   aggregates
    float* u_s = a_s + RB * op_ld;          // [RB][op_ld]   own lower-layer rows (has_in)
    float* gh_s = u_s + RB * op_ld;         // [RB][96]      hidden-side pre-activations of the slice
    float* gi_s = gh_s + RB * 96;           // [RB][96]      input-side
    int4* rec_s = reinterpret_cast<int4*>(gi_s + RB * 96);  // [RB]

    // ---- weights of this slice: issued first, consumed after the aggregate (phase B)
    const int ksl = lane >> 2;              // K-lane 0..15
    const int nchunk = (kpt + KCH - 1) / KCH;
    const int64_t wstride = (int64_t)FT;    // float4 per (chunk, kk) plane: 6 waves x 64 lanes
    const float4* whh = C.whh + (int64_t)sl * kpt * wstride + tid;
    const float4* wih = has_in ? C.wih + (int64_t)sl * kpt * wstride + tid : nullptr;
    float4 wh[KCH], wi[KCH];
    const int n0k = min(KCH, kpt);
#pragma unroll
    for (int kk = 0; kk < KCH; ++kk) {
        wh[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
        wi[kk] = wh[kk];
        if (kk < n0k) {
            wh[kk] = whh[kk * wstride];
            if (has_in) wi[kk] = wih[kk * wstride];
        }
    }

    // ---- phase A: row records, own lower-layer row, aggregate
    if (tid < RB) rec_s[tid] = tid < nr ? *reinterpret_cast<const int4*>(plan + L.rowrec[d] + 4 * (int64_t)(slot0 + tid))
                                        : make_int4(0, 0, 0, 0);
    const int32_t* col = plan + L.col[d];
    const float* eattr = reinterpret_cast<const float*>(plan + L.eattr[d]);
    const int R = C.gain ? S.R : 0;
    const int H4 = H >> 2;
    for (int r = wave; r < RB; r += FT / 64) {
        float* a_row = a_s + r * op_ld;
        float* u_row = u_s + r * op_ld;
        if (r < nr) {
            const int4 rec = *reinterpret_cast<const int4*>(plan + L.rowrec[d] + 4 * (int64_t)(slot0 + r));
            if (has_in) {
                const float4* ur = reinterpret_cast<const float4*>(C.h_in + (int64_t)rec.x * ld_h);
                for (int cc = lane; cc < H4; cc += 64) *reinterpret_cast<float4*>(u_row + apad(4 * cc, kpt)) = ur[cc];
            }
            if (C.has_pred && rec.z > rec.y) {
                aggregate(C, col, eattr, rec.y, rec.z, H, ld_h, NS, R, S.vid_mod, kpt, a_row, lane);
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

    // ---- phase B: slice GEMV, K over the 16 K-lanes of each wave
    float4 acc_h[RB], acc_i[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) { acc_h[r] = make_float4(0.f, 0.f, 0.f, 0.f); acc_i[r] = acc_h[r]; }
    const int kbase = ksl * kpt + 4 * ksl;  // == apad(ksl * kpt, kpt)
    for (int ch = 0; ch < nchunk; ++ch) {
        const int n = min(KCH, kpt - ch * KCH);
        if (ch > 0) {
#pragma unroll
            for (int kk = 0; kk < KCH; ++kk) {
                if (kk < n) {
                    wh[kk] = whh[(int64_t)(ch * KCH + kk) * wstride];
                    if (has_in) wi[kk] = wih[(int64_t)(ch * KCH + kk) * wstride];
                }
            }
        }
        if (C.has_pred) fma_rows(acc_h, wh, a_s, op_ld, kbase + ch * KCH, n);
        if (has_in) fma_rows(acc_i, wi, u_s, op_ld, kbase + ch * KCH, n);
    }
    if (C.has_pred) reduce_k_lanes(acc_h);
    if (has_in) reduce_k_lanes(acc_i);
    if (lane < 4) {  // K-lane 0 holds the sums of column group 4*wave + lane
        const int cg = 4 * wave + lane;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            *reinterpret_cast<float4*>(gh_s + r * 96 + 4 * cg) = acc_h[r];
            *reinterpret_cast<float4*>(gi_s + r * 96 + 4 * cg) = acc_i[r];
        }
    }
    __syncthreads();

    // ---- phase C: gates for 8 rows x 32 units
    if (tid < RB * JS) {
        const int r = tid >> 5, jj = tid & 31;
        if (r < nr) {
            const int v = rec_s[r].x;
            const int j = sl * JS + jj;
            float gr, gz, gn;
            if (has_in) {
                gr = gi_s[r * 96 + jj] + C.bih[j];
                gz = gi_s[r * 96 + 32 + jj] + C.bih[H + j];
                gn = gi_s[r * 96 + 64 + jj] + C.bih[2 * H + j];
            } else {
                const float* g0 = C.gi0 + (int64_t)v * 3 * H;
                gr = g0[j]; gz = g0[H + j]; gn = g0[2 * H + j];
            }
            const float hr = gh_s[r * 96 + jj] + C.bhh[j];
            const float hz = gh_s[r * 96 + 32 + jj] + C.bhh[H + j];
            const float hn = gh_s[r * 96 + 64 + jj] + C.bhh[2 * H + j];
            const float a = a_s[r * op_ld + apad(j, kpt)];
            const float rg = sigm(gr + hr);
            const float zg = sigm(gz + hz);
            const float ng = tanhf(fmaf(rg, hn, gn));
            const float hv = fmaf(zg, a - ng, ng);  // n + z * (a - n)
            C.h_out[(int64_t)v * ld_h + j] = hv;
            float sp = C.wkey[j] * hv;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) sp += __shfl_xor(sp, off, 64);  // 32-lane half-wave
            if (jj == 0) C.spart[(int64_t)v * NS + sl] = sp;
        }
    }
}

// Pack W [3H, K] (torch GRUCell layout: row g*H + j, K contiguous) into slice / lane order:
// out[((sl * kpt + kk) * 6 + w) * 64 + lane] (float4) = the 4 columns of column group cg = 4w + (lane & 3)
// of slice sl at k = (lane >> 2) * kpt + kk, where local column lc = 4cg + q  ->  gate lc/32, unit sl*32 + lc%32.
__global__ void __launch_bounds__(256) pack_slices_kernel(const float* __restrict__ W, float4* __restrict__ out, int H,
                                                           int K, int64_t total) {
    const int kpt = K >> 4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        int64_t rest = idx >> 6;
        const int w = (int)(rest % 6); rest /= 6;
        const int kk = (int)(rest % kpt);
        const int sl = (int)(rest / kpt);
        const int cg = 4 * w + (lane & 3);
        const int k = (lane >> 2) * kpt + kk;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lc = 4 * cg + q;
            const int row = (lc >> 5) * H + sl * JS + (lc & 31);
            v[q] = W[(int64_t)row * K + k];
        }
        out[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

extern "C" int dagnn_pack_slices(const float* w, float* out, int H, int K, void* stream) {
    if (!w || !out || H <= 0 || K <= 0 || (H % JS) || (K % 64)) return DAGNN_EINVAL;
    const int64_t total = (int64_t)(H / JS) * (K / 16) * FT;  // float4 elements == 3H*K/4
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_slices_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(out), H, K, total);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_frontier_run(const dagnn_plan* pl, const dagnn_frontier_args* a, const int32_t* const* layer_ptr,
                                  const int32_t* num_layers, void* stream) {
    if (!pl || !pl->data || !a || !layer_ptr || !num_layers) return DAGNN_EINVAL;
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3;
    if (H <= 0 || (H % 64) || Ls <= 0 || !dir_mask || a->ld_h < H || (a->ld_h & 3)) return DAGNN_EINVAL;
    int ndir = 0, dirs[2];
    for (int d = 0; d < 2; ++d) if ((dir_mask >> d) & 1) dirs[ndir++] = d;
    if (ndir * Ls > DAGNN_MAX_CELLS) return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    int Tmax = 0;
    for (int q = 0; q < ndir; ++q) {
        const int d = dirs[q];
        if (!layer_ptr[d] || num_layers[d] < 0) return DAGNN_EINVAL;
        if (num_layers[d] > Tmax) Tmax = num_layers[d];
        for (int i = 0; i < Ls; ++i) {
            const dagnn_frontier_cell& c = a->cell[d][i];
            if (!c.w_hh_pk || !c.b_hh || !c.w_key || !c.h_out || !c.score_parts) return DAGNN_EINVAL;
            if (i == 0 ? !c.gi0 : (!c.w_ih_pk || !c.b_ih)) return DAGNN_EINVAL;
        }
    }
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    const int NS = H / JS;
    const int op_ld = H + 64;
    const size_t lds = (size_t)(2 * RB * op_ld + 2 * RB * 96) * sizeof(float) + RB * sizeof(int4);
    StepArgs S;
    S.H = H; S.ld_h = a->ld_h; S.NS = NS; S.R = pl->num_edge_feats; S.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
    hipStream_t st = (hipStream_t)stream;
    for (int s = 0; s < Tmax + Ls - 1; ++s) {
        int nc = 0, blocks = 0;
        S.blk_start[0] = 0;
        for (int q = 0; q < ndir; ++q) {
            const int d = dirs[q];
            for (int i = 0; i < Ls; ++i) {
                const int t = s - i;
                if (t < 0 || t >= num_layers[d]) continue;
                const int r0 = layer_ptr[d][t], r1 = layer_ptr[d][t + 1];
                if (r1 <= r0) continue;
                const dagnn_frontier_cell& c = a->cell[d][i];
                Cell& K = S.cell[nc];
                K.whh = (const float4*)c.w_hh_pk; K.wih = i > 0 ? (const float4*)c.w_ih_pk : nullptr;
                K.bhh = c.b_hh; K.bih = c.b_ih; K.wkey = c.w_key;
                K.gain = pl->num_edge_feats > 0 ? c.edge_gain : nullptr;
                K.vid = a->vid_mod > 0 ? c.vid_bias : nullptr;
                K.gi0 = i == 0 ? c.gi0 : nullptr;
                K.h_in = i > 0 ? a->cell[d][i - 1].h_out : nullptr;
                K.h_out = c.h_out; K.spart = c.score_parts;
                K.dir = d; K.row_base = r0; K.row_end = r1; K.has_pred = t > 0;
                blocks += (r1 - r0 + RB - 1) / RB;
                S.blk_start[++nc] = blocks;
            }
        }
        if (nc == 0) continue;
        S.ncell = nc;
        hipLaunchKernelGGL(frontier_step_kernel, dim3((unsigned)(blocks * NS)), dim3(FT), lds, st,
                           (const int32_t*)pl->data, L, S);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return DAGNN_EHIP(e);
    }
    return DAGNN_OK;
}
