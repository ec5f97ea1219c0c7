// wgrad.hip - weight and bias gradients of the GRU cells: the parallel epilogue of the reverse sweep.
//
// Reference path replaced: the parameter-gradient accumulation torch autograd performs for every `nn.GRUCell` call of
// ogbg-code/model/dagnn.py:181 under `loss.backward()` (main_pyg.py:62) - one small GEMM per micro-step there; here the
// sweep (bwd_dataflow.hip / backward.hip) leaves the pre-activation gradients dgi, dgh [N, 3H] of every cell and the
// weight gradients are ONE batch of transposed products over all nodes:
//       dW_ih = dgi^T u   [3H, Din]        dW_hh = dgh^T a   [3H, H]        db_ih = sum_n dgi,  db_hh = sum_n dgh.
// "TN" shape: both operands are node-major, the reduction runs over the N nodes (10^4..10^5) and the output is small
// (768 x 256), so the reduction is split: workgroup (job, 128 output rows, split s) walks its share of the nodes two at
// a time with v_mfma_f32_32x32x2_f32 (exact fp32) and writes a partial tile; a second kernel adds the partials in split
// order (deterministic, no atomics) and drops the padding rows.  Both MFMA operands come straight from global memory in
// fragment order - for a fixed node the 32 x 4 output rows (columns of dg) and the 32 x 2 output columns (columns of u)
// a lane needs are contiguous: one dwordx4 + one dwordx2 load per lane feed 8 MFMAs, no LDS, no transposes.  The bias
// gradients are column sums of the same A fragments (4 adds per load in the waves that own output column 0).
#include "common.h"

namespace {

typedef float wf16 __attribute__((ext_vector_type(16)));

constexpr int WG_TM = 128;       // output rows per workgroup (4 interleaved 32-row MFMA tiles: row = m0 + 4 i + t)
constexpr int WG_TK = 64;        // output columns per wave (2 interleaved 32-column tiles: col = k0 + 2 j + u)
constexpr int WG_WAVES = 4;      // waves per workgroup: WG_WAVES * WG_TK = 256 output columns
constexpr int WG_MAX_JOBS = 32;

struct WgJob {
    const float* A;     // [N, lda] pre-activation gradients (3 gate blocks of Hp columns)
    const float* B;     // [N, ldb] inputs of the product (u or a)
    float* dW;          // [3H, K2] out
    float* db;          // [3H] out, or null
    int lda, ldb, K2;
};

struct WgArgs {
    WgJob job[WG_MAX_JOBS];
    int njob, M, Hp, H, splits, tiles_m, tiles_k, wgs_per;   // tiles_k: 64-column tiles; wgs_per: workgroups per (job, split)
    int64_t N;
    float* part;        // [njob][splits][M][K2max] partial products
    float* bpart;       // [njob][splits][M] partial column sums
    int K2max;
};

__global__ void __launch_bounds__(64 * WG_WAVES, 2) wgrad_partial_kernel(WgArgs S) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A wave owns one 128 x 64 output tile; the waves share nothing, so the tiles of a (job, split) are dealt to waves
    // DENSELY - tile id = 4 workgroup + wave, column tile fastest (round 4: with whole workgroups of 128 x 256 an input width
    // of 320 left three of the four waves of every second workgroup without a tile: 58 % of the launch useful)
    int bid = blockIdx.x;
    const int wg = bid % S.wgs_per; bid /= S.wgs_per;
    const int split = bid % S.splits;
    const int jb = bid / S.splits;
    const WgJob& J = S.job[jb];
    const int tile = wg * WG_WAVES + wave;
    if (tile >= S.tiles_m * S.tiles_k) return;
    const int tm = tile / S.tiles_k, tk = tile - tm * S.tiles_k;
    const int r = lane & 31, kk = lane >> 5;
    const int m0 = tm * WG_TM, k0 = tk * WG_TK;
    const int ma = m0 + 4 * r, kb = k0 + 2 * r;          // first output row / column this lane feeds
    const bool a_on = ma < S.M, b_on = kb < J.K2;
    if (k0 >= J.K2 && tk != 0) return;   // nothing to do for this wave (whole-wave exit; column tile 0 also sums the bias)
    // nodes of this split: pairs (n, n + 1)
    int64_t chunk = (S.N + S.splits - 1) / S.splits;
    chunk += chunk & 1;
    const int64_t n_begin = (int64_t)split * chunk, n_end = min(S.N, n_begin + chunk);
    wf16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = J.db != nullptr && tk == 0;
    const float* ap = J.A + ma;
    const float* bp = J.B + kb;
    constexpr int UN = 4;
    // software pipeline: the fragments of the NEXT group of node pairs are requested before the current group's 32
    // MFMAs issue, so a wave's loads fly under its own matrix work (two waves per SIMD alone leave the pipe half idle)
    float4 a4[2][UN];
    float2 b2[2][UN];
    auto fetch = [&](int buf, int64_t n) {
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const int64_t row = n + 2 * s + kk;
            a4[buf][s] = make_float4(0.f, 0.f, 0.f, 0.f);
            b2[buf][s] = make_float2(0.f, 0.f);
            if (row < n_end) {
                if (a_on) a4[buf][s] = *reinterpret_cast<const float4*>(ap + row * J.lda);
                if (b_on) b2[buf][s] = *reinterpret_cast<const float2*>(bp + row * J.ldb);
            }
        }
    };
    auto work = [&](int buf) {
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const float av[4] = {a4[buf][s].x, a4[buf][s].y, a4[buf][s].z, a4[buf][s].w};
            const float bv[2] = {b2[buf][s].x, b2[buf][s].y};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[u], acc[t][u], 0, 0, 0);
                bs[t] += av[t];
            }
        }
    };
    if (n_begin < n_end) fetch(0, n_begin);
    for (int64_t n = n_begin; n < n_end; n += 4 * UN) {
        fetch(1, n + 2 * UN);
        work(0);
        fetch(0, n + 4 * UN);
        work(1);
    }
    // partial tile: D[i][j] of tile (t, u) is C[m0 + 4 i + t][k0 + 2 j + u], i = (e & 3) + 8 (e >> 2) + 4 kk, j = r
    float* P = S.part + ((int64_t)(jb * S.splits + split) * S.M) * S.K2max;
    if (k0 < J.K2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * kk;
                const int m = m0 + 4 * i + t;
                if (m < S.M && kb < J.K2)
                    *reinterpret_cast<float2*>(P + (int64_t)m * S.K2max + kb) = make_float2(acc[t][0][e], acc[t][1][e]);
            }
    }
    if (do_bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float tot = bs[t] + __shfl_xor(bs[t], 32, 64);
            if (kk == 0 && ma + t < S.M) S.bpart[(int64_t)(jb * S.splits + split) * S.M + ma + t] = tot;
        }
    }
}

// out[g H + j][k] = sum over the splits, in order, of part[.][g Hp + j][k]; the same for the bias sums
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WgArgs S) {
    const int jb = blockIdx.y;
    const WgJob& J = S.job[jb];
    const int H3 = 3 * S.H;
    const int64_t total = (int64_t)H3 * J.K2;
    const float* P = S.part + (int64_t)jb * S.splits * S.M * S.K2max;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / J.K2), k = (int)(idx - (int64_t)row * J.K2);
        const int g = row / S.H, j = row - g * S.H;
        const int64_t off = (int64_t)(g * S.Hp + j) * S.K2max + k;
        float s = 0.f;
        for (int q = 0; q < S.splits; ++q) s += P[(int64_t)q * S.M * S.K2max + off];
        J.dW[idx] = s;
    }
    if (J.db && blockIdx.x == 0) {
        const float* Pb = S.bpart + (int64_t)jb * S.splits * S.M;
        for (int row = threadIdx.x; row < H3; row += blockDim.x) {
            const int g = row / S.H, j = row - g * S.H;
            float s = 0.f;
            for (int q = 0; q < S.splits; ++q) s += Pb[(int64_t)q * S.M + g * S.Hp + j];
            J.db[row] = s;
        }
    }
}

// ---- weighted column sums: out[k] = sum_n w[n] X[n][k] (w = null: plain column sums) for several small jobs at once -
// the attention-key gradients sum_v sigma_v keys_v, the edge-feature sums and sum_v sigma_v of the epilogue.  Two
// stages like the products above: every workgroup sums a chunk of rows, a second kernel adds the chunks in order.
constexpr int CS_MAX_JOBS = 32;
constexpr int CS_CHUNKS = 256;
struct CsJob { const float* X; const float* w; float* out; int ld, K; };
struct CsArgs { CsJob job[CS_MAX_JOBS]; int njob; int64_t N; float* part; int Kmax; };

__global__ void __launch_bounds__(256) colsum_partial_kernel(CsArgs S) {
    const CsJob& J = S.job[blockIdx.y];
    const int chunk = blockIdx.x;
    const int64_t per = (S.N + CS_CHUNKS - 1) / CS_CHUNKS;
    const int64_t n0 = chunk * per, n1 = min(S.N, n0 + per);
    float* P = S.part + ((int64_t)blockIdx.y * CS_CHUNKS + chunk) * S.Kmax;
    // thread = (row phase, column): 256 / K' row phases walk the chunk's rows together, then meet in LDS
    __shared__ float red[256];
    for (int k0 = 0; k0 < J.K; k0 += 256) {
        const int kw = min(256, J.K - k0);
        int cols = 1;
        while (cols < kw) cols <<= 1;          // columns per pass rounded up to a power of two (<= 256)
        const int phases = 256 / cols;
        const int c = threadIdx.x % cols, ph = threadIdx.x / cols;
        float s = 0.f;
        if (c < kw) {   // four rows in flight per thread (a chain of dependent loads would be latency-bound)
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            int64_t n = n0 + ph;
            for (; n + 3 * phases < n1; n += 4 * phases) {
                float xv[4], wv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xv[e] = J.X[(n + e * phases) * J.ld + k0 + c];
                    wv[e] = J.w ? J.w[n + e * phases] : 1.0f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] = fmaf(wv[e], xv[e], s4[e]);
            }
            for (; n < n1; n += phases) s4[0] = fmaf(J.w ? J.w[n] : 1.0f, J.X[n * J.ld + k0 + c], s4[0]);
            s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (ph == 0 && c < kw) {
            float t = 0.f;
            for (int q = 0; q < phases; ++q) t += red[q * cols + c];
            P[k0 + c] = t;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) colsum_reduce_kernel(CsArgs S) {
    const CsJob& J = S.job[blockIdx.x];
    const float* P = S.part + (int64_t)blockIdx.x * CS_CHUNKS * S.Kmax;
    for (int k = threadIdx.x; k < J.K; k += blockDim.x) {
        float t = 0.f;
        for (int q = 0; q < CS_CHUNKS; ++q) t += P[(int64_t)q * S.Kmax + k];
        J.out[k] = t;
    }
}

}  // namespace

extern "C" size_t dagnn_colsum_workspace_bytes(int njob, int max_cols) {
    if (njob <= 0 || max_cols <= 0) return 0;
    return (size_t)njob * CS_CHUNKS * max_cols * sizeof(float);
}

extern "C" int dagnn_colsum_run(const dagnn_colsum_job* jobs, int njob, int64_t N, void* workspace, size_t workspace_bytes,
                                void* stream) {
    if (!jobs || njob <= 0 || njob > CS_MAX_JOBS || N < 0 || !workspace) return DAGNN_EINVAL;
    CsArgs S;
    int Kmax = 0;
    for (int q = 0; q < njob; ++q) {
        if (!jobs[q].x || !jobs[q].out || jobs[q].cols <= 0 || jobs[q].ld_x < jobs[q].cols) return DAGNN_EINVAL;
        S.job[q].X = jobs[q].x; S.job[q].w = jobs[q].weight; S.job[q].out = jobs[q].out;
        S.job[q].ld = jobs[q].ld_x; S.job[q].K = jobs[q].cols;
        if (jobs[q].cols > Kmax) Kmax = jobs[q].cols;
    }
    if (dagnn_colsum_workspace_bytes(njob, Kmax) > workspace_bytes) return DAGNN_ENOSPC;
    S.njob = njob; S.N = N; S.part = (float*)workspace; S.Kmax = Kmax;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(CS_CHUNKS, (unsigned)njob), dim3(256), 0, st, S);
    DAGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)njob), dim3(256), 0, st, S);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

// ---- the rest of the epilogue in ONE launch: per cell the gradients of attn_lin.weight and of the edge encoder from the three
// column sums (dagnn_amd/autograd.py did this with ~10 tiny torch ops per cell: ~40 launches of a host-bound stretch of the step)
struct AgArgs { dagnn_attn_grad_job job[DAGNN_ATTN_GRAD_MAX_JOBS]; };
__global__ void __launch_bounds__(256) attn_grads_kernel(AgArgs S) {
    const dagnn_attn_grad_job& J = S.job[blockIdx.x];
    const float ssum = J.sigma_sum ? J.sigma_sum[0] : 0.f;
    const bool edge = J.edge_w != nullptr && J.feat_sum != nullptr;
    for (int j = threadIdx.x; j < J.attn_len; j += blockDim.x) {
        float v = 0.f;
        const int k = j - J.dq;
        if (k >= 0 && k < J.kd) {
            v = J.key_sum[k];
            if (edge) {
                float e = 0.f;
                for (int r = 0; r < J.R; ++r) e = fmaf(J.edge_w[(int64_t)k * J.R + r], J.feat_sum[r], e);
                v = v + e + J.edge_b[k] * ssum;
            }
        }
        J.g_attn[j] = v;
    }
    if (edge) {
        for (int k = threadIdx.x; k < J.kd; k += blockDim.x) {
            const float wk = J.attn_w[J.dq + k];
            for (int r = 0; r < J.R; ++r) J.g_edge_w[(int64_t)k * J.R + r] = wk * J.feat_sum[r];
            J.g_edge_b[k] = wk * ssum;
        }
    }
}

extern "C" int dagnn_attn_grads_run(const dagnn_attn_grad_job* jobs, int njob, void* stream) {
    if (!jobs || njob <= 0 || njob > DAGNN_ATTN_GRAD_MAX_JOBS) return DAGNN_EINVAL;
    AgArgs S;
    for (int q = 0; q < njob; ++q) {
        const dagnn_attn_grad_job& j = jobs[q];
        if (!j.key_sum || !j.attn_w || !j.g_attn || j.kd <= 0 || j.dq < 0 || j.attn_len < j.dq + j.kd || j.R < 0) return DAGNN_EINVAL;
        if (j.edge_w && (!j.feat_sum || !j.sigma_sum || !j.edge_b || !j.g_edge_w || !j.g_edge_b || j.R <= 0)) return DAGNN_EINVAL;
        S.job[q] = j;
    }
    hipLaunchKernelGGL(attn_grads_kernel, dim3((unsigned)njob), dim3(256), 0, (hipStream_t)stream, S);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" size_t dagnn_wgrad_workspace_bytes(int njob, int Hp, int K2max, int splits) {
    if (njob <= 0 || Hp <= 0 || K2max <= 0 || splits <= 0) return 0;
    return (size_t)njob * splits * 3 * Hp * ((size_t)K2max + 1) * sizeof(float);
}

extern "C" int dagnn_wgrad_splits(int num_cus, int njob, int Hp, int K2max, int64_t N) {
    if (num_cus <= 0 || njob <= 0 || Hp <= 0 || K2max <= 0 || N <= 0) return 0;
    const int tiles = njob * ((((3 * Hp + WG_TM - 1) / WG_TM) * ((K2max + WG_TK - 1) / WG_TK) + WG_WAVES - 1) / WG_WAVES);   // workgroups per split
    int s = 2 * num_cus / tiles;   // two workgroups per CU, all resident at once
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    while (s > 1 && (N + s - 1) / s < 64) --s;   // a split is worth at least a few dozen nodes
    return s;
}

// One transposed product with a SMALL reduction and a LARGE output (the vocabulary heads' weight gradient: 128 graphs reduced,
// 25 010 x 1 024 out - the opposite regime of the cells' products): the same wave tiles, no split, written straight to `out`.
extern "C" int dagnn_tn_product(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t N, int M, int K2, float* out,
                                float* colsum, void* stream) {
    if (!A || !B || !out || N <= 0 || M <= 0 || K2 <= 0 || (K2 % 2) || lda < M || (lda % 4) || ldb < K2 || (ldb % 2) ||
        lda > 0x7fffffff || ldb > 0x7fffffff || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 7) || (((uintptr_t)out) & 7))
        return DAGNN_EINVAL;
    // (the last float4 of a row may reach past column M - 1: lda % 4 == 0 keeps it inside the row's pitch, its products are
    // rows >= M of the output and never stored)
    if ((int64_t)((M + 3) / 4 * 4) > lda) return DAGNN_EINVAL;
    WgArgs S;
    S.job[0].A = A; S.job[0].B = B; S.job[0].dW = out; S.job[0].db = colsum;
    S.job[0].lda = (int)lda; S.job[0].ldb = (int)ldb; S.job[0].K2 = K2;
    S.njob = 1; S.M = M; S.Hp = M; S.H = M; S.splits = 1; S.N = N; S.K2max = K2;
    S.tiles_m = (M + WG_TM - 1) / WG_TM;
    S.tiles_k = (K2 + WG_TK - 1) / WG_TK;
    S.wgs_per = (S.tiles_m * S.tiles_k + WG_WAVES - 1) / WG_WAVES;
    S.part = out;        // partial tile of split 0 of job 0 == the result
    S.bpart = colsum;
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3((unsigned)S.wgs_per), dim3(64 * WG_WAVES), 0, (hipStream_t)stream, S);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_wgrad_run(const dagnn_wgrad_job* jobs, int njob, int64_t N, int Hp, int H, int splits, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!jobs || njob <= 0 || njob > WG_MAX_JOBS || N < 0 || Hp <= 0 || H <= 0 || H > Hp || (Hp % 4) || splits <= 0 ||
        splits > 64 || !workspace)
        return DAGNN_EINVAL;
    WgArgs S;
    int K2max = 0;
    for (int q = 0; q < njob; ++q) {
        const dagnn_wgrad_job& j = jobs[q];
        if (!j.dg || !j.in || !j.d_weight || j.in_dim <= 0 || (j.in_dim % 2) || j.ld_dg < 3 * Hp || (j.ld_dg % 4) ||
            j.ld_in < j.in_dim || (j.ld_in % 2))
            return DAGNN_EINVAL;
        S.job[q].A = j.dg; S.job[q].B = j.in; S.job[q].dW = j.d_weight; S.job[q].db = j.d_bias;
        S.job[q].lda = j.ld_dg; S.job[q].ldb = j.ld_in; S.job[q].K2 = j.in_dim;
        if (j.in_dim > K2max) K2max = j.in_dim;
    }
    if (dagnn_wgrad_workspace_bytes(njob, Hp, K2max, splits) > workspace_bytes) return DAGNN_ENOSPC;
    S.njob = njob; S.M = 3 * Hp; S.Hp = Hp; S.H = H; S.splits = splits; S.N = N; S.K2max = K2max;
    S.tiles_m = (S.M + WG_TM - 1) / WG_TM;
    S.tiles_k = (K2max + WG_TK - 1) / WG_TK;
    S.wgs_per = (S.tiles_m * S.tiles_k + WG_WAVES - 1) / WG_WAVES;
    S.part = (float*)workspace;
    S.bpart = S.part + (size_t)njob * splits * S.M * K2max;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(S.wgs_per * splits * njob);
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(grid), dim3(64 * WG_WAVES), 0, st, S);
    DAGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(96, (unsigned)njob), dim3(256), 0, st, S);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
