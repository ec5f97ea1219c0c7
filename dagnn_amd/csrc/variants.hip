// Constructor-string variants of the message-passing loop, forward pass.
//
// `agg` in {mattn_h, gated_sum, add, max}, `agg_x`, `recurr=0` of ogbg-code/model/dagnn.py select other aggregators
// (dagnn.py:232-276,379-409) or a Linear cell (:83-85,181) on the same topological loop (:144-182).  No BASELINE
// configuration uses them, so they get one generic lock-step pass instead of the tuned kernels of frontier.hip:
//
//   step s, every cell (d, i) at layer t = s - i:
//     variant_aggregate_kernel   one wave per frontier row: messages of its in-edges (plan CSR, original edge order)
//                                -> aggregate row, by node id
//     variant_cell_kernel        16 rows x 64 units per workgroup: GRU or Linear over [input ; aggregate], weights
//                                k-major so a wave reads 256 contiguous bytes per k
//     variant_cell_kernel        (again, as a plain linear map) the per-node projections the aggregator needs of
//                                the rows just produced (MultAttnConv: W_r k + b_r, W_l q + b_l; GatedSumConv:
//                                W_g h + b_g, W_m h + b_m) - projecting once per node instead of once per edge; the
//                                edge-encoder term is folded in per edge through the [dim, R] products of the weights
//
// Every operand is indexed by node id; a row is written exactly once; no atomics (fixed summation order).
#include "common.h"

namespace {

constexpr int VMAXC = DAGNN_MAX_DIRS * DAGNN_MAX_STACKED;

struct VAggArgs {
    dagnn_variant_aggregator a[VMAXC];
    int dir[VMAXC], r0[VMAXC], r1[VMAXC];
    int R;
};

__device__ __forceinline__ float edge_term(const float* __restrict__ m, const float* __restrict__ v, int k, int R,
                                           const float* __restrict__ attr) {
    float e = v ? v[k] : 0.f;
    if (m)
        for (int r = 0; r < R; ++r) e += m[(int64_t)k * R + r] * attr[r];
    return e;
}

// KPL = elements of a message per lane (val_dim <= 64 * KPL)
template <int KPL>
__global__ void __launch_bounds__(256) variant_aggregate_kernel(const int32_t* __restrict__ plan, PlanLayout L,
                                                                 VAggArgs A) {
    const int c = blockIdx.y;
    const dagnn_variant_aggregator& g = A.a[c];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = A.r0[c] + blockIdx.x * 4 + wave;
    if (slot >= A.r1[c]) return;
    const int d = A.dir[c], R = A.R;
    const int32_t* __restrict__ rec = plan + L.rowrec[d] + 16 * (int64_t)slot;
    const int v = rec[0], eb = rec[1], ee = rec[2];
    const int32_t* __restrict__ col = plan + L.col[d];
    const float* __restrict__ ea = reinterpret_cast<const float*>(plan + L.eattr[d]);
    const int dv = g.lands ? g.val_dim : 0, mode = g.mode;
    float* __restrict__ out = g.out + (int64_t)v * g.ld_out;

    float acc[KPL];
#pragma unroll
    for (int q = 0; q < KPL; ++q) acc[q] = 0.f;
    float run_max = -INFINITY, run_sum = 0.f;   // online segment softmax (ATTN, MATTN)
    if (g.lands) {
        // four edges per trip: their index / row loads are independent, so a long in-edge list (out-degree > 100 in
        // the reverse direction) is not one dependent round trip per edge; sums stay in edge order
        for (int e0 = eb; e0 < ee; e0 += 4) {
            int j[4];
            bool ok[4];
            const float* attr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ok[u] = e0 + u < ee;
                const int e = ok[u] ? e0 + u : e0;
                j[u] = col[e];
                attr[u] = ea + (int64_t)e * R;
            }
            if (mode == DAGNN_AGG_ATTN || mode == DAGNN_AGG_MATTN) {
                float logit[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float part = 0.f;
                    if (mode == DAGNN_AGG_ATTN) {
                        const float* key = g.node0 + (int64_t)j[u] * g.ld_node;
                        for (int k = lane; k < g.aux_dim; k += 64) part += g.edge_vec0[k] * key[k];
                    } else {
                        const float* kr = g.node0 + (int64_t)j[u] * g.ld_node;
                        const float* ql = g.node1 + (int64_t)v * g.ld_node;
                        for (int k = lane; k < g.aux_dim; k += 64)
                            part += ql[k] * (kr[k] + edge_term(g.edge_mat0, g.edge_vec0, k, R, attr[u]));
                    }
                    logit[u] = part;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    logit[u] = wave_sum(logit[u]);
                    if (mode == DAGNN_AGG_ATTN && g.edge_mat0)
                        for (int r = 0; r < R; ++r) logit[u] += g.edge_mat0[r] * attr[u][r];
                }
                float m2 = run_max;
#pragma unroll
                for (int u = 0; u < 4; ++u) m2 = ok[u] ? fmaxf(m2, logit[u]) : m2;
                const float scale = __expf(run_max - m2);   // exp(-inf) = 0 on the first trip
                float w[4];
                run_sum *= scale;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w[u] = ok[u] ? __expf(logit[u] - m2) : 0.f;
                    run_sum += w[u];
                }
                run_max = m2;
#pragma unroll
                for (int q = 0; q < KPL; ++q) {
                    const int k = lane + 64 * q;
                    if (k < dv) {
                        float x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[u] = g.vals[(int64_t)j[u] * g.ld_vals + k];
                        float t = acc[q] * scale;
#pragma unroll
                        for (int u = 0; u < 4; ++u) t += w[u] * x[u];
                        acc[q] = t;
                    }
                }
            } else if (mode == DAGNN_AGG_GATED) {
#pragma unroll
                for (int q = 0; q < KPL; ++q) {
                    const int k = lane + 64 * q;
                    if (k < dv) {
                        float gate[4], msg[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            gate[u] = g.node0[(int64_t)j[u] * g.ld_node + k];
                            msg[u] = g.node1[(int64_t)j[u] * g.ld_node + k];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            gate[u] += edge_term(g.edge_mat0, g.edge_vec0, k, R, attr[u]);
                            msg[u] += edge_term(g.edge_mat1, g.edge_vec1, k, R, attr[u]);
                            if (ok[u]) acc[q] += msg[u] / (1.f + __expf(-gate[u]));
                        }
                    }
                }
            } else {   // ADD, MAX
#pragma unroll
                for (int q = 0; q < KPL; ++q) {
                    const int k = lane + 64 * q;
                    if (k < dv) {
                        float msg[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) msg[u] = g.vals[(int64_t)j[u] * g.ld_vals + k];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            msg[u] += edge_term(g.edge_mat0, g.edge_vec0, k, R, attr[u]);
                            if (!ok[u]) continue;
                            if (mode == DAGNN_AGG_ADD) acc[q] += msg[u];
                            else acc[q] = (e0 == eb && u == 0) ? msg[u] : fmaxf(acc[q], msg[u]);
                        }
                    }
                }
            }
        }
    }
    const float norm = (mode == DAGNN_AGG_ATTN || mode == DAGNN_AGG_MATTN) ? 1.f / (run_sum + 1e-16f) : 1.f;
#pragma unroll
    for (int q = 0; q < KPL; ++q) {
        const int k = lane + 64 * q;
        if (k < dv) out[k] = acc[q] * norm;
    }
    for (int k = dv + lane; k < g.out_dim; k += 64) out[k] = 0.f;
}

// ------------------------------------------------------------------------------------------ cell / linear map
struct VJob {
    int dir, in_dim, out_dim, zero_agg;       // zero_agg: the aggregate operand is all zeros (layer 0, or nothing lands)
    const float* input; int64_t ld_in;
    const float* agg; int64_t ld_agg; int agg_dim, pad;
    const float* w_in_t; const float* w_agg_t; const float* b_in; const float* b_agg;
    const float* b_vid; int vid_mod, pad2;    // linear maps: + b_vid[(node % vid_mod) * out_dim + unit] (per-vertex-id bias), or NULL
    float* out; int64_t ld_out;
};
struct VJobArgs {
    VJob j[VMAXC];
    int r0[VMAXC], r1[VMAXC];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// Workgroup = (4 * RPT rows, 64 output units); thread = (unit, row group of RPT rows).  G = 3: GRU gates r,z,n;
// G = 1: plain linear map.  LDS: the rows' inputs and aggregates, zero-padded to a multiple of 4 floats.
template <int G, int RPT>
__global__ void __launch_bounds__(256) variant_cell_kernel(const int32_t* __restrict__ plan, PlanLayout L, VJobArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RB = 4 * RPT;
    const int c = blockIdx.z;
    const VJob& J = A.j[c];
    const int slot0 = A.r0[c] + blockIdx.x * RB;
    if (slot0 >= A.r1[c]) return;
    const int nrows = min(RB, A.r1[c] - slot0);
    const int tid = threadIdx.x;
    const int unit = blockIdx.y * 64 + (tid & 63), rg = tid >> 6;
    const int din = (J.in_dim + 3) & ~3, dag = J.zero_agg ? 0 : ((J.agg_dim + 3) & ~3);
    float* s_in = lds;                   // [RB][din]
    float* s_ag = lds + RB * din;        // [RB][dag]
    __shared__ int s_node[RB];
    const int32_t* __restrict__ rec = plan + L.rowrec[J.dir] + 16 * (int64_t)slot0;
    if (tid < RB) s_node[tid] = tid < nrows ? rec[16 * tid] : -1;
    __syncthreads();
    for (int idx = tid; idx < RB * din; idx += 256) {
        const int r = idx / din, k = idx - r * din;
        const int v = s_node[r];
        s_in[idx] = (v >= 0 && k < J.in_dim) ? J.input[(int64_t)v * J.ld_in + k] : 0.f;
    }
    for (int idx = tid; idx < RB * dag; idx += 256) {
        const int r = idx / dag, k = idx - r * dag;
        const int v = s_node[r];
        s_ag[idx] = (v >= 0 && k < J.agg_dim) ? J.agg[(int64_t)v * J.ld_agg + k] : 0.f;
    }
    __syncthreads();
    if (unit >= J.out_dim) return;
    const int ldw = G * J.out_dim;
    float ai[RPT][G], ah[RPT][G];
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int q = 0; q < G; ++q) ai[r][q] = ah[r][q] = 0.f;
    for (int k = 0; k < din; k += 4) {
        float w[4][G];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < G; ++q)
                w[kk][q] = J.w_in_t[(int64_t)min(k + kk, J.in_dim - 1) * ldw + q * J.out_dim + unit];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(s_in + (rg * RPT + r) * din + k);
#pragma unroll
            for (int q = 0; q < G; ++q)
                ai[r][q] += x.x * w[0][q] + x.y * w[1][q] + x.z * w[2][q] + x.w * w[3][q];
        }
    }
    for (int k = 0; k < dag; k += 4) {
        float w[4][G];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < G; ++q)
                w[kk][q] = J.w_agg_t[(int64_t)min(k + kk, J.agg_dim - 1) * ldw + q * J.out_dim + unit];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(s_ag + (rg * RPT + r) * dag + k);
#pragma unroll
            for (int q = 0; q < G; ++q)
                ah[r][q] += x.x * w[0][q] + x.y * w[1][q] + x.z * w[2][q] + x.w * w[3][q];
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = rg * RPT + r;
        const int v = s_node[row];
        if (v < 0) continue;
        float res;
        if constexpr (G == 3) {
            const int H = J.out_dim;
            const float bi_r = J.b_in[unit], bi_z = J.b_in[H + unit], bi_n = J.b_in[2 * H + unit];
            const float bh_r = J.b_agg[unit], bh_z = J.b_agg[H + unit], bh_n = J.b_agg[2 * H + unit];
            const float rr = sigmoidf_(ai[r][0] + bi_r + ah[r][0] + bh_r);
            const float zz = sigmoidf_(ai[r][1] + bi_z + ah[r][1] + bh_z);
            const float nn = tanhf(ai[r][2] + bi_n + rr * (ah[r][2] + bh_n));
            const float hp = (dag && unit < J.agg_dim) ? s_ag[row * dag + unit] : 0.f;
            res = (1.f - zz) * nn + zz * hp;
        } else {
            res = ai[r][0] + ah[r][0] + (J.b_in ? J.b_in[unit] : 0.f);
            if (J.b_vid) res += J.b_vid[(int64_t)(v % J.vid_mod) * J.out_dim + unit];
        }
        J.out[(int64_t)v * J.ld_out + unit] = res;
    }
}

// Thin launches (few rows): the same arithmetic with K split over the four waves - workgroup = (4 rows, 64 units),
// thread = (unit, quarter of K); the quarters meet in LDS and wave q finishes row q.  A thin launch is a chain of
// dependent L2 reads per workgroup, so a four times shorter chain is a four times shorter launch.
template <int G>
__global__ void __launch_bounds__(256) variant_cell_thin_kernel(const int32_t* __restrict__ plan, PlanLayout L,
                                                                 VJobArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RB = 4, NP = 2 * G;   // partial sums per (row, unit): input side and aggregate side of every gate
    const int c = blockIdx.z;
    const VJob& J = A.j[c];
    const int slot0 = A.r0[c] + blockIdx.x * RB;
    if (slot0 >= A.r1[c]) return;
    const int nrows = min(RB, A.r1[c] - slot0);
    const int tid = threadIdx.x;
    const int ul = tid & 63, unit = blockIdx.y * 64 + ul, kq = tid >> 6;
    const int din = (J.in_dim + 3) & ~3, dag = J.zero_agg ? 0 : ((J.agg_dim + 3) & ~3);
    float* s_in = lds;
    float* s_ag = lds + RB * din;
    float* s_red = s_ag + RB * dag;      // [4 quarters][RB][NP][64]
    __shared__ int s_node[RB];
    const int32_t* __restrict__ rec = plan + L.rowrec[J.dir] + 16 * (int64_t)slot0;
    if (tid < RB) s_node[tid] = tid < nrows ? rec[16 * tid] : -1;
    __syncthreads();
    for (int idx = tid; idx < RB * din; idx += 256) {
        const int r = idx / din, k = idx - r * din;
        const int v = s_node[r];
        s_in[idx] = (v >= 0 && k < J.in_dim) ? J.input[(int64_t)v * J.ld_in + k] : 0.f;
    }
    for (int idx = tid; idx < RB * dag; idx += 256) {
        const int r = idx / dag, k = idx - r * dag;
        const int v = s_node[r];
        s_ag[idx] = (v >= 0 && k < J.agg_dim) ? J.agg[(int64_t)v * J.ld_agg + k] : 0.f;
    }
    __syncthreads();
    const int uc = min(unit, J.out_dim - 1);   // lanes past the last unit compute a duplicate and do not store
    const int ldw = G * J.out_dim;
    float acc[RB][NP];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) acc[r][q] = 0.f;
#pragma unroll 2
    for (int k = 4 * kq; k < din; k += 16) {
        float w[4][G];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < G; ++q) w[kk][q] = J.w_in_t[(int64_t)min(k + kk, J.in_dim - 1) * ldw + q * J.out_dim + uc];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(s_in + r * din + k);
#pragma unroll
            for (int q = 0; q < G; ++q) acc[r][q] += x.x * w[0][q] + x.y * w[1][q] + x.z * w[2][q] + x.w * w[3][q];
        }
    }
#pragma unroll 2
    for (int k = 4 * kq; k < dag; k += 16) {
        float w[4][G];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < G; ++q) w[kk][q] = J.w_agg_t[(int64_t)min(k + kk, J.agg_dim - 1) * ldw + q * J.out_dim + uc];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(s_ag + r * dag + k);
#pragma unroll
            for (int q = 0; q < G; ++q) acc[r][G + q] += x.x * w[0][q] + x.y * w[1][q] + x.z * w[2][q] + x.w * w[3][q];
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) s_red[((kq * RB + r) * NP + q) * 64 + ul] = acc[r][q];
    __syncthreads();
    const int row = kq;
    const int v = s_node[row];
    if (v < 0 || unit >= J.out_dim) return;
    float p[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q)
        p[q] = (s_red[((0 * RB + row) * NP + q) * 64 + ul] + s_red[((1 * RB + row) * NP + q) * 64 + ul]) +
               (s_red[((2 * RB + row) * NP + q) * 64 + ul] + s_red[((3 * RB + row) * NP + q) * 64 + ul]);
    float res;
    if constexpr (G == 3) {
        const int H = J.out_dim;
        const float rr = sigmoidf_(p[0] + J.b_in[unit] + p[3] + J.b_agg[unit]);
        const float zz = sigmoidf_(p[1] + J.b_in[H + unit] + p[4] + J.b_agg[H + unit]);
        const float nn = tanhf(p[2] + J.b_in[2 * H + unit] + rr * (p[5] + J.b_agg[2 * H + unit]));
        const float hp = (dag && unit < J.agg_dim) ? s_ag[row * dag + unit] : 0.f;
        res = (1.f - zz) * nn + zz * hp;
    } else {
        res = p[0] + p[1] + (J.b_in ? J.b_in[unit] : 0.f);
        if (J.b_vid) res += J.b_vid[(int64_t)(v % J.vid_mod) * J.out_dim + unit];
    }
    J.out[(int64_t)v * J.ld_out + unit] = res;
}

int kpl_of(int dv) { return dv <= 64 ? 1 : dv <= 128 ? 2 : dv <= 256 ? 4 : dv <= 512 ? 8 : dv <= 1024 ? 16 : 0; }

int check_agg(const dagnn_variant_aggregator& g, int R) {
    if (g.mode < DAGNN_AGG_ATTN || g.mode > DAGNN_AGG_GIVEN) return DAGNN_EINVAL;
    if (!g.out || g.out_dim <= 0 || g.ld_out < g.out_dim) return DAGNN_EINVAL;
    if (g.mode == DAGNN_AGG_GIVEN || !g.lands) return DAGNN_OK;
    if (g.val_dim <= 0 || g.val_dim > g.out_dim || !kpl_of(g.val_dim)) return DAGNN_EINVAL;
    const bool needs_vals = g.mode != DAGNN_AGG_GATED;
    if (needs_vals && (!g.vals || g.ld_vals < g.val_dim)) return DAGNN_EINVAL;
    if (g.mode == DAGNN_AGG_ATTN && (!g.node0 || !g.edge_vec0 || g.aux_dim <= 0 || g.ld_node < g.aux_dim)) return DAGNN_EINVAL;
    if (g.mode == DAGNN_AGG_MATTN && (!g.node0 || !g.node1 || g.aux_dim <= 0 || g.ld_node < g.aux_dim)) return DAGNN_EINVAL;
    if (g.mode == DAGNN_AGG_GATED && (!g.node0 || !g.node1 || g.ld_node < g.val_dim)) return DAGNN_EINVAL;
    if (R == 0 && (g.edge_mat0 || g.edge_mat1)) return DAGNN_EINVAL;   // the plan carries no edge features
    return DAGNN_OK;
}

int launch_aggregate(const int32_t* plan, const PlanLayout& L, const VAggArgs& A, int n, int max_rows, int max_dv,
                     hipStream_t st) {
    if (n == 0 || max_rows <= 0) return DAGNN_OK;
    const dim3 grid((unsigned)((max_rows + 3) / 4), (unsigned)n);
    switch (kpl_of(max_dv)) {
        case 1: hipLaunchKernelGGL(variant_aggregate_kernel<1>, grid, dim3(256), 0, st, plan, L, A); break;
        case 2: hipLaunchKernelGGL(variant_aggregate_kernel<2>, grid, dim3(256), 0, st, plan, L, A); break;
        case 4: hipLaunchKernelGGL(variant_aggregate_kernel<4>, grid, dim3(256), 0, st, plan, L, A); break;
        case 8: hipLaunchKernelGGL(variant_aggregate_kernel<8>, grid, dim3(256), 0, st, plan, L, A); break;
        case 16: hipLaunchKernelGGL(variant_aggregate_kernel<16>, grid, dim3(256), 0, st, plan, L, A); break;
        default: return DAGNN_EINVAL;
    }
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

template <int G>
int launch_jobs(const int32_t* plan, const PlanLayout& L, const VJobArgs& A, int n, int max_rows, int max_out,
                int max_lds_floats_per_row, hipStream_t st) {
    if (n == 0 || max_rows <= 0) return DAGNN_OK;
    if (max_rows <= 64) {   // thin launch: K split over the waves
        const size_t lds = ((size_t)4 * max_lds_floats_per_row + 4 * 4 * 2 * G * 64) * sizeof(float);
        if (lds > 140 * 1024) return DAGNN_EINVAL;
        if (lds > 65536 - 256 && hipFuncSetAttribute(reinterpret_cast<const void*>(variant_cell_thin_kernel<G>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DAGNN_EHIP(hipGetLastError());
        const dim3 grid((unsigned)((max_rows + 3) / 4), (unsigned)((max_out + 63) / 64), (unsigned)n);
        hipLaunchKernelGGL((variant_cell_thin_kernel<G>), grid, dim3(256), lds, st, plan, L, A);
        DAGNN_CHECK_LAUNCH();
        return DAGNN_OK;
    }
    // 16 rows per workgroup while their operands fit the default 64 KB of LDS (next to the static node list), 8 rows
    // otherwise
    constexpr size_t kDefaultLds = 65536 - 256;
    const bool small = (size_t)16 * max_lds_floats_per_row * sizeof(float) > kDefaultLds;
    const int RB = small ? 8 : 16;
    const size_t lds = (size_t)RB * max_lds_floats_per_row * sizeof(float);
    if (lds > 140 * 1024) return DAGNN_EINVAL;
    const dim3 grid((unsigned)((max_rows + RB - 1) / RB), (unsigned)((max_out + 63) / 64), (unsigned)n);
    if (small) {
        if (lds > kDefaultLds && hipFuncSetAttribute(reinterpret_cast<const void*>(variant_cell_kernel<G, 2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DAGNN_EHIP(hipGetLastError());
        hipLaunchKernelGGL((variant_cell_kernel<G, 2>), grid, dim3(256), lds, st, plan, L, A);
    } else {
        hipLaunchKernelGGL((variant_cell_kernel<G, 4>), grid, dim3(256), lds, st, plan, L, A);
    }
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

}  // namespace

extern "C" int dagnn_variant_aggregate(const dagnn_plan* pl, const dagnn_variant_aggregator* agg, int dir,
                                       int32_t slot_begin, int32_t slot_end, void* stream) {
    if (!pl || !agg || dir < 0 || dir >= DAGNN_MAX_DIRS || slot_begin < 0 || slot_end < slot_begin || slot_end > pl->N)
        return DAGNN_EINVAL;
    if (agg->mode == DAGNN_AGG_GIVEN) return DAGNN_EINVAL;
    if (int rc = check_agg(*agg, pl->num_edge_feats)) return rc;
    if (slot_end == slot_begin) return DAGNN_OK;
    if (!pl->data) return DAGNN_EINVAL;
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    VAggArgs A;
    A.a[0] = *agg;
    A.dir[0] = dir; A.r0[0] = slot_begin; A.r1[0] = slot_end;
    A.R = pl->num_edge_feats;
    return launch_aggregate((const int32_t*)pl->data, L, A, 1, slot_end - slot_begin, agg->lands ? agg->val_dim : 1,
                            (hipStream_t)stream);
}

extern "C" int dagnn_variant_run(const dagnn_plan* pl, const dagnn_variant_args* a, const int32_t* const* layer_ptr,
                                 const int32_t* num_layers, void* stream) {
    if (!pl || !a || !layer_ptr || !num_layers) return DAGNN_EINVAL;
    const int Ls = a->num_stacked, H = a->H;
    if (Ls <= 0 || Ls > DAGNN_MAX_STACKED || H <= 0 || !(a->dir_mask & 3)) return DAGNN_EINVAL;
    if (pl->N == 0) return DAGNN_OK;
    if (!pl->data) return DAGNN_EINVAL;
    const int R = pl->num_edge_feats;
    int maxT = 0;
    for (int d = 0; d < DAGNN_MAX_DIRS; ++d) {
        if (!(a->dir_mask >> d & 1)) continue;
        if (!layer_ptr[d] || num_layers[d] < 0) return DAGNN_EINVAL;
        maxT = num_layers[d] > maxT ? num_layers[d] : maxT;
        for (int i = 0; i < Ls; ++i) {
            const dagnn_variant_cell& c = a->cell[d][i];
            if (int rc = check_agg(c.agg, R)) return rc;
            if (c.agg.out_dim != H && c.agg.mode != DAGNN_AGG_GIVEN) return DAGNN_EINVAL;
            if (c.agg.out_dim < H) return DAGNN_EINVAL;
            if (!c.input || c.in_dim <= 0 || c.ld_input < c.in_dim || !c.w_in_t || !c.w_agg_t || !c.h || c.ld_h < H)
                return DAGNN_EINVAL;
            if (c.recurrent && (!c.b_in || !c.b_agg)) return DAGNN_EINVAL;
            if (c.num_maps < 0 || c.num_maps > 3) return DAGNN_EINVAL;
            for (int m = 0; m < c.num_maps; ++m)
                if (!c.map[m].w_t || !c.map[m].out || c.map[m].out_dim <= 0 || c.map[m].ld_out < c.map[m].out_dim ||
                    (c.map[m].vid_mod > 0 && !c.map[m].vid_bias))
                    return DAGNN_EINVAL;
        }
    }
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, R);
    const int32_t* plan = (const int32_t*)pl->data;
    hipStream_t st = (hipStream_t)stream;
    const int recurrent = a->cell[(a->dir_mask & 1) ? 0 : 1][0].recurrent;
    for (int s = 0; s < maxT + Ls - 1; ++s) {
        VAggArgs AG;
        AG.R = R;
        VJobArgs CJ, MJ;
        int nag = 0, ncj = 0, nmj = 0, ag_rows = 0, ag_dv = 1, cj_rows = 0, cj_lds = 0, mj_rows = 0, mj_out = 0, mj_lds = 0;
        int rc = DAGNN_OK;
        auto flush_maps = [&]() {
            rc = launch_jobs<1>(plan, L, MJ, nmj, mj_rows, mj_out, mj_lds, st);
            nmj = 0; mj_rows = 0; mj_out = 0; mj_lds = 0;
            return rc;
        };
        for (int d = 0; d < DAGNN_MAX_DIRS; ++d) {
            if (!(a->dir_mask >> d & 1)) continue;
            for (int i = 0; i < Ls; ++i) {
                const int t = s - i;
                if (t < 0 || t >= num_layers[d]) continue;
                const dagnn_variant_cell& c = a->cell[d][i];
                if (c.recurrent != recurrent) return DAGNN_EINVAL;
                const int r0 = layer_ptr[d][t], r1 = layer_ptr[d][t + 1];
                if (r1 <= r0) continue;
                const bool zero_agg = c.agg.mode != DAGNN_AGG_GIVEN && (t == 0 || !c.agg.lands);
                if (!zero_agg && c.agg.mode != DAGNN_AGG_GIVEN) {
                    AG.a[nag] = c.agg;
                    AG.dir[nag] = d; AG.r0[nag] = r0; AG.r1[nag] = r1;
                    ++nag;
                    ag_rows = r1 - r0 > ag_rows ? r1 - r0 : ag_rows;
                    ag_dv = c.agg.val_dim > ag_dv ? c.agg.val_dim : ag_dv;
                }
                VJob& J = CJ.j[ncj];
                J.dir = d; J.in_dim = c.in_dim; J.out_dim = H; J.zero_agg = zero_agg ? 1 : 0;
                J.input = c.input; J.ld_in = c.ld_input;
                J.agg = c.agg.out; J.ld_agg = c.agg.ld_out; J.agg_dim = H; J.pad = 0;
                J.w_in_t = c.w_in_t; J.w_agg_t = c.w_agg_t; J.b_in = c.b_in; J.b_agg = c.b_agg;
                J.b_vid = nullptr; J.vid_mod = 1;
                J.out = c.h; J.ld_out = c.ld_h;
                CJ.r0[ncj] = r0; CJ.r1[ncj] = r1;
                ++ncj;
                cj_rows = r1 - r0 > cj_rows ? r1 - r0 : cj_rows;
                const int fl = ((c.in_dim + 3) & ~3) + ((H + 3) & ~3);
                cj_lds = fl > cj_lds ? fl : cj_lds;
            }
        }
        if ((rc = launch_aggregate(plan, L, AG, nag, ag_rows, ag_dv, st))) return rc;
        if (recurrent) rc = launch_jobs<3>(plan, L, CJ, ncj, cj_rows, H, cj_lds, st);
        else rc = launch_jobs<1>(plan, L, CJ, ncj, cj_rows, H, cj_lds, st);
        if (rc) return rc;
        // per-node projections of the rows just produced
        for (int d = 0; d < DAGNN_MAX_DIRS; ++d) {
            if (!(a->dir_mask >> d & 1)) continue;
            for (int i = 0; i < Ls; ++i) {
                const int t = s - i;
                if (t < 0 || t >= num_layers[d]) continue;
                const dagnn_variant_cell& c = a->cell[d][i];
                const int r0 = layer_ptr[d][t], r1 = layer_ptr[d][t + 1];
                if (r1 <= r0) continue;
                for (int m = 0; m < c.num_maps; ++m) {
                    VJob& J = MJ.j[nmj];
                    J.dir = d; J.in_dim = H; J.out_dim = c.map[m].out_dim; J.zero_agg = 1;
                    J.input = c.h; J.ld_in = c.ld_h;
                    J.agg = nullptr; J.ld_agg = 0; J.agg_dim = 0; J.pad = 0;
                    J.w_in_t = c.map[m].w_t; J.w_agg_t = nullptr; J.b_in = c.map[m].bias; J.b_agg = nullptr;
                    J.b_vid = c.map[m].vid_mod > 0 ? c.map[m].vid_bias : nullptr; J.vid_mod = c.map[m].vid_mod > 0 ? c.map[m].vid_mod : 1;
                    J.out = c.map[m].out; J.ld_out = c.map[m].ld_out;
                    MJ.r0[nmj] = r0; MJ.r1[nmj] = r1;
                    ++nmj;
                    mj_rows = r1 - r0 > mj_rows ? r1 - r0 : mj_rows;
                    mj_out = J.out_dim > mj_out ? J.out_dim : mj_out;
                    const int fl = (H + 3) & ~3;
                    mj_lds = fl > mj_lds ? fl : mj_lds;
                    if (nmj == VMAXC && flush_maps()) return rc;
                }
            }
        }
        if (flush_maps()) return rc;
    }
    return DAGNN_OK;
}
