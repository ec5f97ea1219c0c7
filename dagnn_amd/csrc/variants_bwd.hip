// variants_bwd.hip - reverse-mode sweep for the constructor-string aggregators `gated_sum`, `mattn_h`, `add` and `max`
// (SURVEY.md section 8 a12 / f3): what `loss.backward()` (ogbg-code/main_pyg.py:62) does to the loops of
// ogbg-code/model/dagnn.py:144-182 when `agg` selects GatedSumConv (dagnn.py:254-276), MultAttnConv (:379-409) or the
// additive AggConv (:232-251) with GRU cells.  The forward of these variants is the generic lock-step pass of variants.hip;
// this is its mirror: T + L - 1 reverse steps, stacked layer i handling layer t = T - 1 - (s - (L - 1 - i)) in step s
// (the top stacked layer leads), a handful of generic launches per step, every operand indexed by node id.
//
// Per (cell, node v), with the successors w of v in the cell's direction (= the rows of v in the OTHER direction's CSR):
//   g_v      gradient wrt the state h_v: starts as the gradient from outside (read-out), receives the upper cell's input
//            (and, mattn, query) gradient one step earlier, and here the aggregator's pull:
//     gated_sum  message_e = sigma(P_v + gamma_e) (.) (M_v + mu_e),  P = W_g h + b_g,  M = W_m h (+ b_m) per NODE,
//                gamma_e / mu_e = the edge-encoder terms (the [dim, R] products of variants.py):
//                dP_v = sum_e da_w (.) m_e (.) g_e (1 - g_e),  dM_v = sum_e da_w (.) g_e,  g_v += [dP_v ; dM_v] [W_g ; W_m]
//     mattn      logit_e = Ql_w . (Kr_v + rho_e),  alpha = segment softmax over the in-edges of w,  a_w = sum alpha_e h_v:
//                g_v += sum_e alpha_e da_w + (sum_e dlogit_e Ql_w) W_r,   dlogit_e = alpha_e (da_w . h_v - da_w . a_w)
//                (dlogit_e was stored by edge id when w was processed - a later step of the forward is an earlier one here)
//     add        g_v += sum_e da_w            max: only where v's message attained the maximum a_w[k]
//   `recurr=0` (the Linear cell h = W [u ; a] + b, dagnn.py:83-85): no gate algebra, da_v = g_v W[:, in:], du_v = g_v W[:, :in]
//   GRU backward (gates recomputed from gi, gh):  dgi, dgh,  da_v = z (.) g_v + dgh_v W_hh,  du_v = dgi_v W_ih -> gradient
//            of the cell's input (the state one stacked layer down, or the node input x)
//   mattn, target side: dlogit_e for the in-edges of v, dQl_v = sum_e dlogit_e (Kr_j + rho_e), dq_v = dQl_v W_l -> the
//            same input gradient (the query of stacked layer i is the state of layer i - 1, dagnn.py:175-177).
// All matrix work is ONE generic kernel (`vb_map_kernel`: out[rows] += in[rows] W, weights in their torch layouts, which
// are [K][J] for every product of a reverse pass).  No atomics, rows of a step are distinct nodes: deterministic.
// Outputs for the parallel epilogue (weight gradients = transposed products over all nodes, edge-encoder gradients from
// the per-node edge-feature sums): dgi, dgh, the projection gradients (dP|dM, dKr, dQl) and `esum`.
#include "common.h"

namespace {

constexpr int VBC = 8;   // cells per launch (the step table travels as a kernel argument: 8 x ~270 bytes)

__device__ __forceinline__ float vb_sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float vb_edge_term(const float* __restrict__ m, const float* __restrict__ v, int k, int R,
                                              const float* __restrict__ attr) {
    float e = v ? v[k] : 0.f;
    if (m)
        for (int r = 0; r < R; ++r) e += m[(int64_t)k * R + r] * attr[r];
    return e;
}

struct VbStep {
    dagnn_variant_bwd_cell c[VBC];
    int dir[VBC], r0[VBC], r1[VBC];
    int n, R, H;
};

// ---- pull: one wave per frontier row
__global__ void __launch_bounds__(256) vb_pull_kernel(const int32_t* __restrict__ plan, PlanLayout L, VbStep S) {
    const int ci = blockIdx.y;
    const dagnn_variant_bwd_cell& C = S.c[ci];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = S.r0[ci] + blockIdx.x * 4 + wave;
    if (slot >= S.r1[ci]) return;
    const int d = S.dir[ci], od = 1 - d, R = S.R, H = S.H;
    const int v = plan[L.rowrec[d] + 16 * (int64_t)slot];
    const int32_t* __restrict__ orec = plan + L.rowrec[od] + 16 * (int64_t)plan[L.pos[od] + v];
    const int eb = orec[1], ee = orec[2];
    const int32_t* __restrict__ col = plan + L.col[od];
    const int32_t* __restrict__ eidx = plan + L.eidx[od];
    const float* __restrict__ ea = reinterpret_cast<const float*>(plan + L.eattr[od]);
    float* __restrict__ g = C.g + (int64_t)v * H;
    if (C.mode == DAGNN_AGG_GIVEN) return;   // `agg_x`: the aggregate is an input of the recurrence - nothing to pull
    if (C.mode == DAGNN_AGG_GATED) {
        const float* __restrict__ pq = C.node0 + (int64_t)v * 2 * H;
        float* __restrict__ dpq = C.dnode0 + (int64_t)v * 2 * H;
        const bool es = C.esum != nullptr && R > 0 && R <= 2;
        for (int k = lane; k < H; k += 64) {
            const float P = pq[k], M = pq[H + k];
            float aP = 0.f, aM = 0.f, eg[2] = {0.f, 0.f}, em[2] = {0.f, 0.f};
            for (int e = eb; e < ee; ++e) {
                const float* attr = ea + (int64_t)e * R;
                const float gt = vb_sigm(P + vb_edge_term(C.edge_mat0, C.edge_vec0, k, R, attr));
                const float mp = M + vb_edge_term(C.edge_mat1, C.edge_vec1, k, R, attr);
                const float dw = C.da[(int64_t)col[e] * H + k];
                const float dg = dw * mp * gt * (1.0f - gt), dm = dw * gt;
                aP += dg; aM += dm;
                if (es)
                    for (int r = 0; r < R; ++r) { eg[r] = fmaf(dg, attr[r], eg[r]); em[r] = fmaf(dm, attr[r], em[r]); }
            }
            dpq[k] = aP; dpq[H + k] = aM;
            if (es) {
                float* o = C.esum + (int64_t)v * (2 * R * H);
                for (int r = 0; r < R; ++r) { o[r * H + k] = eg[r]; o[(R + r) * H + k] = em[r]; }
            }
        }
    } else if (C.mode == DAGNN_AGG_ADD || C.mode == DAGNN_AGG_MAX) {
        if (!C.lands) return;   // the reference's shared AggConv: in this direction the messages land elsewhere (no gradient)
        const bool es = C.esum != nullptr && R > 0 && R <= 2;
        const bool is_max = C.mode == DAGNN_AGG_MAX;
        for (int k = lane; k < H; k += 64) {
            float acc = 0.f, eg[2] = {0.f, 0.f};
            const float hv = C.h[(int64_t)v * H + k];
            for (int e = eb; e < ee; ++e) {
                float dw = C.da[(int64_t)col[e] * H + k];
                // max: the gradient of a_w[k] goes to the message that attained it - recomputed with the forward's own
                // operations (variants.hip: vals + edge term), so the comparison with the stored maximum is exact
                if (is_max && hv + vb_edge_term(C.edge_mat0, C.edge_vec0, k, R, ea + (int64_t)e * R) != C.a[(int64_t)col[e] * H + k])
                    dw = 0.f;
                acc += dw;
                if (es)
                    for (int r = 0; r < R; ++r) eg[r] = fmaf(dw, ea[(int64_t)e * R + r], eg[r]);
            }
            g[k] += acc;
            if (es) {
                float* o = C.esum + (int64_t)v * ((R + 1) * H);
                for (int r = 0; r < R; ++r) o[r * H + k] = eg[r];
                o[R * H + k] = acc;   // sum of the incoming gradients over the out-edges: the edge-encoder bias' share
            }
        }
    } else if (C.mode == DAGNN_AGG_ATTN) {
        // additive attention (AttnConv / SelfAttnConv, dagnn.py:279-313,347-376): logit_e = w_k . key_v + gain . attr_e (+ terms
        // constant inside a soft-max segment), a_w = sum alpha_e h_v.  g_v += sum_e alpha_e da_w + sigma_v w_k (keys = states),
        // sigma_v = sum_e ds_e, ds_e = alpha_e (da_w . h_v - da_w . a_w); outputs sigma_v and sum_e ds_e attr_e
        const float* __restrict__ hv = C.h + (int64_t)v * H;
        float sig = 0.f, m[2] = {0.f, 0.f};
        float acc[4] = {0.f, 0.f, 0.f, 0.f};   // H <= 256 per lane chunk; wider rows loop below
        const bool narrow = H <= 256;
        for (int e = eb; e < ee; ++e) {
            const int w = col[e];
            const float al = C.alpha[eidx[e]];
            const float* __restrict__ daw = C.da + (int64_t)w * H;
            const float* __restrict__ aw = C.a + (int64_t)w * H;
            float dot = 0.f;
            for (int k = lane, c = 0; k < H; k += 64, ++c) {
                const float dv = daw[k];
                dot = fmaf(dv, hv[k] - aw[k], dot);
                if (narrow) acc[c] = fmaf(al, dv, acc[c]);
                else g[k] = fmaf(al, dv, g[k]);
            }
            const float ds = al * wave_sum(dot);
            sig += ds;
            for (int r = 0; r < R && r < 2; ++r) m[r] = fmaf(ds, ea[(int64_t)e * R + r], m[r]);
        }
        const bool key_state = C.reserved != 0;
        for (int k = lane, c = 0; k < H; k += 64, ++c) {
            float t = narrow ? g[k] + acc[c] : g[k];
            if (key_state) t = fmaf(sig, C.w_node[k], t);
            g[k] = t;
        }
        if (lane == 0) {
            C.dnode0[v] = sig;
            if (C.esum) for (int r = 0; r < R && r < 2; ++r) C.esum[(int64_t)v * R + r] = m[r];
        }
    } else {   // DAGNN_AGG_MATTN
        const int P = C.proj_dim;
        float* __restrict__ dkr = C.dnode0 + (int64_t)v * P;
        // direct path: sum_e alpha_e da_w
        for (int k = lane; k < H; k += 64) {
            float acc = 0.f;
            for (int e = eb; e < ee; ++e) acc = fmaf(C.alpha[eidx[e]], C.da[(int64_t)col[e] * H + k], acc);
            g[k] += acc;
        }
        // key path: dKr_v = sum_e dlogit_e Ql_w (+ the edge-feature sums of dlogit_e Ql_w)
        const bool es = C.esum != nullptr && R > 0 && R <= 2;
        for (int k = lane; k < P; k += 64) {
            float acc = 0.f, er[2] = {0.f, 0.f};
            for (int e = eb; e < ee; ++e) {
                const float t = C.dlogit[eidx[e]] * C.node1[(int64_t)col[e] * P + k];
                acc += t;
                if (es)
                    for (int r = 0; r < R; ++r) er[r] = fmaf(t, ea[(int64_t)e * R + r], er[r]);
            }
            dkr[k] = acc;
            if (es) {
                float* o = C.esum + (int64_t)v * (R * P);
                for (int r = 0; r < R; ++r) o[r * P + k] = er[r];
            }
        }
    }
}

// ---- GRU backward of the frontier rows (elementwise): g -> dgi, dgh, da = z (.) g
__global__ void __launch_bounds__(256) vb_gru_kernel(const int32_t* __restrict__ plan, PlanLayout L, VbStep S) {
    const int ci = blockIdx.y;
    const dagnn_variant_bwd_cell& C = S.c[ci];
    if (!C.recurrent) return;   // Linear cell: its input gradients are plain maps of g
    const int H = S.H, H3 = 3 * H;
    const int rows = S.r1[ci] - S.r0[ci];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)rows * H; idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / H), k = (int)(idx - (int64_t)r * H);
        const int v = plan[L.rowrec[S.dir[ci]] + 16 * (int64_t)(S.r0[ci] + r)];
        const float* gi = C.gi + (int64_t)v * H3;
        const float* gh = C.gh + (int64_t)v * H3;
        const float ghn = gh[2 * H + k];
        const float rr = vb_sigm(gi[k] + gh[k]), zz = vb_sigm(gi[H + k] + gh[H + k]);
        const float nn = tanhf(gi[2 * H + k] + rr * ghn);
        const float G = C.g[(int64_t)v * H + k], av = C.a[(int64_t)v * H + k];
        const float dn = G * (1.0f - zz) * (1.0f - nn * nn);
        const float dz = G * (av - nn) * zz * (1.0f - zz);
        const float dr = dn * ghn * rr * (1.0f - rr);
        float* og = C.dgi + (int64_t)v * H3;
        float* oh = C.dgh + (int64_t)v * H3;
        og[k] = dr; og[H + k] = dz; og[2 * H + k] = dn;
        oh[k] = dr; oh[H + k] = dz; oh[2 * H + k] = dn * rr;
        C.da[(int64_t)v * H + k] = G * zz;
    }
}

// ---- generic rows map: out[v, 0:J] += in[v, 0:K] W[K][J] for the frontier rows of a job
struct VbMapJob { const float* in; const float* W; float* out; int K, J, dir, r0, r1; };
struct VbMaps { VbMapJob j[VBC]; int n; };
constexpr int VB_RT = 8;   // rows per workgroup

__global__ void __launch_bounds__(256) vb_map_kernel(const int32_t* __restrict__ plan, PlanLayout L, VbMaps M, int kchunk) {
    extern __shared__ float s_in[];   // [VB_RT][kchunk]
    __shared__ int s_node[VB_RT];
    const VbMapJob& Jb = M.j[blockIdx.z];
    const int slot0 = Jb.r0 + blockIdx.x * VB_RT;
    if (slot0 >= Jb.r1) return;
    const int nrows = min(VB_RT, Jb.r1 - slot0);
    const int tid = threadIdx.x, j = blockIdx.y * 256 + tid;
    if (tid < VB_RT) s_node[tid] = tid < nrows ? plan[L.rowrec[Jb.dir] + 16 * (int64_t)(slot0 + tid)] : -1;
    float acc[VB_RT];
#pragma unroll
    for (int r = 0; r < VB_RT; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < Jb.K; k0 += kchunk) {
        const int kc = min(kchunk, Jb.K - k0);
        __syncthreads();
        for (int idx = tid; idx < VB_RT * kc; idx += 256) {
            const int r = idx / kc, k = idx - r * kc;
            const int v = s_node[r];
            s_in[r * kchunk + k] = v >= 0 ? Jb.in[(int64_t)v * Jb.K + k0 + k] : 0.f;
        }
        __syncthreads();
        if (j < Jb.J) {
            const float* w = Jb.W + (int64_t)k0 * Jb.J + j;
            for (int k = 0; k < kc; ++k) {
                const float wv = w[(int64_t)k * Jb.J];
#pragma unroll
                for (int r = 0; r < VB_RT; ++r) acc[r] = fmaf(s_in[r * kchunk + k], wv, acc[r]);
            }
        }
    }
    if (j < Jb.J) {
#pragma unroll
        for (int r = 0; r < VB_RT; ++r) {
            const int v = s_node[r];
            if (v >= 0) Jb.out[(int64_t)v * Jb.J + j] += acc[r];
        }
    }
}

// ---- mattn, target side: one wave per frontier row v: dlogit of its in-edges (by edge id), dQl_v
__global__ void __launch_bounds__(256) vb_mattn_target_kernel(const int32_t* __restrict__ plan, PlanLayout L, VbStep S) {
    const int ci = blockIdx.y;
    const dagnn_variant_bwd_cell& C = S.c[ci];
    if (C.mode != DAGNN_AGG_MATTN) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = S.r0[ci] + blockIdx.x * 4 + wave;
    if (slot >= S.r1[ci]) return;
    const int d = S.dir[ci], R = S.R, H = S.H, P = C.proj_dim;
    const int32_t* __restrict__ rec = plan + L.rowrec[d] + 16 * (int64_t)slot;
    const int v = rec[0], eb = rec[1], ee = rec[2];
    const int32_t* __restrict__ col = plan + L.col[d];
    const int32_t* __restrict__ eidx = plan + L.eidx[d];
    const float* __restrict__ ea = reinterpret_cast<const float*>(plan + L.eattr[d]);
    const float* __restrict__ dav = C.da + (int64_t)v * H;
    const float* __restrict__ av = C.a + (int64_t)v * H;
    float* __restrict__ dql = C.dnode1 + (int64_t)v * P;
    float q = 0.f;   // da_v . a_v
    for (int k = lane; k < H; k += 64) q = fmaf(dav[k], av[k], q);
    q = wave_sum(q);
    for (int k = lane; k < P; k += 64) dql[k] = 0.f;
    for (int e = eb; e < ee; ++e) {
        const int j = col[e];
        const float* hj = C.h + (int64_t)j * H;
        float t = 0.f;
        for (int k = lane; k < H; k += 64) t = fmaf(dav[k], hj[k], t);
        t = wave_sum(t);
        const float dl = C.alpha[eidx[e]] * (t - q);
        if (lane == 0) C.dlogit[eidx[e]] = dl;
        const float* attr = ea + (int64_t)e * R;
        for (int k = lane; k < P; k += 64)
            dql[k] = fmaf(dl, C.node0[(int64_t)j * P + k] + vb_edge_term(C.edge_mat0, C.edge_vec0, k, R, attr), dql[k]);
    }
}

// ---- mattn, preparation: alpha of every edge (by edge id) and the aggregates a, all rows of one direction in one launch
__global__ void __launch_bounds__(256) vb_mattn_prepare_kernel(const int32_t* __restrict__ plan, PlanLayout L,
                                                                dagnn_variant_bwd_cell C, int dir, int r0, int r1, int R, int H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = r0 + blockIdx.x * 4 + wave;
    if (slot >= r1) return;
    const int P = C.proj_dim;
    const int32_t* __restrict__ rec = plan + L.rowrec[dir] + 16 * (int64_t)slot;
    const int v = rec[0], eb = rec[1], ee = rec[2];
    const int32_t* __restrict__ col = plan + L.col[dir];
    const int32_t* __restrict__ eidx = plan + L.eidx[dir];
    const float* __restrict__ ea = reinterpret_cast<const float*>(plan + L.eattr[dir]);
    float* __restrict__ alpha = const_cast<float*>(C.alpha);
    float* __restrict__ aout = const_cast<float*>(C.a) + (int64_t)v * H;
    const float* __restrict__ ql = C.node1 + (int64_t)v * P;
    auto logit = [&](int e) {
        const int j = col[e];
        const float* attr = ea + (int64_t)e * R;
        float t = 0.f;
        for (int k = lane; k < P; k += 64)
            t = fmaf(ql[k], C.node0[(int64_t)j * P + k] + vb_edge_term(C.edge_mat0, C.edge_vec0, k, R, attr), t);
        return wave_sum(t);
    };
    float mx = -INFINITY;
    for (int e = eb; e < ee; ++e) mx = fmaxf(mx, logit(e));
    float sum = 0.f;
    for (int e = eb; e < ee; ++e) sum += expf(logit(e) - mx);
    const float den = sum + 1e-16f;   // PyG softmax
    for (int k = lane; k < H; k += 64) aout[k] = 0.f;
    for (int e = eb; e < ee; ++e) {
        const float al = expf(logit(e) - mx) / den;
        if (lane == 0) alpha[eidx[e]] = al;
        const float* hj = C.h + (int64_t)col[e] * H;
        for (int k = lane; k < H; k += 64) aout[k] = fmaf(al, hj[k], aout[k]);
    }
}

// ---- additive attention, preparation: alpha of every edge (by edge id) and the aggregates a
__global__ void __launch_bounds__(256) vb_attn_prepare_kernel(const int32_t* __restrict__ plan, PlanLayout L,
                                                               dagnn_variant_bwd_cell C, int dir, int r0, int r1, int R, int H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = r0 + blockIdx.x * 4 + wave;
    if (slot >= r1) return;
    const int kd = C.proj_dim;
    const int32_t* __restrict__ rec = plan + L.rowrec[dir] + 16 * (int64_t)slot;
    const int v = rec[0], eb = rec[1], ee = rec[2];
    const int32_t* __restrict__ col = plan + L.col[dir];
    const int32_t* __restrict__ eidx = plan + L.eidx[dir];
    const float* __restrict__ ea = reinterpret_cast<const float*>(plan + L.eattr[dir]);
    float* __restrict__ alpha = const_cast<float*>(C.alpha);
    float* __restrict__ aout = const_cast<float*>(C.a) + (int64_t)v * H;
    auto logit = [&](int e) {
        const float* key = C.node0 + (int64_t)col[e] * kd;
        float t = 0.f;
        for (int k = lane; k < kd; k += 64) t = fmaf(C.w_node[k], key[k], t);
        t = wave_sum(t);
        if (C.edge_mat0)
            for (int r = 0; r < R; ++r) t = fmaf(C.edge_mat0[r], ea[(int64_t)e * R + r], t);
        return t;
    };
    float mx = -INFINITY;
    for (int e = eb; e < ee; ++e) mx = fmaxf(mx, logit(e));
    float sum = 0.f;
    for (int e = eb; e < ee; ++e) sum += expf(logit(e) - mx);
    const float den = sum + 1e-16f;
    for (int k = lane; k < H; k += 64) aout[k] = 0.f;
    for (int e = eb; e < ee; ++e) {
        const float al = expf(logit(e) - mx) / den;
        if (lane == 0) alpha[eidx[e]] = al;
        const float* hj = C.h + (int64_t)col[e] * H;
        for (int k = lane; k < H; k += 64) aout[k] = fmaf(al, hj[k], aout[k]);
    }
}

int vb_launch_maps(const int32_t* plan, const PlanLayout& L, VbMaps& M, hipStream_t st) {
    if (M.n == 0) return DAGNN_OK;
    int rows = 0, J = 0, K = 0;
    for (int q = 0; q < M.n; ++q) {
        rows = M.j[q].r1 - M.j[q].r0 > rows ? M.j[q].r1 - M.j[q].r0 : rows;
        J = M.j[q].J > J ? M.j[q].J : J;
        K = M.j[q].K > K ? M.j[q].K : K;
    }
    const int kchunk = K < 1024 ? K : 1024;
    hipLaunchKernelGGL(vb_map_kernel, dim3((unsigned)((rows + VB_RT - 1) / VB_RT), (unsigned)((J + 255) / 256), (unsigned)M.n),
                       dim3(256), (size_t)VB_RT * kchunk * sizeof(float), st, plan, L, M, kchunk);
    DAGNN_CHECK_LAUNCH();
    M.n = 0;
    return DAGNN_OK;
}

}  // namespace

extern "C" int dagnn_variant_mattn_prepare(const dagnn_plan* pl, const dagnn_variant_bwd_cell* c, int dir, int H,
                                           int32_t row_begin, int32_t row_end, void* stream) {
    if (!pl || !pl->data || !c || (dir != 0 && dir != 1) || H <= 0 || row_begin < 0 || row_end > pl->N) return DAGNN_EINVAL;
    if (!c->h || !c->a || !c->alpha || !c->node0 || c->proj_dim <= 0) return DAGNN_EINVAL;
    if (c->mode == DAGNN_AGG_MATTN ? !c->node1 : !c->w_node) return DAGNN_EINVAL;
    if (row_end <= row_begin) return DAGNN_OK;
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    if (c->mode == DAGNN_AGG_ATTN) {
        hipLaunchKernelGGL(vb_attn_prepare_kernel, dim3((unsigned)((row_end - row_begin + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const int32_t*)pl->data, L, *c, dir, row_begin, row_end, pl->num_edge_feats, H);
        DAGNN_CHECK_LAUNCH();
        return DAGNN_OK;
    }
    hipLaunchKernelGGL(vb_mattn_prepare_kernel, dim3((unsigned)((row_end - row_begin + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)pl->data, L, *c, dir, row_begin, row_end, pl->num_edge_feats, H);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_variant_backward_run(const dagnn_plan* pl, const dagnn_variant_bwd_args* a, const int32_t* const* layer_ptr,
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
            const dagnn_variant_bwd_cell& c = a->cell[d][i];
            if (c.mode != DAGNN_AGG_GATED && c.mode != DAGNN_AGG_MATTN && c.mode != DAGNN_AGG_ADD && c.mode != DAGNN_AGG_MAX &&
                c.mode != DAGNN_AGG_ATTN && c.mode != DAGNN_AGG_GIVEN)
                return DAGNN_EINVAL;
            if (c.mode == DAGNN_AGG_ATTN && (!c.alpha || !c.dnode0 || !c.w_node)) return DAGNN_EINVAL;
            if (!c.h || !c.a || !c.w_hh || !c.w_ih || !c.g || !c.g_in || !c.da || c.in_dim <= 0) return DAGNN_EINVAL;
            if (c.recurrent && (!c.gi || !c.gh || !c.dgi || !c.dgh)) return DAGNN_EINVAL;
            if (c.mode == DAGNN_AGG_GATED && (!c.node0 || !c.dnode0 || !c.w_node)) return DAGNN_EINVAL;
            if (c.mode == DAGNN_AGG_MATTN && (!c.node0 || !c.node1 || !c.dnode0 || !c.dnode1 || !c.w_node || !c.w_query ||
                                              !c.alpha || !c.dlogit || c.proj_dim <= 0))
                return DAGNN_EINVAL;
        }
    }
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, R);
    const int32_t* plan = (const int32_t*)pl->data;
    hipStream_t st = (hipStream_t)stream;
    for (int s = 0; s < maxT + Ls - 1; ++s) {
        VbStep S;
        S.n = 0; S.R = R; S.H = H;
        int rows = 0;
        bool any_mattn = false;
        for (int d = 0; d < DAGNN_MAX_DIRS; ++d) {
            if (!(a->dir_mask >> d & 1)) continue;
            for (int i = 0; i < Ls; ++i) {
                const int t = num_layers[d] - 1 - (s - (Ls - 1 - i));
                if (t < 0 || t >= num_layers[d]) continue;
                const int r0 = layer_ptr[d][t], r1 = layer_ptr[d][t + 1];
                if (r1 <= r0) continue;
                if (S.n == VBC) return DAGNN_EINVAL;   // (directions x stacked layers <= 8)
                S.c[S.n] = a->cell[d][i];
                S.dir[S.n] = d; S.r0[S.n] = r0; S.r1[S.n] = r1;
                rows = r1 - r0 > rows ? r1 - r0 : rows;
                any_mattn = any_mattn || a->cell[d][i].mode == DAGNN_AGG_MATTN;
                ++S.n;
            }
        }
        if (S.n == 0) continue;
        const dim3 wgrid((unsigned)((rows + 3) / 4), (unsigned)S.n);
        // 1. pull from the successors
        hipLaunchKernelGGL(vb_pull_kernel, wgrid, dim3(256), 0, st, plan, L, S);
        DAGNN_CHECK_LAUNCH();
        // 2. projection gradients -> g
        VbMaps M;
        M.n = 0;
        for (int q = 0; q < S.n; ++q) {
            const dagnn_variant_bwd_cell& c = S.c[q];
            if (c.mode == DAGNN_AGG_GATED) M.j[M.n++] = VbMapJob{c.dnode0, c.w_node, c.g, 2 * H, H, S.dir[q], S.r0[q], S.r1[q]};
            else if (c.mode == DAGNN_AGG_MATTN) M.j[M.n++] = VbMapJob{c.dnode0, c.w_node, c.g, c.proj_dim, H, S.dir[q], S.r0[q], S.r1[q]};
        }
        if (int rc = vb_launch_maps(plan, L, M, st)) return rc;
        // 3. GRU backward
        int64_t elems = (int64_t)rows * H;
        hipLaunchKernelGGL(vb_gru_kernel, dim3((unsigned)((elems + 255) / 256 > 2048 ? 2048 : (elems + 255) / 256), (unsigned)S.n),
                           dim3(256), 0, st, plan, L, S);
        DAGNN_CHECK_LAUNCH();
        // 4. da += dgh W_hh   (Linear cell: da += g W[:, in_dim:])
        for (int q = 0; q < S.n; ++q)
            M.j[M.n++] = S.c[q].recurrent ? VbMapJob{S.c[q].dgh, S.c[q].w_hh, S.c[q].da, 3 * H, H, S.dir[q], S.r0[q], S.r1[q]}
                                          : VbMapJob{S.c[q].g, S.c[q].w_hh, S.c[q].da, H, H, S.dir[q], S.r0[q], S.r1[q]};
        if (int rc = vb_launch_maps(plan, L, M, st)) return rc;
        // 5. input gradient += dgi W_ih   (Linear cell: += g W[:, :in_dim])
        for (int q = 0; q < S.n; ++q)
            M.j[M.n++] = S.c[q].recurrent
                ? VbMapJob{S.c[q].dgi, S.c[q].w_ih, S.c[q].g_in, 3 * H, S.c[q].in_dim, S.dir[q], S.r0[q], S.r1[q]}
                : VbMapJob{S.c[q].g, S.c[q].w_ih, S.c[q].g_in, H, S.c[q].in_dim, S.dir[q], S.r0[q], S.r1[q]};
        if (int rc = vb_launch_maps(plan, L, M, st)) return rc;
        if (any_mattn) {
            // 6. dlogit of the in-edges, dQl; 7. input gradient += dQl W_l
            hipLaunchKernelGGL(vb_mattn_target_kernel, wgrid, dim3(256), 0, st, plan, L, S);
            DAGNN_CHECK_LAUNCH();
            for (int q = 0; q < S.n; ++q)
                if (S.c[q].mode == DAGNN_AGG_MATTN)
                    M.j[M.n++] = VbMapJob{S.c[q].dnode1, S.c[q].w_query, S.c[q].g_in, S.c[q].proj_dim, S.c[q].in_dim, S.dir[q],
                                          S.r0[q], S.r1[q]};
            if (int rc = vb_launch_maps(plan, L, M, st)) return rc;
        }
    }
    return DAGNN_OK;
}

// `agg_x` (dagnn.py:159-169): the aggregator reads the node inputs only, so its reverse pass is ONE shot over all rows after
// the cells' sweep: `cell` describes the aggregator with h = the inputs x [N, width], a = its output, da = the summed
// gradient of that output over the stacked cells, g = g_in = the gradient of x (accumulated).
extern "C" int dagnn_variant_aggregator_backward(const dagnn_plan* pl, const dagnn_variant_bwd_cell* c, int dir, int width,
                                                 int32_t row_begin, int32_t row_end, void* stream) {
    if (!pl || !pl->data || !c || (dir != 0 && dir != 1) || width <= 0 || row_begin < 0 || row_end > pl->N) return DAGNN_EINVAL;
    if (!c->h || !c->a || !c->da || !c->g || !c->g_in) return DAGNN_EINVAL;
    if (c->mode == DAGNN_AGG_GATED && (!c->node0 || !c->dnode0 || !c->w_node)) return DAGNN_EINVAL;
    if (c->mode == DAGNN_AGG_MATTN && (!c->node0 || !c->node1 || !c->dnode0 || !c->dnode1 || !c->w_node || !c->w_query || !c->alpha ||
                                       !c->dlogit || c->proj_dim <= 0))
        return DAGNN_EINVAL;
    if (c->mode == DAGNN_AGG_ATTN && (!c->alpha || !c->dnode0 || !c->w_node || !c->node0)) return DAGNN_EINVAL;
    if (row_end <= row_begin) return DAGNN_OK;
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    const int32_t* plan = (const int32_t*)pl->data;
    hipStream_t st = (hipStream_t)stream;
    VbStep S;
    S.n = 1; S.R = pl->num_edge_feats; S.H = width;
    S.c[0] = *c; S.dir[0] = dir; S.r0[0] = row_begin; S.r1[0] = row_end;
    const dim3 wgrid((unsigned)((row_end - row_begin + 3) / 4), 1);
    if (c->mode == DAGNN_AGG_MATTN) {   // dlogit of every edge first: the pull below reads it for the out-edges
        hipLaunchKernelGGL(vb_mattn_target_kernel, wgrid, dim3(256), 0, st, plan, L, S);
        DAGNN_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(vb_pull_kernel, wgrid, dim3(256), 0, st, plan, L, S);
    DAGNN_CHECK_LAUNCH();
    VbMaps M;
    M.n = 0;
    if (c->mode == DAGNN_AGG_GATED) M.j[M.n++] = VbMapJob{c->dnode0, c->w_node, c->g, 2 * width, width, dir, row_begin, row_end};
    else if (c->mode == DAGNN_AGG_MATTN) M.j[M.n++] = VbMapJob{c->dnode0, c->w_node, c->g, c->proj_dim, width, dir, row_begin, row_end};
    if (int rc = vb_launch_maps(plan, L, M, st)) return rc;
    if (c->mode == DAGNN_AGG_MATTN) {
        M.j[M.n++] = VbMapJob{c->dnode1, c->w_query, c->g_in, c->proj_dim, c->in_dim, dir, row_begin, row_end};
        if (int rc = vb_launch_maps(plan, L, M, st)) return rc;
    }
    return DAGNN_OK;
}
