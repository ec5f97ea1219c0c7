// misc.hip - HBM-bound row kernels around the recurrence: AST node encoder, weight packing,
// output-node max read-out and the D-VAE fixed-stride row gather.  One wave (64 lanes) moves one
// row; lanes read float4 so a 256-wide fp32 row is one coalesced 1 KiB access.
#include "common.h"

namespace {

// out[v,:] = type_emb[x[v,0]] + attr_emb[x[v,1]] + depth_emb[min(depth[v], max_depth)]
// (ogbg-code/utils.py:26-28).  depth is clamped in place like the reference does (:27).
__global__ void __launch_bounds__(256) encode_ast_kernel(const int64_t* __restrict__ x, int64_t* depth,
                                                          const float* __restrict__ type_emb,
                                                          const float* __restrict__ attr_emb,
                                                          const float* __restrict__ depth_emb, int max_depth,
                                                          float* __restrict__ out, int ld_out, int64_t N, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int H4 = H >> 2;
    for (int64_t v = wave; v < N; v += nwaves) {
        const int64_t t = x[2 * v], a = x[2 * v + 1];
        int64_t dp = depth[v];
        if (dp > max_depth) { dp = max_depth; if (lane == 0) depth[v] = dp; }
        const float4* pt = reinterpret_cast<const float4*>(type_emb + t * H);
        const float4* pa = reinterpret_cast<const float4*>(attr_emb + a * H);
        const float4* pd = reinterpret_cast<const float4*>(depth_emb + dp * H);
        float4* po = reinterpret_cast<float4*>(out + v * ld_out);
        for (int c = lane; c < H4; c += 64) {
            float4 u = pt[c], w = pa[c], z = pd[c], r;
            // same association as the reference: (type + attr) + depth
            r.x = (u.x + w.x) + z.x; r.y = (u.y + w.y) + z.y; r.z = (u.z + w.z) + z.z; r.w = (u.w + w.w) + z.w;
            po[c] = r;
        }
    }
}

// Wt[k, c] = W[c, k]: [3H, H] -> [H, 3H], 32x32 LDS tile transpose.
__global__ void __launch_bounds__(256) pack_whh_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                        int rows /*3H*/, int cols /*H*/) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int r = by + j, c = bx + tx;
        tile[j][tx] = (r < rows && c < cols) ? w[(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = bx + j, r = by + tx;  // output row = c (k), output col = r
        if (c < cols && r < rows) wt[(int64_t)c * rows + r] = tile[tx][j];
    }
}

// out[g, col_off + j] = max over output nodes v of graph g of h[v, j]  (dagnn.py:119-126,184-193).
// Output nodes of direction `dir` are the layer-0 frontier of the OPPOSITE direction, which the
// plan already holds as a contiguous range of order[1-dir].
__global__ void __launch_bounds__(256) readout_max_kernel(const int32_t* __restrict__ plan, PlanLayout L, int dir,
                                                           const float* __restrict__ h, int ld_h, int width,
                                                           float* __restrict__ out, int ld_out, int col_off) {
    const int g = blockIdx.x;
    const int od = 1 - dir;
    const int n0 = plan[L.node_ptr + g];
    const int depth = plan[L.depth[od] + g];
    const int32_t* ls = plan + L.lstart[od] + n0 + g;
    const int32_t* order = plan + L.order[od];
    const int p0 = depth > 0 ? ls[0] : 0, p1 = depth > 0 ? ls[1] : 0;
    for (int j = threadIdx.x; j < width; j += blockDim.x) {
        float m = 0.f;  // PyG scatter-max leaves rows nothing lands on at zero
        for (int p = p0; p < p1; ++p) {
            float v = h[(int64_t)order[p] * ld_h + j];
            m = (p == p0) ? v : fmaxf(m, v);
        }
        out[(int64_t)g * ld_out + col_off + j] = m;
    }
}

struct ReadoutJobs { const float* h[16]; int ld_h[16], width[16], dir[16], col_off[16]; };
__global__ void __launch_bounds__(256) readout_max_batch_kernel(const int32_t* __restrict__ plan, PlanLayout L, ReadoutJobs J,
                                                                 float* __restrict__ out, int ld_out) {
    const int g = blockIdx.x, k = blockIdx.y;
    const int od = 1 - J.dir[k];
    const int n0 = plan[L.node_ptr + g];
    const int depth = plan[L.depth[od] + g];
    const int32_t* ls = plan + L.lstart[od] + n0 + g;
    const int32_t* order = plan + L.order[od];
    const int p0 = depth > 0 ? ls[0] : 0, p1 = depth > 0 ? ls[1] : 0;
    const float* __restrict__ h = J.h[k];
    const int ld_h = J.ld_h[k];
    for (int j = threadIdx.x; j < J.width[k]; j += blockDim.x) {
        float m = 0.f;  // PyG scatter-max leaves rows nothing lands on at zero
        for (int p = p0; p < p1; ++p) {
            float v = h[(int64_t)order[p] * ld_h + j];
            m = (p == p0) ? v : fmaxf(m, v);
        }
        out[(int64_t)g * ld_out + J.col_off[k] + j] = m;
    }
}

// Generic read-out (dagnn.py:194-202 `global_{max,mean,add}_pool`; P_ATTN is a softmax over a size-1 dimension, i.e.
// add): scope 0 / 1 = the output nodes of direction 0 / 1, scope 2 = every node of the graph (`out_pool_all`).
// One workgroup per graph, a thread per column, nodes in id order (a fixed summation order).
__global__ void __launch_bounds__(256) readout_pool_kernel(const int32_t* __restrict__ plan, PlanLayout L, int scope,
                                                            int mode, const float* __restrict__ h, int ld_h, int width,
                                                            float* __restrict__ out, int ld_out, int col_off) {
    const int g = blockIdx.x;
    const int n0 = plan[L.node_ptr + g];
    int p0, p1;
    const int32_t* order = nullptr;
    if (scope == 2) {
        p0 = n0; p1 = plan[L.node_ptr + g + 1];
    } else {
        const int od = 1 - scope;
        const int depth = plan[L.depth[od] + g];
        const int32_t* ls = plan + L.lstart[od] + n0 + g;
        order = plan + L.order[od];
        p0 = depth > 0 ? ls[0] : 0; p1 = depth > 0 ? ls[1] : 0;
    }
    const float inv = mode == DAGNN_POOL_MEAN ? 1.f / (float)max(p1 - p0, 1) : 1.f;
    for (int j = threadIdx.x; j < width; j += blockDim.x) {
        float m = 0.f;  // graphs nothing lands on stay at zero
        for (int p = p0; p < p1; ++p) {
            const float v = h[(int64_t)(order ? order[p] : p) * ld_h + j];
            m = mode == DAGNN_POOL_MAX ? ((p == p0) ? v : fmaxf(m, v)) : m + v;
        }
        out[(int64_t)g * ld_out + col_off + j] = m * inv;
    }
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ h, int ld_h, int width,
                                                           int stride, int node_off, float* __restrict__ out,
                                                           int ld_out, int col_off) {
    const int64_t g = blockIdx.x;
    const float* src = h + (g * stride + node_off) * (int64_t)ld_h;
    for (int j = threadIdx.x; j < width; j += blockDim.x) out[g * ld_out + col_off + j] = src[j];
}

// the same for up to 16 (matrix, node offset, column offset) jobs in one launch (blockIdx.y = job)
struct GatherJobs { const float* h[16]; int ld_h[16], width[16], node_off[16], col_off[16]; };
__global__ void __launch_bounds__(256) gather_rows_batch_kernel(GatherJobs J, int stride, float* __restrict__ out, int ld_out) {
    const int64_t g = blockIdx.x;
    const int q = blockIdx.y;
    const float* src = J.h[q] + (g * stride + J.node_off[q]) * (int64_t)J.ld_h[q];
    for (int j = threadIdx.x; j < J.width[q]; j += blockDim.x) out[g * ld_out + J.col_off[q] + j] = src[j];
}

// ---- decoder-side single-vertex step (dvae/dagnn.py:187-239, dvae/dagnn_bn.py:179-238): for every graph that has
// vertex v, aggregate the states of v's predecessors with the reference's PADDED soft-max (the predecessor lists are
// padded to the longest one with zero rows, and the soft-max runs over the padding as well: a padded key scores
// w_q.q + b, a real one w_q.q + b + w_k.h_j [+ w_vid[j]] - the common term cancels, so padded slots score 0), then
// run the L stacked GRU cells of the propagator on it - the aggregate is computed ONCE, from the layer-0 states, and
// reused by every stacked layer (the reference's `H` is no longer None after the first iteration).  One workgroup per
// graph, one launch per decoder step.
struct IpropLayers {
    const float* w_ih[DAGNN_MAX_STACKED];
    const float* w_hh[DAGNN_MAX_STACKED];
    const float* b_ih[DAGNN_MAX_STACKED];
    const float* b_hh[DAGNN_MAX_STACKED];
    int in_dim[DAGNN_MAX_STACKED];
};

__global__ void __launch_bounds__(256) iprop_step_kernel(const float* __restrict__ values, const int32_t* __restrict__ pred_vid,
                                                          int P, int hs, const float* __restrict__ w_key,
                                                          const float* __restrict__ vid_bias, const float* __restrict__ H_given,
                                                          const float* __restrict__ x, int in0, IpropLayers W, int L,
                                                          float* __restrict__ states, int64_t B) {
    extern __shared__ float lds[];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int din = in0 > hs ? in0 : hs;
    float* Hs = lds;              // [hs]   the aggregate
    float* in = Hs + hs;          // [din]  input of the current stacked layer
    float* gi = in + din;         // [3hs]
    float* gh = gi + 3 * hs;      // [3hs]
    float* sc = gh + 3 * hs;      // [P]
    if (H_given) {
        for (int k = tid; k < hs; k += 256) Hs[k] = H_given[(int64_t)g * hs + k];
    } else {
        const float* vg = values + (int64_t)g * P * hs;
        for (int p = wave; p < P; p += 4) {
            const int id = pred_vid[(int64_t)g * P + p];
            float s = 0.f;
            if (id >= 0) {
                for (int k = lane; k < hs; k += 64) s = fmaf(w_key[k], vg[(int64_t)p * hs + k], s);
                s = wave_sum(s);
                if (vid_bias) s += vid_bias[id];
            }
            if (lane == 0) sc[p] = s;
        }
        __syncthreads();
        float m = -INFINITY, den = 0.f;
        for (int p = 0; p < P; ++p) m = fmaxf(m, sc[p]);
        for (int p = 0; p < P; ++p) den += expf(sc[p] - m);
        for (int k = tid; k < hs; k += 256) {
            float a = 0.f;
            for (int p = 0; p < P; ++p) a = fmaf(expf(sc[p] - m) / den, vg[(int64_t)p * hs + k], a);   // padded rows are zero
            Hs[k] = a;
        }
    }
    for (int k = tid; k < in0; k += 256) in[k] = x[(int64_t)g * in0 + k];
    __syncthreads();
    for (int l = 0; l < L; ++l) {
        const int K = W.in_dim[l];
        for (int j = wave; j < 6 * hs; j += 4) {   // one wave per row of [W_ih ; W_hh]
            const bool hid = j >= 3 * hs;
            const int r = hid ? j - 3 * hs : j, kk = hid ? hs : K;
            const float* wrow = (hid ? W.w_hh[l] : W.w_ih[l]) + (int64_t)r * kk;
            const float* v = hid ? Hs : in;
            float s = 0.f;
            for (int k = lane; k < kk; k += 64) s = fmaf(wrow[k], v[k], s);
            s = wave_sum(s);
            if (lane == 0) (hid ? gh : gi)[r] = s + (hid ? W.b_hh[l] : W.b_ih[l])[r];
        }
        __syncthreads();
        for (int u = tid; u < hs; u += 256) {
            const float r = 1.0f / (1.0f + expf(-(gi[u] + gh[u])));
            const float z = 1.0f / (1.0f + expf(-(gi[hs + u] + gh[hs + u])));
            const float n = tanhf(gi[2 * hs + u] + r * gh[2 * hs + u]);
            const float hv = (1.0f - z) * n + z * Hs[u];
            states[((int64_t)l * B + g) * hs + u] = hv;
            in[u] = hv;
        }
        __syncthreads();
    }
}

}  // namespace

namespace {
// a stand-in for another process's / library's kernels: `ticks` of the 100 MHz clock of pure spinning per workgroup
__global__ void occupy_kernel(long long ticks, float* sink) {
    const unsigned long long t0 = wall_clock64();
    float acc = (float)threadIdx.x;
    while ((long long)(wall_clock64() - t0) < ticks) {
        for (int i = 0; i < 64; ++i) acc = fmaf(acc, 1.0001f, 0.5f);
        __builtin_amdgcn_s_sleep(8);
    }
    if (sink && acc == 12345.678f) *sink = acc;   // (keeps the loop)
}
}  // namespace

extern "C" int dagnn_debug_occupy(int num_wgs, int threads, int64_t ticks, float* sink, void* stream) {
    if (num_wgs <= 0 || threads <= 0 || threads > 1024 || (threads % 64) || ticks < 0 || ticks > (1ll << 31)) return DAGNN_EINVAL;
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)num_wgs), dim3((unsigned)threads), 0, (hipStream_t)stream, (long long)ticks, sink);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

namespace {
struct FpJobs { const unsigned* p[DAGNN_MAX_FP_TENSORS]; long long words[DAGNN_MAX_FP_TENSORS]; };
// one workgroup per tensor: 1024 words spread evenly over it, each weighted by an odd constant of its position, summed mod 2^64
// (in a fixed order: the value is a pure function of the tensor's contents)
__global__ void __launch_bounds__(256) param_fingerprint_kernel(FpJobs J, unsigned long long* fp, int mode, int* err, int bit) {
    __shared__ unsigned long long part[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    const unsigned* __restrict__ p = J.p[t];
    const long long n = J.words[t];
    unsigned long long acc = 0ull;
    if (n > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long k = j * 256 + tid;
            const long long idx = n >= 1024 ? (k * n) >> 10 : (k < n ? k : -1);
            if (idx >= 0) acc += (unsigned long long)p[idx] * (0x9E3779B97F4A7C15ull * (unsigned long long)(2 * k + 1) | 1ull);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long v = part[0] + part[1] + part[2] + part[3];
        if (mode == 0) fp[t] = v;
        else if (fp[t] != v) __hip_atomic_fetch_or(err, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

extern "C" int dagnn_param_fingerprint(const void* const* ptrs, const int64_t* numel, int n, uint64_t* fp, int mode, int* err,
                                       int bit, void* stream) {
    if (!ptrs || !numel || !fp || n < 0 || n > DAGNN_MAX_FP_TENSORS || (mode != 0 && mode != 1) || (mode == 1 && !err)) return DAGNN_EINVAL;
    if (n == 0) return DAGNN_OK;
    FpJobs J;
    for (int t = 0; t < n; ++t) {
        if (numel[t] < 0 || (numel[t] > 0 && !ptrs[t])) return DAGNN_EINVAL;
        J.p[t] = (const unsigned*)ptrs[t]; J.words[t] = numel[t];
    }
    hipLaunchKernelGGL(param_fingerprint_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, J,
                       reinterpret_cast<unsigned long long*>(fp), mode, err, bit);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" const char* dagnn_version(void) { return "dagnn_hip 0.1 gfx950"; }

extern "C" int dagnn_encode_ast(const int64_t* x, int64_t* depth, const float* type_emb, const float* attr_emb,
                                const float* depth_emb, int max_depth, float* out, int ld_out, int64_t N, int H,
                                void* stream) {
    if (N < 0 || H <= 0 || (H & 3) || (ld_out & 3) || ld_out < H) return DAGNN_EINVAL;
    if (N == 0) return DAGNN_OK;
    if (!x || !depth || !type_emb || !attr_emb || !depth_emb || !out) return DAGNN_EINVAL;
    int64_t blocks = (N + 3) / 4;  // 4 waves per block, one row per wave
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(encode_ast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, depth,
                       type_emb, attr_emb, depth_emb, max_depth, out, ld_out, N, H);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_pack_whh(const float* w_hh, float* w_hh_t, int H, void* stream) {
    if (!w_hh || !w_hh_t || H <= 0) return DAGNN_EINVAL;
    dim3 grid((H + 31) / 32, (3 * H + 31) / 32);
    hipLaunchKernelGGL(pack_whh_kernel, grid, dim3(256), 0, (hipStream_t)stream, w_hh, w_hh_t, 3 * H, H);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_readout_max(const dagnn_plan* pl, const float* h, int ld_h, int width, int dir, float* out,
                                 int ld_out, int col_off, void* stream) {
    if (!pl || !pl->data || !h || !out || width <= 0 || (dir != 0 && dir != 1)) return DAGNN_EINVAL;
    if (pl->B == 0) return DAGNN_OK;
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipLaunchKernelGGL(readout_max_kernel, dim3((unsigned)pl->B), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)pl->data, L, dir, h, ld_h, width, out, ld_out, col_off);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_readout_max_batch(const dagnn_plan* pl, const dagnn_readout_job* jobs, int n, float* out, int ld_out,
                                       void* stream) {
    if (!pl || !pl->data || !jobs || !out || n < 0 || n > 16) return DAGNN_EINVAL;
    if (pl->B == 0 || n == 0) return DAGNN_OK;
    ReadoutJobs J;
    for (int k = 0; k < n; ++k) {
        const dagnn_readout_job& q = jobs[k];
        if (!q.h || q.width <= 0 || (q.dir != 0 && q.dir != 1) || q.ld_h < q.width || q.col_off < 0 || ld_out < q.col_off + q.width)
            return DAGNN_EINVAL;
        J.h[k] = q.h; J.ld_h[k] = q.ld_h; J.width[k] = q.width; J.dir[k] = q.dir; J.col_off[k] = q.col_off;
    }
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipLaunchKernelGGL(readout_max_batch_kernel, dim3((unsigned)pl->B, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)pl->data, L, J, out, ld_out);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_readout_pool(const dagnn_plan* pl, const float* h, int ld_h, int width, int scope, int mode,
                                  float* out, int ld_out, int col_off, void* stream) {
    if (!pl || !pl->data || !h || !out || width <= 0 || scope < 0 || scope > 2 || mode < DAGNN_POOL_MAX ||
        mode > DAGNN_POOL_MEAN || ld_h < width || col_off < 0 || ld_out < col_off + width)
        return DAGNN_EINVAL;
    if (pl->B == 0) return DAGNN_OK;
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipLaunchKernelGGL(readout_pool_kernel, dim3((unsigned)pl->B), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)pl->data, L, scope, mode, h, ld_h, width, out, ld_out, col_off);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_gather_rows(const float* h, int ld_h, int width, int64_t num_graphs, int stride, int node_off,
                                 float* out, int ld_out, int col_off, void* stream) {
    if (!h || !out || width <= 0 || num_graphs < 0 || stride <= 0 || node_off < 0 || node_off >= stride)
        return DAGNN_EINVAL;
    if (num_graphs == 0) return DAGNN_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)num_graphs), dim3(256), 0, (hipStream_t)stream, h, ld_h,
                       width, stride, node_off, out, ld_out, col_off);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_gather_rows_batch(const dagnn_gather_job* jobs, int n, int64_t num_graphs, int stride, float* out, int ld_out,
                                       void* stream) {
    if (!jobs || !out || n < 0 || n > 16 || num_graphs < 0 || stride <= 0) return DAGNN_EINVAL;
    if (num_graphs == 0 || n == 0) return DAGNN_OK;
    GatherJobs J;
    for (int q = 0; q < n; ++q) {
        const dagnn_gather_job& j = jobs[q];
        if (!j.h || j.width <= 0 || j.ld_h < j.width || j.node_off < 0 || j.node_off >= stride || j.col_off < 0 ||
            ld_out < j.col_off + j.width)
            return DAGNN_EINVAL;
        J.h[q] = j.h; J.ld_h[q] = j.ld_h; J.width[q] = j.width; J.node_off[q] = j.node_off; J.col_off[q] = j.col_off;
    }
    hipLaunchKernelGGL(gather_rows_batch_kernel, dim3((unsigned)num_graphs, (unsigned)n), dim3(256), 0, (hipStream_t)stream, J,
                       stride, out, ld_out);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_iprop_step(const float* values, const int32_t* pred_vid, int64_t B, int P, int hs, const float* w_key,
                                const float* vid_bias, const float* H_given, const float* x, int in0,
                                const dagnn_iprop_layer* layers, int L, float* states, void* stream) {
    if (B < 0 || P < 0 || hs <= 0 || in0 <= 0 || L <= 0 || L > DAGNN_MAX_STACKED || !layers || !x || !states)
        return DAGNN_EINVAL;
    if (!H_given && P > 0 && (!values || !pred_vid || !w_key)) return DAGNN_EINVAL;
    if (B == 0) return DAGNN_OK;
    IpropLayers W;
    for (int l = 0; l < L; ++l) {
        const dagnn_iprop_layer& q = layers[l];
        if (!q.w_ih || !q.w_hh || !q.b_ih || !q.b_hh || q.in_dim != (l == 0 ? in0 : hs)) return DAGNN_EINVAL;
        W.w_ih[l] = q.w_ih; W.w_hh[l] = q.w_hh; W.b_ih[l] = q.b_ih; W.b_hh[l] = q.b_hh; W.in_dim[l] = q.in_dim;
    }
    const int din = in0 > hs ? in0 : hs;
    const size_t lds = (size_t)(hs + din + 6 * hs + (P > 0 ? P : 1)) * sizeof(float);
    if (lds > 160 * 1024) return DAGNN_EINVAL;
    const void* fn = reinterpret_cast<const void*>(iprop_step_kernel);
    if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DAGNN_EHIP(hipGetLastError());
    hipLaunchKernelGGL(iprop_step_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, values, pred_vid, P, hs,
                       w_key, vid_bias, H_given, x, in0, W, L, states, B);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
