// optim.hip - the tail of the reference's training step as two small pipelines over a table of parameter tensors:
//     torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip);  optimizer.step()      (ogbg-code/main_pyg.py:63-65,
// optimizer = optim.Adam(model.parameters(), lr): main_pyg.py:179).  With torch these are ~12 launches that move the 29.8 M
// gradients three times (norm, scale in place, Adam) and walk the 30 small core tensors at a fraction of the memory rate; here
//   dagnn_grad_norm      sum of squares per 16 K-element chunk, then ONE workgroup adds the chunk sums in index order: the
//                        global 2-norm as a device float (no atomics: bitwise reproducible), 2 launches;
//   dagnn_clip_adam      Adam's update with the clip coefficient min(1, max_norm / (norm + 1e-6)) applied to the gradient as it
//                        is read - g, p, m, v in, p, m, v out: each tensor crosses the memory bus once, 1 launch.
// Same arithmetic as torch.optim.Adam (no amsgrad, L2 weight decay, bias correction on both moments) in fp32.
#include "common.h"
#include <math.h>

namespace {

constexpr int OPT_CHUNK = 16384;   // elements per workgroup (256 threads x 16 float4)
struct OptTable {
    float* p[DAGNN_MAX_OPT_TENSORS]; const float* g[DAGNN_MAX_OPT_TENSORS]; float* m[DAGNN_MAX_OPT_TENSORS]; float* v[DAGNN_MAX_OPT_TENSORS];
    long long n[DAGNN_MAX_OPT_TENSORS];
    int first[DAGNN_MAX_OPT_TENSORS + 1];   // first chunk of every tensor (prefix of ceil(n / OPT_CHUNK))
    int count;
};
struct NormTable { const float* g[DAGNN_MAX_OPT_TENSORS]; long long n[DAGNN_MAX_OPT_TENSORS]; int first[DAGNN_MAX_OPT_TENSORS + 1]; int count; };

template <class T> __device__ __forceinline__ int opt_find(const T& tab, int chunk) {   // tensor of a chunk: the last t with first[t] <= chunk
    int lo = 0, hi = tab.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab.first[mid] <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(256) grad_sq_kernel(NormTable T, float* __restrict__ partial) {
    __shared__ float red[4];
    const int t = opt_find(T, blockIdx.x);
    const long long base = (long long)(blockIdx.x - T.first[t]) * OPT_CHUNK;
    const long long n = T.n[t];
    const float* __restrict__ g = T.g[t] + base;
    const long long left = n - base < OPT_CHUNK ? n - base : OPT_CHUNK;
    float acc = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const long long n4 = left >> 2;
        for (long long i = threadIdx.x; i < n4; i += 256) {
            const float4 x = reinterpret_cast<const float4*>(g)[i];
            acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
        for (long long i = (n4 << 2) + threadIdx.x; i < left; i += 256) acc += g[i] * g[i];
    } else {
        for (long long i = threadIdx.x; i < left; i += 256) acc += g[i] * g[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup: the chunk sums in index order (double accumulators per thread, a fixed tree across the threads)
__global__ void __launch_bounds__(256) grad_norm_finish_kernel(const float* __restrict__ partial, int nchunks, float* __restrict__ norm_sq_acc,
                                                               int accumulate, float* __restrict__ norm) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += 256) acc += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (accumulate) tot += (double)norm_sq_acc[0];
        norm_sq_acc[0] = (float)tot;
        norm[0] = (float)sqrt(tot);
    }
}

__global__ void __launch_bounds__(256) clip_adam_kernel(OptTable T, float lr, float omb1, float beta2, float omb2, float eps, float weight_decay,
                                                        float bias1, float bias2_sqrt, float max_norm, const float* __restrict__ norm) {
    const int t = opt_find(T, blockIdx.x);
    const long long base = (long long)(blockIdx.x - T.first[t]) * OPT_CHUNK;
    const long long n = T.n[t];
    const long long left = n - base < OPT_CHUNK ? n - base : OPT_CHUNK;
    float* __restrict__ p = T.p[t] + base;
    const float* __restrict__ g = T.g[t] + base;
    float* __restrict__ m = T.m[t] + base;
    float* __restrict__ v = T.v[t] + base;
    float coef = 1.0f;
    if (max_norm > 0.f) {   // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
        coef = max_norm / (norm[0] + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
    }
    const float step_size = lr / bias1;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= coef;
        if (weight_decay != 0.f) gg = fmaf(weight_decay, pp, gg);
        mm = fmaf(omb1, gg - mm, mm);               // lerp(m, g, 1 - beta1)
        vv = fmaf(omb2, gg * gg, beta2 * vv);       // (1 - beta in double on the host, as torch rounds them: 1.0f - 0.999f is 1.3e-5 off)
        const float denom = sqrtf(vv) / bias2_sqrt + eps;
        pp -= step_size * (mm / denom);
    };
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    long long done = 0;
    if (vec) {
        const long long n4 = left >> 2;
        for (long long i = threadIdx.x; i < n4; i += 256) {
            float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
            reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
        }
        done = n4 << 2;
    }
    for (long long i = done + threadIdx.x; i < left; i += 256) {
        float pp = p[i], mm = m[i], vv = v[i];
        upd(pp, g[i], mm, vv);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

}  // namespace

extern "C" int64_t dagnn_opt_chunks(const int64_t* numel, int n) {
    if (!numel || n < 0) return -1;
    int64_t c = 0;
    for (int t = 0; t < n; ++t) { if (numel[t] < 0) return -1; c += (numel[t] + OPT_CHUNK - 1) / OPT_CHUNK; }
    return c;
}

extern "C" int dagnn_grad_norm(const float* const* grads, const int64_t* numel, int n, float* partial, int64_t partial_len,
                               float* norm_sq, int accumulate, float* norm, void* stream) {
    if (!grads || !numel || !partial || !norm_sq || !norm || n <= 0 || n > DAGNN_MAX_OPT_TENSORS) return DAGNN_EINVAL;
    NormTable T;
    int c = 0;
    for (int t = 0; t < n; ++t) {
        if (numel[t] <= 0 || !grads[t]) return DAGNN_EINVAL;
        T.g[t] = grads[t]; T.n[t] = numel[t]; T.first[t] = c;
        const int64_t k = (numel[t] + OPT_CHUNK - 1) / OPT_CHUNK;
        if (c + k > (1 << 30)) return DAGNN_EINVAL;
        c += (int)k;
    }
    T.first[n] = c; T.count = n;
    if (c > partial_len) return DAGNN_ENOSPC;
    hipLaunchKernelGGL(grad_sq_kernel, dim3((unsigned)c), dim3(256), 0, (hipStream_t)stream, T, partial);
    DAGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, c, norm_sq, accumulate, norm);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_clip_adam(const dagnn_opt_tensor* tensors, int n, double lr, double beta1, double beta2, double eps, double weight_decay,
                               int64_t step, float max_norm, const float* norm, void* stream) {
    if (!tensors || n <= 0 || n > DAGNN_MAX_OPT_TENSORS || step < 1 || (max_norm > 0.f && !norm)) return DAGNN_EINVAL;
    OptTable T;
    int c = 0;
    for (int t = 0; t < n; ++t) {
        const dagnn_opt_tensor& x = tensors[t];
        if (x.numel <= 0 || !x.param || !x.grad || !x.exp_avg || !x.exp_avg_sq) return DAGNN_EINVAL;
        T.p[t] = x.param; T.g[t] = x.grad; T.m[t] = x.exp_avg; T.v[t] = x.exp_avg_sq; T.n[t] = x.numel; T.first[t] = c;
        const int64_t k = (x.numel + OPT_CHUNK - 1) / OPT_CHUNK;
        if (c + k > (1 << 30)) return DAGNN_EINVAL;
        c += (int)k;
    }
    T.first[n] = c; T.count = n;
    // (hyper-parameters arrive as doubles and are rounded to fp32 where torch rounds them: 1 - beta AFTER the subtraction)
    const double b1 = 1.0 - pow(beta1, (double)step), b2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)c), dim3(256), 0, (hipStream_t)stream, T, (float)lr, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)b1, (float)sqrt(b2), max_norm, norm);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
