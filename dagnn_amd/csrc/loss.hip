// loss.hip - the training loss of the TOK task and its gradient in ONE launch.
//
// Reference path replaced: the caller-side loss loop of ogbg-code/main_pyg.py:55-60 -
//     loss = 0;  for i in range(len(pred_list)): loss += multicls_criterion(pred_list[i].to(torch.float32), batch.y_arr[:, i])
//     loss = loss / len(pred_list)
// with multicls_criterion = torch.nn.CrossEntropyLoss() (main_pyg.py:37): the mean over the batch of -log softmax(pred)[y], per
// head, then the mean over the S heads.  Under autograd that is 5 x (log-softmax, nll, their two backward kernels, a reduction,
// the additions) = ~50 launches of 2-7 us each on a step that is bound by the host's launch rate.  Here: one workgroup per
// (graph, head) row of the [B, S * V] logits - the S heads' outputs side by side, as DAGNN._heads lays them out - computes
// the row's loss AND the row of d loss / d logits = (softmax - onehot) / (B S); the workgroup that finishes last adds the B S row
// losses in index order (no atomics on values: the loss is bitwise reproducible).
// Every head sees the same B rows, so the mean of the per-head means is the mean over all B S rows.  Targets must lie in
// [0, V): CrossEntropyLoss's ignore_index (-100) does not occur in the reference's targets (y_arr holds vocabulary ids,
// utils.py: encode_seq_to_arr) and is not supported - an out-of-range target makes the loss NaN, loudly.
#include "common.h"

namespace {

__device__ __forceinline__ float ce_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float ce_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256) seq_ce_kernel(const float* __restrict__ logits, long long ld, const long long* __restrict__ y,
                                                     int B, int S, int V, float* __restrict__ dlogits, long long ld_d, float* __restrict__ row_loss,
                                                     float* __restrict__ loss, unsigned* __restrict__ counter) {
    __shared__ float red[4];
    __shared__ unsigned last;
    const int row = blockIdx.x, b = row / S, s = row - b * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ x = logits + (long long)b * ld + (long long)s * V;
    const long long t = y[(long long)b * S + s];
    float m = -INFINITY;
    for (int j = tid; j < V; j += 256) m = fmaxf(m, x[j]);
    m = ce_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float z = 0.f;
    for (int j = tid; j < V; j += 256) z += __expf(x[j] - m);
    z = ce_wave_sum(z);
    if (lane == 0) red[wave] = z;
    __syncthreads();
    z = (red[0] + red[1]) + (red[2] + red[3]);
    const float lse = m + __logf(z);
    const bool ok = t >= 0 && t < V;
    if (dlogits) {
        const float scale = 1.0f / ((float)B * (float)S), inv = 1.0f / z;
        float* __restrict__ d = dlogits + (long long)b * ld_d + (long long)s * V;
        for (int j = tid; j < V; j += 256) d[j] = (__expf(x[j] - m) * inv - (j == t ? 1.0f : 0.0f)) * scale;
    }
    if (tid == 0) {
        row_loss[row] = ok ? lse - x[t] : NAN;
        __threadfence();
        last = atomicAdd(counter, 1u) == (unsigned)(B * S - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last) {   // (uniform per workgroup) the last row in: every row loss is visible - sum them in index order
        __threadfence();
        const int n = B * S;
        float acc = 0.f;
        for (int j = tid; j < n; j += 256) acc += __hip_atomic_load(row_loss + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc = ce_wave_sum(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
            counter[0] = 0u;   // ready for the next call
        }
    }
}

}  // namespace

extern "C" int dagnn_seq_ce(const float* logits, int64_t ld, const int64_t* y, int B, int S, int V, float* dlogits, int64_t ld_d,
                            float* row_loss, float* loss, unsigned* counter, void* stream) {
    if (!logits || !y || !row_loss || !loss || !counter || B <= 0 || S <= 0 || V <= 0 || ld < (int64_t)S * V ||
        (dlogits && ld_d < (int64_t)S * V))
        return DAGNN_EINVAL;
    if ((int64_t)B * S >= (1ll << 31)) return DAGNN_EINVAL;
    hipLaunchKernelGGL(seq_ce_kernel, dim3((unsigned)(B * S)), dim3(256), 0, (hipStream_t)stream, logits, (long long)ld,
                       (const long long*)y, B, S, V, dlogits, (long long)ld_d, row_loss, loss, counter);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
