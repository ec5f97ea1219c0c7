// Micro-benchmark: issue cost / dependent latency of the VALU instructions the dataflow kernel's compute waves run
// (v_pk_fma_f32 chains, DPP row-shift adds), and the core clock (s_memtime ticks vs the 100 MHz wall clock).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/ubench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ void pkfma(float* out, int iters, long long* t) {
    v2f acc[CHAINS];
    v2f w = {1.0001f, 0.9999f}, a = {0.5f + threadIdx.x * 1e-6f, 0.25f};
    for (int c = 0; c < CHAINS; ++c) acc[c] = (v2f){(float)c, 1.f};
    long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_elementwise_fma(a, w, acc[c]);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
}

template <int CHAINS>
__global__ void dppadd(float* out, int iters, long long* t) {
    float v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x * 0.001f + c;
    long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
                v[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[c]), 0x111, 0xf, 0xf, true));
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
}

template <typename K> void run(const char* name, K kern, int threads, int per_iter) {
    float* out; long long* t;
    hipMalloc(&out, 1 << 20); hipMalloc(&t, 16);
    const int iters = 20000;
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, iters, t);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, iters, t);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double ns = h[0] * 10.0, instr = (double)iters * per_iter;
    printf("%-44s %4d threads: %.3f ns/instr/wave, s_memtime %.1f MHz -> %.2f memtime ticks/instr\n", name, threads, ns / instr,
           h[1] / (ns * 1e-3), h[1] / instr);
    hipFree(out); hipFree(t);
}

int main() {
    run("v_pk_fma_f32, 1 dependent chain", pkfma<1>, 256, 8);
    run("v_pk_fma_f32, 2 chains", pkfma<2>, 256, 16);
    run("v_pk_fma_f32, 3 chains", pkfma<3>, 256, 24);
    run("v_pk_fma_f32, 6 chains", pkfma<6>, 256, 48);
    run("v_pk_fma_f32, 6 chains, 2 waves/SIMD", pkfma<6>, 512, 48);
    run("v_add_f32_dpp row_shr, 1 chain", dppadd<1>, 256, 4);
    run("v_add_f32_dpp row_shr, 4 chains", dppadd<4>, 256, 16);
    run("v_add_f32_dpp row_shr, 8 chains", dppadd<8>, 256, 32);
    return 0;
}
