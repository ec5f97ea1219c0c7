// Do fp32 MFMAs and the VALU work of ANOTHER wave on the same SIMD overlap, or do they add up?
// 512-thread workgroups: waves w and w + 4 share a SIMD.  Waves 0-3 issue N back-to-back fp32 MFMAs of shape `shape` on 6
// independent accumulators and time them; waves 4-7 run M independent v_fma (4 chains) and time them.  Each role runs alone and
// both together: overlap -> together ~ max(alone), shared datapath -> together ~ sum.
// hipcc --offload-arch=gfx950 -O3 -o mfma_share mfma_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ void k(unsigned long long* out, int do_mfma, int do_valu, int iters, float* sink) {
    const int wave = threadIdx.x >> 6;
    float b = threadIdx.x, c = 1.0f, d = 0.5f, e = 0.25f;
    __syncthreads();
    if (wave < 4) {
        if (!do_mfma) return;
        f4 a4[6]; f16 a16[6];
        for (int i = 0; i < 6; ++i) { a4[i] = f4{0, 0, 0, 0}; for (int j = 0; j < 16; ++j) a16[i][j] = 0.f; }
        unsigned long long c0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    if (SHAPE == 16) a4[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a4[n], 0, 0, 0);
                    else if (SHAPE == 32) a16[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, c, a16[n], 0, 0, 0);
                    else a4[n] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, c, a4[n], 0, 0, 0);
                }
        }
        unsigned long long c1 = clock64();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = c1 - c0;
        float s = 0; for (int n = 0; n < 6; ++n) s += a4[n][0] + a16[n][0];
        if (s == 12345.f) sink[0] = s;
    } else {
        if (!do_valu) return;
        if (do_valu == 2) __builtin_amdgcn_s_setprio(3);
        unsigned long long c0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(b)); asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(c));
                asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(d)); asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(e));
            }
        }
        unsigned long long c1 = clock64();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = c1 - c0;
        if (b + c + d + e == 12345.f) sink[1] = b;
    }
}
// one wave per SIMD: an MFMA followed by K independent v_fma of the SAME wave - how many fit under a 16x16x4?
template <int K>
__global__ void own(unsigned long long* out, int iters, float* sink) {
    float b = threadIdx.x, c = 1.0f, d = 0.5f, e = 0.25f, f = 2.f;
    f4 a4[6];
    for (int i = 0; i < 6; ++i) a4[i] = f4{0, 0, 0, 0};
    unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            a4[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, f, a4[n], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                if (q % 3 == 0) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(c));
                else if (q % 3 == 1) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(d));
                else asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(e));
            }
        }
    }
    unsigned long long c1 = clock64();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
    float s = c + d + e; for (int n = 0; n < 6; ++n) s += a4[n][0];
    if (s == 12345.f) sink[0] = s;
}
template <int K>
void run_own(unsigned long long* d, float* sink) {
    const int iters = 4000;
    hipLaunchKernelGGL(own<K>, dim3(256), dim3(256), 0, 0, d, iters, sink);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 1024; ++i) m += h[i];
    printf("own wave: 16x16x4 + %d v_fma: %.1f cycles per (MFMA + fillers)\n", K, m / 1024 / iters / 6);
}
template <int SHAPE>
void run(unsigned long long* d, float* sink, const char* name) {
    const int iters = 4000;
    double res[4][2];
    for (int mode = 0; mode < 4; ++mode) {
        const int dm = mode != 1, dv = mode == 3 ? 2 : mode != 0;
        hipMemset(d, 0, 256 * 8 * 8);
        hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, d, dm, dv, iters, sink);
        hipDeviceSynchronize();
        unsigned long long h[2048];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int bI = 0; bI < 256; ++bI) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[bI * 8 + w];
        res[mode][0] = m / 1024 / iters / 24; res[mode][1] = v / 1024 / iters / 24;
    }
    printf("%-10s cycles per MFMA alone %.1f, beside the VALU wave %.1f (prio 3: %.1f) | cycles per v_fma alone %.2f, beside the MFMA wave %.2f (prio 3: %.2f)\n",
           name, res[0][0], res[2][0], res[3][0], res[1][1], res[2][1], res[3][1]);
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 64);
    run_own<0>(d, sink); run_own<2>(d, sink); run_own<4>(d, sink); run_own<6>(d, sink); run_own<8>(d, sink);
    run<4>(d, sink, "4x4x1"); run<16>(d, sink, "16x16x4"); run<32>(d, sink, "32x32x2");
    return 0;
}
