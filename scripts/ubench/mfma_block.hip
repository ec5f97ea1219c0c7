// does a wave issuing back-to-back fp32 MFMAs (v_mfma_f32_4x4x1) starve the VALU of the other waves on its SIMD?
// 512-thread workgroups: waves w and w + 4 share a SIMD.  Waves 0-3 run a dependent v_fma chain and time it; waves 4-7 are
// (a) gone, (b) in a 4x4x1 MFMA loop, (c) in an independent-FMA loop, (d) in a 16x16x4 MFMA loop, (e) spinning on LDS + s_sleep 1.
// hipcc --offload-arch=gfx950 -O3 -o mfma_block mfma_block.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* out, int mode, int iters, float* sink) {
    __shared__ int flag[4];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) flag[threadIdx.x] = 0;
    __syncthreads();
    float acc = threadIdx.x;
    if (wave < 4) {
        for (int i = 0; i < 2000; ++i) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(acc));   // let the partners get going
        unsigned long long c0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(acc));
        }
        unsigned long long c1 = clock64();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + wave] = c1 - c0;
        __hip_atomic_store(&flag[wave], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        const int w = wave - 4;
        if (mode == 0) return;
        f4 a = {0.f, 0.f, 0.f, 0.f};
        float b = acc, c = 1.0f, d = 0.5f, e = 0.25f;
        while (__hip_atomic_load(&flag[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
            if (mode == 1) {
#pragma unroll
                for (int q = 0; q < 32; ++q) a = __builtin_amdgcn_mfma_f32_4x4x1f32(b, c, a, 0, 0, 0);
            } else if (mode == 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(b)); asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(c));
                    asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(d)); asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(e));
                }
            } else if (mode == 3) {
#pragma unroll
                for (int q = 0; q < 8; ++q) a = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a, 0, 0, 0);
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (a[0] + b + c + d + e == 12345.f) sink[0] = a[0];
    }
    if (acc == 12345.f) sink[1] = acc;
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 240 * 4 * 8); hipMalloc(&sink, 64);
    const char* names[5] = {"partner gone", "partner: 4x4x1 MFMA loop", "partner: independent FMA loop", "partner: 16x16x4 MFMA loop", "partner: LDS poll + s_sleep 1"};
    for (int mode = 0; mode < 5; ++mode) {
        const int iters = 2000;
        hipLaunchKernelGGL(k, dim3(240), dim3(512), 0, 0, d, mode, iters, sink);
        hipDeviceSynchronize();
        unsigned long long h[960];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double c = 0; for (int i = 0; i < 960; ++i) c += h[i];
        printf("%-34s dependent v_fma: %.2f cycles each\n", names[mode], c / 960 / iters / 32);
    }
    return 0;
}
