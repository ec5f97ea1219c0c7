// shader clock vs the 100 MHz constant clock inside kernels of different activity (is a latency-bound persistent kernel
// running at the boost clock?)   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(unsigned long long* out, int mode, int iters, float* sink) {
    unsigned long long c0 = clock64(), r0 = wall_clock64();
    float acc = threadIdx.x;
    if (mode == 0) {          // mostly sleeping
        for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
    } else if (mode == 1) {   // dependent FMA chain, one wave per CU
        for (int i = 0; i < iters * 64; ++i) acc = fmaf(acc, 1.0001f, 0.5f);
    } else {                  // memory polling of one word
        volatile float* p = sink;
        for (int i = 0; i < iters * 4; ++i) acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (acc == 12345.f) sink[1] = acc;
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 2 * 1024 * 8); hipMalloc(&sink, 64); hipMemset(sink, 0, 64);
    unsigned long long h[2048];
    const char* names[3] = {"sleeping", "fma chain", "polling"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode)
        for (int threads : {64, 768}) {
            for (int iters : {2000, 20000}) {
                hipLaunchKernelGGL(probe, dim3(240), dim3(threads), 0, 0, d, mode, iters, sink);
                hipDeviceSynchronize();
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                double c = 0, r = 0;
                for (int b = 0; b < 240; ++b) { c += h[2 * b]; r += h[2 * b + 1]; }
                printf("%-10s %3d threads x 240 blocks, %6d iters: %.1f us, shader clock %.0f MHz\n", names[mode], threads, iters,
                       r / 240 / 100.0, c / r * 100.0);
            }
        }
    return 0;
}
