// what a taken scalar branch costs a wave on gfx950 (loop back-edges, forward skips), alone on its SIMD and with 3 waves per SIMD
// hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K> __global__ void loop_k(unsigned long long* out, int iters, float* sink) {
    float acc = threadIdx.x;
    unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(acc));
    }
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (acc == 12345.f) sink[1] = acc;
}
// a chain of forward branches: each s_cbranch_scc1 jumps over one dummy instruction (taken) or falls through (not taken)
template <int TAKEN> __global__ void fwd(unsigned long long* out, int iters, float* sink) {
    float acc = threadIdx.x;
    unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "s_cmp_eq_u32 %1, %1\n\t"
            ".rept 16\n\t"
            "s_cbranch_scc%c2 1f\n\t"
            "v_fma_f32 %0, %0, 1.0, 0.5\n\t"
            "1:\n\t"
            ".endr\n\t" : "+v"(acc) : "s"(i), "n"(TAKEN) : "scc");
    }
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (acc == 12345.f) sink[1] = acc;
}
template <typename F> void run(const char* name, F launch, int per_iter, unsigned long long* d) {
    for (int threads : {64, 768}) {
        const int iters = 20000;
        launch(threads, iters);
        hipDeviceSynchronize();
        unsigned long long h[240];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double c = 0; for (int b = 0; b < 240; ++b) c += h[b];
        printf("%-34s %3d threads: %.1f cycles per iteration (%d ops)\n", name, threads, c / 240 / iters, per_iter);
    }
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 1024 * 8); hipMalloc(&sink, 64);
    run("loop, 1 dependent fma", [&](int t, int it) { hipLaunchKernelGGL(loop_k<1>, dim3(240), dim3(t), 0, 0, d, it, sink); }, 1, d);
    run("loop, 4 dependent fma", [&](int t, int it) { hipLaunchKernelGGL(loop_k<4>, dim3(240), dim3(t), 0, 0, d, it, sink); }, 4, d);
    run("loop, 16 dependent fma", [&](int t, int it) { hipLaunchKernelGGL(loop_k<16>, dim3(240), dim3(t), 0, 0, d, it, sink); }, 16, d);
    run("loop, 64 dependent fma", [&](int t, int it) { hipLaunchKernelGGL(loop_k<64>, dim3(240), dim3(t), 0, 0, d, it, sink); }, 64, d);
    run("16 forward branches NOT taken", [&](int t, int it) { hipLaunchKernelGGL(fwd<0>, dim3(240), dim3(t), 0, 0, d, it, sink); }, 16, d);
    run("16 forward branches taken", [&](int t, int it) { hipLaunchKernelGGL(fwd<1>, dim3(240), dim3(t), 0, 0, d, it, sink); }, 16, d);
    return 0;
}
