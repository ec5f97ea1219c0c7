// Micro-benchmark: issue rate of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1: 256 MACs) and v_mfma_f32_16x16x4_f32
// with distinct B registers (resident weights), one wave per SIMD - the candidates for the dataflow kernel's products.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ void k(const float* in, float* out, int iters, long long* t) {
    float w[96], a[8];
    for (int i = 0; i < 96; ++i) w[i] = in[threadIdx.x + i];
    for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x + 100 + i];
    f4 acc[NACC];
    for (int c = 0; c < NACC; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 96; ++q) {
            if (MODE == 0) acc[q % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q & 7], w[q], acc[q % NACC], 0, 0, 0);
            else acc[q % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 7], w[q], acc[q % NACC], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    }
    long long w1 = wall_clock64();
    float s = 0;
    for (int c = 0; c < NACC; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[0] = w1 - w0;
}
template <typename K> void run(const char* name, K kern, int threads, double macs) {
    float* in; float* out; long long* t;
    hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&t, 16);
    const int iters = 20000;
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, in, out, iters, t);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, in, out, iters, t);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    const double ns = h * 10.0 / (iters * 96.0);
    printf("%-44s %3d threads: %.2f ns per instruction per wave = %.1f MAC/ns/SIMD\n", name, threads, ns, macs / ns);
}
int main() {
    run("mfma 4x4x1 16b, 3 accumulators", k<0, 3>, 256, 256);
    run("mfma 4x4x1 16b, 6 accumulators", k<0, 6>, 256, 256);
    run("mfma 4x4x1 16b, 6 acc, 2 waves/SIMD", k<0, 6>, 512, 256);
    run("mfma 16x16x4, 6 accumulators", k<1, 6>, 256, 1024);
    return 0;
}
