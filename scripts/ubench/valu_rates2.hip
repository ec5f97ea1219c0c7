// v_pk_fma_f32 with distinct operand registers per instruction (as in the dataflow kernel: weights resident in 96
// registers, operands from LDS) vs v_fma_f32, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void k(const float* in, float* out, int iters, long long* t) {
    v2f w[48]; v2f a[8]; v2f acc[6];
    for (int i = 0; i < 48; ++i) w[i] = (v2f){in[threadIdx.x + i], in[threadIdx.x + 64 + i]};
    for (int i = 0; i < 8; ++i) a[i] = (v2f){in[threadIdx.x + 128 + i], in[threadIdx.x + 136 + i]};
    for (int c = 0; c < 6; ++c) acc[c] = (v2f){0.f, 0.f};
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (PK) acc[c] = __builtin_elementwise_fma(a[q], w[6 * q + c], acc[c]);
                else { acc[c].x = fmaf(a[q].x, w[6 * q + c].x, acc[c].x); acc[c].y = fmaf(a[q].y, w[6 * q + c].y, acc[c].y); }
            }
        }
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    }
    long long w1 = wall_clock64();
    float s = 0;
    for (int c = 0; c < 6; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[0] = w1 - w0;
}

template <typename K> void run(const char* name, K kern, int threads) {
    float* in; float* out; long long* t;
    hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&t, 16);
    const int iters = 200000;
    hipLaunchKernelGGL(kern, dim3(getenv("WGS") ? atoi(getenv("WGS")) : 1), dim3(threads), 0, 0, in, out, iters, t);
    hipLaunchKernelGGL(kern, dim3(getenv("WGS") ? atoi(getenv("WGS")) : 1), dim3(threads), 0, 0, in, out, iters, t);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-40s %3d threads: %.3f ns per 2 FMA lanes-op (pk = one instr)\n", name, threads, h * 10.0 / (iters * 48.0));
}
int main() {
    run("v_pk_fma_f32 distinct regs", k<1>, 256);
    run("2 x v_fma_f32 distinct regs", k<0>, 256);
    run("v_pk_fma_f32 distinct regs", k<1>, 512);
    run("2 x v_fma_f32 distinct regs", k<0>, 512);
    return 0;
}
