// Micro-benchmark: LDS read rate of the broadcast patterns the dataflow kernel's compute waves use for their operand
// rows (every 8-lane / 16-lane group reads the same 16-byte words) vs a conflict-free distinct-address read.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bcast scripts/ubench/lds_bcast.hip && /tmp/lds_bcast
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void rd(float* out, int iters, long long* t) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int base;
    if (MODE == 0) base = (lane & 7) * 36;          // 8 distinct 16-B words, 8 groups read the same (K over 8 lanes)
    else if (MODE == 1) base = (lane & 15) * 20;    // 16 distinct words, 4 groups (K over 16 lanes)
    else if (MODE == 2) base = lane * 4;            // 64 distinct words, contiguous
    else base = 0;                                  // one word for the whole wave
    float4 acc = make_float4(0, 0, 0, 0);
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const float* p = lds + base + (it & 7) * 1280;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = *reinterpret_cast<const float4*>(p + 4 * u * (MODE == 2 ? 64 : 1));
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) t[0] = w1 - w0;
}

template <typename K> void run(const char* name, K kern, int threads) {
    float* out; long long* t;
    hipMalloc(&out, 1 << 20); hipMalloc(&t, 16);
    const int iters = 20000;
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, iters, t);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, iters, t);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    const double ns = h * 10.0, instr = (double)iters * 8 * (threads / 64);
    printf("%-40s %3d threads: %.2f ns per ds_read_b128 per CU = %.0f B/clk returned @2.4 GHz (4 VALU adds each)\n", name, threads,
           ns / instr, 1024.0 / (ns / instr * 2.4));
    hipFree(out); hipFree(t);
}

int main() {
    for (int th : {64, 256, 512}) {
        run("8 words x 8 groups (K over 8 lanes)", rd<0>, th);
        run("16 words x 4 groups (K over 16 lanes)", rd<1>, th);
        run("64 distinct words", rd<2>, th);
        run("1 word, whole wave", rd<3>, th);
    }
    return 0;
}
