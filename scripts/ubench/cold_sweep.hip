// Micro-benchmark: latency of ONE sweep (one wave, 4 x 8-byte agent-scope loads per lane = a 2 KB granule row) of rows
// another workgroup wrote a few microseconds earlier - what a loader wave of the dataflow kernel pays per block when
// nothing is pending - by store flavour and placement.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define XCC_ID_REG (20 | (0 << 6) | (3 << 11))
constexpr int ROWS = 256;   // rows swept; PITCH (granules between rows) is a template parameter: 272 = the kernel's, large = one page per row

template <int ST_SCOPE, size_t PITCH, int PIECE>
__global__ void k(u64* rows, int* flag, int partner, int* xcc, long long* out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) xcc[b] = __builtin_amdgcn_s_getreg(XCC_ID_REG);
    if (b == 0) {   // producer: all rows, then a flag a little later
        for (int r = 0; r < ROWS; ++r) {
            if (PIECE == 64) {   // whole 512-byte runs per store instruction
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __hip_atomic_store(rows + (size_t)r * PITCH + 64 * q + lane, (1ull << 32) | (unsigned)(r + q), __ATOMIC_RELAXED, ST_SCOPE);
            } else {             // PIECE granules (8 -> 64 bytes) per store instruction, as the kernel's compute waves write
                for (int c = 0; c < 256; c += PIECE)
                    if (lane < PIECE)
                        __hip_atomic_store(rows + (size_t)r * PITCH + c + lane, (1ull << 32) | (unsigned)(r + c), __ATOMIC_RELAXED, ST_SCOPE);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __threadfence();
        long long t = wall_clock64();
        while (wall_clock64() - t < 500) {}   // 5 us
        if (lane == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else if (b == partner) {
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {}
        u64 acc = 0;
        int bad = 0;
        long long t0 = wall_clock64();
        for (int r = 0; r < ROWS; ++r) {
            u64 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = __hip_atomic_load(rows + (size_t)r * PITCH + 64 * q + lane + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < 4; ++q) { bad += (x[q] >> 32) != 1; acc += x[q] & 2; }
            acc = 0;
            asm volatile("" : "+v"(acc), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));   // dependent: one sweep at a time
        }
        long long t1 = wall_clock64();
        if (lane == 0) { out[0] = t1 - t0; out[1] = bad; }
    }
}

template <int SS, size_t PITCH, int PIECE = 64> void run(const char* name, int partner) {
    u64* rows; int* flag; int* xcc; long long* out;
    hipMalloc(&rows, (size_t)ROWS * PITCH * 8 + 4096); hipMemset(rows, 0, (size_t)ROWS * PITCH * 8 + 4096);
    hipMalloc(&flag, 4); hipMemset(flag, 0, 4); hipMalloc(&xcc, 64 * 4); hipMalloc(&out, 16); hipMemset(out, 0, 16);
    hipLaunchKernelGGL((k<SS, PITCH, PIECE>), dim3(16), dim3(64), 0, 0, rows, flag, partner, xcc, out);
    hipDeviceSynchronize();
    int hx[16]; long long h[2];
    hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-26s reader wg %2d (xcc %d / %d): %.0f ns per sweep, %lld stale granules\n", name, partner, hx[0], hx[partner], h[0] * 10.0 / ROWS, h[1]);
    hipFree(rows); hipFree(flag); hipFree(xcc); hipFree(out);
}
int main() {
    for (int partner : {8, 1}) {
        run<__HIP_MEMORY_SCOPE_AGENT, 272>("sc1 stores, pitch 2 KB", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 272 + 8192>("sc1 stores, pitch 66 KB", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 272 + 262144>("sc1 stores, pitch 2 MB", partner);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, 272>("plain stores, pitch 2 KB", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 272, 8>("sc1 64-B pieces, 2 KB", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 272, 16>("sc1 128-B pieces, 2 KB", partner);
        run<__HIP_MEMORY_SCOPE_AGENT, 272 + 8192, 8>("sc1 64-B pieces, 66 KB", partner);
    }
    return 0;
}
