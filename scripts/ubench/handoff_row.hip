// Micro-benchmark: a 2 KB row of 8-byte {epoch, value} granules bounced between two workgroups (one polling wave each,
// 4 granules per lane), by store flavour and placement.  What one hop of the dataflow kernel pays for the hand-off.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/handoff_row scripts/ubench/handoff_row.hip && /tmp/handoff_row
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define XCC_ID_REG (20 | (0 << 6) | (3 << 11))

template <int ST_SCOPE, int LD_SCOPE>
__global__ void bounce(u64* rows, int partner, int iters, int* xcc, long long* ticks, long long* polls, int* fail) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) xcc[b] = __builtin_amdgcn_s_getreg(XCC_ID_REG);
    if (b != 0 && b != partner) return;
    u64* mine = rows + (b == 0 ? 0 : 512);     // two 2 KB rows, 4 KB apart
    u64* theirs = rows + (b == 0 ? 512 : 0);
    long long t0 = wall_clock64(), np = 0;
    for (int it = 1; it <= iters; ++it) {
        if (b == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) __hip_atomic_store(mine + 4 * lane + q, ((u64)it << 32) | (unsigned)(lane + q), __ATOMIC_RELAXED, ST_SCOPE);
        }
        unsigned spins = 0;
        for (;;) {
            u64 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = __hip_atomic_load(theirs + 4 * lane + q, __ATOMIC_RELAXED, LD_SCOPE);
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(x[q] >> 32) == (unsigned)it;
            ++np;
            if (__all(ok)) break;
            if (++spins > 4000000u) { *fail = 1; return; }
        }
        if (b != 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) __hip_atomic_store(mine + 4 * lane + q, ((u64)it << 32) | (unsigned)(lane + q), __ATOMIC_RELAXED, ST_SCOPE);
        }
    }
    if (b == 0 && lane == 0) { *ticks = wall_clock64() - t0; *polls = np; }
}

// latency of ONE sweep (4 x 8-byte agent loads per lane) of a row nobody is writing: L2-resident vs dropped from L2
template <int LD_SCOPE>
__global__ void sweep(u64* rows, int iters, long long* ticks) {
    const int lane = threadIdx.x;
    u64 acc = 0;
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        u64 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = __hip_atomic_load(rows + 4 * lane + q + (acc & 1), __ATOMIC_RELAXED, LD_SCOPE);
        acc += (x[0] ^ x[1] ^ x[2] ^ x[3]) & 2;   // dependent: one sweep at a time
    }
    if (lane == 0) { ticks[0] = wall_clock64() - t0; ticks[1] = (long long)acc; }
}

template <int SS, int SL> void run(const char* name, int partner) {
    u64* rows; int* xcc; long long* ticks; long long* polls; int* fail;
    hipMalloc(&rows, 16384); hipMemset(rows, 0, 16384);
    hipMalloc(&xcc, 64 * 4); hipMalloc(&ticks, 16); hipMalloc(&polls, 8); hipMalloc(&fail, 4);
    hipMemset(fail, 0, 4); hipMemset(ticks, 0, 16); hipMemset(polls, 0, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((bounce<SS, SL>), dim3(16), dim3(64), 0, 0, rows, partner, iters, xcc, ticks, polls, fail);
    hipDeviceSynchronize();
    int hx[16]; long long ht, hp; int hf;
    hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost); hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hp, polls, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("%-28s wg 0 <-> %2d (xcc %d / %d): %s one-way %.0f ns, %.2f sweeps per hop\n", name, partner, hx[0], hx[partner],
           hf ? "FAILED" : "ok", hf ? 0.0 : ht * 10.0 / iters / 2.0, (double)hp / iters);
    hipFree(rows); hipFree(xcc); hipFree(ticks); hipFree(polls); hipFree(fail);
}

int main() {
    for (int partner : {8, 1}) {
        run<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_AGENT>("store agent / load agent", partner);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_AGENT>("store wg / load agent", partner);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_WORKGROUP>("store wg / load wg", partner);
    }
    u64* rows; long long* ticks;
    hipMalloc(&rows, 16384); hipMemset(rows, 0, 16384); hipMalloc(&ticks, 16);
    long long ht[2];
    hipLaunchKernelGGL((sweep<__HIP_MEMORY_SCOPE_AGENT>), dim3(1), dim3(64), 0, 0, rows, 2000, ticks);
    hipMemcpy(ht, ticks, 16, hipMemcpyDeviceToHost);
    printf("one sweep, agent loads, quiet row: %.0f ns\n", ht[0] * 10.0 / 2000);
    hipLaunchKernelGGL((sweep<__HIP_MEMORY_SCOPE_WORKGROUP>), dim3(1), dim3(64), 0, 0, rows, 2000, ticks);
    hipMemcpy(ht, ticks, 16, hipMemcpyDeviceToHost);
    printf("one sweep, plain loads, quiet row: %.0f ns\n", ht[0] * 10.0 / 2000);
    return 0;
}
