// Micro-benchmark: one-way hand-off latency between two workgroups through global memory, by scope
// of the store/load pair and by placement (same XCD / different XCD).  MI355X (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pingpong scripts/ubench/pingpong.hip && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>

#define XCC_ID_REG (20 | (0 << 6) | (3 << 11))   // hwreg(HW_REG_XCC_ID, 0, 4)

template <int SCOPE_ST, int SCOPE_LD>
__global__ void pingpong(unsigned long long* flags, int partner, int iters, int* xcc, long long* ticks, int* fail) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg(XCC_ID_REG);
    if (b != 0 && b != partner) return;
    if (threadIdx.x != 0) return;
    unsigned long long* mine = flags + (b == 0 ? 0 : 64);     // separate 512 B regions
    unsigned long long* theirs = flags + (b == 0 ? 64 : 0);
    long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (b == 0) __hip_atomic_store(mine, (unsigned long long)it, __ATOMIC_RELAXED, SCOPE_ST);
        unsigned spins = 0;
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE_LD) != (unsigned long long)it) {
            if (++spins > 20000000u) { *fail = 1; return; }
        }
        if (b != 0) __hip_atomic_store(mine, (unsigned long long)it, __ATOMIC_RELAXED, SCOPE_ST);
    }
    if (b == 0) *ticks = wall_clock64() - t0;
}

template <int SS, int SL> void run(const char* name, int partner) {
    unsigned long long* flags; int* xcc; long long* ticks; int* fail;
    hipMalloc(&flags, 4096); hipMemset(flags, 0, 4096);
    hipMalloc(&xcc, 64 * 4); hipMalloc(&ticks, 8); hipMalloc(&fail, 4); hipMemset(fail, 0, 4); hipMemset(ticks, 0, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((pingpong<SS, SL>), dim3(16), dim3(64), 0, 0, flags, partner, iters, xcc, ticks, fail);
    hipDeviceSynchronize();
    int hx[16]; long long ht; int hf;
    hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost); hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("%-34s partner wg %2d (xcc %d vs %d): %s one-way %.0f ns\n", name, partner, hx[0], hx[partner],
           hf ? "FAILED (never seen)" : "ok", hf ? 0.0 : ht * 10.0 / iters / 2.0);
    hipFree(flags); hipFree(xcc); hipFree(ticks); hipFree(fail);
}

int main() {
    for (int partner : {8, 1, 4}) {
        run<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_AGENT>("store agent / load agent", partner);
        run<__HIP_MEMORY_SCOPE_SYSTEM, __HIP_MEMORY_SCOPE_SYSTEM>("store system / load system", partner);
    }
    // workgroup scope only makes sense on the same XCD (L2 is the meeting point); bounded spins report failure
    run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_WORKGROUP>("store wg / load wg", 8);
    run<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_WORKGROUP>("store agent / load wg", 8);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_AGENT>("store wg / load agent", 8);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_WORKGROUP>("store wg / load wg", 1);
    return 0;
}
