#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
// reduced form of fat_persistent_kernel's ticket loop (bodies skipped)
struct P_t { int* ws; int tcap, Ls, ndir; unsigned spin_limit; int NS; };
__global__ void __launch_bounds__(256, 2) loop_kernel(P_t P, int* err) {
    __shared__ int s_ticket, s_ok;
    const int NS = P.NS, ncell = P.ndir * P.Ls;
    int* ws = P.ws;
    int* done = ws + 4;
    const int* need = done + ncell * P.tcap;
    const int4* desc = reinterpret_cast<const int4*>(need + ncell * P.tcap);
    const int total = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned long long t_start = wall_clock64();
    if (ws[3] != 0) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(err, 16); return; }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (;;) {
        if (wave == 0) {   // every lane of wave 0 walks the same path: no lane-divergent loop anywhere near a barrier
            int v = 0;
            if (lane == 0) v = atomicAdd(ws, 1);
            v = __builtin_amdgcn_readfirstlane(v);
            int ok = 1;
            if (v < total) {
                const int4 dv = desc[v / NS];
                const int c = __builtin_amdgcn_readfirstlane(dv.x), t = __builtin_amdgcn_readfirstlane(dv.y);
                const int i = c % P.Ls;
                const int* wa = done + c * P.tcap + (t > 0 ? t - 1 : 0);
                const int na = t > 0 ? __builtin_amdgcn_readfirstlane(need[c * P.tcap + (t - 1)]) : 0;
                const int* wb = done + (i > 0 ? c - 1 : c) * P.tcap + t;
                const int nb = i > 0 ? __builtin_amdgcn_readfirstlane(need[(c - 1) * P.tcap + t]) : 0;
                unsigned spins = 0;
                for (;;) {
                    const int a = __builtin_amdgcn_readfirstlane(__hip_atomic_load(wa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const int b = __builtin_amdgcn_readfirstlane(__hip_atomic_load(wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (a >= na && b >= nb) break;
                    __builtin_amdgcn_s_sleep(32);
                    const int e = (++spins & 63u) == 0u ? __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
                    if (spins > P.spin_limit || wall_clock64() - t_start > 300000000ull || e != 0) {
                        if (lane == 0) atomicOr(err, 2);
                        ok = 0;
                        break;
                    }
                }
            }
            if (lane == 0) { s_ticket = v; s_ok = ok; }
        }
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
        if (ticket >= total || !__builtin_amdgcn_readfirstlane(s_ok)) break;
        const int4 dsc_v = desc[ticket / NS];
        const int c = __builtin_amdgcn_readfirstlane(dsc_v.x), t = __builtin_amdgcn_readfirstlane(dsc_v.y);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done + c * P.tcap + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    const int Ls = 5, ndir = 2, ncell = 10, T = 50, NS = 16;
    std::vector<int> h(4 + 2 * ncell * T, 0);
    std::vector<int> d4;
    int nd = 0;
    for (int s = 0; s < T + Ls - 1; ++s)
        for (int c = 0; c < ncell; ++c) {
            int i = c % Ls, t = s - i;
            if (t < 0 || t >= T) continue;
            int cnt = 1 + (t % 3);
            h[4 + ncell * T + c * T + t] = cnt * NS;
            for (int k = 0; k < cnt; ++k) { d4.push_back(c); d4.push_back(t); d4.push_back(0); d4.push_back(1); ++nd; }
        }
    h[0] = 0; h[1] = nd * NS; h[2] = nd; h[3] = 0;
    h.insert(h.end(), d4.begin(), d4.end());
    int *ws, *err;
    hipMalloc(&ws, h.size() * 4); hipMalloc(&err, 4);
    hipMemcpy(ws, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(err, 0, 4);
    P_t P{ws, T, Ls, ndir, 20000u, NS};
    hipLaunchKernelGGL(loop_kernel, dim3(512), dim3(256), 40000, 0, P, err);
    hipError_t e = hipDeviceSynchronize();
    int he = -1, q = -1; hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&q, ws, 4, hipMemcpyDeviceToHost);
    printf("sync %d err %d queue %d total %d\n", (int)e, he, q, nd * NS);
    return 0;
}
