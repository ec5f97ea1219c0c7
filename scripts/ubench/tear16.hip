// Does a 16-byte write-through store ever TEAR against a 16-byte load on gfx950, across XCDs?  (The guide promises single-copy
// atomicity for 8 bytes; the dataflow kernel's 16-byte projection granules {tag, r, z, n} need it for 16.)
// Writers (the even workgroups) store {k, k ^ A, k ^ B, k ^ C} into 16-byte slots, k = 1, 2, ...; readers (the odd workgroups: with
// the dispatch rule "workgroup b runs on XCD b % 8" every reader's XCD differs from the XCD of the writer whose slots it reads)
// load the slots with the same instruction forms the kernel uses - global_store_dwordx4 / global_load_dwordx4 with sc1, and the
// plain-store form of the same-XCD hand-off - and count words that do not belong together.  Also counted: loads that saw a NEW
// value (the test only means something if readers really observe the writers' progress).
// hipcc --offload-arch=gfx950 -O3 -o tear16 tear16.hip && ./tear16 [iters] [plain_store]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr unsigned A = 0x9e3779b9u, B = 0x7f4a7c15u, C = 0x94d049bbu;
template <int PLAIN, int MISALIGN>
__global__ void __launch_bounds__(256) k(unsigned* buf, int iters, unsigned long long* tears, unsigned long long* news, unsigned long long* loads) {
    const int pair = blockIdx.x >> 1;
    // slot of thread t of pair p: 16 bytes each; MISALIGN: slots 8 bytes off a 16-byte boundary (a control: these MAY tear)
    unsigned* slot = buf + ((size_t)pair * 256 + threadIdx.x) * (MISALIGN ? 6 : 4) + (MISALIGN ? 2 : 0);
    if ((blockIdx.x & 1) == 0) {
        for (int i = 1; i <= iters; ++i) {
            const u4v v = {(unsigned)i, (unsigned)i ^ A, (unsigned)i ^ B, (unsigned)i ^ C};
            if (PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(slot), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slot), "v"(v) : "memory");
            if ((i & 15) == 0) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        unsigned long long t = 0, n = 0, l = 0;
        unsigned last = 0;
        for (int i = 0; i < iters; ++i) {
            u4v v;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
            ++l;
            if (v.x != 0 || v.y != 0) {
                if ((v.x ^ A) != v.y || (v.x ^ B) != v.z || (v.x ^ C) != v.w) ++t;
                if (v.x != last) { ++n; last = v.x; }
            }
        }
        atomicAdd(tears, t); atomicAdd(news, n); atomicAdd(loads, l);
    }
}
template <int PLAIN, int MIS>
void run(int iters, const char* what) {
    unsigned* buf;
    unsigned long long* cnt;
    const int wgs = 240;
    hipMalloc(&buf, (size_t)wgs * 256 * 32);
    hipMemset(buf, 0, (size_t)wgs * 256 * 32);
    hipMalloc(&cnt, 3 * sizeof(unsigned long long));
    hipMemset(cnt, 0, 3 * sizeof(unsigned long long));
    hipLaunchKernelGGL((k<PLAIN, MIS>), dim3(wgs), dim3(256), 0, 0, buf, iters, cnt, cnt + 1, cnt + 2);
    hipDeviceSynchronize();
    unsigned long long h[3];
    hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s loads %llu, saw a new value %llu times, TORN %llu\n", what, h[2], h[1], h[0]);
    hipFree(buf); hipFree(cnt);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    run<0, 0>(iters, "sc1 store / sc1 load, 16-byte aligned:");
    run<1, 0>(iters, "plain store / sc1 load, 16-byte aligned:");
    run<0, 1>(iters, "sc1 store / sc1 load, 8 bytes off (control):");
    return 0;
}
