// Micro-benchmark: does hipExtStreamCreateWithCUMask confine a stream's workgroups, and how are the mask bits numbered?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask scripts/ubench/cumask.hip && /tmp/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>

#define XCC_ID_REG (20 | (0 << 6) | (3 << 11))   // hwreg(HW_REG_XCC_ID, 0, 4)
#define HW_ID_REG (4 | (0 << 6) | (31 << 11))     // hwreg(HW_REG_HW_ID, 0, 32)

__global__ void where(int* out, int spin) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(XCC_ID_REG);
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(HW_ID_REG);
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

static void run(const char* name, hipStream_t s, int wgs) {
    int* d; hipMalloc(&d, wgs * 8);
    hipLaunchKernelGGL(where, dim3(wgs), dim3(64), 0, s, d, 20000);   // 200 us of spinning: all workgroups co-resident
    hipStreamSynchronize(s);
    std::vector<int> h(2 * wgs);
    hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
    std::set<int> cus; int per_xcc[8] = {0};
    for (int i = 0; i < wgs; ++i) {
        const int xcc = h[2 * i] & 15, hw = h[2 * i + 1];
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
        cus.insert((xcc << 8) | (se << 5) | (sh << 4) | cu);
        per_xcc[xcc & 7]++;
    }
    printf("%-28s %4d workgroups on %3zu distinct CUs; per XCC:", name, wgs, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    hipFree(d);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    hipStream_t plain; hipStreamCreate(&plain);
    run("plain stream", plain, 256);
    const uint32_t pats[3][8] = {{0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u},
                                 {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0},
                                 {0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu}};
    const char* names[3] = {"mask 0x55.. (alternating)", "mask low 128 bits", "mask low 16 of every 32"};
    for (int k = 0; k < 3; ++k) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, pats[k]);
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", names[k], hipGetErrorString(e)); continue; }
        run(names[k], s, 256);
        run(names[k], s, 128);
        hipStreamDestroy(s);
    }
    return 0;
}
