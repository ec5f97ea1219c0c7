// What does the compute wave's state store cost, by lane layout?  (round 6: with the stores compiled out a free-running compute
// wave of the dataflow kernel is 24 % faster.)  4 waves per workgroup (one per SIMD), 240 workgroups; every iteration = WORK
// dependent v_fma (the rest of a block) + one 8-byte store per live lane into 4 "node rows" of a [N, 256] granule buffer (+ one
// 4-byte store into a [N, 272] float buffer), the rows picked pseudo-randomly per iteration like the schedule's nodes.
//   pattern 0  no stores
//   pattern 1  32 live lanes, lane x = lane & 3 picks the ROW, (lane >> 2) the unit: every lane of an instruction in another row
//   pattern 2  lane x picks the unit inside a quad of 4, lane bits 2-3 the row: 32 contiguous bytes per 4 lanes
//   pattern 3  8 consecutive lanes = the 8 units of one row (64 contiguous bytes), lanes 0-31 live
//   pattern 4  all 64 lanes live, 16 lanes per row (128 contiguous bytes) - what a 16-unit wave would store
// hipcc --offload-arch=gfx950 -O3 -o store_cost store_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PAT, int SC1>
__global__ void k(unsigned long long* gran, float* hrow, int N, int iters, int work, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float c = lane;
    unsigned seed = blockIdx.x * 977u + wave * 131u + 7u;
    int row_sel, unit;
    bool live;
    if (PAT == 1) { row_sel = lane & 3; unit = 8 * wave + 4 * (lane >> 5) + ((lane >> 2) & 3); live = (lane & 16) == 0; }
    else if (PAT == 2) { row_sel = (lane >> 2) & 3; unit = 8 * wave + 4 * (lane >> 5) + (lane & 3); live = (lane & 16) == 0; }
    else if (PAT == 3) { row_sel = (lane >> 3) & 3; unit = 8 * wave + (lane & 7); live = lane < 32; }
    else { row_sel = lane >> 4; unit = 8 * wave + (lane & 15); live = true; }
    const int slice = (blockIdx.x % 8) * 32;
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        for (int q = 0; q < work; ++q) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(c));
        if (PAT != 0) {
            seed = seed * 1664525u + 1013904223u;
            const unsigned base = (seed >> 8) % (unsigned)(N - 4);
            const unsigned v = base + row_sel;
            if (live) {
                const unsigned long long g = ((unsigned long long)(i + 1) << 32) | __float_as_uint(c);
                unsigned long long* p = gran + (size_t)v * 256 + slice + unit;
                if (SC1) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = g;
                hrow[(size_t)v * 272 + slice + unit] = c;
            }
        }
    }
    const unsigned long long c1 = clock64();
    if (lane == 0) out[blockIdx.x * 4 + wave] = c1 - c0;
    if (c == 12345.f) sink[0] = c;
}
template <int PAT, int SC1>
double run(unsigned long long* gran, float* hrow, int N, int work, unsigned long long* d_out, float* sink) {
    const int iters = 2000, grid = 240;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<PAT, SC1>), dim3(grid), dim3(256), 0, 0, gran, hrow, N, iters, work, d_out, sink);
        hipDeviceSynchronize();
    }
    static unsigned long long h[240 * 4];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid * 4; ++i) s += (double)h[i];
    return s / (grid * 4) / iters;   // clock64 ticks (100 MHz) per iteration
}
int main() {
    const int N = 16561;
    unsigned long long* gran; float* hrow; unsigned long long* d_out; float* sink;
    hipMalloc(&gran, (size_t)N * 256 * 8); hipMalloc(&hrow, (size_t)N * 272 * 4); hipMalloc(&d_out, 240 * 4 * 8); hipMalloc(&sink, 64);
    hipMemset(gran, 0, (size_t)N * 256 * 8);
    for (int work : {0, 100, 300}) {
        const double t0 = run<0, 0>(gran, hrow, N, work, d_out, sink);
        printf("work %3d v_fma: no stores %.1f ns per iteration\n", work, t0 * 10);
        printf("   plain stores : rows-per-lane %.1f | quad %.1f | 8 lanes %.1f | 16 lanes x 64 live %.1f   (ns per iteration beyond no stores)\n",
               (run<1, 0>(gran, hrow, N, work, d_out, sink) - t0) * 10, (run<2, 0>(gran, hrow, N, work, d_out, sink) - t0) * 10,
               (run<3, 0>(gran, hrow, N, work, d_out, sink) - t0) * 10, (run<4, 0>(gran, hrow, N, work, d_out, sink) - t0) * 10);
        printf("   sc1 granules : rows-per-lane %.1f | quad %.1f | 8 lanes %.1f | 16 lanes x 64 live %.1f\n",
               (run<1, 1>(gran, hrow, N, work, d_out, sink) - t0) * 10, (run<2, 1>(gran, hrow, N, work, d_out, sink) - t0) * 10,
               (run<3, 1>(gran, hrow, N, work, d_out, sink) - t0) * 10, (run<4, 1>(gran, hrow, N, work, d_out, sink) - t0) * 10);
    }
    return 0;
}
