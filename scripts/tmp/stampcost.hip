#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long* out) {
    long long t0 = wall_clock64(); long long acc = 0;
    for (int i = 0; i < 1000; ++i) { acc += wall_clock64(); }
    long long t1 = wall_clock64();
    out[0] = t1 - t0; out[1] = acc;
}
int main() { long long* o; hipMalloc(&o, 16); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); printf("wall_clock64: %.1f ns each\n", h[0] * 10.0 / 1000); return 0; }
