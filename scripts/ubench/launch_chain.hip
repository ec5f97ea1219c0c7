// Micro-benchmark: cost of a chain of dependent small launches on one stream (MI355X).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_chain scripts/ubench/launch_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Small { float* p; int n; };
struct Big { float* p; int n; long pad[240]; };  // ~1.9 KB

template <class A> __global__ void touch(A a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) a.p[i] += 1.0f;
}

template <class A> float run(int wgs, int chain, bool graph) {
    float* d; hipMalloc(&d, 1 << 20);
    A a{}; a.p = d; a.n = wgs * 64;
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    if (graph) {
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(touch<A>, dim3(wgs), dim3(384), 0, s, a);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    }
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, s);
        if (graph) hipGraphLaunch(ge, s);
        else for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(touch<A>, dim3(wgs), dim3(384), 0, s, a);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipFree(d);
    return best * 1000.f / chain;
}

int main() {
    const char* env = getenv("HIP_FORCE_DEV_KERNARG");
    printf("HIP_FORCE_DEV_KERNARG=%s\n", env ? env : "(unset)");
    for (int wgs : {1, 32, 256}) {
        printf("wgs=%3d  small-args eager %.2f us/launch  graph %.2f | big-args eager %.2f  graph %.2f\n", wgs,
               run<Small>(wgs, 400, false), run<Small>(wgs, 400, true), run<Big>(wgs, 400, false), run<Big>(wgs, 400, true));
    }
    return 0;
}
