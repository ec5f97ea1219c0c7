// Does an XCD's L2 keep read-only data across kernel launches?  8 x 32 workgroups (round-robin over the XCDs) read the
// same 2 MB buffer in 6 back-to-back launches; rocprofv3 --pmc FETCH_SIZE per launch answers it.
//   hipcc -O3 --offload-arch=gfx950 -o l2_persist l2_persist.hip && rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o l -- ./l2_persist
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) read_kernel(const float4* __restrict__ p, int n4, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x + (blockIdx.x / 8) * 256; i < n4; i += 256 * (gridDim.x / 8)) {
        const float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
__global__ void write_kernel(float* p, int n) {   // another kernel in between, as in the real schedule
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 1.f;
}
int main() {
    const int bytes = 2 << 20, n4 = bytes / 16;
    float4* buf; float* out; float* other;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4); hipMalloc(&other, 1 << 20);
    hipMemset(buf, 0, bytes);
    for (int it = 0; it < 6; ++it) {
        hipLaunchKernelGGL(read_kernel, dim3(256), dim3(256), 0, 0, buf, n4, out);
        if (it >= 3) hipLaunchKernelGGL(write_kernel, dim3(64), dim3(256), 0, 0, other, 1 << 18);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
