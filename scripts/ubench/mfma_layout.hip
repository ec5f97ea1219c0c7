// Register layout of v_mfma_f32_4x4x1_16b_f32 and of v_permlane16_swap on gfx950, checked against the expectation the
// dataflow kernel's product stage is written to:
//   block = lane / 4;  A: lane holds A[i = lane % 4];  B: lane holds B[j = lane % 4];  D: register r of lane = D[i = r][j = lane % 4]
//   permlane16_swap(a, b): a.row1 <-> b.row0, a.row3 <-> b.row2 (rows of 16 lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, unsigned* sw) {
    const int l = threadIdx.x;
    const float a = 1.0f + l, b = 100.0f * (1 + l);
    f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, (f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    out[4 * l + 0] = d.x; out[4 * l + 1] = d.y; out[4 * l + 2] = d.z; out[4 * l + 3] = d.w;
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(1000 + l), false, false);
    sw[2 * l] = r[0]; sw[2 * l + 1] = r[1];
}
int main() {
    float* out; unsigned* sw; hipMalloc(&out, 64 * 16); hipMalloc(&sw, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sw);
    float h[256]; unsigned s[128];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(s, sw, sizeof(s), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4;
            const float want = (1.0f + 4 * blk + r) * 100.0f * (1 + 4 * blk + j);
            if (h[4 * l + r] != want) { if (bad < 8) printf("lane %d reg %d: %g want %g\n", l, r, h[4 * l + r], want); ++bad; }
        }
    printf("mfma 4x4x1 layout: %s\n", bad ? "DIFFERENT" : "as expected");
    printf("permlane16_swap(a = lane, b = 1000 + lane): a' at lanes 0,16,32,48 = %u %u %u %u; b' = %u %u %u %u\n",
           s[0], s[32], s[64], s[96], s[1], s[33], s[65], s[97]);
    return 0;
}
