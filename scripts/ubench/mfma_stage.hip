// The dataflow kernel's product stage in isolation: 4 rows x H = 256 operand block, one wave = 8 units x 3 gates,
// v_mfma_f32_4x4x1 + reduce-scatter, against a CPU sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float df_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float df_row_pair_sum(float x) {
    // (inline asm: with both operands holding the same value hipcc 7.2 folds the builtin's two results into one and
    // emits v1 + v1 behind the swap; volatile also keeps it out of the divergent branch that uses the sum)
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// W [3][8 units][256], a [4 rows][256]  ->  out [3][8 units][4 rows]
__global__ void k(const float* W, const float* a, float* out, float* dbg) {
    const int lane = threadIdx.x, quad = lane >> 5, ks = (lane >> 2) & 7, x = lane & 3;
    const bool s0 = ks & 1, s1 = ks & 2;
    f4v acc[3] = {(f4v){0, 0, 0, 0}, (f4v){0, 0, 0, 0}, (f4v){0, 0, 0, 0}};
    for (int t = 0; t < 32; ++t) {
        const float b = a[x * 256 + ks * 32 + t];
        for (int g = 0; g < 3; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(W[(g * 8 + 4 * quad + x) * 256 + ks * 32 + t], b, acc[g], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) dbg[lane * 4 + i] = acc[0][i];
    for (int g = 0; g < 3; ++g) {
        // (every DPP op outside the selects: inside a divergent branch its source lanes would be switched off)
        const float u0 = acc[g][0] + df_dpp<0x104>(acc[g][0]), u1 = acc[g][1] + df_dpp<0x104>(acc[g][1]);   // + the lane 4 up
        const float u2 = acc[g][2] + df_dpp<0x114>(acc[g][2]), u3 = acc[g][3] + df_dpp<0x114>(acc[g][3]);   // + the lane 4 down
        const float e0 = s0 ? u2 : u0, e1 = s0 ? u3 : u1;   // units (2, 3) | (0, 1) of the quad
        const float f0 = e0 + df_dpp<0x108>(e0), f1 = e1 + df_dpp<0x118>(e1);
        const float f = s1 ? f1 : f0;
        if (g == 0) { dbg[256 + lane] = e0; dbg[320 + lane] = e1; dbg[384 + lane] = f; }
        const float tot = df_row_pair_sum(f);
        const int unit = 4 * quad + 2 * (ks & 1) + ((ks >> 1) & 1);
        if ((lane & 16) == 0) out[(g * 8 + unit) * 4 + x] = tot;
    }
}
int main() {
    static float W[3 * 8 * 256], a[4 * 256], o[96];
    for (float& v : W) v = (rand() % 2001 - 1000) / 1000.0f;
    for (float& v : a) v = (rand() % 2001 - 1000) / 1000.0f;
    float *dW, *da, *dout, *ddbg; static float dbg[512]; hipMalloc(&ddbg, sizeof(dbg));
    hipMalloc(&dW, sizeof(W)); hipMalloc(&da, sizeof(a)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(dW, W, sizeof(W), hipMemcpyHostToDevice); hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice);
    hipMemset(dout, 0, sizeof(o));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dW, da, dout, ddbg);
    hipMemcpy(dbg, ddbg, sizeof(dbg), hipMemcpyDeviceToHost);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    {   // stage by stage, gate 0
        double wm = 0, w0 = 0, wf = 0;
        static double part[64][4];
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
            const int quad = l >> 5, ks = (l >> 2) & 7, x = l & 3;
            double r = 0;
            for (int t = 0; t < 32; ++t) r += (double)W[(4 * quad + i) * 256 + ks * 32 + t] * a[x * 256 + ks * 32 + t];
            part[l][i] = r;
            wm = fmax(wm, fabs(r - dbg[l * 4 + i]));
        }
        for (int l = 0; l < 64; ++l) {
            const int ks = (l >> 2) & 7; const bool s0 = ks & 1, s1 = ks & 2;
            const double e0 = s0 ? part[l][2] + part[l - 4][2] : part[l][0] + part[l + 4][0];
            w0 = fmax(w0, fabs(e0 - dbg[256 + l]));
        }
        static double E0[64], E1[64], F[64];
        for (int l = 0; l < 64; ++l) {
            const int ks = (l >> 2) & 7; const bool s0 = ks & 1;
            E0[l] = s0 ? part[l][2] + part[l - 4][2] : part[l][0] + part[l + 4][0];
            E1[l] = s0 ? part[l][3] + part[l - 4][3] : part[l][1] + part[l + 4][1];
        }
        double w1 = 0;
        for (int l = 0; l < 64; ++l) {
            const int ks = (l >> 2) & 7; const bool s1 = ks & 2;
            F[l] = s1 ? E1[l] + E1[l - 8] : E0[l] + E0[l + 8];
            wf = fmax(wf, fabs(F[l] - dbg[384 + l]));
            w1 = fmax(w1, fabs(E1[l] - dbg[320 + l]));
        }
        printf("after mfma: worst %.3g; after step 1 (e0): worst %.3g (e1) %.3g; after step 2: %.3g\n", wm, w0, w1, wf);
        double wt = 0;
        for (int l = 0; l < 64; ++l) if (!(l & 16)) {
            const int quad = l >> 5, ks = (l >> 2) & 7, x = l & 3, unit = 4 * quad + 2 * (ks & 1) + ((ks >> 1) & 1);
            wt = fmax(wt, fabs(F[l] + F[l ^ 16] - o[unit * 4 + x]));
        }
        printf("after step 3 (gate 0): %.3g\n", wt);
    }
    double worst = 0;
    for (int g = 0; g < 3; ++g) for (int u = 0; u < 8; ++u) for (int r = 0; r < 4; ++r) {
        double ref = 0;
        for (int kk = 0; kk < 256; ++kk) ref += (double)W[(g * 8 + u) * 256 + kk] * a[r * 256 + kk];
        const double e = fabs(ref - o[(g * 8 + u) * 4 + r]);
        if (e > worst) worst = e;
        if (e > 1e-3) printf("gate %d unit %d row %d: %g want %g\n", g, u, r, o[(g * 8 + u) * 4 + r], ref);
    }
    printf("worst |error| %.3g\n", worst);
    return 0;
}
