// gemm_f32.hip - C[M,Nc] = A[M,K] * W[Nc,K]^T + bias, fp32 in / fp32 MFMA accumulate.
//
// This is the input-side half of nn.GRUCell (W_ih u + b_ih, ogbg-code/model/dagnn.py:181) hoisted
// out of the recurrence: u (the embedding x for the first stacked layer, the previous layer's
// hidden states afterwards) is known for every node before that layer's recurrence starts, so it
// is ONE batched GEMM per layer instead of one GEMV per frontier node.
//
// gfx950 design: v_mfma_f32_32x32x2_f32 (exact fp32, bit-equal to an fmaf chain - bf16/xf32 are
// ruled out by the 1e-4 parity bound after hundreds of recurrent steps).  128x128x16 block tile,
// 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator registers).  Both operands
// are K-contiguous in HBM ("NT"), loaded as float4 and stored k-major in LDS with a +4 pad so that
// the per-MFMA fragment reads (lane&31 -> consecutive rows, lane>>5 -> k) are conflict-free
// ds_read_b32.  Register prefetch + double-buffered LDS: one barrier per K tile.  Blocks are
// remapped so that the Nc/128 column tiles of one row tile run on the same XCD and share A in L2.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmGroups {
    const float* A[DAGNN_MAX_GROUPS];
    const float* W[DAGNN_MAX_GROUPS];
    const float* bias[DAGNN_MAX_GROUPS];
    float* C[DAGNN_MAX_GROUPS];
};

// load one [128 x 16] K-contiguous tile slice owned by this thread: 2 float4 (rows r0, r0+64)
template <bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t rows, int K, int ld, int64_t row0,
                                          int k0, int tid, float4 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        const int64_t r = row0 + (idx >> 2);
        const int k = k0 + (idx & 3) * 4;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
            const float* p = P + r * ld + k;
            if (VEC && k + 3 < K) {
                t = *reinterpret_cast<const float4*>(p);
            } else {
                if (k < K) t.x = p[0];
                if (k + 1 < K) t.y = p[1];
                if (k + 2 < K) t.z = p[2];
                if (k + 3 < K) t.w = p[3];
            }
        }
        v[i] = t;
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const float4 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 2, kq = (idx & 3) * 4;
        S[(kq + 0) * LDT + r] = v[i].x;
        S[(kq + 1) * LDT + r] = v[i].y;
        S[(kq + 2) * LDT + r] = v[i].z;
        S[(kq + 3) * LDT + r] = v[i].w;
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) gemm_nt_bias_kernel(GemmGroups G, int64_t M, int Nc, int K, int lda, int ldw,
                                                            int ldc, int tiles_m, int tiles_n) {
    __shared__ float As[2][BK * LDT];
    __shared__ float Bs[2][BK * LDT];
    const int g = blockIdx.y;
    const float* __restrict__ A = G.A[g];
    const float* __restrict__ W = G.W[g];
    const float* __restrict__ bias = G.bias[g];
    float* __restrict__ C = G.C[g];

    // XCD-aware, bijective remap: hardware puts block b on XCD b % 8; give each XCD a contiguous
    // run of tiles so the tiles_n column tiles of a row tile hit the same L2.
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = swz / tiles_n, tn = swz - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fr = lane & 31, fk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[2], rb[2];
    load_tile<VEC>(A, M, K, lda, m0, 0, tid, ra);
    load_tile<VEC>(W, Nc, K, ldw, n0, 0, tid, rb);
    store_tile(As[0], tid, ra);
    store_tile(Bs[0], tid, rb);
    __syncthreads();

    const int nk = (K + BK - 1) / BK;
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) {
            load_tile<VEC>(A, M, K, lda, m0, (t + 1) * BK, tid, ra);
            load_tile<VEC>(W, Nc, K, ldw, n0, (t + 1) * BK, tid, rb);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = (2 * kk + fk) * LDT;
            const float a0 = as[krow + wm + fr], a1 = as[krow + wm + 32 + fr];
            const float b0 = bs[krow + wn + fr], b1 = bs[krow + wn + 32 + fr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (t + 1 < nk) {
            store_tile(As[cur ^ 1], tid, ra);
            store_tile(Bs[cur ^ 1], tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + j * 32 + fr;
        if (col >= Nc) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
                if (row < M) C[row * ldc + col] = acc[i][j][e] + bv;
            }
        }
    }
}

// ---- K a multiple of 32, 16-byte aligned rows (every GRU matrix of the models here): the same 128 x 128 block tile on
// 128-bit LDS traffic.  The kernel above stores every operand float with its own ds_write_b32 (k-major transpose), reads one
// ds_read_b32 per MFMA operand and meets a barrier every 16 k: 52 % (headline, K = 256) to 63 % (cfg 5, K = 512) of the
// fp32 matrix peak.  Here a stage is 32 k, both operands stay ROW-major in LDS (pitch 36 floats: the 16 rows of a
// ds_read_b128 lane group fall on 16 different bank quads, no swizzle), a thread moves four 16-byte chunks per operand and
// stage (global_load_dwordx4 -> ds_write_b128, one stage ahead in registers), and a lane's A / B fragment for FOUR
// consecutive MFMAs is one ds_read_b128: lane (i, hh) holds k = 8 g + 4 hh + q, q = 0..3 - instruction q multiplies k = 8 g + q
// (lanes 0-31) and 8 g + 4 + q (lanes 32-63).  The sum over k is the same set of exact fp32 fmaf steps in another order
// (deterministic; which kernel runs depends on (K, alignment) only, never on M or the group count).
constexpr int GK = 32, GP = GK + 4;

__device__ __forceinline__ void load_rows4(const float* __restrict__ P, int64_t rows, int ld, int64_t row0, int k0, int tid,
                                           float4 (&v)[4]) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t r = row0 + (tid >> 3) + 32 * i;
        r = r < rows ? r : rows - 1;   // rows past the end repeat the last one (their outputs are never stored)
        v[i] = *reinterpret_cast<const float4*>(P + r * ld + k0 + 4 * c);
    }
}

__device__ __forceinline__ void store_rows4(float* __restrict__ S, int tid, const float4 (&v)[4]) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(S + ((tid >> 3) + 32 * i) * GP + 4 * c) = v[i];
}

__global__ void __launch_bounds__(256, 2) gemm_nt_bias_k32_kernel(GemmGroups G, int64_t M, int Nc, int K, int lda, int ldw,
                                                                  int ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];   // [2][A: 128 x GP | B: 128 x GP]
    const int g = blockIdx.y;
    const float* __restrict__ A = G.A[g];
    const float* __restrict__ W = G.W[g];
    const float* __restrict__ bias = G.bias[g];
    float* __restrict__ C = G.C[g];
    const int nwg = tiles_m * tiles_n;   // XCD-aware remap, as above
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = swz / tiles_n, tn = swz - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fr = lane & 31, fk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[4], rb[4];
    load_rows4(A, M, lda, m0, 0, tid, ra);
    load_rows4(W, Nc, ldw, n0, 0, tid, rb);
    store_rows4(gsm, tid, ra);
    store_rows4(gsm + BM * GP, tid, rb);
    __syncthreads();
    const int nk = K / GK;
    for (int t = 0; t < nk; ++t) {
        const int tn1 = min(t + 1, nk - 1);   // unconditional (the last stage re-reads itself): no load behind a branch
        load_rows4(A, M, lda, m0, tn1 * GK, tid, ra);
        load_rows4(W, Nc, ldw, n0, tn1 * GK, tid, rb);
        const float* as = gsm + (t & 1) * (2 * BM * GP);
        const float* bs = as + BM * GP;
#pragma unroll
        for (int kg = 0; kg < GK / 8; ++kg) {
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const float4*>(as + (wm + 32 * i + fr) * GP + 8 * kg + 4 * fk);
                bf[i] = *reinterpret_cast<const float4*>(bs + (wn + 32 * i + fr) * GP + 8 * kg + 4 * fk);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float av = q == 0 ? af[i].x : q == 1 ? af[i].y : q == 2 ? af[i].z : af[i].w;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float bv = q == 0 ? bf[j].x : q == 1 ? bf[j].y : q == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        float* ns = gsm + ((t + 1) & 1) * (2 * BM * GP);
        store_rows4(ns, tid, ra);
        store_rows4(ns + BM * GP, tid, rb);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + j * 32 + fr;
        if (col >= Nc) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fk;
                if (row < M) C[row * ldc + col] = acc[i][j][e] + bv;
            }
        }
    }
}

// ---- small weight matrices (Nc * K <= 32 K words, or K <= 16; K a multiple of 4, 16-byte aligned rows): one output per thread.  The D-VAE encoders' input
// products have K = 8 / 10 (one-hot vertex types) and their final projection is 64 x 256 x 128: the 128 x 128 MFMA tile
// kernel above runs those on 1-12 workgroups in ~21 us (rocprofv3: 42 of the 125 us of kernels in a cfg 1 forward).
// A workgroup = 64 columns x 4 rows: the four waves walk the same 64 rows of W (L1 reuse), a row of A is a broadcast.
// The choice depends on (Nc, K) and the alignment only - never on M or the group count - so a batch and its sub-batches, a grouped
// launch and separate launches, take the same kernel and give the same bits (bias first, then k ascending).
constexpr int SMALL_WK = 1 << 15;
__global__ void __launch_bounds__(256) gemm_small_kernel(GemmGroups G, int64_t M, int Nc, int K, int lda, int ldw, int ldc,
                                                         int col_blocks) {
    const int g = blockIdx.y;
    const int cb = blockIdx.x % col_blocks;
    const int64_t rb = blockIdx.x / col_blocks;
    const int n = cb * 64 + (threadIdx.x & 63);
    const int64_t m = rb * 4 + (threadIdx.x >> 6);
    if (m >= M || n >= Nc) return;
    const float* a = G.A[g] + m * lda;
    const float* w = G.W[g] + (int64_t)n * ldw;
    float acc = G.bias[g] ? G.bias[g][n] : 0.f;
    for (int k = 0; k < K; k += 4) {   // (K % 4 == 0, 16-byte aligned rows: checked by the caller)
        const float4 av = *reinterpret_cast<const float4*>(a + k), wv = *reinterpret_cast<const float4*>(w + k);
        acc = fmaf(av.x, wv.x, acc); acc = fmaf(av.y, wv.y, acc); acc = fmaf(av.z, wv.z, acc); acc = fmaf(av.w, wv.w, acc);
    }
    G.C[g][m * ldc + n] = acc;
}

}  // namespace

extern "C" int dagnn_gemm_nt_bias(const dagnn_gemm_group* groups, int num_groups, int64_t M, int Nc, int K,
                                  int lda, int ldw, int ldc, void* stream) {
    if (!groups || num_groups <= 0 || num_groups > DAGNN_MAX_GROUPS) return DAGNN_EINVAL;
    if (M < 0 || Nc <= 0 || K <= 0 || lda < K || ldw < K || ldc < Nc) return DAGNN_EINVAL;
    if (M == 0) return DAGNN_OK;
    GemmGroups G;
    bool vec = (lda % 4 == 0) && (ldw % 4 == 0);
    for (int g = 0; g < DAGNN_MAX_GROUPS; ++g) {
        const dagnn_gemm_group& s = groups[g < num_groups ? g : 0];
        if (!s.A || !s.W || !s.C) return DAGNN_EINVAL;
        G.A[g] = s.A; G.W[g] = s.W; G.bias[g] = s.bias; G.C[g] = s.C;
        vec = vec && (((uintptr_t)s.A & 15) == 0) && (((uintptr_t)s.W & 15) == 0);
    }
    // (16-byte rows only: with scalar loads - cfg 4's K = 10 - the tile kernel is the faster one, 13.9 against 23.2 us)
    if (vec && (K & 3) == 0 && (K <= 16 || (int64_t)Nc * K <= SMALL_WK)) {
        const int col_blocks = (Nc + 63) / 64;
        const int64_t nblk = (M + 3) / 4 * col_blocks;
        if (nblk >= (int64_t(1) << 31)) return DAGNN_EINVAL;
        dim3 grid((unsigned)nblk, (unsigned)num_groups);
        hipLaunchKernelGGL(gemm_small_kernel, grid, dim3(256), 0, (hipStream_t)stream, G, M, Nc, K, lda, ldw, ldc, col_blocks);
        DAGNN_CHECK_LAUNCH();
        return DAGNN_OK;
    }
    const int64_t tiles_m64 = (M + BM - 1) / BM;
    const int tiles_n = (Nc + BN - 1) / BN;
    if (tiles_m64 * tiles_n >= (int64_t(1) << 31)) return DAGNN_EINVAL;
    const int tiles_m = (int)tiles_m64;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)num_groups);
    static const bool k32_off = getenv("DAGNN_AMD_GEMM_K32") && getenv("DAGNN_AMD_GEMM_K32")[0] == '0';   // A/B knob
    if (vec && K % GK == 0 && !k32_off) {
        constexpr size_t lds = (size_t)2 * 2 * BM * GP * sizeof(float);   // 72 KB: two blocks per CU
        static std::atomic<unsigned long long> attr_done{0ull};
        if (dagnn_lds_attr_once(attr_done, reinterpret_cast<const void*>(gemm_nt_bias_k32_kernel), (int)lds) != hipSuccess)
            return DAGNN_EHIP(hipGetLastError());
        hipLaunchKernelGGL(gemm_nt_bias_k32_kernel, grid, dim3(256), lds, (hipStream_t)stream, G, M, Nc, K, lda, ldw, ldc,
                           tiles_m, tiles_n);
    } else if (vec)
        hipLaunchKernelGGL(gemm_nt_bias_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, G, M, Nc, K, lda, ldw,
                           ldc, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL(gemm_nt_bias_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, G, M, Nc, K, lda,
                           ldw, ldc, tiles_m, tiles_n);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
