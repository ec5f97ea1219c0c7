// fat.hip - the fat launches of the lock-step schedule on the matrix cores: ONE kernel per launch, 64 frontier rows
// x one 32-unit slice of one cell per workgroup, the gather / segment soft-max fused into the staging of the A tile.
//
// Reference path replaced (ogbg-code/model/dagnn.py:144-182 at one topological layer, every cell of the launch):
//   ps_h = AttnConv(...)[layer]   (:175-179, message :366-373, PyG softmax + scatter_add)
//   inp  = GRUCell(inp, ps_h)     (:181)      h[d][i][layer] += inp (:182)
//
// Shape of the work (cfg 5: H = 512, L = 5, both directions): a launch holds 10 cells x 40..3000 rows, a row of a
// cell is [1 x 1024] x [1024 x 1536] (input | hidden side) = 3.1 MFLOP of exact-fp32 products.  Rounds 1-4 ran these
// launches as a gather kernel (aggregates through HBM) + 32-row x 32-unit tiles that re-streamed a 393 KB weight
// slice per 32 rows and staged K in 256-wide chunks behind __syncthreads without a second buffer.  Here:
//   * a tile is 64 rows x 32 units (x 3 gates): a weight pass is amortised over 64 rows, 16 slices x ceil(rows / 64)
//     workgroups per cell, two workgroups (4 waves each) per CU so that one's prologue / epilogue / barriers hide
//     behind the other's products (a launch that fits one workgroup per CU asks for more LDS than two can share);
//   * K is walked in stages of 64: 32 k of the input side + 32 k of the hidden side (or 64 of the one side a cell has:
//     stacked layer 0 reads gi0 from the batched GEMM, layer 0 of a direction has no predecessors).  Wave w = (side
//     hs = w >> 1, k half sub = w & 1) multiplies ALL rows of the tile by its 16 k of side hs on
//     v_mfma_f32_16x16x4_f32: 4 row blocks x 6 column blocks = 24 accumulators, every B fragment is fetched ONCE per
//     workgroup - straight from global memory in fragment order (dagnn_pack_mfma: one contiguous 1 KiB per wave-load)
//     into one of two register sets, a whole stage ahead.  16x16x4 and not 32x32x2: an fp32 MFMA holds the SIMD's issue
//     port for its whole duration and the OTHER waves of the SIMD get exactly one vector instruction in per MFMA
//     (scripts/ubench/mfma_share.hip: a partner's v_fma takes 5.8 cycles alone, 37.8 beside 16x16x4, 69.8 beside
//     32x32x2) - the partner workgroup's prologue / epilogue runs at that rate;
//   * the A tile of a stage ([64 rows x 64 k], 16 KB, two LDS slots) is built by all 256 threads: a thread owns a
//     16-byte chunk of two rows, the loads of stage s + 1 are in flight while the products of stage s run; hidden
//     side: a = sum_e alpha_e h[pred_e] over a row's one or two predecessors (alpha from the prologue: PyG's
//     exp(x - max) / (sum + 1e-16) over the partial scores stored behind the state rows), input side: the node's
//     lower-layer row.  The aggregate never goes to memory.  A row with three or four predecessors becomes a scratch row,
//     built in the prologue by all threads from its four (pointer, alpha) pairs; a row with more than four is
//     aggregated once per workgroup by one wave (the generic routine of the per-layer kernels); both are then read
//     like a single predecessor with alpha = 1.  Every load of the stage loop is unconditional: a load behind a branch
//     makes hipcc's wait counts assume the shortest queue, and the products then wait for loads they do not use;
//   * A fragments are ONE ds_read_b128 per four MFMAs (lane (i, kq) holds k = 16 g + 4 kq + q for q = 0..3; the B
//     fragments are packed with the same k map); rows are 256 B apart in LDS with the 16-byte chunk index XOR-ed by
//     (row & 15): reads and writes are bank-conflict free;
//   * epilogue: the two k halves of a side are added through LDS (the tile's outputs alias the A slots; plain read -
//     add - write a barrier apart: ds_add_f32 measured 0.3 us per wave instruction), gates (r, z, n) on the hardware
//     exp / rcp, h', partial scores as the per-layer kernels store them, row stores; the operands of the gate algebra
//     that do not depend on the products (aggregate, gi0) are loaded before the merge.
// What binds it (scripts/fat_stamps.py on a -DFAT_STAMPS build, cfg 5): a workgroup spends ~15 us in the prologue, ~29
// in the stage loop (two workgroups share the CU's matrix pipes: 2 x 11.2 us of products at the measured 2.2 GHz), ~9 in
// merge + gates; with the products compiled out (-DFAT_EXP_NOMFMA) the fat launches of a forward still take 7.0 of 12.3 ms
// - launch-level fill / drain and the per-tile dependent round trips, not the matrix pipe, are what is left.
// Exact fp32 (v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain per k quarter); results are bitwise run-to-run deterministic.
#include "frontier_dev.h"

namespace {

typedef float mf32x4 __attribute__((ext_vector_type(4)));

constexpr int FTM = 64;               // rows per tile
constexpr int FAT_THREADS = 256;
constexpr int FAT_SLOT = FTM * 64;    // floats per A slot: 64 rows x 64 k
constexpr int FAT_OP = 96;            // row pitch of the epilogue tile
constexpr int FAT_OUT = 2 * FTM * FAT_OP;   // floats of the epilogue tile: [side][row][3 gates x 32 units]
constexpr int FAT_MAIN = FAT_OUT > 2 * FAT_SLOT ? FAT_OUT : 2 * FAT_SLOT;

struct FatArgs {
    Cell cell[DAGNN_MAX_CELLS];
    int tile_start[DAGNN_MAX_CELLS + 1];   // prefix sums of 64-row tiles over the active cells
    int ncell, H, ld_h, R, vid_mod;
    unsigned epoch;
};

// Row pointers travel through LDS as plain integers and are read through the GLOBAL address space (a generic pointer
// loaded from memory would make every row load a flat_load, which waits on the LDS counter as well).
typedef float fat_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(uint64_t a) {
    const fat_f4v v = *reinterpret_cast<const __attribute__((address_space(1))) fat_f4v*>(a);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ldg1(uint64_t a) { return *reinterpret_cast<const __attribute__((address_space(1))) float*>(a); }

// gate non-linearities on the hardware exp / rcp (as dataflow.hip): sigma(x) = 1 / (1 + e^-x), tanh(x) = 1 - 2 / (1 + e^2x)
__device__ __forceinline__ float fat_sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fat_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// LDS float index of 16-byte chunk c (0..15) of row r inside an A slot
__device__ __forceinline__ int a_idx(int r, int c) { return r * 64 + ((c ^ (r & 15)) << 2); }

#ifdef FAT_STAMPS   // experiment build (scripts/fat_stamps.py): phase sums over the workgroups of every launch
__device__ unsigned long long fat_stamp_sum[8];
__device__ int fat_cu_res[4096];              // workgroups resident per (XCD, SE, SH, CU)
__device__ unsigned long long fat_res_hist[8];   // histogram: co-resident workgroups seen at a workgroup's start / end
#define FAT_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&fat_stamp_sum[i], t_ - T.t0); const_cast<FatTile&>(T).t0 = t_; } } while (0)
#else
#define FAT_STAMP(i) do { } while (0)
#endif

struct FatTile {            // what the prologue hands to the main part (all uniform over the workgroup)
    const Cell* C;
    int sl, slot0, nr;
    bool two, has_in, has_hid;
#ifdef FAT_STAMPS
    unsigned long long t0;
#endif
};

struct FatLds {
    float* ring;            // [2][64][64] A slots | [2][64][96] outputs
    uint64_t* hptr;         // [64][2] predecessor rows (or a finite dummy), as addresses
    uint64_t* iptr;         // [64] lower-layer row
    float* alpha;           // [64][2]
    int* node_s;            // [64]
    int* gen_s;             // [64] rows with more than 4 predecessors
    int* gen_n;             // [0] their count, [1] some row gathers 3-4 inline predecessors
};

__device__ __forceinline__ FatLds fat_lds(float* smem) {
    FatLds M;
    M.ring = smem;
    M.hptr = reinterpret_cast<uint64_t*>(smem + FAT_MAIN);
    M.iptr = M.hptr + FTM * 2;
    M.alpha = reinterpret_cast<float*>(M.iptr + FTM);
    M.node_s = reinterpret_cast<int*>(M.alpha + FTM * 2);
    M.gen_s = M.node_s + FTM;
    M.gen_n = M.gen_s + FTM;
    return M;
}

// ---- prologue.  P0: thread (row r, edge slot e) - record, score of inline predecessor e, soft-max over the quad.
// Rows with one or two predecessors are gathered by the stage loop itself; a row with three or four becomes a scratch
// row, built by all 256 threads from its four (pointer, alpha) pairs (P1: one round trip for every such row of the
// tile together); a row with more than four is aggregated by one wave (the generic routine of the per-layer kernels:
// its predecessors beyond the fourth sit in the plan's CSR, two more dependent round trips).
__device__ __forceinline__ void fat_prologue(const int32_t* __restrict__ plan, const PlanLayout& L, const FatArgs& S,
                                             const FatTile& T, const FatLds& M) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Cell& C = *T.C;
    const int H = S.H, ld_h = S.ld_h, d = C.dir;
    const int4* __restrict__ recs = reinterpret_cast<const int4*>(plan + L.rowrec[d]);
    const float* dummy = reinterpret_cast<const float*>(C.whh_m);               // finite floats: alpha = 0 slots read these
    const int nparts = H / PU;
    // the A slots are free until the stage loop starts: the materialised rows' operands live there
    uint64_t* mptr = reinterpret_cast<uint64_t*>(M.ring);        // [64][4]
    float* malpha = M.ring + FTM * 8;                            // [64][4]
    int* mat_s = reinterpret_cast<int*>(M.ring + FTM * 12);      // [64] rows with 3-4 predecessors
    if (tid < 2) M.gen_n[tid] = 0;
    __syncthreads();
    {
        const int r = tid >> 2, e = tid & 3;
        const bool live = r < T.nr;
        int4 rec0 = make_int4(0, 0, 0, 0), rec1 = rec0, rec2 = rec0, rec3 = rec0;
        if (live) {
            const int4* rp = recs + 4 * (int64_t)(T.slot0 + r);
            rec0 = rp[0]; rec1 = rp[1]; rec2 = rp[2]; rec3 = rp[3];
        }
        const int deg = (live && T.has_hid) ? rec0.z - rec0.y : 0;
        const int pj = e == 0 ? rec1.x : e == 1 ? rec1.y : e == 2 ? rec1.z : rec1.w;
        const float f0 = __int_as_float(e == 0 ? rec2.x : e == 1 ? rec2.z : e == 2 ? rec3.x : rec3.z);
        const float f1 = __int_as_float(e == 0 ? rec2.y : e == 1 ? rec2.w : e == 2 ? rec3.y : rec3.w);
        const bool mine = e < deg && deg <= 4;
        float lg = -INFINITY;
        if (mine && deg > 1) {
            float s = C.sscore ? C.sscore[pj] : score_of(C.h_out + (int64_t)pj * ld_h + H, nparts);
            if (C.vid) s += C.vid[pj % S.vid_mod];
            if (C.gain) {
                if (S.R >= 1) s = fmaf(C.gain[0], f0, s);
                if (S.R >= 2) s = fmaf(C.gain[1], f1, s);
            }
            lg = s;
        }
        // the four logits of the row in every lane of its quad; same operation order as the per-layer kernels
        const int q0 = lane & ~3;
        float l4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) l4[k] = __shfl(lg, q0 + k, 64);
        float al = 0.f;
        if (deg == 1) al = e == 0 ? 1.f : 0.f;
        else if (deg >= 2 && deg <= 4) {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) mx = fmaxf(mx, l4[k]);
            float sum = 0.f, mine_e = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ex = k < deg ? expf(l4[k] - mx) : 0.f;
                sum += ex;
                if (k == e) mine_e = ex;
            }
            al = mine_e / (sum + 1e-16f);
        }
        const float* p = mine ? C.h_out + (int64_t)pj * ld_h : dummy;
        if (deg > 2) {   // a scratch row stands in for the whole segment
            if (deg <= 4) { mptr[r * 4 + e] = reinterpret_cast<uint64_t>(p); malpha[r * 4 + e] = al; }
            if (e == 0) {
                if (deg <= 4) mat_s[atomicAdd(M.gen_n + 1, 1)] = r;
                else M.gen_s[atomicAdd(M.gen_n, 1)] = r;
            }
            p = e == 0 ? C.a_pre + (int64_t)(T.slot0 - C.row_base + r) * H : dummy;
            al = e == 0 ? 1.f : 0.f;
        }
        if (e < 2) {
            M.hptr[r * 2 + e] = reinterpret_cast<uint64_t>(p);
            M.alpha[r * 2 + e] = al;
        }
        if (e == 0) {
            M.node_s[r] = rec0.x;
            M.iptr[r] = reinterpret_cast<uint64_t>(T.has_in ? C.h_in + (int64_t)rec0.x * ld_h : dummy);
        }
    }
    __syncthreads();
    const int ng = M.gen_n[0], nm = M.gen_n[1];
    if (ng + nm > 0) {   // plain stores; the same workgroup reads the rows back behind the barrier
        const int H4 = H >> 2;
        for (int it = tid; it < nm * H4; it += FAT_THREADS) {   // P1: (row, 16-byte chunk) items, four loads in flight each
            const int r = mat_s[it / H4], cc = it % H4;
            float4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ldg4(mptr[r * 4 + e] + 16 * (uint64_t)cc);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) fma4(acc, malpha[r * 4 + e], v[e]);
            reinterpret_cast<float4*>(const_cast<float*>(C.a_pre) + (int64_t)(T.slot0 - C.row_base + r) * H)[cc] = acc;
        }
        if (ng > 0) {
            GranCtx G;
            G.epoch = S.epoch; G.err = nullptr;
            for (int g = wave; g < ng; g += FAT_THREADS / 64) {
                const int r = M.gen_s[g];
                const int4* rp = recs + 4 * (int64_t)(T.slot0 + r);
                const int4 rec0 = rp[0];
                float* out = const_cast<float*>(C.a_pre) + (int64_t)(T.slot0 - C.row_base + r) * H;
                aggregate<false>(C, plan + L.col[d], reinterpret_cast<const float*>(plan + L.eattr[d]), rec0.y, rec0.z, rp[1], rp[2],
                                 rp[3], H, ld_h, C.gain ? S.R : 0, S.vid_mod, H, out, lane, G);
            }
        }
        __syncthreads();
    }
}

// SIDES = which products the cell has at this layer: bit 0 hidden side (the layer has predecessors), bit 1 input side
// (stacked layers above 0); 0: stacked layer 0 at layer 0 of a direction - gates of gi0 and the biases only.
// TB = 16-row MFMA blocks of the tile: 4 (64 rows) or 2 (a tile of <= 32 rows).  Wave = (side hs, k half sub): it multiplies
// ALL rows of the tile by its 16 k of every stage, so each B fragment is fetched ONCE per workgroup; the two k halves of a
// side meet in LDS behind the loop.
template <int SIDES, int TB>
__device__ __forceinline__ void fat_main(const FatArgs& S, const FatTile& T, const FatLds& M) {
    constexpr int MD = 2;                 // predecessor rows a hidden-side item gathers (more: scratch row, see the prologue)
    constexpr int MB = TB;                // 16-row blocks per wave: the whole tile
    constexpr bool two = TB == 4;         // rows 32..63 of the tile exist
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Cell& C = *T.C;
    const int H = S.H, ld_h = S.ld_h, sl = T.sl, nr = T.nr;
    constexpr bool has_in = (SIDES & 2) != 0, has_hid = (SIDES & 1) != 0;
    constexpr bool IN0 = has_in, IN1 = SIDES == 2;   // side 0 / 1 of a stage is an input side (else a hidden side)
    float* ring = M.ring;

    // ---- stage plan: side h of a stage covers k [kmul * s + koff_h, +32) of its matrix
    // side 0 = input when both exist (its sums are gi), side 1 = hidden (gh); one side only: the two k halves of it
    constexpr bool both = IN0 != IN1;
    const int nstage = SIDES == 0 ? 0 : both ? (H >> 5) : (H >> 6);
    constexpr int kmul = both ? 32 : 64;
    constexpr bool side_in[2] = {IN0, IN1};
    constexpr int koff[2] = {0, both ? 0 : 32};

    // staging role: chunk c8 of rows rr and rr + 32, both sides
    const int c8 = tid & 7, rr = tid >> 3;
    uint64_t sp_in[2];             // input rows of my two staging rows (addresses)
    uint64_t sp_h[2][MD];          // predecessor rows
    float sa[2][MD];
    auto row_meta = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = rr + 32 * it;
            sp_in[it] = M.iptr[r] + 16 * c8;
#pragma unroll
            for (int e = 0; e < MD; ++e) { sp_h[it][e] = M.hptr[r * 2 + e] + 16 * c8; sa[it][e] = M.alpha[r * 2 + e]; }
        }
    };
    row_meta();
    // compute role: wave = (side hs, k half sub); lane = (row / column fi, k quarter kq) of a 16x16x4 fragment
    const int hs = wave >> 1, sub = wave & 1;
    const int fi = lane & 15, kq = lane >> 4;
    const bool my_in = hs == 0 ? side_in[0] : side_in[1];
    const int my_koff = hs == 0 ? koff[0] : koff[1];
    const int64_t blk_stride = (int64_t)(H >> 4) * 64;   // float4 elements between the 16-column blocks of a slice
    const float4* wp = (my_in ? C.wih_m : C.whh_m) + (int64_t)(sl * 6) * blk_stride + lane;

    mf32x4 acc[MB][6];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m][n][e] = 0.f;

    float4 areg[2][2][MD];   // [row half][side][predecessor] loads in flight
    auto a_issue = [&](int s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kmul * s + koff[h];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (it == 0 || two) {
                    if (side_in[h]) areg[it][h][0] = ldg4(sp_in[it] + 4 * k);
                    else {
#pragma unroll
                        for (int e = 0; e < MD; ++e) areg[it][h][e] = ldg4(sp_h[it][e] + 4 * k);
                    }
                }
            }
        }
    };
    auto a_commit = [&](int slot) {
        float* base = ring + slot * FAT_SLOT;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (it == 0 || two) {
                    float4 v;
                    if (side_in[h]) v = areg[it][h][0];
                    else {
                        v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int e = 0; e < MD; ++e) fma4(v, sa[it][e], areg[it][h][e]);
                    }
                    *reinterpret_cast<float4*>(base + a_idx(rr + 32 * it, 8 * h + c8)) = v;
                }
            }
        }
    };
    // B fragments (this wave's 16 k x the slice's six 16-column blocks): two register sets used in turn (even / odd stages) - a
    // whole stage of lead for the weight loads, no copies
    float4 bA[6], bB[6];
    auto b_issue = [&](int s, float4 (&dst)[6]) {
        const int k16 = ((kmul * s + my_koff) >> 4) + sub;
#pragma unroll
        for (int n = 0; n < 6; ++n) dst[n] = wp[n * blk_stride + (int64_t)k16 * 64];
    };
    // one stage: products of slot s & 1 with `cur`; meanwhile the next stage's B fragments -> `nxt`, its A tile -> the other
    // slot, the A rows of stage s + 2 -> registers
    auto stage = [&](int s, float4 (&cur)[6], float4 (&nxt)[6]) {
        const float* slot = ring + (s & 1) * FAT_SLOT;
        // every load below is unconditional (the last stages re-read the final stage's operands): a load behind a
        // branch makes hipcc's wait counts assume the shortest queue, and the products then wait for loads they do not use
        const int s1 = min(s + 1, nstage - 1), s2 = min(s + 2, nstage - 1);
        b_issue(s1, nxt);
        float4 af[MB];   // A fragments of the tile's row blocks: k = 16 sub + 4 kq + q for q = 0..3
#pragma unroll
        for (int m = 0; m < MB; ++m)
            af[m] = *reinterpret_cast<const float4*>(slot + a_idx(m * 16 + fi, 8 * hs + 4 * sub + kq));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float4 aw = af[m];
                const float aq = q == 0 ? aw.x : q == 1 ? aw.y : q == 2 ? aw.z : aw.w;
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    const float4 bw = cur[n];
                    const float bq = q == 0 ? bw.x : q == 1 ? bw.y : q == 2 ? bw.z : bw.w;
#ifdef FAT_EXP_NOMFMA   // timing experiment: everything but the products
                    acc[m][n][0] += aq * bq;
#else
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq, acc[m][n], 0, 0, 0);
#endif
                }
            }
        }
        a_commit((s + 1) & 1);
        a_issue(s2);
        __syncthreads();
    };

    if constexpr (SIDES != 0) {
#ifdef FAT_STAMPS
        const unsigned long long cyc0 = clock64();
#endif
        a_issue(0);
        b_issue(0, bA);
        a_commit(0);
        a_issue(min(1, nstage - 1));
        __syncthreads();
        for (int s = 0; s < nstage; s += 2) {
            stage(s, bA, bB);
            if (s + 1 < nstage) stage(s + 1, bB, bA);
        }
#ifdef FAT_STAMPS
        if (threadIdx.x == 0) atomicAdd(&fat_stamp_sum[6], (unsigned long long)(clock64() - cyc0));
#endif
    }

    FAT_STAMP(1);
    // ---- epilogue operands that do not depend on the products: in flight while the tile goes to LDS.  Element
    // (row r = 8 p + tid / 32, unit jj = tid % 32) for p = 0..NP-1; 16 consecutive lanes = 16 units of a row.  Every load is
    // unconditional (dead rows carry node 0 and the finite dummy row with alpha = 0).
    constexpr int NP = 2 * TB;
    const int jj = tid & 31, j = sl * 32 + jj, r0 = tid >> 5;
    int gvv[NP];
    float av[NP], g0r[NP], g0z[NP], g0n[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = p * 8 + r0;
        gvv[p] = M.node_s[r];
        av[p] = 0.f; g0r[p] = 0.f; g0z[p] = 0.f; g0n[p] = 0.f;
        if constexpr (has_hid) {
#pragma unroll
            for (int e = 0; e < MD; ++e) av[p] = fmaf(M.alpha[r * 2 + e], ldg1(M.hptr[r * 2 + e] + 4 * (uint64_t)j), av[p]);
        }
        if constexpr (!has_in) {
            const float* g0 = C.gi0 + (int64_t)gvv[p] * 3 * H;
            g0r[p] = g0[j]; g0z[p] = g0[H + j]; g0n[p] = g0[2 * H + j];
        }
    }
    const float bh_r = C.bhh[j], bh_z = C.bhh[H + j], bh_n = C.bhh[2 * H + j];
    float bi_r = 0.f, bi_z = 0.f, bi_n = 0.f;
    if constexpr (has_in) { bi_r = C.bih[j]; bi_z = C.bih[H + j]; bi_n = C.bih[2 * H + j]; }
    const float wk = C.wkey ? C.wkey[j] : 0.f;

    // ---- the two k halves of a side meet in LDS: out[side][row][gate * 32 + unit].  Pass 1: the sub = 0 wave of a side stores the
    // first half of its row blocks, the sub = 1 wave the second half; pass 2: each adds its other half to what the partner stored
    // (plain read - add - write: the two passes are a barrier apart; LDS float atomics cost ~0.3 us per wave instruction here).
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + e.
    float* out = ring;
    if constexpr (SIDES != 0) {
        auto tile_rw = [&](int half, bool add) {
#pragma unroll
            for (int mm = 0; mm < MB / 2; ++mm) {
                const int m = half * (MB / 2) + mm;
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    float* o = out + hs * (FTM * FAT_OP) + (m * 16 + 4 * kq) * FAT_OP + n * 16 + fi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = half == 0 ? acc[mm][n][e] : acc[MB / 2 + mm][n][e];
                        o[e * FAT_OP] = add ? o[e * FAT_OP] + v : v;
                    }
                }
            }
        };
        if (sub == 0) tile_rw(0, false); else tile_rw(1, false);
        __syncthreads();
        if (sub == 0) tile_rw(1, true); else tile_rw(0, true);
        __syncthreads();
    }

    FAT_STAMP(2);
    // ---- gates (hardware exp / rcp, as the dataflow kernels)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = p * 8 + r0;
        const bool live = r < nr;
        float hv = 0.f, spart = 0.f;
        const int gv = gvv[p];
        if (live) {
            float gr, gz, gn;
            const float* o0 = out + r * FAT_OP;                      // side 0
            const float* o1 = out + FTM * FAT_OP + r * FAT_OP;       // side 1
            if constexpr (has_in) {
                float pr = o0[jj], pz = o0[32 + jj], pn = o0[64 + jj];
                if constexpr (!has_hid) { pr += o1[jj]; pz += o1[32 + jj]; pn += o1[64 + jj]; }
                gr = pr + bi_r; gz = pz + bi_z; gn = pn + bi_n;
            } else {
                gr = g0r[p]; gz = g0z[p]; gn = g0n[p];
            }
            float hr = bh_r, hz = bh_z, hn = bh_n;
            if constexpr (has_hid) {
                float pr, pz, pn;
                if constexpr (has_in) { pr = o1[jj]; pz = o1[32 + jj]; pn = o1[64 + jj]; }
                else { pr = o0[jj] + o1[jj]; pz = o0[32 + jj] + o1[32 + jj]; pn = o0[64 + jj] + o1[64 + jj]; }
                hr += pr; hz += pz; hn += pn;
            }
            const float rg = fat_sigm(gr + hr);
            const float zg = fat_sigm(gz + hz);
            const float ng = fat_tanh(fmaf(rg, hn, gn));
            hv = fmaf(zg, av[p] - ng, ng);
            spart = wk * hv;
        }
        spart = dpp_row_sum16(spart);
        if (live) {
            float* po = C.h_out + (int64_t)gv * ld_h;
            po[j] = hv;
            if ((tid & 15) == 15) po[H + (j >> 4)] = spart;
            if (C.g_out) {
                gran_t* pg = C.g_out + (int64_t)gv * (H + H / PU);
                pg[j] = gran_pack(S.epoch, hv);
                if ((tid & 15) == 15) pg[H + (j >> 4)] = gran_pack(S.epoch, spart);
            }
        }
    }
    FAT_STAMP(3);
}

template <int TB>
__device__ __forceinline__ void fat_pick(const FatArgs& S, const FatTile& T, const FatLds& M) {
    if (T.has_in && T.has_hid) fat_main<3, TB>(S, T, M);
    else if (T.has_hid) fat_main<1, TB>(S, T, M);
    else if (T.has_in) fat_main<2, TB>(S, T, M);
    else fat_main<0, TB>(S, T, M);
}

__global__ void __launch_bounds__(FAT_THREADS, 2) fat_layer_kernel(const int32_t* __restrict__ plan, PlanLayout L, FatArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NS = S.H >> 5;
    const int gb = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gb >= S.tile_start[c + 1]) ++c;
    FatTile T;
    T.C = &S.cell[c];
    T.sl = blockIdx.x % NS;
    T.slot0 = T.C->row_base + (gb - S.tile_start[c]) * FTM;
    T.nr = min(FTM, T.C->row_end - T.slot0);
    T.two = T.nr > 32;
    T.has_in = T.C->wih_m != nullptr;
    T.has_hid = T.C->has_pred != 0;
    const FatLds M = fat_lds(smem);
    __builtin_amdgcn_s_setprio(3);
#ifdef FAT_STAMPS
    T.t0 = wall_clock64();
    int fat_cu_key = 0;
    if (threadIdx.x == 0) {
        atomicAdd(&fat_stamp_sum[7], 1ull);
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        fat_cu_key = (int)(((xcc & 15) << 8) | ((hw >> 8) & 0xff));
        const int seen = atomicAdd(&fat_cu_res[fat_cu_key], 1);
        atomicAdd(&fat_res_hist[min(seen, 3)], 1ull);
    }
#endif
    fat_prologue(plan, L, S, T, M);
    FAT_STAMP(0);
    if (T.two) fat_pick<4>(S, T, M);
    else fat_pick<2>(S, T, M);
#ifdef FAT_STAMPS
    if (threadIdx.x == 0) {
        const int seen = atomicAdd(&fat_cu_res[fat_cu_key], -1) - 1;
        atomicAdd(&fat_res_hist[4 + min(seen, 3)], 1ull);
    }
#endif
}

constexpr size_t FAT_LDS_ALONE = 96 * 1024;   // more than half a CU's 160 KB

inline size_t fat_lds_bytes() {
    return (size_t)FAT_MAIN * sizeof(float) + (size_t)FTM * (2 * sizeof(float*) + sizeof(float*) + 2 * sizeof(float) + 2 * sizeof(int)) +
           16 * sizeof(int);
}

}  // namespace

#ifdef FAT_STAMPS
extern "C" int dagnn_fat_debug_read(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(fat_stamp_sum), sizeof(fat_stamp_sum)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16 + 8, HIP_SYMBOL(fat_res_hist), sizeof(fat_res_hist)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fat_stamp_sum), z, sizeof(z));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fat_res_hist), z, sizeof(z));
    }
    return 0;
}
#endif

// One fat launch: every active cell of `cells` ([row_base, row_end) = the rows of its layer) in 64-row tiles.
// `scratch` = fp32 [>= total rows of the launch, H]: aggregates of the rows with more than four predecessors.
int dagnn_fat_launch(const int32_t* plan, const PlanLayout& L, const Cell* cells, int ncell, int H, int ld_h, int R, int vid_mod,
                     unsigned epoch, float* scratch, int num_cus, hipStream_t st) {
    if (ncell <= 0 || ncell > DAGNN_MAX_CELLS || (H % 64) || !scratch) return DAGNN_EINVAL;
    static std::atomic<unsigned long long> attr_done{0ull};
    if (dagnn_lds_attr_once(attr_done, reinterpret_cast<const void*>(fat_layer_kernel), (int)FAT_LDS_ALONE) != hipSuccess)
        return DAGNN_EHIP(hipGetLastError());
    FatArgs A;
    int off = 0, tiles = 0;
    A.tile_start[0] = 0;
    for (int k = 0; k < ncell; ++k) {
        A.cell[k] = cells[k];
        const int n = cells[k].row_end - cells[k].row_base;
        A.cell[k].a_pre = scratch + (int64_t)off * H;
        off += n;
        tiles += (n + FTM - 1) / FTM;
        A.tile_start[k + 1] = tiles;
    }
    A.ncell = ncell; A.H = H; A.ld_h = ld_h; A.R = R; A.vid_mod = vid_mod > 0 ? vid_mod : 1; A.epoch = epoch;
    if (tiles == 0) return DAGNN_OK;
    // a launch that fits one workgroup per CU asks for more LDS than two workgroups can share: the dispatcher would
    // otherwise pair workgroups on some CUs (they halve each other's matrix pipe) while other CUs stay empty
    const int wgs = tiles * (H / 32);
    const size_t lds = wgs <= num_cus ? FAT_LDS_ALONE : fat_lds_bytes();
    hipLaunchKernelGGL(fat_layer_kernel, dim3((unsigned)wgs), dim3(FAT_THREADS), lds, st, plan, L, A);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DAGNN_OK : DAGNN_EHIP(e);
}
