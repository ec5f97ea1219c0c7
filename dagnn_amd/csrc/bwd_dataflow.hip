// bwd_dataflow.hip - the reverse-mode sweep of the recurrence as ONE persistent, graph-affine dataflow launch (H <= 256).
//
// Reference path replaced: what `loss.backward()` (ogbg-code/main_pyg.py:62; dvae/train.py through `encode`) does to the
// loops of ogbg-code/model/dagnn.py:144-182 (dvae/dagnn.py:112-146, dvae/dagnn_bn.py:110-137) through torch autograd: one
// autograd node per (direction, topological layer, stacked layer) micro-step.  backward.hip (round 1) walks them as
// T + L - 1 reverse lock-step launches plus a persistent head; here the whole sweep is the mirror image of dataflow.hip:
// the SAME schedule (graphs dealt to independent groups, dagnn_dataflow_schedule), walked block by block from the deepest
// layer of every group back to layer 0, one workgroup per (workgroup set, kernel cell, 32-unit slice) with its matrix
// slice resident in registers, rows handed between workgroups as tagged granules.
//
// Math (per direction d, stacked layer i, node v; see backward.hip's header for the derivation):
//   G_v   = Gext_v + du_v(i+1) + sum_{e = (v -> w)} [ alpha_e da_w + ds_e w_k ],   ds_e = alpha_e (da_w . h_v - q_w)
//           with q_w = da_w . a_w.  q_w never needs the row a_w: da_w = z_w (.) G_w + W_hh^T dgh_w, hence
//           q_w = G_w . c_q,w with c_q = z a + c_r (gh_r - b_r) + c_z (gh_z - b_z) + c_nr (gh_n - b_n)  (W_hh a = gh - b_hh):
//           ONE scalar per node, published by the workgroup that computes G_w;
//   GRU backward is linear in G_v with coefficients that depend on the forward pass only, so they are computed for all
//   nodes beforehand (dagnn_bwd_dataflow_prepare, off the dependent chain):
//           dgh_v = G_v (.) (c_r, c_z, c_nr),  dgi_v = G_v (.) (c_r, c_z, c_n),  zg_v = G_v (.) z
//           c_n = (1 - z)(1 - n^2), c_z = (a - n) z (1 - z), c_r = c_n gh_n r (1 - r), c_nr = c_n r;
//   da_v  = zg_v + W_hh^T dgh_v          (K = 3H: three gate blocks of K = H, i.e. the forward kernel's product shape
//                                          on the gate-wise transposed matrix)
//   du_v  = W_ih^T dgi_v  -> G of stacked layer i - 1 at the same node.
// Kernel cells of a direction: L "state-gradient" cells (W_hh^T -> da granules) and L - 1 "input-gradient" cells (W_ih^T of
// the stacked layers above the first -> du granules, off the dependent chain like the forward projection cells).
//   loader wave (da cell)  row w of every block of its stream: static rows of the node (Gext, h, the coefficient rows: one
//                   contiguous 8 KB record of dagnn_bwd_dataflow_prepare), poll the successors' da rows + q scalars (and
//                   the node's du row), pull, coefficients -> dgh into the stream's LDS ring slot (3 operand rows), zg
//                   slice; its slice of dgi / dgh to memory (the weight-gradient epilogue reads them), dgi also as
//                   granules for the input-gradient cell; slice 0 publishes q_v.  (Round 4: the row's outputs can ride
//                   the LDS slot to compute wave r instead - BD_*_LOADER, chosen per workgroup shape; sigma_v and the
//                   edge-feature sums always leave through slice 1's compute waves.)  A row's edge scalars (alpha of every
//                   stacked layer, the edge features of its first four successors) come with its 64-word record;
//   loader wave (du cell)  polls the dgi granules of the node, no arithmetic;
//   compute wave    v_mfma_f32_4x4x1 products exactly as in dataflow.hip (A = resident weights, B = the block's operand
//                   rows), the three gate accumulators summed BEFORE the K reduce-scatter, + zg, granule store.
// No atomics on values, a fixed order of additions: gradients are bitwise reproducible run to run.
#include "df_common.h"
#include <type_traits>

namespace {

typedef float bf4 __attribute__((ext_vector_type(4)));

#ifndef BD_NSLOT_V
#define BD_NSLOT_V 3
#endif
constexpr int BD_NSLOT = BD_NSLOT_V;   // LDS ring depth per stream (a slot holds 3 operand rows per block row; 2 / 3 / 4: the same time)
#ifndef BD_WPS_V
#define BD_WPS_V 4
#endif
constexpr int BD_WPS = BD_WPS_V;     // loader waves per stream (4: one row of a block each; 2: the 8-wave shape of H = 320, two rows each)
constexpr int BD_RPW = DF_RB / BD_WPS;   // rows of a block per loader wave, one after the other
constexpr int BD_NLW = DF_NLS * BD_WPS;
constexpr int BD_THREADS = 64 * (DF_NCW + BD_NLW);
constexpr int BD_SP = DF_STAT_SP;    // pitch of a static row (floats): lane l holds columns {l, 64 + l, 128 + l, 192 + l} at [4l, 4l + 4)
// H = 320 (round 4): a fifth column block {256 + l}.  The record of a (cell, node) is then part A = the eight 256-float rows as
// before, part B = eight 64-float rows behind them (row r, lane l at 8 * 256 + 64 r + l): 10 KB instead of 8
__host__ __device__ constexpr int bd_stat_floats(int H) { return df_stat_floats(H); }
constexpr int BD_NSTAT = DF_NSTAT;   // static rows per (cell, node): Gext, h, c_r, c_z, c_nr, c_n, z, c_q (df_common.h)
constexpr int BD_RD = 6;             // a loader wave requests a record this many of its blocks ahead (ring: 8 entries)
// a successor record (bd_records_kernel): 64 words = one DMA of a full wave
//   [0..3] v, first / last CSR slot of v's row in direction 1 - d, 0     [4..7] first four successors   [8..11] their edge ids
//   [16..19] / [20..23] first / second edge feature of those edges       [24 + 4 i + e] alpha of edge e in stacked layer i
constexpr int BD_RECW = 64;
// Who stores a row's outputs: its LOADER wave, behind the ready flag - q_v, the slice's columns of dgi (granules and plain
// rows) and dgh; a block costs the compute waves more than a row costs its loader, and they serve two streams (measured per
// workgroup shape in round 4; the compute-wave forms of these stores live in scripts/experiments/).  The compute
// wave of row r keeps sigma_v and the edge-feature sums of slice BD_SCAL_SL.
static_assert(BD_WPS_V == 4, "the workgroup shape: 4 compute + 2 x 4 loader waves (the 8-wave shape of round 4 is in scripts/experiments/)");
constexpr int BD_SCAL_SL = 1;        // slice whose compute waves store sigma_v and the edge-feature sums (slice 0 stores q_v)
enum { BD_DA = 0, BD_DU = 1 };
enum { ST_GEXT = DF_ST_GEXT, ST_H = DF_ST_H, ST_CR = DF_ST_CR, ST_CZ = DF_ST_CZ, ST_CNR = DF_ST_CNR, ST_CN = DF_ST_CN, ST_Z = DF_ST_Z, ST_CQ = DF_ST_CQ };

struct BdCell {            // (104 bytes: 30 kernel cells - 8 stacked layers, both directions - fit the kernel-argument segment)
    const float4* w;       // packed slices (dagnn_pack_dataflow of the gate-wise transposed matrix)
    const float* wkey;     // da: [H] key weights (zeros when the scores are static)
    const float* alpha;    // da: [E] attention weights by original edge id
    const float* stat;     // da: [N, 8 * 256] static rows
    const gran_t* du_in;   // da: [N, gld] du arriving from stacked layer i + 1, or null (top layer)
    gran_t* out_g;         // da: [N, gld] granules of da (also read by this cell's own loaders); du: [N, gld] granules of du
                           //     for stacked layer i - 1
    gran_t* q_g;           // da: [N] granules of q
    gran_t* dgi_g;         // da: [N, 3 gld] granules of dgi for the input-gradient cell, or null (stacked layer 0);
                           // du: the same buffer (its input)
    float* dgi;            // da: [N, 3H]
    float* dgh;            // da: [N, 3H]
    float* sig;            // da: [N]
    float* mrel;           // da: [N, R] or null
    int dir, kind;
    int partner, layer;    // da: index of the input-gradient cell that reads this cell's dgi granules, or -1; the stacked layer
};

#define BD_MAX_KCELLS 24
#define BD_MAX_WGS 320
#define BD_IDLE_ROLE 0xffffu

struct BdArgs {
    BdCell cell[BD_MAX_KCELLS];
    const int32_t* sched;   // schedule workspace (dagnn_dataflow_schedule)
    const int32_t* brecs;   // successor records in schedule order (dagnn_bwd_dataflow_prepare)
    int64_t gtab[2], brec[2];   // word offsets into sched / brecs
    int64_t col[2], eidx[2], eattr[2];   // word offsets into the plan
    int ncell, H, gld, R, groups, N;
    unsigned epoch, spin_limit;
    const int32_t* status;
    int* err;
    // XCD-aware placement, verified at run time (see DfArgs::role in dataflow.hip): unit = a state-gradient cell's slices +
    // the slices of its input-gradient cell; da / q / dgi granules stay in the XCD's L2 when all of them run there
    gran_t* xcc_tab;
    int nroles;
    unsigned short role[BD_MAX_WGS];
#ifdef BD_STAMPS
    unsigned long long* dbg;   // [grid][2] start / end of every workgroup, then [blocks][NLS][16] stamps of the workgroup with role dbg_role
    unsigned dbg_role;
#endif
};
// per-block stamps (scripts/bd_hops.py) only in a build with -DBD_STAMPS (scripts/build_variant.sh stamps SRC=bwd_dataflow -DBD_STAMPS)
#ifdef BD_STAMPS
#define BD_STAMP(on, blk, st, k) do { if (on) S.dbg[2 * (int64_t)gridDim.x + 16 * (int64_t)(DF_NLS * (blk) + (st)) + (k)] = wall_clock64(); } while (0)
#define BD_STAMP_V(on, blk, st, k, val) do { if (on) S.dbg[2 * (int64_t)gridDim.x + 16 * (int64_t)(DF_NLS * (blk) + (st)) + (k)] = (val); } while (0)
#else
#define BD_STAMP(on, blk, st, k) do { } while (0)
#define BD_STAMP_V(on, blk, st, k, val) do { } while (0)
#endif
static_assert(sizeof(BdArgs) + 8 <= 4096, "the cell and role tables must fit the kernel-argument segment");

template <int KPT> struct BdPad { static constexpr int kp8 = 2 * KPT; static constexpr int seg = kp8 + 4; static constexpr int row = 8 * seg + 8; };

template <int KPT> struct BdSlot {
    static constexpr int AP = BdPad<KPT>::row;
    static constexpr int op_off = 0;                          // [3][RB][AP] operand rows: gate block g of row r at (g * RB + r) * AP
    static constexpr int zg_off = 3 * DF_RB * AP;             // [RB][32]    z (.) G of the slice's units
    static constexpr int dn_off = zg_off + DF_RB * DF_JS;     // [RB][32]    c_n (.) G of the slice's units (the n block of dgi)
    static constexpr int sc_off = dn_off + DF_RB * DF_JS;     // [RB][4]     sigma_v and the two edge-feature sums (slice 1)
    static constexpr int v_off = sc_off + DF_RB * 4;          // [RB] ints
    static constexpr int words = v_off + 4;
};

struct BdLds {
    float* ring;   // [NLS][NSLOT] slots
    int* rec;      // [NLS * RB][8][BD_RECW] row records, landed by LDS-DMA BD_RD blocks ahead
    int* rdy;      // [NLS][WPS]
    int* dn;       // [NLS][NCW]
    int* dump;     // [NLS * RB][64] landing area of the L2 warm-up DMAs (never read)
    int* local;    // [1] every reader of this cell's granules runs on this workgroup's XCD
};

__device__ __forceinline__ int bd_flag_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void bd_flag_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ bool bd_wait4(const int* f, int target, int* err, unsigned limit) {
    unsigned spins = 0;
    for (;;) {
        const int a = bd_flag_ld(f), b = bd_flag_ld(f + 1), c = bd_flag_ld(f + 2), d = bd_flag_ld(f + 3);
        if (min(min(a, b), min(c, d)) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4 * limit) { __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        if ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
}

__device__ __forceinline__ bool bd_retry(unsigned& spins, int* err, unsigned limit) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > limit) { __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    if ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    return true;
}

// a pointer / value the "s" operand of an asm statement can take (the compiler does not always see that it is wave-uniform)
__device__ __forceinline__ const void* bd_sptr(const void* p) {
    const unsigned long long u = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const void*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float bd_dpp_row_sum16(float v) {
#define BD_DPP_ADD(ctrl) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    BD_DPP_ADD(0x111); BD_DPP_ADD(0x112); BD_DPP_ADD(0x114); BD_DPP_ADD(0x118);
#undef BD_DPP_ADD
    return v;
}
// sum over the 64 lanes, broadcast as a wave-uniform value (row scans, then row_bcast15 / row_bcast31)
__device__ __forceinline__ float bd_wave_sum(float v) {
    v = bd_dpp_row_sum16(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <int CTRL> __device__ __forceinline__ float bd_dpp(float v) {   // 0 where the source lane is outside the DPP row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float bd_row_pair_sum(float x) {   // (see df_row_pair_sum in dataflow.hip: the swap must be inline asm)
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __host__ __forceinline__ int bd_stream_group(int pair, int set, int groups) {
    return df_group_of_stream(pair, set, groups);
}

// ---------------------------------------------------------------- preparation kernels
// successor records in schedule order: for schedule record r of direction d (node v, or -1 = padding) the 64 bytes
// {v, first / last CSR slot of v's row in direction 1 - d, 0, first four successors, their original edge ids, 0 x 4}
struct BdRecAlpha { const float* alpha[DAGNN_MAX_DIRS][DAGNN_MAX_STACKED]; int Ls, R; };
static_assert(24 + 4 * DAGNN_MAX_STACKED <= BD_RECW, "a record holds the attention weights of every stacked layer");

__global__ void __launch_bounds__(256) bd_records_kernel(const int32_t* __restrict__ plan, PlanLayout L,
                                                          const int32_t* __restrict__ sched, DfLayout S,
                                                          int32_t* __restrict__ brecs, int64_t nrec, int groups,
                                                          const int32_t* __restrict__ status, BdRecAlpha AL) {
    if (status && status[0] != 0) return;
    const int d = blockIdx.y, od = 1 - d;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    // the schedule defines the records its groups use (first record of the last group + its blocks); behind them: padding
    const int32_t* last = sched + S.gtab[d] + 2 * (groups - 1);
    const int64_t used = (int64_t)last[0] + (int64_t)DF_RB * last[1];
    const int v = r < used ? sched[S.grec[d] + 16 * r] : -1;
    int4* out = reinterpret_cast<int4*>(brecs + ((int64_t)d * nrec + r) * BD_RECW);
    if (v < 0) {
        out[0] = make_int4(-1, 0, 0, 0);
        for (int i = 1; i < BD_RECW / 4; ++i) out[i] = make_int4(0, 0, 0, 0);
        return;
    }
    const int4* orec = reinterpret_cast<const int4*>(plan + L.rowrec[od]) + 4 * (int64_t)plan[L.pos[od] + v];
    const int4 r0 = orec[0], r1 = orec[1];
    const int eb = r0.y, ee = r0.z;
    const int32_t* eidx = plan + L.eidx[od];
    out[0] = make_int4(v, eb, ee, 0);
    out[1] = r1;
    const int e4[4] = {eb < ee ? eidx[eb] : 0, eb + 1 < ee ? eidx[eb + 1] : 0, eb + 2 < ee ? eidx[eb + 2] : 0, eb + 3 < ee ? eidx[eb + 3] : 0};
    out[2] = make_int4(e4[0], e4[1], e4[2], e4[3]);
    out[3] = make_int4(0, 0, 0, 0);
    // what the sweep's pull needs per edge, so that a row costs its loader no trip to memory beyond the record: the edge
    // features (the sweep sums ds_e f_e) and the attention weights of every stacked layer
    const float* eattr = reinterpret_cast<const float*>(plan + L.eattr[od]);
    float f[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) f[k][e] = (k < AL.R && eb + e < ee) ? eattr[(int64_t)(eb + e) * AL.R + k] : 0.f;
    float4* outf = reinterpret_cast<float4*>(out);
    outf[4] = make_float4(f[0][0], f[0][1], f[0][2], f[0][3]);
    outf[5] = make_float4(f[1][0], f[1][1], f[1][2], f[1][3]);
    for (int i = 0; i < DAGNN_MAX_STACKED; ++i) {
        const float* al = i < AL.Ls ? AL.alpha[d][i] : nullptr;
        outf[6 + i] = al ? make_float4(eb < ee ? al[e4[0]] : 0.f, eb + 1 < ee ? al[e4[1]] : 0.f, eb + 2 < ee ? al[e4[2]] : 0.f,
                                       eb + 3 < ee ? al[e4[3]] : 0.f)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = 6 + DAGNN_MAX_STACKED; i < BD_RECW / 4; ++i) out[i] = make_int4(0, 0, 0, 0);
}

struct BdStatCell {
    const float* gi;     // [N,3H]
    const float* gh;     // [N,3H]
    const float* a;      // [N,H]
    const float* b_hh;   // [3H]
    const float* h;      // [N,ld_h]
    const float* gext;   // [N,ld_g]
    float* stat;         // [N, 8 * 256]
};
struct BdStatArgs { BdStatCell cell[DAGNN_MAX_DIRS * DAGNN_MAX_STACKED]; int ncell, H, ld_h, ld_g; };

__device__ __forceinline__ float bd_sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// one wave per (cell, node): the eight static rows in the lane order the sweep's loader waves read them in
__global__ void __launch_bounds__(256) bd_stat_kernel(BdStatArgs A, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;
    const BdStatCell& C = A.cell[blockIdx.y];
    const int H = A.H, H3 = 3 * H;
    float o[BD_NSTAT][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int c = 64 * q + lane;
#pragma unroll
        for (int r = 0; r < BD_NSTAT; ++r) o[r][q] = 0.f;
        if (c < H) {
            const float* gi = C.gi + v * H3;
            const float* gh = C.gh + v * H3;
            const float ghr = gh[c], ghz = gh[H + c], ghn = gh[2 * H + c];
            const float rr = bd_sigm(gi[c] + ghr), zz = bd_sigm(gi[H + c] + ghz);
            const float nn = tanhf(gi[2 * H + c] + rr * ghn);
            const float av = C.a[v * H + c];
            const float cn = (1.0f - zz) * (1.0f - nn * nn);
            const float cz = (av - nn) * zz * (1.0f - zz);
            const float cr = cn * ghn * rr * (1.0f - rr);
            const float cnr = cn * rr;
            o[ST_GEXT][q] = C.gext[v * A.ld_g + c];
            o[ST_H][q] = C.h[v * A.ld_h + c];
            o[ST_CR][q] = cr; o[ST_CZ][q] = cz; o[ST_CNR][q] = cnr; o[ST_CN][q] = cn; o[ST_Z][q] = zz;
            o[ST_CQ][q] = zz * av + cr * (ghr - C.b_hh[c]) + cz * (ghz - C.b_hh[H + c]) + cnr * (ghn - C.b_hh[2 * H + c]);
        }
    }
    float* rec = C.stat + v * bd_stat_floats(H);
    float4* dst = reinterpret_cast<float4*>(rec) + lane;
#pragma unroll
    for (int r = 0; r < BD_NSTAT; ++r) dst[r * (BD_SP / 4)] = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
    if (H > 256) {
#pragma unroll
        for (int r = 0; r < BD_NSTAT; ++r) rec[BD_NSTAT * BD_SP + 64 * r + lane] = o[r][4];
    }
}

// ... and when the forward kernel's training epilogue wrote rows 1..7 already (dagnn_dataflow_args.stat_rows): row 0 only
__global__ void __launch_bounds__(256) bd_gext_kernel(BdStatArgs A, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;
    const BdStatCell& C = A.cell[blockIdx.y];
    const int H = A.H;
    const float* g = C.gext + v * A.ld_g;
    float o[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) o[q] = 64 * q + lane < H ? g[64 * q + lane] : 0.f;
    float* rec = C.stat + v * bd_stat_floats(H);
    reinterpret_cast<float4*>(rec)[ST_GEXT * (BD_SP / 4) + lane] = make_float4(o[0], o[1], o[2], o[3]);
    if (H > 256) rec[BD_NSTAT * BD_SP + 64 * ST_GEXT + lane] = o[4];
}

// ---------------------------------------------------------------- loader waves
// Memory traffic by hand, as in dataflow.hip: ONE asm statement per trip to memory - the polls (granule rows with
// `sc1`), on the first trip of a row also its static rows and the LDS-DMA of the record BD_RD blocks ahead, and the counted
// wait (the DMA stays in flight).  A destination register the compiler can see while its load is in flight gets copied
// sooner or later; loads and wait in the same statement leave it nothing to see.
struct BdSweep {
    gran_t x[4][5];   // da rows of up to four successors: columns {lane, 64 + lane, 128 + lane, 192 + lane} (+ 256 + lane: H = 320)
    gran_t q[4];      // their q scalars (every lane loads the same granule)
    gran_t u[5];      // the node's du row
};

#define BD_ROW_LD(e)                                                            \
    "global_load_dwordx2 %[x" #e "0], %[vo], %[b" #e "] offset:0 sc1\n\t"        \
    "global_load_dwordx2 %[x" #e "1], %[vo], %[b" #e "] offset:%[o1] sc1\n\t"    \
    "global_load_dwordx2 %[x" #e "2], %[vo], %[b" #e "] offset:%[o2] sc1\n\t"    \
    "global_load_dwordx2 %[x" #e "3], %[vo], %[b" #e "] offset:%[o3] sc1\n\t"    \
    "global_load_dwordx2 %[q" #e "], %[vz], %[c" #e "] offset:0 sc1\n\t"
#define BD_ROWS_0 "v_mov_b32 %[vz], 0\n\t"
#define BD_ROWS_1 BD_ROWS_0 BD_ROW_LD(0)
#define BD_ROWS_2 BD_ROWS_1 BD_ROW_LD(1)
#define BD_ROWS_3 BD_ROWS_2 BD_ROW_LD(2)
#define BD_ROWS_4 BD_ROWS_3 BD_ROW_LD(3)
#define BD_DU_0 ""
#define BD_DU_1                                                      \
    "global_load_dwordx2 %[u0], %[vo], %[ub] offset:0 sc1\n\t"        \
    "global_load_dwordx2 %[u1], %[vo], %[ub] offset:%[o1] sc1\n\t"    \
    "global_load_dwordx2 %[u2], %[vo], %[ub] offset:%[o2] sc1\n\t"    \
    "global_load_dwordx2 %[u3], %[vo], %[ub] offset:%[o3] sc1\n\t"
// first trip of a row: its static rows, the LDS-DMA of the record BD_RD blocks ahead, and a one-instruction warm-up of the
// NEXT block's static rows (lane l fetches one dword of cache line l of that 8 KB record into an LDS dump area nobody
// reads: the lines are in this XCD's L2 when the next block's trip asks for them - a cold static row costs ~1.5 us, an L2
// hit less than the polls next to it); both DMAs stay in flight behind the counted wait
#define BD_STAT_1                                                     \
    "v_lshlrev_b32 %[vs], 1, %[vo]\n\t"                               \
    "global_load_dwordx4 %[s0], %[vs], %[sa] offset:0\n\t"            \
    "global_load_dwordx4 %[s1], %[vs], %[sa] offset:1024\n\t"         \
    "global_load_dwordx4 %[s2], %[vs], %[sa] offset:2048\n\t"         \
    "global_load_dwordx4 %[s3], %[vs], %[sa] offset:3072\n\t"         \
    "global_load_dwordx4 %[s4], %[vs], %[sb] offset:0\n\t"            \
    "global_load_dwordx4 %[s5], %[vs], %[sb] offset:1024\n\t"         \
    "global_load_dwordx4 %[s6], %[vs], %[sb] offset:2048\n\t"         \
    "global_load_dwordx4 %[s7], %[vs], %[sb] offset:3072\n\t"         \
    "s_mov_b32 %[km], m0\n\ts_mov_b32 m0, %[pl]\n\tv_lshlrev_b32 %[vs], 4, %[vo]\n\tglobal_load_lds_dword %[vs], %[pa]\n\t" \
    "s_mov_b32 m0, %[rl]\n\t" \
    "v_lshrrev_b32 %[vs], 1, %[vo]\n\tglobal_load_lds_dword %[vs], %[ra]\n\t"                 \
    "s_mov_b32 m0, %[km]\n\ts_waitcnt vmcnt(2)"
// H = 320: the fifth column block of every polled row, and part B of the static record
#define BD_ROW_LD5(e) BD_ROW_LD(e) "global_load_dwordx2 %[x" #e "4], %[vo], %[b" #e "] offset:2048 sc1\n\t"
#define BD_ROWS5_0 BD_ROWS_0
#define BD_ROWS5_1 BD_ROWS5_0 BD_ROW_LD5(0)
#define BD_ROWS5_2 BD_ROWS5_1 BD_ROW_LD5(1)
#define BD_ROWS5_3 BD_ROWS5_2 BD_ROW_LD5(2)
#define BD_ROWS5_4 BD_ROWS5_3 BD_ROW_LD5(3)
#define BD_DU5_0 ""
#define BD_DU5_1 BD_DU_1 "global_load_dwordx2 %[u4], %[vo], %[ub] offset:2048 sc1\n\t"
#define BD_STAT5_1                                                    \
    "v_lshrrev_b32 %[vt], 1, %[vo]\n\t"                               \
    "global_load_dword %[t0], %[vt], %[sc] offset:0\n\t"              \
    "global_load_dword %[t1], %[vt], %[sc] offset:256\n\t"            \
    "global_load_dword %[t2], %[vt], %[sc] offset:512\n\t"            \
    "global_load_dword %[t3], %[vt], %[sc] offset:768\n\t"            \
    "global_load_dword %[t4], %[vt], %[sc] offset:1024\n\t"           \
    "global_load_dword %[t5], %[vt], %[sc] offset:1280\n\t"           \
    "global_load_dword %[t6], %[vt], %[sc] offset:1536\n\t"           \
    "global_load_dword %[t7], %[vt], %[sc] offset:1792\n\t"           \
    BD_STAT_1
#define BD_ROW_OUTS_C(CS)                                                                                                        \
                   [x00] CS(W.x[0][0]), [x01] CS(W.x[0][1]), [x02] CS(W.x[0][2]), [x03] CS(W.x[0][3]),              \
                   [x10] CS(W.x[1][0]), [x11] CS(W.x[1][1]), [x12] CS(W.x[1][2]), [x13] CS(W.x[1][3]),              \
                   [x20] CS(W.x[2][0]), [x21] CS(W.x[2][1]), [x22] CS(W.x[2][2]), [x23] CS(W.x[2][3]),              \
                   [x30] CS(W.x[3][0]), [x31] CS(W.x[3][1]), [x32] CS(W.x[3][2]), [x33] CS(W.x[3][3]),              \
                   [q0] CS(W.q[0]), [q1] CS(W.q[1]), [q2] CS(W.q[2]), [q3] CS(W.q[3]),                              \
                   [u0] CS(W.u[0]), [u1] CS(W.u[1]), [u2] CS(W.u[2]), [u3] CS(W.u[3]), [vz] "=&v"(tmp_vz)
#define BD_ROW_OUTS BD_ROW_OUTS_C("=v")
#define BD_ROW_INS                                                                                                          \
                   [vo] "v"(lane8), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [b3] "s"(b3),                \
                   [c0] "s"(c0p), [c1] "s"(c1p), [c2] "s"(c2p), [c3] "s"(c3p), [ub] "s"(ub),                                \
                   [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3)
#define BD_TRIP_FIRST(n_, du_)                                                                                              \
    asm volatile(BD_ROWS_##n_ BD_DU_##du_ BD_STAT_1                                                                         \
                 : BD_ROW_OUTS,                                                                                             \
                   [s0] "=v"(ST[0]), [s1] "=v"(ST[1]), [s2] "=v"(ST[2]), [s3] "=v"(ST[3]),                                  \
                   [s4] "=v"(ST[4]), [s5] "=v"(ST[5]), [s6] "=v"(ST[6]), [s7] "=v"(ST[7]),                                  \
                   [km] "=&s"(keep_m0), [vs] "=&v"(tmp_vs)                                                                         \
                 : BD_ROW_INS, [sa] "s"(sa), [sb] "s"(sb), [rl] "s"(rl), [ra] "s"(ra), [pa] "s"(pa),          \
                   [pl] "s"(pl)                                                                               \
                 : "memory")
// (a poll keeps the sweep in the registers it arrived in: in / out operands, see DF_TRIP in dataflow.hip)
#define BD_TRIP_POLL(n_, du_)                                                                                               \
    asm volatile(BD_ROWS_##n_ BD_DU_##du_ "s_waitcnt vmcnt(0)" : BD_ROW_OUTS : BD_ROW_INS : "memory")
#define BD_TRIP_REPOLL(n_, du_)                                                                                             \
    asm volatile(BD_ROWS_##n_ BD_DU_##du_ "s_waitcnt vmcnt(0)" : BD_ROW_OUTS_C("+v") : BD_ROW_INS : "memory")
#define BD_ROW_OUTS5_C(CS) BD_ROW_OUTS_C(CS), [x04] CS(W.x[0][4]), [x14] CS(W.x[1][4]), [x24] CS(W.x[2][4]), [x34] CS(W.x[3][4]), [u4] CS(W.u[4])
#define BD_ROW_OUTS5 BD_ROW_OUTS5_C("=v")
#define BD_TRIP_FIRST5(n_, du_)                                                                                             \
    asm volatile(BD_ROWS5_##n_ BD_DU5_##du_ BD_STAT5_1                                                                      \
                 : BD_ROW_OUTS5,                                                                                            \
                   [s0] "=v"(ST[0]), [s1] "=v"(ST[1]), [s2] "=v"(ST[2]), [s3] "=v"(ST[3]),                                  \
                   [s4] "=v"(ST[4]), [s5] "=v"(ST[5]), [s6] "=v"(ST[6]), [s7] "=v"(ST[7]),                                  \
                   [t0] "=v"(ST5[0]), [t1] "=v"(ST5[1]), [t2] "=v"(ST5[2]), [t3] "=v"(ST5[3]),                              \
                   [t4] "=v"(ST5[4]), [t5] "=v"(ST5[5]), [t6] "=v"(ST5[6]), [t7] "=v"(ST5[7]),                              \
                   [km] "=&s"(keep_m0), [vs] "=&v"(tmp_vs), [vt] "=&v"(tmp_vt)                                                     \
                 : BD_ROW_INS, [sa] "s"(sa), [sb] "s"(sb), [rl] "s"(rl), [ra] "s"(ra), [pa] "s"(pa),          \
                   [pl] "s"(pl), [sc] "s"(sc)                                                                 \
                 : "memory")
#define BD_TRIP_POLL5(n_, du_)                                                                                              \
    asm volatile(BD_ROWS5_##n_ BD_DU5_##du_ "s_waitcnt vmcnt(0)" : BD_ROW_OUTS5 : BD_ROW_INS : "memory")
#define BD_TRIP_REPOLL5(n_, du_)                                                                                            \
    asm volatile(BD_ROWS5_##n_ BD_DU5_##du_ "s_waitcnt vmcnt(0)" : BD_ROW_OUTS5_C("+v") : BD_ROW_INS : "memory")
#define BD_SEL1(n_, du_) if constexpr (NN == (n_) && DUK == (du_)) {                                                          \
        if constexpr (NQ4 == 5) { if constexpr (FST == 1) BD_TRIP_FIRST5(n_, du_); else if constexpr (FST == 2) BD_TRIP_REPOLL5(n_, du_); else BD_TRIP_POLL5(n_, du_); } \
        else { if constexpr (FST == 1) BD_TRIP_FIRST(n_, du_); else if constexpr (FST == 2) BD_TRIP_REPOLL(n_, du_); else BD_TRIP_POLL(n_, du_); } } else
#define BD_SELS(n_) BD_SEL1(n_, 0) BD_SEL1(n_, 1)
#define BD_TRIP_SEL(nn_, du_, fst_) do { constexpr int DUK = (du_); constexpr int FST = (fst_); (void)DUK; (void)FST;   /* 1 first trip of a row, 2 re-poll, 0 first poll of a later chunk */ \
        BD_SELS(0) BD_SELS(1) BD_SELS(2) BD_SELS(3) BD_SELS(4) {} } while (0)
#define BD_CASE(n_, du_) case (n_) * 2 + (du_):                                                                             \
        if constexpr (NQ4 == 5) { if (stat_pending) BD_TRIP_FIRST5(n_, du_); else BD_TRIP_POLL5(n_, du_); }                 \
        else { if (stat_pending) BD_TRIP_FIRST(n_, du_); else BD_TRIP_POLL(n_, du_); }                                      \
        break;
#define BD_CASES(n_) BD_CASE(n_, 0) BD_CASE(n_, 1)

template <int KPT, bool HAS_DU>
__device__ __forceinline__ void bd_loader_da(const int32_t* __restrict__ plan, const BdArgs& S, const BdCell& C, int sl,
                                             int group, const BdLds& lds, int w, int set) {
    constexpr int H = 16 * KPT;
    constexpr int SEG = BdPad<KPT>::seg, KP8 = BdPad<KPT>::kp8;
    constexpr int NQ4 = H / 64;
    typedef BdSlot<KPT> Slot;
    const int lane = threadIdx.x & 63;
    const int d = C.dir, od = 1 - d;
    const int32_t* tab = S.sched + S.gtab[d] + 2 * group;
    const int rec_base = tab[0], nblk = tab[1];
    const int32_t* __restrict__ recs = S.brecs + S.brec[d] + BD_RECW * (int64_t)rec_base;
    const int32_t* __restrict__ col = plan + S.col[od];
    const int32_t* __restrict__ eidx = plan + S.eidx[od];
    const float* __restrict__ eattr = reinterpret_cast<const float*>(plan + S.eattr[od]);
    const int R = C.mrel ? S.R : 0;
    const int r1 = R > 1 ? 1 : 0;
    const int al_w = 24 + 4 * C.layer;   // word of the record that holds this stacked layer's alpha of the first edge
    const float* __restrict__ ea = R > 0 ? eattr : C.alpha;
    const unsigned epoch = S.epoch, spin_limit = S.spin_limit;
    int* const err = S.err;
    const int gld = S.gld;
    const gran_t* const da_g = C.out_g;
    const gran_t* const q_in = C.q_g;
    const gran_t* const du_in = C.du_in;
    const float* const stat = C.stat;
    const float* const alpha = C.alpha;
    int* const dn = lds.dn + set * DF_NCW;
    constexpr int NC = NQ4 > 4 ? NQ4 : 4;   // column blocks a lane carries
    constexpr int SREC = bd_stat_floats(H);
    float wk[NC];
    int cpos[NC];   // (64 % KP8 == 0: cpos[q] = cpos[0] + a constant - one address register, immediate offsets)
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = 64 * q + lane;
        cpos[q] = (64 % KP8 == 0) ? lane + (SEG - KP8) * (lane / KP8) + q * (64 + (SEG - KP8) * (64 / KP8)) : c + (SEG - KP8) * (c / KP8);
        wk[q] = q < NQ4 ? C.wkey[c] : 0.f;
    }
    const unsigned lane8 = 8u * lane;   // (the 16- and 4-byte lane offsets and the zero offset of the q loads are temporaries of the trips)
    constexpr int O1 = (NQ4 > 1 ? 1 : 0) * 512, O2 = (NQ4 > 2 ? 2 : NQ4 - 1) * 512, O3 = (NQ4 > 3 ? 3 : NQ4 - 1) * 512;
    int lw = w * BD_RPW;   // this wave's row(s) of every block
    int* rec_ring;
    unsigned rec_ring_a, pl;
    const int32_t* rec_w;
    auto set_row = [&](int row) {
        lw = row;
        rec_ring = lds.rec + (set * DF_RB + lw) * (8 * BD_RECW);
        rec_ring_a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)rec_ring);
        rec_w = recs + BD_RECW * lw;   // (uniform; lane l adds its 4 l bytes inside the DMA statements)
        // L2 warm-up (see BD_STAT_1): lane l asks for line l of the next block's static record; the dump area is this row's
        pl = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(reinterpret_cast<unsigned*>(lds.dump) + ((set * DF_RB + lw) * 64)));
    };
    set_row(lw);
    const int64_t wstride = BD_RECW * DF_RB;
    // block j of this wave (the j-th from the END of the stream's record list: the sweep runs the layers in reverse)
    auto rec_src = [&](int j) -> const void* { return bd_sptr(rec_w + (int64_t)(nblk - 1 - min(j, nblk - 1)) * wstride); };
    auto rec_dst = [&](int j) -> unsigned { return rec_ring_a + (j & 7) * (BD_RECW * 4); };
    auto glds4 = [&](const void* gsrc, unsigned lds_dst) {   // (lane l moves word l of the record)
        unsigned keep, off;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tv_lshrrev_b32 %1, 1, %4\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&v"(off) : "s"(gsrc), "s"(lds_dst), "v"(lane8) : "memory");
    };
    if (nblk > 0) {
        for (int rr = 0; rr < BD_RPW; ++rr) {
            set_row(w * BD_RPW + rr);
#pragma unroll
            for (int j = 0; j < BD_RD; ++j) glds4(rec_src(j), rec_dst(j));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // the slice's share of a full row in the loader's column layout: units [32 sl, 32 sl + 32) = column block q = sl / 2,
    // lanes [32 (sl & 1), + 32)
    const bool local_st = lds.local[0] != 0;
    (void)local_st;
    const int myq = __builtin_amdgcn_readfirstlane(sl >> 1), odd = __builtin_amdgcn_readfirstlane(sl & 1);
    const bool mine = (lane >> 5) == odd;
    // the slice's column block of a row array.  (As `myq == q ? a[q] : r` the compiler turns the chain into a dynamically indexed
    // array in SCRATCH memory - five round trips to memory per row, two of them in front of the ready flag: v_cndmask by hand.)
    // (the lane masks are made per row from the 32-bit `myq` / `odd`: 64-bit masks kept across the sweep end up parked in VGPRs)
    unsigned long long pmask[5];
    auto make_masks = [&]() {
        asm volatile("s_cmp_eq_u32 %[q], 1\n\ts_cselect_b64 %[p1], -1, 0\n\ts_cmp_eq_u32 %[q], 2\n\ts_cselect_b64 %[p2], -1, 0\n\t"
                     "s_cmp_eq_u32 %[q], 3\n\ts_cselect_b64 %[p3], -1, 0\n\ts_cmp_eq_u32 %[q], 4\n\ts_cselect_b64 %[p4], -1, 0"
                     : [p1] "=&s"(pmask[1]), [p2] "=&s"(pmask[2]), [p3] "=&s"(pmask[3]), [p4] "=&s"(pmask[4])
                     : [q] "s"(myq) : "scc");
    };
    auto pick = [&](const float (&a)[NC]) -> float {
        float r = a[0];
#pragma unroll
        for (int q = 1; q < NC; ++q) asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(a[q]), "s"(pmask[q]));
        return r;
    };

    BdSweep A;
#ifdef BD_STAMPS
    const bool prof = S.dbg && lds.local[1] != 0 && w == 0 && lane == 0;
    unsigned long long npoll = 0;
#endif
    for (int b = 0; b < nblk; ++b) {
        const int j = b;
#pragma unroll 1
      for (int rr = 0; rr < BD_RPW; ++rr) {
        if (BD_RPW > 1) set_row(w * BD_RPW + rr);
        BD_STAMP(prof, b, set, 0);   // row start
        const int cur = rec_ring[(j & 7) * BD_RECW + lane];
#define BD_W(i) __builtin_amdgcn_readlane(cur, i)
        const int v = BD_W(0), eb = BD_W(1), ee = BD_W(2);
        const int s4[4] = {BD_W(4), BD_W(5), BD_W(6), BD_W(7)};
        // the first four edges' features and attention weights ride in the record (bd_records_kernel): no loads in front of the trip
        const float rf0[4] = {__int_as_float(BD_W(16)), __int_as_float(BD_W(17)), __int_as_float(BD_W(18)), __int_as_float(BD_W(19))};
        const float rf1[4] = {__int_as_float(BD_W(20)), __int_as_float(BD_W(21)), __int_as_float(BD_W(22)), __int_as_float(BD_W(23))};
        const float ral[4] = {__int_as_float(BD_W(al_w)), __int_as_float(BD_W(al_w + 1)), __int_as_float(BD_W(al_w + 2)),
                              __int_as_float(BD_W(al_w + 3))};
#undef BD_W
        const int slot = b % BD_NSLOT;
        float* sbase = lds.ring + (set * BD_NSLOT + slot) * Slot::words;
        int* v_s = reinterpret_cast<int*>(sbase + Slot::v_off);
        const void* ra = rec_src(j + BD_RD);
        const unsigned rl = rec_dst(j + BD_RD);
        const int vnext = __builtin_amdgcn_readfirstlane(rec_ring[((j + 1) & 7) * BD_RECW]);   // (past the end: the last block's again)
        const void* pa = bd_sptr(stat + (int64_t)max(vnext, 0) * SREC);   // (+ 128 lane bytes in the statement)
        if (v >= 0) {
            const int deg = ee - eb;
            const float* sa = stat + (int64_t)v * SREC;
            const float* sb = sa + 4 * BD_SP;
            const float* sc = sa + BD_NSTAT * BD_SP;   // part B of the record (H = 320)
            float ST5[BD_NSTAT];                       // ... the lane's fifth column of the eight static rows
            const gran_t* ub = HAS_DU ? du_in + (unsigned)v * (unsigned)gld : da_g;
            float acc[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) acc[q] = 0.f;
            float sig = 0.f, m0 = 0.f, m1 = 0.f;
            bf4 ST[BD_NSTAT];   // the node's static rows: outputs of the FIRST trip only, so they stay put across re-polls
            float du[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) du[q] = 0.f;
            if (NC == 4) { (void)sc; }
            // Chunks of <= 4 successors, each ONE statically shaped body (round 4; see df_loader_fast in dataflow.hip for what a
            // wave's time is made of): the trip is a fixed asm statement per (successors, du, first), the re-poll loop is
            // [trip, minimum of the tags, compare, branch] - tags never exceed the pass's epoch, so "all landed" is "the minimum
            // equals" - and the pull runs over exactly NN rows.
            auto chunk = [&](auto nn_c, auto first_c, int c0) {
                constexpr int NN = decltype(nn_c)::value;
                constexpr bool FIRST = decltype(first_c)::value;
                constexpr int DU = (FIRST && HAS_DU) ? 1 : 0;
                int pj[4] = {0, 0, 0, 0};
                float al[4] = {0.f, 0.f, 0.f, 0.f}, f0[4] = {0.f, 0.f, 0.f, 0.f}, f1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < NN; ++e) {
                    if (FIRST) { pj[e] = s4[e]; al[e] = ral[e]; f0[e] = rf0[e]; f1[e] = rf1[e]; }
                    else {
                        pj[e] = col[eb + c0 + e];
                        al[e] = alpha[eidx[eb + c0 + e]];
                        // (unconditional: `R >= 1` as a hoisted condition ends up as a 0 / 1 VGPR in scratch memory, reloaded in
                        // front of every trip; with R = 0 the loads read alpha[0], with R = 1 f1 = f0 - neither sum is stored then)
                        f0[e] = ea[(int64_t)(eb + c0 + e) * R];
                        f1[e] = ea[(int64_t)(eb + c0 + e) * R + r1];
                    }
                }
                // (N * 3 gld granules fit 32 bits - host check: one 32-bit multiply per row base)
                const gran_t* b0 = da_g + (unsigned)pj[0] * (unsigned)gld;
                const gran_t* b1 = da_g + (unsigned)pj[1] * (unsigned)gld;
                const gran_t* b2 = da_g + (unsigned)pj[2] * (unsigned)gld;
                const gran_t* b3 = da_g + (unsigned)pj[3] * (unsigned)gld;
                const gran_t* c0p = q_in + pj[0], * c1p = q_in + pj[1], * c2p = q_in + pj[2], * c3p = q_in + pj[3];
                unsigned keep_m0, tmp_vz, tmp_vs, tmp_vt;
                (void)tmp_vs; (void)tmp_vt;
                BdSweep& W = A;
                auto landed = [&]() -> bool {
                    unsigned m = epoch;
#pragma unroll
                    for (int e = 0; e < NN; ++e) {
#pragma unroll
                        for (int q = 0; q < NC; ++q) m = min(m, (unsigned)(A.x[e][q] >> 32));
                        m = min(m, (unsigned)(A.q[e] >> 32));
                    }
                    if (DU) {
#pragma unroll
                        for (int q = 0; q < NC; ++q) m = min(m, (unsigned)(A.u[q] >> 32));
                    }
                    return __builtin_amdgcn_uicmp(m, epoch, 33 /* ICMP_NE */) == 0ull;
                };
                if (FIRST) { BD_TRIP_SEL(NN, DU, 1); } else { BD_TRIP_SEL(NN, 0, 0); }
#ifdef BD_STAMPS
                if (FIRST) { BD_STAMP(prof, b, set, 2); npoll = 1; }   // first trip back (static rows + the first look at the successors)
                unsigned long long t_issue = 0;
#endif
                if (NN + DU > 0 && !landed()) {
                    unsigned spins = 0;
                    // (a lost pass leaves the loop BEHIND its trip: with a way out in front of it the rows of the previous
                    // trip stay live across the statement and the register allocator copies the whole sweep every turn)
                    bool more;
                    do {
                        more = bd_retry(spins, err, spin_limit);
#ifdef BD_STAMPS
                        t_issue = wall_clock64(); ++npoll;
#endif
                        BD_TRIP_SEL(NN, DU, 0);
                    } while (more && !landed());
                }
#ifdef BD_STAMPS
                if (FIRST) { BD_STAMP(prof, b, set, 3); BD_STAMP_V(prof, b, set, 7, t_issue); }   // the successors' rows are here; when the winning poll was issued
#endif
                if (DU) {
#pragma unroll
                    for (int q = 0; q < NC; ++q) du[q] = __uint_as_float((unsigned)A.u[q]);
                }
                // pull: ds_e = alpha_e (da_w . h_v - q_w), G += alpha_e da_w
#pragma unroll
                for (int e = 0; e < NN; ++e) {
#define BD_X(q) __uint_as_float((unsigned)A.x[e][q])
                    float dot = BD_X(0) * ST[ST_H].x;
                    if (NQ4 > 1) dot = fmaf(BD_X(1), ST[ST_H].y, dot);
                    if (NQ4 > 2) dot = fmaf(BD_X(2), ST[ST_H].z, dot);
                    if (NQ4 > 3) dot = fmaf(BD_X(3), ST[ST_H].w, dot);
                    if (NQ4 > 4) dot = fmaf(BD_X(4), ST5[ST_H], dot);
                    const float ds = al[e] * (bd_wave_sum(dot) - __uint_as_float((unsigned)A.q[e]));
                    sig += ds;
                    m0 = fmaf(ds, f0[e], m0);
                    m1 = fmaf(ds, f1[e], m1);
#pragma unroll
                    for (int q = 0; q < NC; ++q) acc[q] = fmaf(al[e], BD_X(q), acc[q]);
#undef BD_X
                }
            };
            typedef std::integral_constant<bool, true> first_t;
            typedef std::integral_constant<bool, false> later_t;
            if (b >= BD_NSLOT) bd_wait4(dn, b - BD_NSLOT + 1, err, spin_limit);   // the ring slot is free again (before the poll: off the dependent chain)
            BD_STAMP(prof, b, set, 1);   // slot free
            if (deg <= 0) chunk(std::integral_constant<int, 0>(), first_t(), 0);
            else if (deg == 1) chunk(std::integral_constant<int, 1>(), first_t(), 0);
            else if (deg == 2) chunk(std::integral_constant<int, 2>(), first_t(), 0);
            else if (deg == 3) chunk(std::integral_constant<int, 3>(), first_t(), 0);
            else {
                chunk(std::integral_constant<int, 4>(), first_t(), 0);
                for (int c0 = 4; c0 < deg; c0 += 4) {
                    const int nn = min(4, deg - c0);
                    if (nn == 4) chunk(std::integral_constant<int, 4>(), later_t(), c0);
                    else if (nn == 3) chunk(std::integral_constant<int, 3>(), later_t(), c0);
                    else if (nn == 2) chunk(std::integral_constant<int, 2>(), later_t(), c0);
                    else chunk(std::integral_constant<int, 1>(), later_t(), c0);
                }
            }
#ifdef BD_STAMPS
            { asm volatile("" :: "v"(acc[0]), "v"(acc[NC - 1]), "v"(sig)); BD_STAMP(prof, b, set, 4); BD_STAMP_V(prof, b, set, 8, npoll); }   // pulls done
#endif
            make_masks();
            // G_v, then everything that is linear in it
            auto stv = [&](int r, int q) -> float { return q < 4 ? ST[r][q] : ST5[r]; };   // (q is a constant after unrolling)
            float G[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) G[q] = stv(ST_GEXT, q) + du[q] + acc[q] + sig * wk[q];
            float dr[NC], dz[NC], dnr[NC], dnn[NC], zg[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                dr[q] = G[q] * stv(ST_CR, q); dz[q] = G[q] * stv(ST_CZ, q); dnr[q] = G[q] * stv(ST_CNR, q);
                dnn[q] = G[q] * stv(ST_CN, q); zg[q] = G[q] * stv(ST_Z, q);
            }
            float* op = sbase + Slot::op_off + lw * Slot::AP;
#pragma unroll
            for (int q = 0; q < NQ4; ++q) {
                op[cpos[q]] = dr[q];
                op[DF_RB * Slot::AP + cpos[q]] = dz[q];
                op[2 * DF_RB * Slot::AP + cpos[q]] = dnr[q];
            }
            if (mine) {
                sbase[Slot::zg_off + lw * DF_JS + (lane & 31)] = pick(zg);
                sbase[Slot::dn_off + lw * DF_JS + (lane & 31)] = pick(dnn);
            }
            v_s[lw] = v;   // (every lane: same word, same value)
            // The row's outputs to memory (q, sigma, the edge-feature sums, the slice's columns of dgi / dgh) are the COMPUTE
            // waves' job (round 4: a row costs its loader wave ~3 us of single-wave instruction issue, the compute waves idle
            // two thirds of the sweep): compute wave r stores row r, from the operand rows it reads anyway + what follows
            float qd = G[0] * ST[ST_CQ].x;   // q_v = G_v . c_q,v: the per-lane parts
            if (NQ4 > 1) qd = fmaf(G[1], ST[ST_CQ].y, qd);
            if (NQ4 > 2) qd = fmaf(G[2], ST[ST_CQ].z, qd);
            if (NQ4 > 3) qd = fmaf(G[3], ST[ST_CQ].w, qd);
            if (NQ4 > 4) qd = fmaf(G[4], ST5[ST_CQ], qd);
            const float mr = pick(dr), mz = pick(dz), mn = pick(dnn), mnr = pick(dnr);
            if (sl == BD_SCAL_SL) {   // (every lane: same words, same values)
                float* sc = sbase + Slot::sc_off + lw * 4;
                sc[0] = sig; sc[1] = m0; sc[2] = m1;
            }
            if (rr == BD_RPW - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bd_flag_st(lds.rdy + set * BD_WPS + w, b + 1);
            }
            BD_STAMP(prof, b, set, 5);   // flag raised
            if (sl == 0) {   // (next to the compute waves' products: off the dependent chain)
                qd = bd_wave_sum(qd);
                if (lane == 0) {
                    if (local_st) C.q_g[v] = gran_pack(epoch, qd);
                    else __hip_atomic_store(C.q_g + v, gran_pack(epoch, qd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            {   // the slice's 32 columns: row bases in SGPRs (v is uniform), ONE per-lane byte offset, the half-wave that owns the
                // columns under an exec mask - as C++ the compiler keeps per-lane 64-bit base pointers alive across the sweep
                unsigned long long keep_e, mine_mask;
                unsigned c8, msh;   // byte offset of column 32 sl + (lane & 31) = 64 myq + lane in a granule row, made per row
                asm volatile("v_lshl_add_u32 %[c8], %[q], 9, %[l8]\n\ts_lshl_b32 %[sh], %[o], 5\n\ts_bfm_b64 %[mm], 32, %[sh]"
                             : [c8] "=&v"(c8), [mm] "=&s"(mine_mask), [sh] "=&s"(msh) : [l8] "v"(lane8), [q] "s"(myq), [o] "s"(odd));
                (void)msh;
                if (C.dgi_g) {
                    const char* pg0 = reinterpret_cast<const char*>(C.dgi_g) + (uint64_t)((unsigned)v * (unsigned)(3 * gld)) * 8u;
                    const char* pg1 = pg0 + (uint64_t)(unsigned)gld * 8u;
                    const char* pg2 = pg1 + (uint64_t)(unsigned)gld * 8u;
                    const gran_t g0 = gran_pack(epoch, mr), g1 = gran_pack(epoch, mz), g2 = gran_pack(epoch, mn);
                    if (local_st)   // (readers on this XCD: the lines stay in its L2)
                        asm volatile("s_mov_b64 %[ke], exec\n\ts_mov_b64 exec, %[mm]\n\t"
                                     "global_store_dwordx2 %[c8], %[g0], %[p0]\n\tglobal_store_dwordx2 %[c8], %[g1], %[p1]\n\t"
                                     "global_store_dwordx2 %[c8], %[g2], %[p2]\n\ts_mov_b64 exec, %[ke]"
                                     : [ke] "=&s"(keep_e)
                                     : [mm] "s"(mine_mask), [c8] "v"(c8), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [p0] "s"(pg0), [p1] "s"(pg1), [p2] "s"(pg2)
                                     : "memory");
                    else
                        asm volatile("s_mov_b64 %[ke], exec\n\ts_mov_b64 exec, %[mm]\n\t"
                                     "global_store_dwordx2 %[c8], %[g0], %[p0] sc1\n\tglobal_store_dwordx2 %[c8], %[g1], %[p1] sc1\n\t"
                                     "global_store_dwordx2 %[c8], %[g2], %[p2] sc1\n\ts_mov_b64 exec, %[ke]"
                                     : [ke] "=&s"(keep_e)
                                     : [mm] "s"(mine_mask), [c8] "v"(c8), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [p0] "s"(pg0), [p1] "s"(pg1), [p2] "s"(pg2)
                                     : "memory");
                }
                const char* og = reinterpret_cast<const char*>(C.dgi) + (uint64_t)(unsigned)v * (unsigned)(3 * H * 4);
                const char* oh = reinterpret_cast<const char*>(C.dgh) + (uint64_t)(unsigned)v * (unsigned)(3 * H * 4);
                const unsigned c4 = c8 >> 1;
                asm volatile("s_mov_b64 %[ke], exec\n\ts_mov_b64 exec, %[mm]\n\t"
                             "global_store_dword %[c4], %[mr], %[og]\n\tglobal_store_dword %[c4], %[mz], %[og] offset:%[h1]\n\t"
                             "global_store_dword %[c4], %[mn], %[og] offset:%[h2]\n\t"
                             "global_store_dword %[c4], %[mr], %[oh]\n\tglobal_store_dword %[c4], %[mz], %[oh] offset:%[h1]\n\t"
                             "global_store_dword %[c4], %[mnr], %[oh] offset:%[h2]\n\ts_mov_b64 exec, %[ke]"
                             : [ke] "=&s"(keep_e)
                             : [mm] "s"(mine_mask), [c4] "v"(c4), [mr] "v"(mr), [mz] "v"(mz), [mn] "v"(mn), [mnr] "v"(mnr),
                               [og] "s"(og), [oh] "s"(oh), [h1] "n"(4 * H), [h2] "n"(8 * H)
                             : "memory");
            }
            BD_STAMP(prof, b, set, 6);   // row outputs issued
        } else {
            glds4(ra, rl);   // an idle row keeps the record ring moving
            if (b >= BD_NSLOT) bd_wait4(dn, b - BD_NSLOT + 1, err, spin_limit);
            v_s[lw] = v;
            if (rr == BD_RPW - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bd_flag_st(lds.rdy + set * BD_WPS + w, b + 1);
            }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef BD_TRIP_FIRST
#undef BD_TRIP_POLL
#undef BD_TRIP_FIRST5
#undef BD_TRIP_POLL5
#undef BD_CASE
#undef BD_CASES

// ---- loader wave of an input-gradient cell: the node's dgi row (3 gate blocks of granules) -> the three operand rows
#define BD_G_LD(g)                                                               \
    "global_load_dwordx2 %[y" #g "0], %[vo], %[g" #g "] offset:0 sc1\n\t"        \
    "global_load_dwordx2 %[y" #g "1], %[vo], %[g" #g "] offset:%[o1] sc1\n\t"    \
    "global_load_dwordx2 %[y" #g "2], %[vo], %[g" #g "] offset:%[o2] sc1\n\t"    \
    "global_load_dwordx2 %[y" #g "3], %[vo], %[g" #g "] offset:%[o3] sc1\n\t"
#define BD_DMA_0 "s_waitcnt vmcnt(0)"
#define BD_DMA_1                                                                                   \
    "s_mov_b32 %[km], m0\n\ts_mov_b32 m0, %[rl]\n\t" \
    "v_lshrrev_b32 %[vs], 1, %[vo]\n\tglobal_load_lds_dword %[vs], %[ra]\n\t"                     \
    "s_mov_b32 m0, %[km]\n\ts_waitcnt vmcnt(1)"
#define BD_G_LD5(g) BD_G_LD(g) "global_load_dwordx2 %[y" #g "4], %[vo], %[g" #g "] offset:2048 sc1\n\t"
#define BD_GTRIP5(dm)                                                                                                        \
    asm volatile(BD_G_LD5(0) BD_G_LD5(1) BD_G_LD5(2) BD_DMA_##dm                                                             \
                 : [y00] "=v"(Y[0][0]), [y01] "=v"(Y[0][1]), [y02] "=v"(Y[0][2]), [y03] "=v"(Y[0][3]), [y04] "=v"(Y[0][4]),  \
                   [y10] "=v"(Y[1][0]), [y11] "=v"(Y[1][1]), [y12] "=v"(Y[1][2]), [y13] "=v"(Y[1][3]), [y14] "=v"(Y[1][4]),  \
                   [y20] "=v"(Y[2][0]), [y21] "=v"(Y[2][1]), [y22] "=v"(Y[2][2]), [y23] "=v"(Y[2][3]), [y24] "=v"(Y[2][4]),  \
                   [km] "=&s"(keep_m0), [vs] "=&v"(tmp_vs)                                                           \
                 : [vo] "v"(lane8), [g0] "s"(g0), [g1] "s"(g1), [g2] "s"(g2), [rl] "s"(rl), [ra] "s"(ra),      \
                   [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3)                                                                  \
                 : "memory")
#define BD_GTRIP(dm)                                                                                                         \
    asm volatile(BD_G_LD(0) BD_G_LD(1) BD_G_LD(2) BD_DMA_##dm                                                                \
                 : [y00] "=v"(Y[0][0]), [y01] "=v"(Y[0][1]), [y02] "=v"(Y[0][2]), [y03] "=v"(Y[0][3]),                       \
                   [y10] "=v"(Y[1][0]), [y11] "=v"(Y[1][1]), [y12] "=v"(Y[1][2]), [y13] "=v"(Y[1][3]),                       \
                   [y20] "=v"(Y[2][0]), [y21] "=v"(Y[2][1]), [y22] "=v"(Y[2][2]), [y23] "=v"(Y[2][3]),                       \
                   [km] "=&s"(keep_m0), [vs] "=&v"(tmp_vs)                                                           \
                 : [vo] "v"(lane8), [g0] "s"(g0), [g1] "s"(g1), [g2] "s"(g2), [rl] "s"(rl), [ra] "s"(ra),      \
                   [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3)                                                                  \
                 : "memory")

template <int KPT>
__device__ __forceinline__ void bd_loader_du(const BdArgs& S, const BdCell& C, int group, const BdLds& lds, int w, int set) {
    constexpr int H = 16 * KPT;
    constexpr int SEG = BdPad<KPT>::seg, KP8 = BdPad<KPT>::kp8;
    constexpr int NQ4 = H / 64;
    typedef BdSlot<KPT> Slot;
    const int lane = threadIdx.x & 63;
    const int d = C.dir;
    const int32_t* tab = S.sched + S.gtab[d] + 2 * group;
    const int rec_base = tab[0], nblk = tab[1];
    const int32_t* __restrict__ recs = S.brecs + S.brec[d] + BD_RECW * (int64_t)rec_base;
    const unsigned epoch = S.epoch, spin_limit = S.spin_limit;
    int* const err = S.err;
    const int gld = S.gld;
    const gran_t* const dgi_in = C.dgi_g;
    int* const dn = lds.dn + set * DF_NCW;
    constexpr int NC = NQ4 > 4 ? NQ4 : 4;
    int cpos[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = 64 * q + lane;
        cpos[q] = (64 % KP8 == 0) ? lane + (SEG - KP8) * (lane / KP8) + q * (64 + (SEG - KP8) * (64 / KP8)) : c + (SEG - KP8) * (c / KP8);
    }
    const unsigned lane8 = 8u * lane;
    constexpr int O1 = (NQ4 > 1 ? 1 : 0) * 512, O2 = (NQ4 > 2 ? 2 : NQ4 - 1) * 512, O3 = (NQ4 > 3 ? 3 : NQ4 - 1) * 512;
    int lw = w * BD_RPW;
    int* rec_ring;
    unsigned rec_ring_a;
    const int32_t* rec_w;
    auto set_row = [&](int row) {
        lw = row;
        rec_ring = lds.rec + (set * DF_RB + lw) * (8 * BD_RECW);
        rec_ring_a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)rec_ring);
        rec_w = recs + BD_RECW * lw;   // (uniform; lane l adds its 4 l bytes inside the DMA statements)
    };
    set_row(lw);
    const int64_t wstride = BD_RECW * DF_RB;
    auto rec_src = [&](int j) -> const void* { return bd_sptr(rec_w + (int64_t)(nblk - 1 - min(j, nblk - 1)) * wstride); };
    auto rec_dst = [&](int j) -> unsigned { return rec_ring_a + (j & 7) * (BD_RECW * 4); };
    auto glds4 = [&](const void* gsrc, unsigned lds_dst) {   // (lane l moves word l of the record)
        unsigned keep, off;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tv_lshrrev_b32 %1, 1, %4\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&v"(off) : "s"(gsrc), "s"(lds_dst), "v"(lane8) : "memory");
    };
    if (nblk > 0) {
        for (int rr = 0; rr < BD_RPW; ++rr) {
            set_row(w * BD_RPW + rr);
#pragma unroll
            for (int j = 0; j < BD_RD; ++j) glds4(rec_src(j), rec_dst(j));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    gran_t Y[3][5];
    unsigned tmp_vs;
    for (int b = 0; b < nblk; ++b) {
        const int j = b;
#pragma unroll 1
      for (int rr = 0; rr < BD_RPW; ++rr) {
        if (BD_RPW > 1) set_row(w * BD_RPW + rr);
        const int v = __builtin_amdgcn_readfirstlane(rec_ring[(j & 7) * BD_RECW]);
        const int slot = b % BD_NSLOT;
        float* sbase = lds.ring + (set * BD_NSLOT + slot) * Slot::words;
        int* v_s = reinterpret_cast<int*>(sbase + Slot::v_off);
        const void* ra = rec_src(j + BD_RD);
        const unsigned rl = rec_dst(j + BD_RD);
        if (v >= 0) {
            const gran_t* g0 = dgi_in + (unsigned)v * (unsigned)(3 * gld);
            const gran_t* g1 = g0 + gld, * g2 = g0 + 2 * gld;
            unsigned spins = 0;
            bool dma = true;
            for (;;) {
                unsigned keep_m0;
                if constexpr (NQ4 == 5) { if (dma) BD_GTRIP5(1); else BD_GTRIP5(0); } else { if (dma) BD_GTRIP(1); else BD_GTRIP(0); }
                dma = false;
                bool ok = true;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int q = 0; q < NC; ++q) ok = ok && (unsigned)(Y[g][q] >> 32) == epoch;
                if (__all(ok) || !bd_retry(spins, err, spin_limit)) break;
            }
            if (b >= BD_NSLOT) bd_wait4(dn, b - BD_NSLOT + 1, err, spin_limit);
            float* op = sbase + Slot::op_off + lw * Slot::AP;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int q = 0; q < NQ4; ++q) op[g * DF_RB * Slot::AP + cpos[q]] = __uint_as_float((unsigned)Y[g][q]);
        } else {
            glds4(ra, rl);
            if (b >= BD_NSLOT) bd_wait4(dn, b - BD_NSLOT + 1, err, spin_limit);
        }
        if (lane == 0) v_s[lw] = v;
      }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bd_flag_st(lds.rdy + set * BD_WPS + w, b + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef BD_GTRIP
#undef BD_GTRIP5

// ---- compute wave `cw`: output units [8 cw, 8 cw + 8) of the slice (lane layout and reduce-scatter of dataflow.hip's
// df_compute; the A operands are the packed gate-wise transposed matrix, the B operands differ per gate block)
template <int KPT>
__device__ __forceinline__ void bd_compute(const BdArgs& S, const BdCell& C, int sl, int pair, const BdLds& lds, int cw) {
    constexpr int H = 16 * KPT;
    constexpr int SEG = BdPad<KPT>::seg, KP8 = BdPad<KPT>::kp8, NK4 = KP8 / 4;
    typedef BdSlot<KPT> Slot;
    const int tc = threadIdx.x & 255;
    const int lane = tc & 63;
    const int quad = lane >> 5, ks = (lane >> 2) & 7, x = lane & 3;
    const bool s0 = (ks & 1) != 0, s1 = (ks & 2) != 0;
    const bool is_da = C.kind == BD_DA;
    const int d = C.dir;
    int nb[DF_NLS];
#pragma unroll
    for (int q = 0; q < DF_NLS; ++q) {
        const int grp = bd_stream_group(pair, q, S.groups);
        nb[q] = grp >= 0 ? S.sched[S.gtab[d] + 2 * grp + 1] : 0;
    }
    float wr[KP8], wz[KP8], wn[KP8];
    {
        const float4* wp = C.w + (int64_t)sl * (3 * NK4) * 256 + tc;
#pragma unroll
        for (int q = 0; q < NK4; ++q) {
            const float4 x0 = wp[(0 * NK4 + q) * 256], x1 = wp[(1 * NK4 + q) * 256], x2 = wp[(2 * NK4 + q) * 256];
            wr[4 * q] = x0.x; wr[4 * q + 1] = x0.y; wr[4 * q + 2] = x0.z; wr[4 * q + 3] = x0.w;
            wz[4 * q] = x1.x; wz[4 * q + 1] = x1.y; wz[4 * q + 2] = x1.z; wz[4 * q + 3] = x1.w;
            wn[4 * q] = x2.x; wn[4 * q + 1] = x2.y; wn[4 * q + 2] = x2.z; wn[4 * q + 3] = x2.w;
        }
#pragma unroll
        for (int k = 0; k < KP8; ++k) asm volatile("" : "+v"(wr[k]), "+v"(wz[k]), "+v"(wn[k]));
    }
    const int unit_l = 8 * cw + 4 * quad + 2 * (ks & 1) + ((ks >> 1) & 1), unit = sl * DF_JS + unit_l;
    const int gr = x;
    const unsigned epoch = S.epoch, spin_limit = S.spin_limit;
    int* const err = S.err;
    gran_t* const out_g = C.out_g;
    const int gld = S.gld, num_nodes = S.N;
    const bool local_st = is_da && lds.local[0] != 0;
    const int R = C.mrel ? S.R : 0;
#ifdef BD_STAMPS
    const bool prof = S.dbg && lds.local[1] != 0 && cw == 0 && lane == 0;
#endif

    // The block loop, shaped like dataflow.hip's df_compute (round 6: ONE wave pays for every instruction it issues, so the
    // bookkeeping is scalar and branch-free, the next look at the ready flags leaves behind the last product, a lane reads the
    // one node id it needs, the done flag is stored without a predicate - the other lanes write into the DMA dump area - and
    // the store variants are hoisted out of the loop).
    typedef int i4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const volatile i4v* lds_i4p;
    const lds_i4p rdy_p = (lds_i4p)(uintptr_t)(unsigned)(uintptr_t)lds.rdy;   // (an LDS address is the low 32 bits of the generic pointer)
    static_assert(DF_NLS == 2 && BD_WPS == 4, "two streams, four ready flags each");
    const bool lane_st = (lane & 16) == 0;   // (lanes 16 away hold the same sums)
    int* const dn_or_dump = lane == 0 ? lds.dn + cw : lds.dump + lane;
    const int dn_step = lane == 0 ? DF_NCW : 0;
    const int nb0 = nb[0], nb1 = nb[DF_NLS - 1];
    const bool scal = sl == BD_SCAL_SL;

    auto run = [&](auto local_c, auto da_c) {
        constexpr bool LOCAL = decltype(local_c)::value, DA = decltype(da_c)::value;
        int done0 = 0, done1 = 0, pref = 0;
        int m0 = 0, m1 = 0;   // blocks the streams' loaders had finished at the last look (wave-uniform)
        auto flags_min = [&](const i4v& r0, const i4v& r1) {
            m0 = __builtin_amdgcn_readfirstlane(min(min(r0.x, r0.y), min(r0.z, r0.w)));
            m1 = __builtin_amdgcn_readfirstlane(min(min(r1.x, r1.y), min(r1.z, r1.w)));
        };
        int slot0 = 0, slot1 = 0;   // ring slots of the streams' next blocks (BD_NSLOT need not be a power of two)
        for (int left = nb0 + nb1; left > 0; --left) {
            int l0 = m0 - done0, l1 = m1 - done1;   // leads (>= 0: a loader stops at its stream's last block)
            if (l0 <= 0 && l1 <= 0) {   // nothing known to be ready: look until there is
                unsigned spins = 0;
                for (;;) {
                    const i4v r0 = rdy_p[0], r1 = rdy_p[1];
                    flags_min(r0, r1);
                    l0 = m0 - done0; l1 = m1 - done1;
                    if (l0 > 0 || l1 > 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    bool give_up = false;
                    if (++spins > 4 * spin_limit) {
                        __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        give_up = true;
                    }
                    if ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = true;
                    if (give_up) { l0 = nb0 - done0; l1 = nb1 - done1; break; }
                }
            }
            // smallest positive lead first, ties alternate (one unsigned comparison: dataflow.hip)
            const unsigned k0 = ((unsigned)(l0 - 1) << 1) | (unsigned)pref, k1 = ((unsigned)(l1 - 1) << 1) | (unsigned)(pref ^ 1);
            const int st = k1 < k0 ? 1 : 0;
            pref = st ^ 1;
            const int b = st ? done1 : done0;
            const int slot = st ? slot1 : slot0;
            done0 += st ^ 1;
            done1 += st;
            { const int nx = slot + 1 == BD_NSLOT ? 0 : slot + 1; if (st) slot1 = nx; else slot0 = nx; }
            BD_STAMP(prof, b, st, 9);   // block seen
            const float* sbase = lds.ring + (st * BD_NSLOT + slot) * Slot::words;
            // every read of the slot leaves in one go (the values read for rows that turn out idle are never stored)
            const int gv = reinterpret_cast<const int*>(sbase + Slot::v_off)[gr];
            const float zgv = DA ? sbase[Slot::zg_off + gr * DF_JS + unit_l] : 0.f;   // (input-gradient cells: no such term)
            int row_v = -1;
            float o_s = 0.f;
            if (DA && scal) {   // row `cw` of the block: this wave stores its sigma_v and edge-feature sums
                row_v = reinterpret_cast<const int*>(sbase + Slot::v_off)[cw];
                o_s = sbase[Slot::sc_off + cw * 4 + (lane & 3)];
            }
            const float* a_seg = sbase + Slot::op_off + x * Slot::AP + ks * SEG;   // gate block 0, row x, K slice ks
            float gsum;
            {
                bf4 acc[3] = {(bf4){0.f, 0.f, 0.f, 0.f}, (bf4){0.f, 0.f, 0.f, 0.f}, (bf4){0.f, 0.f, 0.f, 0.f}};
#ifndef BD_EXP_NOMFMA   // (timing experiment, scripts/build_variant.sh)
#pragma unroll
                for (int q = 0; q < NK4; ++q) {
                    const float4 b0 = *reinterpret_cast<const float4*>(a_seg + 4 * q);
                    const float4 b1 = *reinterpret_cast<const float4*>(a_seg + DF_RB * Slot::AP + 4 * q);
                    const float4 b2 = *reinterpret_cast<const float4*>(a_seg + 2 * DF_RB * Slot::AP + 4 * q);
                    const float q0[4] = {b0.x, b0.y, b0.z, b0.w}, q1[4] = {b1.x, b1.y, b1.z, b1.w}, q2[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * q + e], q0[e], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wz[4 * q + e], q1[e], acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wn[4 * q + e], q2[e], acc[2], 0, 0, 0);
                    }
                }
#endif
                // the look for the NEXT block: issued behind the last product, read at the top of the next iteration
                __builtin_amdgcn_sched_barrier(0);
                const i4v r0 = rdy_p[0], r1 = rdy_p[1];
                const bf4 t = acc[0] + acc[1] + acc[2];   // the three gate blocks of K: one reduce-scatter for their sum
#ifdef BD_STAMPS
                { asm volatile("" :: "v"(t[0]), "v"(t[3])); BD_STAMP(prof, b, st, 10); }   // products done
#endif
                const float u0 = t[0] + bd_dpp<0x104>(t[0]), u1 = t[1] + bd_dpp<0x104>(t[1]);
                const float u2 = t[2] + bd_dpp<0x114>(t[2]), u3 = t[3] + bd_dpp<0x114>(t[3]);
                const float e0 = s0 ? u2 : u0, e1 = s0 ? u3 : u1;
                const float f0 = e0 + bd_dpp<0x108>(e0), f1 = e1 + bd_dpp<0x118>(e1);
                gsum = bd_row_pair_sum(s1 ? f1 : f0);
                flags_min(r0, r1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bd_flag_st(dn_or_dump + st * dn_step, b + 1);   // this wave is done with the slot
            // (the bound on the node id also covers padding rows, id -1, and whatever a slot holds once a wait has expired)
            if (lane_st && (unsigned)gv < (unsigned)num_nodes) {
                gran_t* po = out_g + (__umul24((unsigned)gv, (unsigned)gld) + (unsigned)unit);   // (N * gld < 2^31, N < 2^24: host check)
                if (LOCAL) *po = gran_pack(epoch, gsum + zgv);
                else __hip_atomic_store(po, gran_pack(epoch, gsum + zgv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (DA && scal && (unsigned)row_v < (unsigned)num_nodes) {   // off the dependent chain
                const int o_si = __float_as_int(o_s);
                const float s_sig = __int_as_float(__builtin_amdgcn_readlane(o_si, 0)), s_m0 = __int_as_float(__builtin_amdgcn_readlane(o_si, 1)),
                            s_m1 = __int_as_float(__builtin_amdgcn_readlane(o_si, 2));
                if (lane == 0) {
                    C.sig[row_v] = s_sig;
                    if (R >= 1) C.mrel[(int64_t)row_v * R] = s_m0;
                    if (R >= 2) C.mrel[(int64_t)row_v * R + 1] = s_m1;
                }
            }
            BD_STAMP(prof, b, st, 11);   // stores issued
        }
    };
    if (is_da) { if (local_st) run(std::true_type(), std::true_type()); else run(std::false_type(), std::true_type()); }
    else run(std::false_type(), std::false_type());
}

template <int KPT>
__global__ void __launch_bounds__(BD_THREADS, BD_THREADS / 256) bwd_dataflow_kernel(const int32_t* __restrict__ plan, BdArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef BdSlot<KPT> Slot;
    constexpr int NS = 16 * KPT / DF_JS;
    const int tid = threadIdx.x;
    if (S.status && S.status[0] != 0) {
        if (tid == 0) __hip_atomic_fetch_or(S.err, 4 | ((S.status[0] & 0xff) << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (S.sched[0] != S.groups || S.sched[1] != DF_MAGIC || S.sched[2] != DF_RB) {
        if (tid == 0) __hip_atomic_fetch_or(S.err, 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pair, c, sl;
    if (S.nroles > 0) {
        const unsigned role = S.role[blockIdx.x];
        if (role == BD_IDLE_ROLE) return;
        pair = (int)(role >> 10); c = (int)((role >> 5) & 31u); sl = (int)(role & 31u);
    } else {
        const int per_pair = S.ncell * NS;
        pair = blockIdx.x / per_pair;
        const int rem = blockIdx.x - pair * per_pair;
        c = rem / NS; sl = rem - c * NS;
    }
    const BdCell& C = S.cell[c];
    BdLds lds;
    lds.ring = smem;
    lds.rec = reinterpret_cast<int*>(lds.ring + DF_NLS * BD_NSLOT * Slot::words);
    int* flags = lds.rec + DF_NLS * DF_RB * 8 * BD_RECW;
    lds.rdy = flags;
    lds.dn = flags + BD_NLW;
    lds.dump = flags + BD_NLW + DF_NLS * DF_NCW;
    lds.local = lds.dump + DF_NLS * DF_RB * 64;
    if (tid < BD_NLW + DF_NLS * DF_NCW) flags[tid] = 0;
    if (tid == 0) lds.local[0] = 0;
#ifdef BD_STAMPS
    if (tid == 0) lds.local[1] = S.dbg && (unsigned)((pair << 10) | (c << 5) | sl) == S.dbg_role;
    if (S.dbg && tid == 0) S.dbg[2 * blockIdx.x] = wall_clock64();
    if (S.dbg && tid == 0 && blockIdx.x == 0) S.dbg[(1 << 20) - 1] = gridDim.x;
#endif
    if (S.nroles > 0 && wave == 0) {   // publish where this workgroup runs; state-gradient cells look at their unit
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        if ((tid & 63) == 0)
            __hip_atomic_store(S.xcc_tab + blockIdx.x, gran_pack(S.epoch, __uint_as_float(xcc)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (C.kind == BD_DA) {
            bool same = true;
            for (int b0 = 0; b0 < S.nroles; b0 += 64) {
                const int b = b0 + (tid & 63);
                const unsigned r = b < S.nroles ? S.role[b] : BD_IDLE_ROLE;
                const int rc = (int)((r >> 5) & 31u);
                const bool member = r != BD_IDLE_ROLE && (int)(r >> 10) == pair && (rc == c || rc == C.partner);
                unsigned spins = 0;
                bool have = !member;
                unsigned theirs = xcc;
                for (;;) {
                    if (!have) {
                        const gran_t g = gran_ld(S.xcc_tab + b);
                        if ((unsigned)(g >> 32) == S.epoch) { theirs = (unsigned)g; have = true; }
                    }
                    if (__all(have) || !bd_retry(spins, S.err, S.spin_limit)) break;
                }
                same = same && have && theirs == xcc;
            }
            if (__all(same) && (tid & 63) == 0) lds.local[0] = 1;
        }
    }
    __syncthreads();
    if (wave < DF_NCW) {
        bd_compute<KPT>(S, C, sl, pair, lds, wave);
#ifdef BD_STAMPS
        if (S.dbg && tid == 0) S.dbg[2 * blockIdx.x + 1] = wall_clock64();
#endif
    } else {
        const int set = (wave - DF_NCW) / BD_WPS, w = (wave - DF_NCW) % BD_WPS;
        const int grp = bd_stream_group(pair, set, S.groups);
        if (grp >= 0) {
            if (C.kind == BD_DU) bd_loader_du<KPT>(S, C, grp, lds, w, set);
            else if (C.du_in) bd_loader_da<KPT, true>(plan, S, C, sl, grp, lds, w, set);
            else bd_loader_da<KPT, false>(plan, S, C, sl, grp, lds, w, set);
        }
    }
}

template <int KPT> size_t bd_lds_bytes() {
    return (size_t)(DF_NLS * (BD_NSLOT * BdSlot<KPT>::words + DF_RB * 8 * BD_RECW) + BD_NLW + DF_NLS * DF_NCW + DF_NLS * DF_RB * 64 + 4) * 4 + 256;
}

}  // namespace

#ifdef BD_STAMPS
static unsigned long long* g_bd_dbg = nullptr;
static unsigned g_bd_dbg_role = 0;
// (stamps build only: the buffer and the role - (set << 10) | (cell << 5) | slice - of the stamped workgroup)
extern "C" void dagnn_debug_bwd_stamps(void* buf, unsigned role) { g_bd_dbg = (unsigned long long*)buf; g_bd_dbg_role = role; }
#endif
#ifdef BD_WIDE_TU
constexpr int BD_TU_MAX_H = 320;   // csrc/bwd_dataflow_w.hip: the 8-wave workgroup shape of H = 320
#else
constexpr int BD_TU_MAX_H = 256;
extern "C" size_t dagnn_bwd_dataflow_record_bytes(int64_t N) {
    if (N < 0) return 0;
    return (size_t)2 * BD_RECW * (4 * N + 4) * sizeof(int32_t);
}

extern "C" size_t dagnn_bwd_dataflow_static_bytes(int64_t N) {
    if (N < 0) return 0;
    return (size_t)N * BD_NSTAT * BD_SP * sizeof(float);
}

extern "C" size_t dagnn_bwd_dataflow_static_bytes_h(int64_t N, int H) {   // H = 320: 10 KB per (cell, node)
    if (N < 0 || H <= 0) return 0;
    return (size_t)N * bd_stat_floats(H) * sizeof(float);
}

extern "C" int dagnn_bwd_dataflow_prepare(const dagnn_plan* pl, const dagnn_bwd_dataflow_args* a, void* stream) {
    if (!pl || !pl->data || !a || !a->schedule || !a->records) return DAGNN_EINVAL;
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3, G = a->groups;
    if (H <= 0 || (H % 64) || H > 320 || Ls <= 0 || Ls > DAGNN_MAX_STACKED || !dir_mask || G < 1 || G > DF_MAX_GROUPS ||
        a->ld_h < H || a->ld_g < H)
        return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    const DfLayout SL = df_layout_words(pl->N, pl->B, G);
    const int64_t nrec = 4 * pl->N + 4;
    BdRecAlpha AL = {};
    AL.Ls = Ls; AL.R = pl->num_edge_feats;
    for (int d = 0; d < 2; ++d)
        for (int i = 0; i < Ls; ++i) {
            AL.alpha[d][i] = ((dir_mask >> d) & 1) ? a->cell[d][i].alpha : nullptr;
            if (((dir_mask >> d) & 1) && !AL.alpha[d][i]) return DAGNN_EINVAL;
        }
    hipLaunchKernelGGL(bd_records_kernel, dim3((unsigned)((nrec + 255) / 256), 2), dim3(256), 0, st, (const int32_t*)pl->data, L,
                       (const int32_t*)a->schedule, SL, (int32_t*)a->records, nrec, G, (const int32_t*)a->plan_status, AL);
    DAGNN_CHECK_LAUNCH();
    BdStatArgs A;
    A.ncell = 0; A.H = H; A.ld_h = a->ld_h; A.ld_g = a->ld_g;
    for (int d = 0; d < 2; ++d) {
        if (!((dir_mask >> d) & 1)) continue;
        for (int i = 0; i < Ls; ++i) {
            const dagnn_bwd_dataflow_cell& c = a->cell[d][i];
            if (!c.g_ext || !c.stat) return DAGNN_EINVAL;
            if (!a->stat_rows_written && (!c.gi || !c.gh || !c.a || !c.b_hh || !c.h)) return DAGNN_EINVAL;
            BdStatCell& K = A.cell[A.ncell++];
            K.gi = c.gi; K.gh = c.gh; K.a = c.a; K.b_hh = c.b_hh; K.h = c.h; K.gext = c.g_ext; K.stat = c.stat;
        }
    }
    if (a->stat_rows_written)   // (the forward pass of this step ran with dagnn_dataflow_args.stat_rows on these buffers)
        hipLaunchKernelGGL(bd_gext_kernel, dim3((unsigned)((pl->N + 3) / 4), (unsigned)A.ncell), dim3(256), 0, st, A, pl->N);
    else
        hipLaunchKernelGGL(bd_stat_kernel, dim3((unsigned)((pl->N + 3) / 4), (unsigned)A.ncell), dim3(256), 0, st, A, pl->N);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

#endif   // !BD_WIDE_TU

#ifdef BD_WIDE_TU
extern "C" int dagnn_bwd_dataflow_run_wide(const dagnn_plan* pl, const dagnn_bwd_dataflow_args* a, void* stream) {
    if (!pl || !pl->data || !a || !a->schedule || !a->records) return DAGNN_EINVAL;
    if (a->H <= 256) return DAGNN_EINVAL;
#else
extern "C" int dagnn_bwd_dataflow_run(const dagnn_plan* pl, const dagnn_bwd_dataflow_args* a, void* stream) {
    if (!pl || !pl->data || !a || !a->schedule || !a->records) return DAGNN_EINVAL;
    if (a->H > 256) return dagnn_bwd_dataflow_run_wide(pl, a, stream);   // H = 320: csrc/bwd_dataflow_w.hip
#endif
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3, G = a->groups;
    if (H <= 0 || (H % 64) || H > BD_TU_MAX_H || Ls <= 0 || Ls > DAGNN_MAX_STACKED || !dir_mask || a->gld < H || G < 1 ||
        G > DF_MAX_GROUPS || a->epoch == 0 || !a->err)
        return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    if ((int64_t)pl->N * 3 * a->gld >= (1ll << 31)) return DAGNN_EINVAL;   // row offsets inside the granule buffers are 32-bit
    BdArgs S;
    BdCell* cells = S.cell;
    int nc = 0;
    for (int d = 0; d < 2; ++d) {
        if (!((dir_mask >> d) & 1)) continue;
        for (int i = Ls - 1; i >= 0; --i) {   // the top stacked layer leads
            const dagnn_bwd_dataflow_cell& c = a->cell[d][i];
            if (!c.w_hh_t || !c.w_key || !c.alpha || !c.stat || !c.da_granules || !c.q_granules || !c.dgi || !c.dgh || !c.sigma ||
                (pl->num_edge_feats > 0 && !c.edge_feat_grad) || (i > 0 && (!c.w_ih_t || !c.dgi_granules)) ||
                (i + 1 < Ls && !c.du_granules))
                return DAGNN_EINVAL;
            if (nc + (i > 0 ? 2 : 1) > BD_MAX_KCELLS) return DAGNN_EINVAL;
            BdCell K = {};
            K.w = (const float4*)c.w_hh_t; K.wkey = c.w_key; K.alpha = c.alpha; K.stat = c.stat;
            K.du_in = i + 1 < Ls ? (const gran_t*)c.du_granules : nullptr;
            K.out_g = (gran_t*)c.da_granules; K.q_g = (gran_t*)c.q_granules;
            K.dgi_g = i > 0 ? (gran_t*)c.dgi_granules : nullptr;
            K.dgi = c.dgi; K.dgh = c.dgh; K.sig = c.sigma; K.mrel = pl->num_edge_feats > 0 ? c.edge_feat_grad : nullptr;
            K.dir = d; K.kind = BD_DA; K.partner = i > 0 ? nc + 1 : -1; K.layer = i;
            cells[nc++] = K;
            if (i > 0) {
                BdCell U = {};
                U.w = (const float4*)c.w_ih_t;
                U.dgi_g = (gran_t*)c.dgi_granules;
                U.out_g = (gran_t*)a->cell[d][i - 1].du_granules;
                U.dir = d; U.kind = BD_DU; U.partner = -1; U.layer = i;
                cells[nc++] = U;
            }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int NS = H / DF_JS;
    const int sets = (G + DF_NLS - 1) / DF_NLS;
    S.sched = (const int32_t*)a->schedule;
    S.brecs = (const int32_t*)a->records;
    const DfLayout SL = df_layout_words(pl->N, pl->B, G);
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    const int64_t nrec = 4 * pl->N + 4;
    for (int d = 0; d < 2; ++d) {
        S.gtab[d] = SL.gtab[d]; S.brec[d] = (int64_t)d * BD_RECW * nrec;
        S.col[d] = L.col[d]; S.eidx[d] = L.eidx[d]; S.eattr[d] = L.eattr[d];
    }
    S.ncell = nc; S.H = H; S.gld = a->gld; S.R = pl->num_edge_feats; S.groups = G; S.N = (int)pl->N;
    S.epoch = a->epoch; S.spin_limit = a->spin_limit ? a->spin_limit : (1u << 22);
    S.status = (const int32_t*)a->plan_status;
    S.err = (int*)a->err;
    unsigned grid = (unsigned)(sets * nc * NS);
    S.nroles = 0; S.xcc_tab = (gran_t*)a->xcc_table;
    if (a->num_cus >= 8 && a->num_cus <= BD_MAX_WGS && a->xcc_table && sets < 64 && nc <= 31) {   // (dataflow.hip: same packing)
        const int cap = a->num_cus / 8;
        int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int xrot = (a->xcd_first % 8 + 8) % 8;   // bin x of the packing = XCD (x + xcd_first) mod 8
        for (int b = 0; b < BD_MAX_WGS; ++b) S.role[b] = BD_IDLE_ROLE;
        bool ok = true;
        int top = 0;
        for (int pass = 0; pass < 2 && ok; ++pass)
            for (int set = 0; set < sets && ok; ++set)
                for (int c = 0; c < nc && ok; ++c) {
                    if (S.cell[c].kind != BD_DA || (S.cell[c].partner >= 0) != (pass == 0)) continue;
                    const int members[2] = {c, S.cell[c].partner};
                    const int size = NS * (members[1] >= 0 ? 2 : 1);
                    int x = 0;
                    while (x < 8 && fill[x] + size > cap) ++x;
                    if (x == 8) { ok = false; break; }
                    for (int m = 0; m < 2; ++m) {
                        if (members[m] < 0) continue;
                        for (int sl = 0; sl < NS; ++sl) {
                            const int b = fill[x]++ * 8 + (x + xrot) % 8;
                            S.role[b] = (unsigned short)((set << 10) | (members[m] << 5) | sl);
                            if (b + 1 > top) top = b + 1;
                        }
                    }
                }
        if (ok) { S.nroles = top; grid = (unsigned)top; }
    }
#ifdef BD_STAMPS
    S.dbg = g_bd_dbg; S.dbg_role = g_bd_dbg_role;
#endif
    const int32_t* plan = (const int32_t*)pl->data;
#define BD_LAUNCH(KPT)                                                                                                   \
    do {                                                                                                                 \
        const void* fn = reinterpret_cast<const void*>(bwd_dataflow_kernel<KPT>);                                        \
        const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bd_lds_bytes<KPT>()); \
        if (ea != hipSuccess) return DAGNN_EHIP(ea);                                                                     \
        hipLaunchKernelGGL((bwd_dataflow_kernel<KPT>), dim3(grid), dim3(BD_THREADS), bd_lds_bytes<KPT>(), st, plan, S);  \
    } while (0)
#ifdef BD_WIDE_TU
    BD_LAUNCH(20);
#else
    switch (H / 16) {
        case 4: BD_LAUNCH(4); break;
        case 8: BD_LAUNCH(8); break;
        case 12: BD_LAUNCH(12); break;
        default: BD_LAUNCH(16); break;
    }
#endif
#undef BD_LAUNCH
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
