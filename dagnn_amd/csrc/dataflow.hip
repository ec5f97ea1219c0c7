// dataflow.hip - the recurrence as ONE persistent, graph-affine dataflow launch (H <= 256).
//
// Reference path replaced: the three nested loops of ogbg-code/model/dagnn.py:144-182 (frontier selection :146-149,
// per-node edge scan :151-157, AttnConv :362-373, GRUCell :181, state write :182) and their D-VAE twins
// (dvae/dagnn.py:112-146, dvae/dagnn_bn.py:110-137), for every (direction, stacked layer) cell at once.
//
// Why this shape.  The batch is hundreds of topological layers deep with a handful of rows per layer: the time of
// the recurrence is (dependent hops) x (latency of one hop) for the deepest graph and (rows) x (cost of a row) for
// everything else.  Graphs share nothing, so the schedule is cut along GRAPHS, not along layers:
//   * the graphs are dealt to G independent groups (longest-processing-time first on cost = c_layer * depth +
//     c_row * nodes, csrc: df_assign_kernel; the deepest graph ends up almost alone in its group);
//   * a workgroup SET is ncell x H/32 workgroups, one per (kernel cell, 32-unit slice of the hidden dimension), each on
//     its own CU for the whole pass with its matrix slice in registers.  Kernel cells of a direction: its L recurrent
//     cells (W_hh) plus one PROJECTION cell per stacked layer above the first (W_ih u + b_ih, published as [N, 3H]
//     granules: the input-side product leaves the dependent chain).  A set serves TWO groups ("streams"): a stream
//     inside a thin dependent chain is ready once per hop (~3.5 us), the other stream's blocks fill the gap;
//   * a group walks ITS graphs layer by layer in blocks of <= 4 rows (records re-sorted by (group, layer, graph),
//     every group-layer padded to whole blocks, so block b of a group is records [4b, 4b + 4) - no indirection on
//     the dependent chain);
//   * rows travel between the workgroups of a set as 8-byte {epoch, fp32} granules (one write-through store each,
//     polled with relaxed agent-scope loads: cdna_hip_programming.md Guideline 16, form R2) - no barrier, no fence,
//     nothing placement-dependent.  Groups never exchange anything, and workgroup ids are set-major: with in-order
//     dispatch a partially resident grid still makes progress set by set.
// Inside a workgroup the waves are specialised (2 x 4 loader + 4 compute waves, coupled only through LDS flags):
//   loader wave w of set s   row w of every block of stream s: row record and gi0 slice by LDS-DMA a few blocks
//                   ahead, poll the predecessor rows (and the node's projection granules), attention softmax over the
//                   in-edges (scores = w_key . h_j computed HERE from the polled row: a DPP wave reduction, so the
//                   producers publish no score parts), aggregate -> the stream's LDS ring slot, ready flag;
//   compute wave c  its 8 hidden units of the slice, blocks of either stream as they become ready: products on
//                   v_mfma_f32_4x4x1 (the resident weights are the A operands, the block's rows the B operands), K
//                   slices summed by a DPP / v_permlane16_swap reduce-scatter, gates in the same lanes (no workgroup
//                   barrier anywhere), h' -> plain row store + granule store.
// The loaders run up to NSLOT blocks ahead, so wide layers stream while a thin dependent chain costs one hand-off +
// ~1 us of compute per hop.  DESIGN.md section 4a has the measurements.
#include "sched_dev.h"
#include <type_traits>


namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- the schedule's kernels: one body of sched_dev.h each (csrc/prepare.hip runs the same bodies several per launch)
// LPT assignment by workgroup 0; workgroups 1.. of the launch initialise the rest of the workspace meanwhile (tables and
// counters = 0, records = -1: two memsets less on a launch-bound path)
__global__ void __launch_bounds__(256) df_assign_kernel(const int32_t* __restrict__ plan, PlanLayout L, int32_t* ws,
                                                         DfLayout S, int B, int G, int c_layer, int c_row, const int32_t* __restrict__ status) {
    if (status && status[0] != 0) return;   // the batch violates the plan contract: nothing here can be trusted
    if (blockIdx.x > 0) {
        df_fill_body(ws, S, (int64_t)(blockIdx.x - 1) * blockDim.x + threadIdx.x, (int64_t)(gridDim.x - 1) * blockDim.x);
        return;
    }
    constexpr int CAP = 4096;
    __shared__ int32_t s_g[CAP], s_d[CAP], s_n[CAP];
    df_assign_block<CAP>(plan, L, ws, S, B, G, c_layer, c_row, s_g, s_d, s_n);
}

__global__ void __launch_bounds__(256) df_count_kernel(const int32_t* __restrict__ plan, PlanLayout L, int32_t* ws,
                                                        DfLayout S, const int32_t* __restrict__ status) {
    if (status && status[0] != 0) return;   // the batch violates the plan contract: nothing here can be trusted
    df_count_body(plan, L, ws, S, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(256) df_prefix_kernel(int32_t* ws, DfLayout S, int G, const int32_t* __restrict__ status) {
    if (status && status[0] != 0) return;   // the batch violates the plan contract: nothing here can be trusted
    df_prefix_body(ws, S, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(64) df_base_kernel(int32_t* ws, DfLayout S, int G, const int32_t* __restrict__ status) {
    if (status && status[0] != 0) return;   // the batch violates the plan contract: nothing here can be trusted
    const int base = df_base_wave(ws, S, G, blockIdx.x, threadIdx.x);
    if ((int)threadIdx.x < G) ws[S.gtab[blockIdx.x] + 2 * threadIdx.x] = base;
}

__global__ void __launch_bounds__(256) df_lbase_kernel(const int32_t* __restrict__ plan, PlanLayout L, int32_t* ws,
                                                        DfLayout S, int B, int G, const int32_t* __restrict__ status) {
    if (status && status[0] != 0) return;   // the batch violates the plan contract: nothing here can be trusted
    df_lbase_body(plan, L, ws, S, B, G, blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, blockIdx.y);
}

__global__ void __launch_bounds__(256) df_records_kernel(const int32_t* __restrict__ plan, PlanLayout L, int32_t* ws,
                                                          DfLayout S, int N, const int32_t* __restrict__ status) {
    if (status && status[0] != 0) return;   // the batch violates the plan contract: nothing here can be trusted
    df_records_body(plan, L, ws, S, N, blockIdx.x, blockIdx.y, ws + S.gtab[blockIdx.y], 2);
}

// ---------------------------------------------------------------- the persistent kernel
// Kernel cells.  A GRU cell of stacked layer i > 0 has two matrix products per node: W_hh x (aggregate of the
// predecessors' states) - on the dependent chain - and W_ih x (the node's own state one layer down), which is known
// one hop earlier.  They run as two kinds of kernel cell with the SAME compute path (one [3H x H] slice resident in
// registers, no weight stream):
//   RECURRENT  aggregate -> W_hh product -> gates -> h' (states + granules); its input-side pre-activations come
//              from gi0 (stacked layer 0: the batched GEMM) or from the granules of its PROJECTION cell;
//   PROJECTION lower-layer row -> W_ih product + b_ih -> [N, 3H] granules, off the dependent chain.
enum { DF_RECURRENT = 0, DF_PROJECTION = 1 };
// Code variants of a workgroup (template parameters of the loader / compute paths: the constructor-dependent branches,
// their operands and the scalar registers that carried them leave the per-block path).
//   KIND  0 recurrent cell of stacked layer 0 (input side from the gi0 ring), 1 recurrent cell above it (input side from
//         its projection cell's granules), 2 projection cell;
//   RR    2: exactly two edge features, gains folded inline (ogbg-code: `edge_attr` [E,2]); -1: any other count (run-time);
//   EXTRA static scores or vertex-id key biases present (`*_x` aggregators, D-VAE NA).
enum { DFK_REC0 = 0, DFK_RECP = 1, DFK_PROJ = 2 };
// per-block stamps (scripts/df_stamps.py) only in a build with -DDF_STAMPS: the disabled form still costs a branch and
// its scalar state in every block of every wave
#ifdef DF_STAMPS
constexpr bool DF_PROF = true;
#else
constexpr bool DF_PROF = false;
#endif

// A projection granule: the three input-side pre-activations of one unit of one node behind ONE tag - 16 bytes {tag, r, z, n},
// stored and loaded as a whole (global_store_dwordx4 / global_load_dwordx4 on 16-byte aligned addresses, write-through stores).
// The guide promises single-copy atomicity for 8 bytes; for an aligned 16 it was measured (scripts/ubench/tear16.hip: 1.4e10
// observed value changes across XCDs without a torn read, against 1 % torn on the 8-bytes-off control).  Against three
// 8-byte granules: a third less hand-off traffic, one load / store / tag check instead of three.
typedef unsigned pgran_t __attribute__((ext_vector_type(4)));

struct DfCell {
    const float4* w;      // packed slices (dagnn_pack_dataflow): W_hh (recurrent) or W_ih (projection)
    const float* bias;    // [3H] b_hh (recurrent; + b_ih folded by the projection) or b_ih (projection)
    const float* wkey;    // [H] or null (static scores)
    const float* sscore;  // [N] or null
    const float* gain;    // [R] or null
    const float* vid;     // [vid_mod] or null
    const float* gi0;     // recurrent, stacked layer 0: [N,3H] input-side pre-activations, else null
    const pgran_t* p_in;  // recurrent, stacked layers > 0: [N,pld] projection granules of its projection cell, else null
    float* h_out;         // recurrent: [N,ld_h]
    gran_t* g_out;        // recurrent: [N,gld] granules of h_out; projection: [N,pld] projection granules (pgran_t) of the pre-activations
    const gran_t* g_in;   // projection: granules of the lower stacked layer's states
    float* aux_out;       // training passes: [N,3H] plain copy of the pre-activations this cell computes (recurrent: W_hh a +
                          // b_hh; projection: W_ih u + b_ih), or null
    const float* agg_w;   // plain aggregators (`add` / `max`, dagnn.py:232-251): edge_encoder.weight [H, R] or null
    const float* agg_b;   //                                                       edge_encoder.bias [H] or null
    int dir;
    int kind;
    int variant;          // AGG * 16 + KIND * 4 + (RR == 2 ? 2 : 0) + EXTRA
    int partner;          // recurrent: index of the projection cell that reads this cell's state rows, or -1
    int agg;              // DAGNN_DF_AGG_*: 0 attention (the soft-max fold), 1 add, 2 max, 3 none (the aggregate is zero)
};

#define DF_MAX_KCELLS 24   // (24 x 136 bytes of cell table + the role table + the rest stay inside the 4 KB kernel-argument segment)
#define DF_MAX_WGS 320     // workgroups the XCD-aware placement table covers (one per CU)
#define DF_IDLE_ROLE 0xffffu

struct DfArgs {
    DfCell cell[DF_MAX_KCELLS];
    const int32_t* sched;   // schedule workspace (dagnn_dataflow_schedule)
    int64_t gtab[2], grec[2];   // word offsets into sched
    int64_t col[2], eattr[2];   // word offsets into the plan
    int ncell, H, ld_h, gld, pld, R, vid_mod, groups, N;
    unsigned epoch, spin_limit;
    int dbg_wg;                 // workgroup whose blocks are stamped
    const int32_t* status;      // plan status word (dagnn_plan_build), or null
    int* err;
    unsigned long long* dbg;    // optional: [grid][2] start / end stamps, then [blocks][8] stamps of workgroup dbg_wg (100 MHz)
    // XCD-aware placement (nroles > 0): role[b] = set << 10 | cell << 5 | slice of workgroup b (DF_IDLE_ROLE: none), chosen so
    // that - with the observed dispatch rule "workgroup b runs on XCD b % 8" - a recurrent cell's 8 slices and the
    // projection cell reading its rows share one XCD, i.e. one L2.  Nothing is ASSUMED about the placement: every
    // workgroup publishes the XCD it actually runs on (xcc_tab, tagged granules) and a recurrent cell keeps its state rows
    // in that L2 (plain stores instead of write-through ones) only when it SEES all their readers there.
    gran_t* xcc_tab;
    int nroles;
    int aux_stat;               // recurrent cells: aux_out is the reverse sweep's static-record buffer (rows 1..7 written here)
    unsigned short role[DF_MAX_WGS];
};

static_assert(sizeof(DfArgs) + 8 <= 4096, "the cell and role tables must fit the kernel-argument segment");

__device__ __forceinline__ float df_dpp_row_sum16(float v) {
#define DF_DPP_ADD(ctrl) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    DF_DPP_ADD(0x111); DF_DPP_ADD(0x112); DF_DPP_ADD(0x114); DF_DPP_ADD(0x118);
#undef DF_DPP_ADD
    return v;
}

// sum over the 64 lanes, broadcast as a wave-uniform value (row scans, then row_bcast15 / row_bcast31)
__device__ __forceinline__ float df_wave_sum(float v) {
    v = df_dpp_row_sum16(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// gate non-linearities on the hardware exp / rcp (both ~1 ulp): sigma(x) = 1 / (1 + e^-x), tanh(x) = 1 - 2 / (1 + e^2x)
// (saturates correctly: e^2x = inf -> 1, e^2x = 0 -> -1)
__device__ __forceinline__ float df_sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float df_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// operand rows in LDS: the K dimension is split over 8 lanes (KP8 = H / 8 = 2 KPT values each); K-lane segment s
// starts at s * (KP8 + 4): the 8 segments a half DPP row reads concurrently (ds_read_b128) fall on disjoint banks
// The forward kernel's compute shape: one compute wave per SIMD, 8 hidden units each (3 H/8 resident weights per lane), K split over
// 8 lanes.  (Round 6 measured the other shape the register file allows - two compute waves per SIMD with 4 units each, K over 16
// lanes, 16 waves per workgroup at <= 128 VGPRs: 1.50 against 1.33 ms on the headline batch; the complete kernel
// path is scripts/experiments/dataflow_split_r06.patch.)
// The slice shape of a translation unit (dataflow_x.hip sets the macros): DFF_JS hidden units per workgroup = 8 per compute wave,
// DFF_NLS streams per workgroup, 12 waves in all.  The default - 32 units, 4 compute waves, 2 streams x 4 loader waves - is the
// reverse sweep's shape too (df_common.h); the 64-unit shape has 8 compute waves (two per SIMD) and ONE stream of 4 loader waves.
#ifndef DFF_JS_V
#define DFF_JS_V DF_JS
#endif
#ifndef DFF_NLS_V
#define DFF_NLS_V DF_NLS
#endif
constexpr int DFF_JS = DFF_JS_V;
constexpr int DFF_NCW = DFF_JS / 8;
constexpr int DFF_NLS = DFF_NLS_V;
constexpr int DFF_NLW = DFF_JS == DF_JS ? DF_NLW : 12 - DFF_NCW;
constexpr int DFF_WPS = DFF_NLW / DFF_NLS;      // loader waves per stream
constexpr int DFF_RPW = DF_RB / DFF_WPS;        // rows of a block per loader wave (one after the other)
constexpr int DFF_THREADS = 64 * (DFF_NCW + DFF_NLW);
static_assert((DFF_NCW == 4 || DFF_NCW == 8) && DFF_NCW * 8 == DFF_JS && DFF_NLW % DFF_NLS == 0 && DF_RB % DFF_WPS == 0 && (DFF_NLS == 1 || DFF_NLS == 2), "workgroup shape");
// operand rows in LDS: the K dimension is split over 8 lanes (KP8 = H / 8 values each); K-lane segment s starts at
// s * (KP8 + 4): the 8 segments a half DPP row reads concurrently (ds_read_b128) fall on disjoint banks
template <int KPT> struct DfPad { static constexpr int kp8 = 2 * KPT; static constexpr int seg = kp8 + 4; static constexpr int row = 8 * seg + 8; };

constexpr int DF_CSLEEP_N = 1;   // s_sleep argument (x 64 cycles) of the compute waves' look at the ready flags
#ifndef DF_WSLEEP_V
#define DF_WSLEEP_V 1
#endif
constexpr int DF_WSLEEP_N = DF_WSLEEP_V;   // ... of a loader's wait for its ring slot
constexpr bool DF_LEAN_COMPUTE = true;   // (the 12-wave shape: the 2 x 4 ready flags are one trip to LDS; the 8-wave shape of H = 320 has 2 x 2)
constexpr int DF_RD = 6;       // a loader wave requests a row record this many of ITS blocks ahead (record ring: 8 entries)
constexpr int DF_GD = 2;       // a gi0 slice this many (its node id must have landed: DF_RD >= DF_GD + 2)
constexpr int DF_GI_LANES = 3 * DFF_JS / 4;          // lanes of a gi0 slice's DMA: 16 bytes each, gate-major
constexpr int DF_GIRING = DF_NSLOT + DF_GD + 2;   // blocks in a stream's gi0 ring: slots in use + prefetch distance + slack

// group served by stream `set` of workgroup set `pair` (-1: none).  (Measured and dropped: the first set serving the
// deepest graph's group alone, 9 groups on 5 sets - 2.15 ms against 1.93: what binds the pass is the sets'
// throughput, not that one chain.)
__device__ __host__ __forceinline__ int df_stream_group(int pair, int set, int groups) {
    const int g = DFF_NLS * pair + set;
    return g < groups ? g : -1;
}
__host__ inline int df_sets_for(int groups) { return (groups + DFF_NLS - 1) / DFF_NLS; }

struct DfLds {
    float* ring;     // [NLS][NSLOT] slots: a ring per stream
    float* giring;   // [NLS][DF_GIRING][RB][96]: gi0 slices of the slice's rows, landed by LDS-DMA two blocks ahead
    int* rec;        // [NLS * RB][8][16]: row records of the loader waves, landed by LDS-DMA DF_RD of their blocks ahead
    int* rdy;        // [NLS][WPS]  per loader wave: blocks it has finished (relaxed workgroup-scope atomics: plain ds_ accesses;
    int* dn;         // [NLS][NCW]  per stream and compute wave likewise   a volatile access here compiles to a FLAT load + vmcnt(0))
    int* local;      // [1] every reader of this cell's state rows runs on this workgroup's XCD (see DfArgs::role)
    float* bias;     // [3][DFF_JS] the slice's biases, gate-major (the thin-block path evaluates other units per lane than the
                     // MFMA path, whose lanes keep their three biases in registers)
};

template <int KPT> struct DfSlot {
    static constexpr int AP = DfPad<KPT>::row;
    static constexpr int a_off = 0;                       // [RB][AP]  operand rows (aggregates / lower-layer rows)
    static constexpr int gi_off = DF_RB * AP;             // [RB][96]  input-side pre-activations of the slice
    static constexpr int v_off = gi_off + DF_RB * 3 * DFF_JS;   // [RB] ints (16 B)
    static constexpr int words = v_off + 4;
};

__device__ __forceinline__ int df_flag_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void df_flag_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ bool df_wait4(const int* f, int target, int* err, unsigned limit) {
    unsigned spins = 0;
    for (;;) {
        int m = min(min(df_flag_ld(f), df_flag_ld(f + 1)), min(df_flag_ld(f + 2), df_flag_ld(f + 3)));   // (every compute wave's flag)
        if (DFF_NCW == 8) m = min(m, min(min(df_flag_ld(f + 4), df_flag_ld(f + 5)), min(df_flag_ld(f + 6), df_flag_ld(f + 7))));
        if (m >= target) return true;
        __builtin_amdgcn_s_sleep(DF_WSLEEP_N);
        if (++spins > 4 * limit) { __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        // once any wait of the launch has failed, nobody waits long again (the pass is lost; it must still end)
        if ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
}

// bounded global poll: false (and error bit 0) once the budget is spent
__device__ __forceinline__ bool df_retry(unsigned& spins, int* err, unsigned limit) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > limit) { __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    if ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    return true;
}

// ---- loader wave: row `lw` of every block of this group
// PLAIN: the cell's aggregator is not an attention soft-max but `add` / `max` over the messages h_j + e_j, e_j =
// edge_encoder(edge_attr_j) (AggConv, dagnn.py:232-251; C.agg picks the fold at run time; 3: no message lands on these rows -
// the reference's shared AggConv in the reverse direction - and the aggregate is zero without a poll)
template <int KPT, int KIND, int RR, bool EXTRA, bool PLAIN = false>
__device__ __forceinline__ void df_loader(const int32_t* __restrict__ plan, const DfArgs& S,
                                          const DfCell& C, int sl, int group, const DfLds& lds, int w, int set) {
    constexpr int H = 16 * KPT;
    constexpr int SEG = DfPad<KPT>::seg, KP8 = DfPad<KPT>::kp8;
    typedef DfSlot<KPT> Slot;
    const int lane = threadIdx.x & 63;
    const int d = C.dir;
    const int32_t* tab = S.sched + S.gtab[d] + 2 * group;
    const int rec_base = tab[0], nblk = tab[1];
    const int32_t* __restrict__ recs = S.sched + S.grec[d] + 16 * (int64_t)rec_base;   // records of this stream (16 words each)
    const int32_t* __restrict__ col = plan + S.col[d];
    const float* __restrict__ eattr = reinterpret_cast<const float*>(plan + S.eattr[d]);
    constexpr bool proj = KIND == DFK_PROJ;
    const int R = proj ? 0 : (RR >= 0 ? RR : (C.gain ? S.R : 0));
    // everything the loop needs from the argument structs, read ONCE: a field access inside the loop is a scalar load
    // from the kernel-argument segment plus an lgkmcnt(0) wait on the dependent chain
    const unsigned epoch = S.epoch, spin_limit = S.spin_limit;
    int* const err = S.err;
    const gran_t* const g_src = proj ? C.g_in : C.g_out;   // rows this cell reads: the lower layer's / its own states
    const int gld = S.gld, pld = S.pld;
    const pgran_t* const p_in = KIND == DFK_RECP ? C.p_in : nullptr;
    const float* const gi0 = KIND == DFK_REC0 ? C.gi0 : nullptr;
    const float* const sscore = EXTRA ? C.sscore : nullptr;
    const float* const vid = EXTRA ? C.vid : nullptr;
    const float* const gainp = proj ? nullptr : C.gain;
    const int vid_mod = EXTRA ? S.vid_mod : 1;
    const float gain0 = R >= 1 ? C.gain[0] : 0.f, gain1 = R >= 2 ? C.gain[1] : 0.f;
    unsigned long long* const dbg = (DF_PROF && S.dbg) ? S.dbg + 2 * gridDim.x : nullptr;
    const gran_t ready = (gran_t)epoch << 32;
    int* const dn = lds.dn + set * DFF_NCW;   // this stream's slots
    // a lane holds columns {lane, 64 + lane, 128 + lane, 192 + lane} of a row (the first NQ4 = H / 64 of them): load
    // instruction q of a sweep then covers 512 contiguous bytes of the row (granules [64q, 64q + 64)) - a quarter of
    // the cache lines a lane-owns-4-consecutive-granules sweep asks for
    constexpr int NQ4 = H / 64;
    float wk[4] = {0.f, 0.f, 0.f, 0.f};
    int cpos[4];   // LDS position of the lane's columns in an operand row
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 64 * q + lane;
        cpos[q] = c + (SEG - KP8) * (c / KP8);
        if (!proj && !PLAIN && !(EXTRA && C.sscore) && q < NQ4) wk[q] = C.wkey[c];
    }
    // plain aggregators: the lane's columns of the edge encoder (two edge features at most: the host checks)
    const int agg = PLAIN ? C.agg : 0;
    const int Rp = (PLAIN && C.agg_w) ? S.R : 0;
    float ew0[4] = {0.f, 0.f, 0.f, 0.f}, ew1[4] = {0.f, 0.f, 0.f, 0.f}, ebv[4] = {0.f, 0.f, 0.f, 0.f};
    if (PLAIN && C.agg_w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 64 * q + lane;
            if (q < NQ4) {
                if (Rp >= 1) ew0[q] = C.agg_w[(int64_t)c * Rp];
                if (Rp >= 2) ew1[q] = C.agg_w[(int64_t)c * Rp + 1];
                ebv[q] = C.agg_b ? C.agg_b[c] : 0.f;
            }
        }
    }
    const bool prof_wave = DF_PROF && dbg != nullptr && (int)blockIdx.x == S.dbg_wg && w == 0 && lane == 0;
    bool prof = prof_wave;

    // ---- memory traffic of this wave, by hand.  Three streams share the wave's in-order vmcnt counter: the granule
    // sweeps (on the dependent chain), the static row records and the gi0 slices (cold lines: HBM latency).  Left to
    // the compiler every wait inside this loop is a vmcnt(0), i.e. each sweep would also wait for the prefetches
    // issued next to it (measured: 1.3 us per block instead of 0.2).  So:
    //  * records and gi0 slices travel by LDS-DMA (global_load_lds: no destination register) into small LDS rings,
    //    DF_RD resp. DF_GD blocks ahead;
    //  * ONE asm statement per trip to memory: the sweep's register loads (wave-uniform row pointer in SGPRs + one
    //    per-lane byte offset + immediates), behind them the prefetch group P(b) = {record of block b + RD, gi0 slice
    //    of block b + GD}, and the counted wait - vmcnt(|P|) completes the sweep and everything older (P(b - 1)
    //    included) while P(b) stays in flight.  The loads and their wait sit in the SAME statement because a
    //    destination register the compiler can see while its load is in flight gets copied sooner or later (a
    //    loop-carried value, a phi behind a branch): the copy reads the stale register and the wait then protects the
    //    wrong one - measured twice, a memory fault and a silent 0.5 error.
    //    One statement per shape: NN rows x 4 loads (H < 256 repeats the last 512 bytes: same instruction count for
    //    every H), the projection slice or not, no / record / record + gi0 prefetch.
    struct Sweep { gran_t x[4][4]; pgran_t xp; };
    const unsigned lane8 = 8u * lane, lanepx16 = 16u * (lane & (DFF_JS - 1));
    constexpr bool has_gi0 = KIND == DFK_REC0;
#define DF_ROW_LD(e)                                                            \
    "global_load_dwordx2 %[x" #e "0], %[vo], %[b" #e "] offset:0 sc1\n\t"        \
    "global_load_dwordx2 %[x" #e "1], %[vo], %[b" #e "] offset:%[o1] sc1\n\t"    \
    "global_load_dwordx2 %[x" #e "2], %[vo], %[b" #e "] offset:%[o2] sc1\n\t"    \
    "global_load_dwordx2 %[x" #e "3], %[vo], %[b" #e "] offset:%[o3] sc1\n\t"
#define DF_ROWS_0 ""
#define DF_ROWS_1 DF_ROW_LD(0)
#define DF_ROWS_2 DF_ROWS_1 DF_ROW_LD(1)
#define DF_ROWS_3 DF_ROWS_2 DF_ROW_LD(2)
#define DF_ROWS_4 DF_ROWS_3 DF_ROW_LD(3)
#define DF_PROJ_0 ""
#define DF_PROJ_1 "global_load_dwordx4 %[p0], %[vp], %[c0] offset:0 sc1\n\t"
    // LDS-DMA: lane l of the first 16 (3 JS / 4) lanes moves 4 (16) bytes to M0 + 4 l (16 l); all lanes are active here
#define DF_DMA_0 "s_waitcnt vmcnt(0)"
#define DF_DMA_1                                                                                   \
    "s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %[rl]\n\t" \
    "s_nop 0\n\tglobal_load_lds_dword %[ra], off\n\t"                                              \
    "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]\n\ts_waitcnt vmcnt(1)"
#define DF_DMA_2                                                                                   \
    "s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %[rl]\n\t" \
    "s_nop 0\n\tglobal_load_lds_dword %[ra], off\n\t"                                              \
    "s_bfm_b64 exec, %[gx], 0\n\ts_mov_b32 m0, %[gl]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[ga], off\n\t" \
    "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]\n\ts_waitcnt vmcnt(2)"
#define DF_TRIP(n, p, d)                                                                                                   \
    asm volatile(DF_ROWS_##n DF_PROJ_##p DF_DMA_##d                                                                        \
                 : [x00] "=v"(W.x[0][0]), [x01] "=v"(W.x[0][1]), [x02] "=v"(W.x[0][2]), [x03] "=v"(W.x[0][3]),             \
                   [x10] "=v"(W.x[1][0]), [x11] "=v"(W.x[1][1]), [x12] "=v"(W.x[1][2]), [x13] "=v"(W.x[1][3]),             \
                   [x20] "=v"(W.x[2][0]), [x21] "=v"(W.x[2][1]), [x22] "=v"(W.x[2][2]), [x23] "=v"(W.x[2][3]),             \
                   [x30] "=v"(W.x[3][0]), [x31] "=v"(W.x[3][1]), [x32] "=v"(W.x[3][2]), [x33] "=v"(W.x[3][3]),             \
                   [p0] "=v"(W.xp), [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec), [ra] "+v"(ra), [ga] "+v"(ga)  \
                 : [vo] "v"(lane8), [vp] "v"(lanepx16), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [b3] "s"(b3),            \
                   [c0] "s"(c0p), [rl] "s"(rl), [gl] "s"(gl),    \
                   [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3), [gx] "n"(DF_GI_LANES)                                          \
                 : "memory")
#define DF_CASE(n, p, d) case (n) * 6 + (p) * 3 + (d): DF_TRIP(n, p, d); break;
#define DF_CASES(n) DF_CASE(n, 0, 0) DF_CASE(n, 0, 1) DF_CASE(n, 0, 2) DF_CASE(n, 1, 0) DF_CASE(n, 1, 1) DF_CASE(n, 1, 2)

    // (a wave serves DFF_RPW rows of every block, one after the other: `lw` and the ring addresses below follow the row)
    int lw = w * DFF_RPW;
    int* rec_ring;
    unsigned rec_ring_a, gi_ring_a;
    const int32_t* rec_w;
    auto set_row = [&](int row) {
        lw = row;
        rec_ring = lds.rec + (set * DF_RB + lw) * (8 * 16);
        rec_ring_a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)rec_ring);
        gi_ring_a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds.giring + (set * DF_GIRING * DF_RB + lw) * (3 * DFF_JS)));
        rec_w = recs + 16 * lw + (lane & 15);
    };
    set_row(lw);
    const int64_t wstride = 16 * DF_RB;   // words per block
    const int gi_lane_off = ((lane % DF_GI_LANES) / (DFF_JS / 4)) * H + sl * DFF_JS + 4 * (lane % (DFF_JS / 4));
    // addresses of the prefetch group: record of this wave's j-th block (past the end: the last block's again) -> ring
    // entry j & 7; gi0 slice of `node` -> gi ring entry blk % DF_GIRING, row lw
    auto rec_src = [&](int j) -> const void* { return rec_w + (int64_t)min(j, nblk - 1) * wstride; };
    auto rec_dst = [&](int j) -> unsigned { return rec_ring_a + (j & 7) * 64; };
    auto gi_src = [&](int node) -> const void* { return gi0 + (int64_t)max(node, 0) * 3 * H + gi_lane_off; };
    auto gi_dst = [&](int blk) -> unsigned { return gi_ring_a + (blk % DF_GIRING) * (DF_RB * 3 * DFF_JS * 4); };
    auto glds4 = [&](const void* gsrc, unsigned lds_dst) {   // lane l: 4 bytes from gsrc -> LDS lds_dst + 4 l
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    auto glds16 = [&](const void* gsrc, unsigned lds_dst) {   // lane l: 16 bytes from gsrc -> LDS lds_dst + 16 l
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    auto rec_dma = [&](int j) { if (lane < 16) glds4(rec_src(j), rec_dst(j)); };
    auto gi_dma = [&](int blk, int node) { if (lane < DF_GI_LANES) glds16(gi_src(node), gi_dst(blk)); };
    if (nblk > 0) {   // prologue: records of this wave's rows of blocks 0..RD-1, gi0 slices of blocks 0..GD-1
        for (int rr = 0; rr < DFF_RPW; ++rr) {
            set_row(w * DFF_RPW + rr);
#pragma unroll
            for (int j = 0; j < DF_RD; ++j) rec_dma(j);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (has_gi0) {
#pragma unroll
                for (int j = 0; j < DF_GD; ++j) gi_dma(j, __builtin_amdgcn_readfirstlane(rec_ring[j * 16]));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }

    Sweep A;
    for (int b = 0; b < nblk; ++b) {
      const int j = b;
      bool slot_free = b < DF_NSLOT;   // the ring slot has been handed back (checked once per block, before the first write)
#pragma unroll 1
      for (int rr = 0; rr < DFF_RPW; ++rr) {
        if (DFF_RPW > 1) { set_row(w * DFF_RPW + rr); prof = prof_wave && rr == 0; }
        const int cur = rec_ring[(j & 7) * 16 + (lane & 15)];
#define DF_W(i) __builtin_amdgcn_readlane(cur, i)
        const int4 r0 = make_int4(DF_W(0), DF_W(1), DF_W(2), DF_W(3));
        const int4 r1 = make_int4(DF_W(4), DF_W(5), DF_W(6), DF_W(7));
        const int4 r2 = make_int4(DF_W(8), DF_W(9), DF_W(10), DF_W(11));
        const int4 r3 = make_int4(DF_W(12), DF_W(13), DF_W(14), DF_W(15));
#undef DF_W
        const int v2 = __builtin_amdgcn_readfirstlane(rec_ring[((j + DF_GD) & 7) * 16]);   // node of this wave's block j + GD (landed long ago)
        const int slot = b % DF_NSLOT;
        float* sbase = lds.ring + (set * DF_NSLOT + slot) * Slot::words;
        int* v_s = reinterpret_cast<int*>(sbase + Slot::v_off);
        const int v = r0.x;
        if (prof) dbg[8 * (int64_t)(DFF_NLS * b + set) + 4] = wall_clock64();
        unsigned polls = 0;
        constexpr int O1 = (NQ4 > 1 ? 1 : 0) * 512, O2 = (NQ4 > 2 ? 2 : NQ4 - 1) * 512, O3 = (NQ4 - 1) * 512;
        // one trip to memory: the rows pj[0..nn), the projection slice if `pp`, the prefetch group P(b) if `dma`
        // (1: record, 2: record + gi0 slice); returns with every register it loaded valid
        auto trip = [&](Sweep& W, int nn, bool pp, int dma, const int (&pj)[4], const pgran_t* gp_in) {
            // wave-uniform row bases (unused slots: row 0, not loaded); N * gld granules fit 32 bits (host check): one
            // 32-bit multiply + a 64-bit add per row instead of the five-instruction 64-bit product
            const gran_t* b0 = g_src + (unsigned)pj[0] * (unsigned)gld;
            const gran_t* b1 = g_src + (unsigned)pj[1] * (unsigned)gld;
            const gran_t* b2 = g_src + (unsigned)pj[2] * (unsigned)gld;
            const gran_t* b3 = g_src + (unsigned)pj[3] * (unsigned)gld;
            const pgran_t* c0p = gp_in;
            const void* ra = rec_src(j + DF_RD);
            const unsigned rl = rec_dst(j + DF_RD);
            const void* ga = has_gi0 ? gi_src(v2) : ra;
            const unsigned gl = gi_dst(b + DF_GD);
            unsigned keep_m0;
            unsigned long long keep_exec;
            switch (nn * 6 + (pp ? 3 : 0) + dma) {
                DF_CASES(0) DF_CASES(1) DF_CASES(2) DF_CASES(3)
                default: DF_CASES(4)
            }
        };
        if (v >= 0) {
            const int eb = r0.y;
            const int deg = proj ? 1 : ((PLAIN && agg == 3) ? 0 : r0.z - r0.y);   // a projection reads ONE row: the node's own state one layer down
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            float m = -INFINITY, l = 0.f;
            // input-side pre-activations of the slice from the projection cell: 3 gates x 32 units, lanes 0..31
            bool p_pending = KIND == DFK_RECP;
            float pv[3] = {0.f, 0.f, 0.f};
            const pgran_t* gp_in = p_pending ? p_in + (unsigned)v * (unsigned)pld + sl * DFF_JS : reinterpret_cast<const pgran_t*>(g_src);   // wave-uniform (g_src: never loaded)
            // in-edges in chunks of <= 4 (ids and features of the first chunk came with the record).  A node with more
            // than 4 in-edges takes two chunks per trip to memory (all of them finished long ago: the trips, not the
            // data, are what such a row waits for)
            float ff0[4] = {0.f, 0.f, 0.f, 0.f}, ff1[4] = {0.f, 0.f, 0.f, 0.f};   // (PLAIN) the chunk's raw edge features
            auto chunk_ids = [&](int c0, int (&pj)[4], float (&fe)[4]) -> int {
                const int nn = max(0, min(4, deg - c0));
#pragma unroll
                for (int e = 0; e < 4; ++e) { pj[e] = 0; fe[e] = 0.f; }
                if (PLAIN && Rp > 0) {
                    if (c0 == 0) {
                        ff0[0] = __int_as_float(r2.x); ff0[1] = __int_as_float(r2.z); ff0[2] = __int_as_float(r3.x); ff0[3] = __int_as_float(r3.z);
                        if (Rp >= 2) { ff1[0] = __int_as_float(r2.y); ff1[1] = __int_as_float(r2.w); ff1[2] = __int_as_float(r3.y); ff1[3] = __int_as_float(r3.w); }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ff0[e] = ff1[e] = 0.f;
                            if (e < nn) {
                                ff0[e] = eattr[(int64_t)(eb + c0 + e) * Rp];
                                if (Rp >= 2) ff1[e] = eattr[(int64_t)(eb + c0 + e) * Rp + 1];
                            }
                        }
                    }
                }
                if (proj) {
                    pj[0] = v;
                } else if (c0 == 0) {
                    pj[0] = r1.x; pj[1] = r1.y; pj[2] = r1.z; pj[3] = r1.w;
                    if (R >= 1 && R <= 2) {
                        fe[0] = gain0 * __int_as_float(r2.x) + gain1 * __int_as_float(r2.y);
                        fe[1] = gain0 * __int_as_float(r2.z) + gain1 * __int_as_float(r2.w);
                        fe[2] = gain0 * __int_as_float(r3.x) + gain1 * __int_as_float(r3.y);
                        fe[3] = gain0 * __int_as_float(r3.z) + gain1 * __int_as_float(r3.w);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nn) pj[e] = col[eb + c0 + e];
                }
                if (R > 2 || (R > 0 && c0 > 0)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        fe[e] = 0.f;
                        if (e < nn) for (int r = 0; r < R; ++r) fe[e] = fmaf(gainp[r], eattr[(int64_t)(eb + c0 + e) * R + r], fe[e]);
                    }
                }
                return nn;
            };
            // every row of the trip carries this pass's tag (nn is wave-uniform)
            auto arrived = [&](const Sweep& W, int nn) -> bool {
                bool all = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < nn) {
                        bool ok = true;
#pragma unroll
                        for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(W.x[e][q] >> 32) == epoch;
                        all = all && __all(ok);
                    }
                }
                return all;
            };
            // online softmax over one chunk of NN in-edges (NN is wave-uniform: one straight-line variant per count, so
            // the independent reductions of a chunk interleave and an absent slot costs nothing - the loader waves
            // share their SIMD's issue slots with the compute wave, every instruction here is paid twice)
            auto fold_n = [&](auto nn_c, const int (&pj)[4], const float (&fe)[4], const Sweep& W) {
                constexpr int NN = decltype(nn_c)::value;
#define DF_ROW(e, q) __uint_as_float((unsigned)W.x[e][q])
                float s[NN];
#pragma unroll
                for (int e = 0; e < NN; ++e) s[e] = DF_ROW(e, 0) * wk[0] + DF_ROW(e, 1) * wk[1] + DF_ROW(e, 2) * wk[2] + DF_ROW(e, 3) * wk[3];
                if (!sscore) {
#pragma unroll
                    for (int e = 0; e < NN; ++e) s[e] = df_wave_sum(s[e]);
                }
                float mc = m;
#pragma unroll
                for (int e = 0; e < NN; ++e) {
                    float sv = sscore ? sscore[pj[e]] : s[e];
                    if (vid) sv += vid[pj[e] % vid_mod];
                    sv += fe[e];
                    s[e] = sv;
                    mc = fmaxf(mc, sv);
                }
                const float sc = __expf(m - mc);   // 0 on the first chunk (m = -inf)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] *= sc;
                l *= sc;
#pragma unroll
                for (int e = 0; e < NN; ++e) {
                    const float p = __expf(s[e] - mc);
                    l += p;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(p, DF_ROW(e, q), acc[q]);
                }
                m = mc;
#undef DF_ROW
            };
            auto fold = [&](const int (&pj)[4], const float (&fe)[4], int nn, const Sweep& W) {
                switch (nn) {
                    case 1: fold_n(std::integral_constant<int, 1>(), pj, fe, W); break;
                    case 2: fold_n(std::integral_constant<int, 2>(), pj, fe, W); break;
                    case 3: fold_n(std::integral_constant<int, 3>(), pj, fe, W); break;
                    default: fold_n(std::integral_constant<int, 4>(), pj, fe, W); break;
                }
            };
            int c0 = 0;
            do {
                int pj[4];
                float fe[4];
                const int nn = chunk_ids(c0, pj, fe);
                unsigned spins = 0;
                unsigned long long t_issue = 0ull;
                int dma = c0 == 0 ? (has_gi0 ? 2 : 1) : 0;   // P(b) rides behind the block's first sweep
                for (;;) {
                    if (prof) t_issue = wall_clock64();
                    trip(A, nn, p_pending, dma, pj, gp_in);
                    dma = 0;
                    const bool rows_ok = arrived(A, nn);
                    if (p_pending) {
                        if (__all(A.xp.x == epoch)) {
                            pv[0] = __uint_as_float(A.xp.y); pv[1] = __uint_as_float(A.xp.z); pv[2] = __uint_as_float(A.xp.w);
                            p_pending = false;
                        }
                    }
                    ++polls;
                    if ((rows_ok && !p_pending) || !df_retry(spins, err, spin_limit)) break;
                }
                if (prof && c0 == 0) {
                    dbg[8 * (int64_t)(DFF_NLS * b + set) + 5] = wall_clock64(); dbg[8 * (int64_t)(DFF_NLS * b + set) + 6] = polls;
                    if (DFF_NLS * b + set >= 8) dbg[8 * (int64_t)(DFF_NLS * b + set) + 7] = t_issue;   // when the poll that found the row was issued
                }
                if (PLAIN) {   // messages h_j + e_j, summed or maximised column by column (rows without a message stay zero)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e < nn) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float mv = __uint_as_float((unsigned)A.x[e][q]) + fmaf(ff0[e], ew0[q], fmaf(ff1[e], ew1[q], ebv[q]));
                                acc[q] = agg == 2 ? ((c0 == 0 && e == 0) ? mv : fmaxf(acc[q], mv)) : acc[q] + mv;
                            }
                        }
                    }
                } else if (deg == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __uint_as_float((unsigned)A.x[0][q]);
                    l = 1.f;
                } else if (nn > 0) {
                    fold(pj, fe, nn, A);
                }
                c0 += 4;
            } while (c0 < deg);
            if (prof) { asm volatile("" :: "v"(acc[0]), "v"(acc[3]), "v"(l)); dbg[(1 << 19) + 8 * (int64_t)(DFF_NLS * b + set) + 4] = wall_clock64(); }   // fold done
            if (deg > 1 && !PLAIN) {   // PyG softmax: exp(x - max) / (sum + 1e-16)
                const float inv = __builtin_amdgcn_rcpf(l + 1e-16f);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] *= inv;
            }
            if (!slot_free) { df_wait4(dn, b - DF_NSLOT + 1, err, spin_limit); slot_free = true; }   // the ring slot is free again
            float* a_row = sbase + Slot::a_off + lw * Slot::AP;
#pragma unroll
            for (int q = 0; q < NQ4; ++q) a_row[cpos[q]] = acc[q];
            if (KIND == DFK_RECP && lane < DFF_JS) {
#pragma unroll
                for (int g = 0; g < 3; ++g) sbase[Slot::gi_off + lw * (3 * DFF_JS) + g * DFF_JS + lane] = pv[g];
            }
        } else {
            const int none[4] = {0, 0, 0, 0};
            trip(A, 0, false, has_gi0 ? 2 : 1, none, reinterpret_cast<const pgran_t*>(g_src));   // an idle row keeps the cadence: P(b) out, P(b - 1) landed
            if (!slot_free) { df_wait4(dn, b - DF_NSLOT + 1, err, spin_limit); slot_free = true; }
        }
        if (lane == 0) v_s[lw] = v;
      }
        if (prof_wave) dbg[(1 << 19) + 8 * (int64_t)(DFF_NLS * b + set) + 5] = wall_clock64();   // LDS writes issued
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) df_flag_st(lds.rdy + set * DFF_WPS + w, b + 1);
        if (prof_wave) dbg[8 * (int64_t)(DFF_NLS * b + set) + 3] = wall_clock64();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of ours is in flight when the wave ends
#undef DF_TRIP
#undef DF_CASE
#undef DF_CASES
}

// ---- the loader of the configurations that are benchmarked (two edge features or a projection cell, no static scores, no
// vertex-id key biases, one row per loader wave), rebuilt around what a wave's time is made of (round 4,
// scripts/ubench/branch_cost.hip: ONE wave issues a dependent VALU operation every ~9 cycles, an independent one every ~5,
// a taken branch costs ~22, a not-taken one ~8 - a hop of the dependent chain is ~8000 cycles of exactly that).  Same
// protocol, same arithmetic as df_loader above (the rows it writes are bitwise the same); what differs is the shape of the code:
//   * the in-degree of a row picks ONE straight-line body (0 / 1 / 2 / 3 / 4 predecessors: >= 97 % of the rows of an AST
//     batch) - a trip is one statically shaped asm statement, its re-poll loop is [trip, minimum of the tags, compare,
//     branch], the soft-max of a single chunk needs no running rescale, nothing is predicated per lane;
//   * everything that does not depend on the polled rows happens BEFORE the poll (ring-slot wait, addresses, edge gains);
//   * rows with more than 4 in-edges take the general chunk loop (online soft-max), as before.
struct DfSweep { gran_t x[4][5]; pgran_t xp; };   // (the fifth column block: H = 320 only)

#define DF_TRIP(n, p, d)                                                                                                   \
    asm volatile(DF_ROWS_##n DF_PROJ_##p DF_DMA_##d                                                                        \
                 : [x00] "=v"(W.x[0][0]), [x01] "=v"(W.x[0][1]), [x02] "=v"(W.x[0][2]), [x03] "=v"(W.x[0][3]),             \
                   [x10] "=v"(W.x[1][0]), [x11] "=v"(W.x[1][1]), [x12] "=v"(W.x[1][2]), [x13] "=v"(W.x[1][3]),             \
                   [x20] "=v"(W.x[2][0]), [x21] "=v"(W.x[2][1]), [x22] "=v"(W.x[2][2]), [x23] "=v"(W.x[2][3]),             \
                   [x30] "=v"(W.x[3][0]), [x31] "=v"(W.x[3][1]), [x32] "=v"(W.x[3][2]), [x33] "=v"(W.x[3][3]),             \
                   [p0] "=v"(W.xp), [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec), [ra] "+v"(ra), [ga] "+v"(ga)  \
                 : [vo] "v"(lane8), [vp] "v"(lanepx16), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [b3] "s"(b3),            \
                   [c0] "s"(c0p), [rl] "s"(rl), [gl] "s"(gl),    \
                   [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3), [gx] "n"(DF_GI_LANES)                                          \
                 : "memory")
// H = 320: five column blocks per lane
#define DF_ROW_LD5(e) DF_ROW_LD(e) "global_load_dwordx2 %[x" #e "4], %[vo], %[b" #e "] offset:%[o4] sc1\n\t"
#define DF_ROWS5_0 ""
#define DF_ROWS5_1 DF_ROW_LD5(0)
#define DF_ROWS5_2 DF_ROWS5_1 DF_ROW_LD5(1)
#define DF_ROWS5_3 DF_ROWS5_2 DF_ROW_LD5(2)
#define DF_ROWS5_4 DF_ROWS5_3 DF_ROW_LD5(3)
#define DF_TRIP5(n, p, d)                                                                                                  \
    asm volatile(DF_ROWS5_##n DF_PROJ_##p DF_DMA_##d                                                                       \
                 : [x00] "=v"(W.x[0][0]), [x01] "=v"(W.x[0][1]), [x02] "=v"(W.x[0][2]), [x03] "=v"(W.x[0][3]), [x04] "=v"(W.x[0][4]), \
                   [x10] "=v"(W.x[1][0]), [x11] "=v"(W.x[1][1]), [x12] "=v"(W.x[1][2]), [x13] "=v"(W.x[1][3]), [x14] "=v"(W.x[1][4]), \
                   [x20] "=v"(W.x[2][0]), [x21] "=v"(W.x[2][1]), [x22] "=v"(W.x[2][2]), [x23] "=v"(W.x[2][3]), [x24] "=v"(W.x[2][4]), \
                   [x30] "=v"(W.x[3][0]), [x31] "=v"(W.x[3][1]), [x32] "=v"(W.x[3][2]), [x33] "=v"(W.x[3][3]), [x34] "=v"(W.x[3][4]), \
                   [p0] "=v"(W.xp), [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec), [ra] "+v"(ra), [ga] "+v"(ga)  \
                 : [vo] "v"(lane8), [vp] "v"(lanepx16), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [b3] "s"(b3),            \
                   [c0] "s"(c0p), [rl] "s"(rl), [gl] "s"(gl),    \
                   [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3), [o4] "n"(2048), [gx] "n"(DF_GI_LANES)                          \
                 : "memory")
#define DF_IFC(n, p, d) if constexpr (NN == (n) && PP == (p) && DMA == (d)) { if constexpr (NQ4 == 5) { DF_TRIP5(n, p, d); } else { DF_TRIP(n, p, d); } } else
#define DF_IFCS(n) DF_IFC(n, 0, 0) DF_IFC(n, 0, 1) DF_IFC(n, 0, 2) DF_IFC(n, 1, 0) DF_IFC(n, 1, 1) DF_IFC(n, 1, 2)

struct DfTripArgs {
    const gran_t* b[4];        // wave-uniform row bases
    const pgran_t* c;          // projection slice of the node (its DFF_JS projection granules)
    const void* ra; unsigned rl;   // prefetch group: record (global source per lane, LDS destination)
    const void* ga; unsigned gl;   //                 gi0 slice
};

// one trip to memory: NN rows (+ the projection slice) into W, the prefetch group behind them, the counted wait - ONE asm
// statement of a static shape (see df_loader)
template <int NQ4, int H, int NN, int PP, int DMA>
__device__ __forceinline__ void df_trip(DfSweep& W, const DfTripArgs& T, unsigned lane8, unsigned lanepx16) {
    constexpr int O1 = (NQ4 > 1 ? 1 : 0) * 512, O2 = (NQ4 > 2 ? 2 : NQ4 - 1) * 512, O3 = (NQ4 > 3 ? 3 : NQ4 - 1) * 512;
    const gran_t* b0 = T.b[0]; const gran_t* b1 = T.b[1]; const gran_t* b2 = T.b[2]; const gran_t* b3 = T.b[3];
    const pgran_t* c0p = T.c;
    const void* ra = T.ra; const unsigned rl = T.rl; const void* ga = T.ga; const unsigned gl = T.gl;
    unsigned keep_m0;
    unsigned long long keep_exec;
    DF_IFCS(0) DF_IFCS(1) DF_IFCS(2) DF_IFCS(3) DF_IFCS(4) {}
}

// every granule of the trip carries this pass's tag.  Tags never exceed the current epoch (the arena hands out strictly
// increasing ones and starts over on zeroed buffers), so "all equal" is "the minimum equals": one v_min3 per two tags
template <int NN, int PP, int NQ4 = 4>
__device__ __forceinline__ bool df_landed(const DfSweep& W, unsigned epoch) {
    unsigned m = epoch;
#pragma unroll
    for (int e = 0; e < NN; ++e) {
        m = min(min(m, min((unsigned)(W.x[e][0] >> 32), (unsigned)(W.x[e][1] >> 32))), min((unsigned)(W.x[e][2] >> 32), (unsigned)(W.x[e][3] >> 32)));
        if (NQ4 == 5) m = min(m, (unsigned)(W.x[e][4] >> 32));
    }
    if (PP) m = min(m, W.xp.x);
    return __builtin_amdgcn_uicmp(m, epoch, 33 /* ICMP_NE */) == 0ull;
}

template <int KPT, int KIND>
__device__ __forceinline__ void df_loader_fast(const int32_t* __restrict__ plan, const DfArgs& S,
                                               const DfCell& C, int sl, int group, const DfLds& lds, int w, int set) {
    constexpr int H = 16 * KPT;
    constexpr int SEG = DfPad<KPT>::seg, KP8 = DfPad<KPT>::kp8;
    typedef DfSlot<KPT> Slot;
    constexpr bool proj = KIND == DFK_PROJ, has_gi0 = KIND == DFK_REC0;
    constexpr int PPK = KIND == DFK_RECP ? 1 : 0;      // the node's projection slice rides in every trip
    constexpr int DMA1 = has_gi0 ? 2 : 1;              // prefetch group of a block's first trip
    constexpr int NQ4 = H / 64;
    const int lane = threadIdx.x & 63;
    const int d = C.dir;
    const int32_t* tab = S.sched + S.gtab[d] + 2 * group;
    const int rec_base = tab[0], nblk = tab[1];
    const int32_t* __restrict__ recs = S.sched + S.grec[d] + 16 * (int64_t)rec_base;
    const int32_t* __restrict__ col = plan + S.col[d];
    const float* __restrict__ eattr = reinterpret_cast<const float*>(plan + S.eattr[d]);
    const unsigned epoch = S.epoch, spin_limit = S.spin_limit;
    int* const err = S.err;
    const gran_t* const g_src = proj ? C.g_in : C.g_out;
    const unsigned gld = (unsigned)S.gld, pld = (unsigned)S.pld;
    const pgran_t* const p_in = KIND == DFK_RECP ? C.p_in : nullptr;
    const float* const gi0 = has_gi0 ? C.gi0 : nullptr;
    const float gain0 = proj ? 0.f : C.gain[0], gain1 = proj ? 0.f : C.gain[1];
    int* const dn = lds.dn + set * DFF_NCW;
    constexpr int NC = NQ4 > 4 ? NQ4 : 4;   // column blocks a lane carries (H < 256 repeats the last one: same trip shape for every H <= 256)
    float wk[NC];
    int cpos[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = 64 * q + lane;
        cpos[q] = c + (SEG - KP8) * (c / KP8);
        wk[q] = (!proj && q < NQ4) ? C.wkey[c] : 0.f;
    }
    const unsigned lane8 = 8u * lane, lanepx16 = 16u * (lane & (DFF_JS - 1));
    // (a wave serves DFF_RPW rows of every block, one after the other: `lw` and the ring addresses follow the row)
    int lw = w * DFF_RPW;
    int* rec_ring;
    unsigned rec_ring_a, gi_ring_a;
    const int32_t* rec_w;
    auto set_row = [&](int row) {
        lw = row;
        rec_ring = lds.rec + (set * DF_RB + lw) * (8 * 16);
        rec_ring_a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)rec_ring);
        gi_ring_a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds.giring + (set * DF_GIRING * DF_RB + lw) * (3 * DFF_JS)));
        rec_w = recs + 16 * lw + (lane & 15);
    };
    set_row(lw);
    const int64_t wstride = 16 * DF_RB;
    const int gi_lane_off = ((lane % DF_GI_LANES) / (DFF_JS / 4)) * H + sl * DFF_JS + 4 * (lane % (DFF_JS / 4));
    auto rec_src = [&](int j) -> const void* { return rec_w + (int64_t)min(j, nblk - 1) * wstride; };
    auto rec_dst = [&](int j) -> unsigned { return rec_ring_a + (j & 7) * 64; };
    auto gi_src = [&](int node) -> const void* { return gi0 + (int64_t)max(node, 0) * 3 * H + gi_lane_off; };
    auto gi_dst = [&](int blk) -> unsigned { return gi_ring_a + (blk % DF_GIRING) * (DF_RB * 3 * DFF_JS * 4); };
    if (nblk > 0) {   // prologue: records of blocks 0..RD-1, gi0 slices of blocks 0..GD-1
        auto glds4 = [&](const void* gsrc, unsigned lds_dst) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        };
        auto glds16 = [&](const void* gsrc, unsigned lds_dst) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        };
        for (int rr = 0; rr < DFF_RPW; ++rr) {
            set_row(w * DFF_RPW + rr);
#pragma unroll
            for (int j = 0; j < DF_RD; ++j) if (lane < 16) glds4(rec_src(j), rec_dst(j));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (has_gi0) {
#pragma unroll
                for (int j = 0; j < DF_GD; ++j) {
                    const int node = __builtin_amdgcn_readfirstlane(rec_ring[j * 16]);
                    if (lane < DF_GI_LANES) glds16(gi_src(node), gi_dst(j));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }

    unsigned long long* const dbg = (DF_PROF && S.dbg) ? S.dbg + 2 * gridDim.x : nullptr;
    const bool prof = DF_PROF && dbg != nullptr && (int)blockIdx.x == S.dbg_wg && w == 0 && lane == 0;
    unsigned long long t_issue = 0ull;
    unsigned polls = 0;
    DfSweep A;
    for (int b = 0; b < nblk; ++b) {
        // the ring slot has been handed back (off the dependent chain: the loaders run ahead of the compute waves only
        // where rows are already waiting)
        if (b >= DF_NSLOT) df_wait4(dn, b - DF_NSLOT + 1, err, spin_limit);
#pragma unroll 1
      for (int rr = 0; rr < DFF_RPW; ++rr) {
        if (DFF_RPW > 1) set_row(w * DFF_RPW + rr);
        if (prof) { dbg[8 * (int64_t)(DFF_NLS * b + set) + 4] = wall_clock64(); polls = 0; }
        const int cur = rec_ring[(b & 7) * 16 + (lane & 15)];
        const int v2 = has_gi0 ? __builtin_amdgcn_readfirstlane(rec_ring[((b + DF_GD) & 7) * 16]) : 0;   // node of block b + GD (landed long ago)
#define DF_W(i) __builtin_amdgcn_readlane(cur, i)
        const int v = DF_W(0);
        float* const sbase = lds.ring + (set * DF_NSLOT + b % DF_NSLOT) * Slot::words;
        DfTripArgs T;
        T.ra = rec_src(b + DF_RD); T.rl = rec_dst(b + DF_RD);
        T.ga = has_gi0 ? gi_src(v2) : T.ra; T.gl = gi_dst(b + DF_GD);
        T.b[0] = T.b[1] = T.b[2] = T.b[3] = g_src; T.c = reinterpret_cast<const pgran_t*>(g_src);
        float acc[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[q] = 0.f;
#ifdef DF_EXP_NOLOAD   // timing experiment: the loaders only keep the records and the flags moving (no polls, no fold)
        if (false) {
#else
        if (v >= 0) {
#endif
            const int eb = DF_W(1);
            const int deg = proj ? 1 : DF_W(2) - eb;
            if (PPK) T.c = p_in + (unsigned)v * pld + sl * DFF_JS;
            // poll until every granule of the trip carries this pass's tag: first trip with the prefetch group behind it
            auto poll = [&](auto nn_c, auto dma_c) {
                constexpr int NN = decltype(nn_c)::value, DM = decltype(dma_c)::value;
                if (prof) { t_issue = wall_clock64(); ++polls; }
                df_trip<NQ4, H, NN, PPK, DM>(A, T, lane8, lanepx16);
                if (NN + PPK > 0 && !df_landed<NN, PPK, NQ4>(A, epoch)) {
                    unsigned spins = 0;
                    // (a lost pass leaves the loop BEHIND its trip: with a way out in front of it the rows of the previous trip
                    // stay live across the statement and the register allocator copies the whole sweep on every turn)
                    bool more;
                    do {
                        more = df_retry(spins, err, spin_limit);
                        if (prof) { t_issue = wall_clock64(); ++polls; }
                        df_trip<NQ4, H, NN, PPK, 0>(A, T, lane8, lanepx16);
                    } while (more && !df_landed<NN, PPK, NQ4>(A, epoch));
                }
                if (prof && DM != 0) {
                    dbg[8 * (int64_t)(DFF_NLS * b + set) + 5] = wall_clock64(); dbg[8 * (int64_t)(DFF_NLS * b + set) + 6] = polls;
                    if (DFF_NLS * b + set >= 8) dbg[8 * (int64_t)(DFF_NLS * b + set) + 7] = t_issue;
                }
            };
#define DF_ROWF(e, q) __uint_as_float((unsigned)A.x[e][q])
            // single chunk: s_e = w_key . h_e + gain . feat_e, alpha = exp(s - max) / (sum + 1e-16) (PyG), a = sum alpha_e h_e
            auto row_n = [&](auto nn_c) {
                constexpr int NN = decltype(nn_c)::value;
                static_assert(NN >= 2 && NN <= 4, "");
                T.b[0] = g_src + (unsigned)DF_W(4) * gld;
                T.b[1] = g_src + (unsigned)DF_W(5) * gld;
                if (NN > 2) T.b[2] = g_src + (unsigned)DF_W(6) * gld;
                if (NN > 3) T.b[3] = g_src + (unsigned)DF_W(7) * gld;
                float fe[NN];
#pragma unroll
                for (int e = 0; e < NN; ++e)
                    fe[e] = gain0 * __int_as_float(DF_W(8 + 2 * e)) + gain1 * __int_as_float(DF_W(9 + 2 * e));
                poll(nn_c, std::integral_constant<int, DMA1>());
#ifdef DF_EXP_NOFOLD   // timing experiment: trips and polls, no soft-max / aggregate arithmetic
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[q] = DF_ROWF(0, q) + DF_ROWF(NN - 1, q);
                return;
#endif
                float s[NN];
#pragma unroll
                for (int e = 0; e < NN; ++e) { s[e] = DF_ROWF(e, 0) * wk[0] + DF_ROWF(e, 1) * wk[1] + DF_ROWF(e, 2) * wk[2] + DF_ROWF(e, 3) * wk[3]; if (NC > 4) s[e] = fmaf(DF_ROWF(e, 4), wk[NC - 1], s[e]); }
#pragma unroll
                for (int e = 0; e < NN; ++e) s[e] = df_wave_sum(s[e]);
                float mc = -INFINITY;
#pragma unroll
                for (int e = 0; e < NN; ++e) { s[e] += fe[e]; mc = fmaxf(mc, s[e]); }
                float l = 0.f;
#pragma unroll
                for (int e = 0; e < NN; ++e) {
                    const float pe = __expf(s[e] - mc);
                    l += pe;
#pragma unroll
                    for (int q = 0; q < NC; ++q) acc[q] = fmaf(pe, DF_ROWF(e, q), acc[q]);
                }
                const float inv = __builtin_amdgcn_rcpf(l + 1e-16f);
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[q] *= inv;
            };
            if (deg == 1) {          // the aggregate IS the predecessor's row (alpha = 1 / (1 + 1e-16) = 1)
                T.b[0] = g_src + (unsigned)(proj ? v : DF_W(4)) * gld;
                poll(std::integral_constant<int, 1>(), std::integral_constant<int, DMA1>());
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[q] = DF_ROWF(0, q);
            } else if (deg == 2) {
                row_n(std::integral_constant<int, 2>());
            } else if (deg == 3) {
                row_n(std::integral_constant<int, 3>());
            } else if (deg == 4) {
                row_n(std::integral_constant<int, 4>());
            } else if (deg <= 0) {   // a node without predecessors: zero aggregate (its projection slice still has to land)
                poll(std::integral_constant<int, 0>(), std::integral_constant<int, DMA1>());
            } else {                 // more than 4 in-edges: chunks of <= 4 under an online soft-max (ids / features of the
                                     // chunks behind the first through the plan's CSR)
                float m = -INFINITY, l = 0.f;
                for (int c0 = 0; c0 < deg; c0 += 4) {
                    const int nn = min(4, deg - c0);
                    int pj[4] = {0, 0, 0, 0};
                    float fe[4] = {0.f, 0.f, 0.f, 0.f};
                    if (c0 == 0) {
                        pj[0] = DF_W(4); pj[1] = DF_W(5); pj[2] = DF_W(6); pj[3] = DF_W(7);
#pragma unroll
                        for (int e = 0; e < 4; ++e) fe[e] = gain0 * __int_as_float(DF_W(8 + 2 * e)) + gain1 * __int_as_float(DF_W(9 + 2 * e));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (e < nn) {
                                pj[e] = col[eb + c0 + e];
                                fe[e] = fmaf(gain1, eattr[(int64_t)(eb + c0 + e) * 2 + 1], fmaf(gain0, eattr[(int64_t)(eb + c0 + e) * 2], 0.f));
                            }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) T.b[e] = g_src + (unsigned)pj[e] * gld;
                    if (c0 == 0) poll(std::integral_constant<int, 4>(), std::integral_constant<int, DMA1>());
                    else if (nn == 4) poll(std::integral_constant<int, 4>(), std::integral_constant<int, 0>());
                    else if (nn == 3) poll(std::integral_constant<int, 3>(), std::integral_constant<int, 0>());
                    else if (nn == 2) poll(std::integral_constant<int, 2>(), std::integral_constant<int, 0>());
                    else poll(std::integral_constant<int, 1>(), std::integral_constant<int, 0>());
                    float s[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s[e] = DF_ROWF(e, 0) * wk[0] + DF_ROWF(e, 1) * wk[1] + DF_ROWF(e, 2) * wk[2] + DF_ROWF(e, 3) * wk[3]; if (NC > 4) s[e] = fmaf(DF_ROWF(e, 4), wk[NC - 1], s[e]); }
                    float mc = m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[e] = df_wave_sum(s[e]) + fe[e];
                        if (e < nn) mc = fmaxf(mc, s[e]);
                    }
                    const float sc = __expf(m - mc);   // 0 on the first chunk (m = -inf)
#pragma unroll
                    for (int q = 0; q < NC; ++q) acc[q] *= sc;
                    l *= sc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e < nn) {
                            const float pe = __expf(s[e] - mc);
                            l += pe;
#pragma unroll
                            for (int q = 0; q < NC; ++q) acc[q] = fmaf(pe, DF_ROWF(e, q), acc[q]);
                        }
                    }
                    m = mc;
                }
                const float inv = __builtin_amdgcn_rcpf(l + 1e-16f);
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[q] *= inv;
            }
#undef DF_ROWF
            if (prof) { asm volatile("" :: "v"(acc[0]), "v"(acc[NC - 1])); dbg[(1 << 19) + 8 * (int64_t)(DFF_NLS * b + set) + 4] = wall_clock64(); }   // fold done
            float* a_row = sbase + Slot::a_off + lw * Slot::AP;
#pragma unroll
            for (int q = 0; q < NQ4; ++q) a_row[cpos[q]] = acc[q];
            if (PPK) {   // input-side pre-activations of the slice: lanes l and l + 32 loaded the same granules (same words, same place)
#pragma unroll
                for (int g = 0; g < 3; ++g) sbase[Slot::gi_off + lw * (3 * DFF_JS) + g * DFF_JS + (lane & (DFF_JS - 1))] = __uint_as_float(A.xp[1 + g]);
            }
        } else {
            df_trip<NQ4, H, 0, 0, DMA1>(A, T, lane8, lanepx16);   // an idle row keeps the cadence: P(b) out, P(b - 1) landed
        }
#undef DF_W
        reinterpret_cast<int*>(sbase + Slot::v_off)[lw] = v;   // (every lane: same word, same value - no lane-0 predicate)
      }
        if (prof) dbg[(1 << 19) + 8 * (int64_t)(DFF_NLS * b + set) + 5] = wall_clock64();   // LDS writes issued
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        df_flag_st(lds.rdy + set * DFF_WPS + w, b + 1);
        if (prof) dbg[8 * (int64_t)(DFF_NLS * b + set) + 3] = wall_clock64();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of ours is in flight when the wave ends
}

#undef DF_TRIP
#undef DF_TRIP5
#undef DF_IFC
#undef DF_IFCS

typedef float f4v __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float df_dpp(float v) {   // 0 where the source lane is outside the DPP row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// x + (x of the lane 16 away, i.e. the neighbouring DPP row of the pair): v_permlane16_swap on two copies of x leaves
// rows (0, 0, 2, 2) in one and (1, 1, 3, 3) in the other
__device__ __forceinline__ float df_row_pair_sum(float x) {
    // (inline asm: with both operands holding the same value hipcc 7.2 folds the builtin's two results into one and
    // emits v1 + v1 behind the swap; volatile also keeps it out of the divergent branch that uses the sum)
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// element index of (row v, column c) in a buffer of `ld` elements per row: ONE full-rate v_mad_u32_u24 instead of the 64-bit
// multiply (quarter rate, on the compute wave's serial chain) - v < 2^24 and v * ld + c < 2^31 (the host checks N * ld < 2^31)
__device__ __forceinline__ unsigned df_idx(int v, int ld, int c) { return __umul24((unsigned)v, (unsigned)ld) + (unsigned)c; }

// ---- compute wave `cw`: hidden units [8 cw, 8 cw + 8) of the slice, every block of the workgroup's streams.
// The products run on the matrix cores as v_mfma_f32_4x4x1 (16 independent 4 x 4 x 1 outer products per instruction:
// the only MFMA shape a 4-row block fills).  Lane = (unit quad = lane >> 5, K slice ks = (lane >> 2) & 7, x = lane & 3):
//   A operand  W[unit 4 quad + x][k]   - this lane's 3 x H/8 resident weights (k in [ks H/8, (ks + 1) H/8))
//   B operand  a[row x][k]             - the block's operand rows from LDS: H/8 floats per lane and block (the whole
//                                        4 x H block is read ONCE per wave; the FMA layout read it 8 times)
//   D[i][x] (register i) += W[unit 4 quad + i][k] a[row x][k]   for the lane's K slice,
// 3 H/8 instructions per block whatever the number of live rows.  The 8 K slices are then summed by a reduce-scatter
// over the lanes (ks bit 0: DPP row_shl/shr 4 keeps units {0,1} / {2,3}; bit 1: row_shl/shr 8 keeps one unit; bit 2:
// v_permlane16_swap adds the neighbouring row): lane (quad, ks, x) ends with all three gate sums of unit
// 4 quad + 2 (ks & 1) + ((ks >> 1) & 1) for row x, evaluates the gates and stores h' itself - no LDS exchange, no
// barrier.  Always the same order of additions -> deterministic.
template <int KPT, int KIND>
__device__ __forceinline__ void df_compute(const DfArgs& S, const DfCell& C, int sl, int pair, const DfLds& lds, int cw) {
    constexpr int H = 16 * KPT;
    constexpr int SEG = DfPad<KPT>::seg, KP8 = DfPad<KPT>::kp8, NK4 = KP8 / 4;
    typedef DfSlot<KPT> Slot;
    const int tc = threadIdx.x & 255;   // position among the compute waves
    const int lane = tc & 63;
    const int quad = lane >> 5, ks = (lane >> 2) & 7, x = lane & 3;
    const bool s0 = (ks & 1) != 0, s1 = (ks & 2) != 0;
    constexpr bool proj = KIND == DFK_PROJ;
    constexpr bool has_gi = !proj;
    constexpr bool gi_ring = KIND == DFK_REC0;
    const int d = C.dir;
    // the two streams of this workgroup: groups NLS * pair and NLS * pair + 1 (the second may not exist)
    int nb[DFF_NLS];   // blocks of the workgroup's streams (0: no such group)
#pragma unroll
    for (int q = 0; q < DFF_NLS; ++q) {
        const int grp = df_stream_group(pair, q, S.groups);
        nb[q] = grp >= 0 ? S.sched[S.gtab[d] + 2 * grp + 1] : 0;
    }
    float wr[KP8], wz[KP8], wn[KP8];   // the lane's K slice of the r / z / n rows of its unit
    {
        // (the packed matrix is in 32-unit slices of 256 lanes: a 64-unit workgroup's compute waves 4..7 take the odd one)
        const float4* wp = C.w + (int64_t)(sl * (DFF_JS / 32) + (cw >> 2)) * (3 * NK4) * 256 + tc;
#pragma unroll
        for (int q = 0; q < NK4; ++q) {
            const float4 x0 = wp[(0 * NK4 + q) * 256], x1 = wp[(1 * NK4 + q) * 256], x2 = wp[(2 * NK4 + q) * 256];
            wr[4 * q] = x0.x; wr[4 * q + 1] = x0.y; wr[4 * q + 2] = x0.z; wr[4 * q + 3] = x0.w;
            wz[4 * q] = x1.x; wz[4 * q + 1] = x1.y; wz[4 * q + 2] = x1.z; wz[4 * q + 3] = x1.w;
            wn[4 * q] = x2.x; wn[4 * q + 1] = x2.y; wn[4 * q + 2] = x2.z; wn[4 * q + 3] = x2.w;
        }
        // the weights have landed before the block loop starts: otherwise the first use inside the loop carries a
        // vmcnt(0), which - every block - also waits for the acknowledgement of the previous block's state stores
#pragma unroll
        for (int k = 0; k < KP8; ++k) asm volatile("" : "+v"(wr[k]), "+v"(wz[k]), "+v"(wn[k]));
    }
    // after the reduction: lane (quad, ks, x) holds unit 4 quad + x of the wave's eight and ROW 2 (ks & 1) + ((ks >> 1) & 1) of the block
    // (the activations are the MFMA's A operand, the weights its B operand: D register i = row i, lane x = unit x - the four lanes
    // of a quad store four consecutive units of one row, 32 contiguous bytes of granules; with the operands the other way round
    // every lane of a store went to another row: 32 eight-byte transactions per instruction, a quarter of the wave's time)
    const int unit_l = 8 * cw + 4 * quad + x, unit = sl * DFF_JS + unit_l;
    const int row_l = 2 * (ks & 1) + ((ks >> 1) & 1);
    float b_r = C.bias[unit], b_z = C.bias[H + unit], b_n = C.bias[2 * H + unit];
    asm volatile("" : "+v"(b_r), "+v"(b_z), "+v"(b_n));   // landed before the loop (see the weights above)
    const int apos = unit + (SEG - KP8) * (unit / KP8);   // LDS position of column `unit` of an operand row
    // byte position of column `unit` inside a row of the reverse sweep's static record (df_common.h)
    const unsigned stat_col = unit >= 256 ? 4u * (DF_NSTAT * DF_STAT_SP + (unit - 256)) : 4u * (4 * (unit & 63) + (unit >> 6));
    const unsigned epoch = S.epoch, spin_limit = S.spin_limit;
    int* const err = S.err;
    float* const h_out = C.h_out;
    float* const aux_out = C.aux_out;
    gran_t* const g_out = C.g_out;
    const bool local_st = !proj && lds.local[0] != 0;
    const int ld_h = S.ld_h, gld = S.gld, pld = S.pld, num_nodes = S.N;
    unsigned long long* const dbg = (DF_PROF && S.dbg) ? S.dbg + 2 * gridDim.x : nullptr;
    const bool prof = DF_PROF && dbg != nullptr && (int)blockIdx.x == S.dbg_wg && cw == 0 && lane == 0;

    // Blocks of the two streams in whatever order they become ready.  A stream inside a thin dependent chain is ready
    // once per hop (~3 us, of which this wave works ~0.8): the other stream's blocks fill the gap.  When both have a
    // block, the one whose loader is LESS far ahead goes first (it is the latency-bound one); ties alternate.
    //
    // Round 6: what a block costs this wave was measured with subtractive builds on a free-running compute wave (-DDF_EXP_*,
    // B = 1024): 0.89 us = products 0.40 + reduction and gates 0.11 + the look at the flags 0.11 + ~0.25 of bookkeeping, and
    // NOT latencies - moving 40 vector instructions into the products' shadow, or the stores' lanes onto contiguous bytes,
    // changed nothing: ONE wave issues an instruction every ~8 cycles whatever it is, so the block costs its instruction
    // count.  Hence the shape of this loop: all bookkeeping scalar and branch-free (m0 / m1 = the ready counts as last seen;
    // a loader never passes its stream's block count, so m - done is the lead without further conditions), the next look at
    // the flags as two ds_read_b128 issued behind the last product (its verdict is there when the block ends), one LDS word
    // per lane for the row's node id (no select chain, no live-row count), the done flag stored without a predicate (the
    // other lanes write into a dump area), the gates evaluated by every lane, the store variants hoisted out of the loop.
    typedef int i4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const volatile i4v* lds_i4p;
    const lds_i4p rdy_p = (lds_i4p)(uintptr_t)(unsigned)(uintptr_t)lds.rdy;   // (an LDS address is the low 32 bits of the generic pointer)
    const bool lane_st = (lane & 16) == 0;   // (lanes 16 away hold the same sums)
    // done flag: lane 0 writes dn[st][cw], the others a word of their own in the dump area
    int* const dn_or_dump = lane == 0 ? lds.dn + cw : reinterpret_cast<int*>(lds.bias) + lane;
    const int dn_step = lane == 0 ? DFF_NCW : 0;   // (words between the two streams' flags)
    const int nb0 = nb[0], nb1 = DFF_NLS > 1 ? nb[DFF_NLS - 1] : 0;
    static_assert(DFF_NLS == 2 || (DFF_NLS == 1 && DFF_WPS == 4), "two streams per workgroup, or one (its counters below stay zero)");

    auto run = [&](auto local_c, auto aux_c) {
        constexpr bool LOCAL = decltype(local_c)::value;
        constexpr int AUX = decltype(aux_c)::value;   // 0: none, 1: the pre-activations, 2: the reverse sweep's static rows
        int done0 = 0, done1 = 0, pref = 0;
        int m0 = 0, m1 = 0;   // blocks the streams' loaders had finished at the last look (wave-uniform)
        auto flags_min = [&](const i4v& r0, const i4v& r1) {   // DFF_WPS flags per stream, stream-major
            if (DFF_WPS == 4) {
                m0 = __builtin_amdgcn_readfirstlane(min(min(r0.x, r0.y), min(r0.z, r0.w)));
                if (DFF_NLS > 1) m1 = __builtin_amdgcn_readfirstlane(min(min(r1.x, r1.y), min(r1.z, r1.w)));
            } else {
                m0 = __builtin_amdgcn_readfirstlane(min(r0.x, r0.y));
                m1 = __builtin_amdgcn_readfirstlane(min(r0.z, r0.w));
            }
        };
        static_assert(DFF_WPS == 4 || DFF_WPS == 2, "");
        for (int left = nb0 + nb1; left > 0; --left) {
            int l0 = m0 - done0, l1 = m1 - done1;   // leads (>= 0: a loader stops at its stream's last block)
#ifdef DF_EXP_NOLOOK   // timing experiment: no look at the ready flags (only meaningful with DF_EXP_NOLOAD)
            l0 = nb0 - done0; l1 = nb1 - done1;
#endif
            if (l0 <= 0 && l1 <= 0) {   // nothing known to be ready: look until there is
                unsigned spins = 0;
                for (;;) {
                    const i4v r0 = rdy_p[0], r1 = (DFF_WPS == 4 && DFF_NLS > 1) ? rdy_p[1] : r0;
                    flags_min(r0, r1);
                    l0 = m0 - done0; l1 = m1 - done1;
                    if (l0 > 0 || l1 > 0) break;
                    __builtin_amdgcn_s_sleep(DF_CSLEEP_N);
                    bool give_up = false;
                    if (++spins > 4 * spin_limit) {   // the pass is lost; it must still end (node ids are bounded below)
                        __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        give_up = true;
                    }
                    if ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = true;
                    if (give_up) { l0 = nb0 - done0; l1 = nb1 - done1; break; }
                }
            }
            // smallest positive lead first, ties alternate - as ONE unsigned comparison: key = (lead - 1) << 1 | (1 if the
            // stream was served last); no lead wraps to the largest keys
            const unsigned k0 = ((unsigned)(l0 - 1) << 1) | (unsigned)pref, k1 = ((unsigned)(l1 - 1) << 1) | (unsigned)(pref ^ 1);
            const int st = k1 < k0 ? 1 : 0;
            pref = st ^ 1;
            const int b = st ? done1 : done0;
            done0 += st ^ 1;
            done1 += st;
            const float* sbase = lds.ring + (st * DF_NSLOT + (b & (DF_NSLOT - 1))) * Slot::words;
            static_assert((DF_NSLOT & (DF_NSLOT - 1)) == 0 && (DF_GIRING & (DF_GIRING - 1)) == 0, "ring sizes are powers of two");
            if (prof) dbg[8 * (int64_t)(DFF_NLS * b + st) + 0] = wall_clock64();
            // every LDS read of the block leaves in one go: the row's node id, the gate operands, the operand rows (a dead
            // row's lanes read their slot's stale words and drop them)
            const int gv = reinterpret_cast<const int*>(sbase + Slot::v_off)[row_l];
            float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, aval = 0.f;
            if (!proj) {   // from the gi0 ring (stacked layer 0) or from the slot (projection granules)
                const float* gp = (gi_ring ? lds.giring + (st * DF_GIRING + (b & (DF_GIRING - 1))) * (DF_RB * 3 * DFF_JS) : sbase + Slot::gi_off) +
                                  row_l * (3 * DFF_JS) + unit_l;
                gi_r = gp[0]; gi_z = gp[DFF_JS]; gi_n = gp[2 * DFF_JS];
                aval = sbase[Slot::a_off + row_l * Slot::AP + apos];
            }
            float g3[3];
            float rg = 0.f, zg = 0.f;
            {
                const float* a_seg = sbase + Slot::a_off + x * Slot::AP + ks * SEG;   // A operand: row x, K slice ks
                float4 bv[NK4];
#pragma unroll
                for (int q = 0; q < NK4; ++q) bv[q] = *reinterpret_cast<const float4*>(a_seg + 4 * q);
                if (KPT <= 16) __builtin_amdgcn_sched_barrier(0);   // (every read in flight before the first product: left alone the
                                                     // scheduler sinks them between the products, two registers ahead of their use -
                                                     // which is what H = 320 needs: its 120 weight registers leave no room for 40 more)
#ifdef DF_EXP_NOOPLD   // timing experiment: no operand reads from LDS
#pragma unroll
                for (int q = 0; q < NK4; ++q) bv[q] = make_float4(wr[q], wz[q], wn[q], aval);
#endif
                f4v acc[3] = {(f4v){0.f, 0.f, 0.f, 0.f}, (f4v){0.f, 0.f, 0.f, 0.f}, (f4v){0.f, 0.f, 0.f, 0.f}};
                // reduce-scatter of one gate's accumulators over the 8 K slices
                // (six DPP additions whose bank masks do the selecting: lanes with ks bit 0 clear keep rows (0, 1) and add the
                // lane 4 up, the others rows (2, 3) and the lane 4 down; then the same over ks bit 1 and 8 lanes.  Inline asm: the
                // destination is written under a mask, which the builtins cannot express; the wait states in front of a DPP
                // read of a fresh result - 6 behind an MFMA, 2 behind a vector instruction - are written out, the compiler
                // inserts none around asm.  Same additions in the same order as the select form.)
                auto reduce = [&](const f4v& A) -> float {
                    float e0, e1, f;
                    asm("s_nop 5\n\t"
                        "v_add_f32_dpp %0, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                        "v_add_f32_dpp %1, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                        "v_add_f32_dpp %0, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                        "v_add_f32_dpp %1, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                        "s_nop 1\n\t"
                        "v_add_f32_dpp %2, %0, %0 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                        "v_add_f32_dpp %2, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xc"
                        : "=&v"(e0), "=&v"(e1), "=&v"(f) : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]));
                    return df_row_pair_sum(f);
                };
#pragma unroll
                for (int q = 0; q < NK4; ++q) {
                    const float bq[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};
#ifdef DF_EXP_NOMFMA   // timing experiment: everything but the products
                    if (q < 3) acc[q] += (f4v){bq[0] * wr[q], bq[1] * wz[q], bq[2] * wn[q], bq[3]};
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(bq[e], wr[4 * q + e], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(bq[e], wz[4 * q + e], acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(bq[e], wn[4 * q + e], acc[2], 0, 0, 0);
                    }
#endif
                }
#ifndef DF_EXP_NOLOOK
                {   // the look for the NEXT block: issued behind the last product, read at the top of the next iteration
                    __builtin_amdgcn_sched_barrier(0);
                    const i4v r0 = rdy_p[0], r1 = (DFF_WPS == 4 && DFF_NLS > 1) ? rdy_p[1] : r0;
#endif
#ifdef DF_EXP_NOEPI   // timing experiment: no reduction, no gates
#pragma unroll
                for (int a = 0; a < 3; ++a) g3[a] = acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
#else
#pragma unroll
                for (int a = 0; a < 3; ++a) g3[a] = reduce(acc[a]);
#endif
#ifndef DF_EXP_NOLOOK
                    flags_min(r0, r1);
                }
#endif
            }
            if (prof) dbg[8 * (int64_t)(DFF_NLS * b + st) + 1] = wall_clock64();
            const float p_r = g3[0] + b_r, p_z = g3[1] + b_z, p_n = g3[2] + b_n;   // W a + b: the cell's pre-activations
            float hv = 0.f, ng = 0.f;
            if (!proj) {   // (every lane: no branch around the gates)
#ifdef DF_EXP_NOEPI
                hv = p_r + gi_r + p_z + gi_z + p_n + gi_n + aval;
#else
                rg = df_sigm(p_r + gi_r);
                zg = df_sigm(p_z + gi_z);
                ng = df_tanh(fmaf(rg, p_n, gi_n));
                hv = fmaf(zg, aval - ng, ng);   // n + z * (a - n)
#endif
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            df_flag_st(dn_or_dump + st * dn_step, b + 1);   // this wave is done with the slot (LDS executes a wave's accesses in order)
#ifdef DF_EXP_NOSTORE   // timing experiment: no global stores (only meaningful with DF_EXP_NOLOAD: nobody polls)
            asm volatile("" :: "v"(hv), "v"(p_r), "v"(p_z), "v"(p_n), "v"(gv));
            if (num_nodes > 0) continue;
#endif
            // (the bound on the node id also covers padding rows, id -1, and a slot that may hold anything once a wait has
            // expired: the pass is lost then, but it must not write outside its buffers)
            if (lane_st && (unsigned)gv < (unsigned)num_nodes) {
                if (proj) {   // input-side pre-activations of the upper cell: W_ih u + b_ih - the unit's three gates behind one tag
                    pgran_t* po = reinterpret_cast<pgran_t*>(g_out) + df_idx(gv, pld, unit);
                    const pgran_t pv = {epoch, __float_as_uint(p_r), __float_as_uint(p_z), __float_as_uint(p_n)};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(po), "v"(pv) : "memory");
                } else {
                    // hand-off store first (the consumers poll it): write-through (sc1: the line leaves this XCD's L2, any XCD's
                    // sc1 load finds it in memory), or - all readers are on this XCD - a plain 8-byte store that leaves the line in
                    // the shared L2; the plain state row behind it
                    if (LOCAL) g_out[df_idx(gv, gld, unit)] = gran_pack(epoch, hv);
                    else __hip_atomic_store(g_out + df_idx(gv, gld, unit), gran_pack(epoch, hv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h_out[df_idx(gv, ld_h, unit)] = hv;
                }
                if (AUX == 1) {   // (behind the hand-off stores: nobody waits for these)
                    float* ao = aux_out + df_idx(gv, 3 * H, unit);
                    ao[0] = p_r; ao[H] = p_z; ao[2 * H] = p_n;
                }
                if (AUX == 2 && !proj) {
                    // the reverse sweep's static rows of this node (csrc/bwd_dataflow.hip, bd_stat_kernel: the same terms in
                    // the same order, from the gates this pass applied): the state and the six coefficient rows in the
                    // record's lane order - column c at 4 (c % 64) + c / 64 of a 256-float row, columns 256.. (H = 320) in part
                    // B.  The row of external gradients is the reverse pass's to fill.
                    const float cn = (1.0f - zg) * (1.0f - ng * ng);
                    const float cz = (aval - ng) * zg * (1.0f - zg);
                    const float cr = cn * p_n * rg * (1.0f - rg);
                    const float cnr = cn * rg;
                    const float cq = zg * aval + cr * (p_r - b_r) + cz * (p_z - b_z) + cnr * (p_n - b_n);
                    // (a uniform base and a 32-bit byte offset - the host checks N x record bytes < 2^32 -, row offsets as immediates)
                    char* const rec = reinterpret_cast<char*>(aux_out) + ((unsigned)gv * (unsigned)(4 * df_stat_floats(H)) + stat_col);
                    auto put = [&](auto rs_c) {
                        constexpr int RS = 4 * decltype(rs_c)::value;
                        *reinterpret_cast<float*>(rec + DF_ST_H * RS) = hv; *reinterpret_cast<float*>(rec + DF_ST_CR * RS) = cr;
                        *reinterpret_cast<float*>(rec + DF_ST_CZ * RS) = cz; *reinterpret_cast<float*>(rec + DF_ST_CNR * RS) = cnr;
                        *reinterpret_cast<float*>(rec + DF_ST_CN * RS) = cn; *reinterpret_cast<float*>(rec + DF_ST_Z * RS) = zg;
                        *reinterpret_cast<float*>(rec + DF_ST_CQ * RS) = cq;
                    };
                    if (H > 256 && sl >= 256 / DFF_JS) put(std::integral_constant<int, 64>());   // (wave-uniform)
                    else put(std::integral_constant<int, DF_STAT_SP>());
                }
            }
            if (prof) dbg[8 * (int64_t)(DFF_NLS * b + st) + 2] = wall_clock64();
        }
    };
    typedef std::integral_constant<int, 0> aux0;
    typedef std::integral_constant<int, 1> aux1;
    typedef std::integral_constant<int, 2> aux2;
    if (!aux_out) { if (local_st) run(std::true_type(), aux0()); else run(std::false_type(), aux0()); }
    else if (proj || !S.aux_stat) { if (local_st) run(std::true_type(), aux1()); else run(std::false_type(), aux1()); }
    else { if (local_st) run(std::true_type(), aux2()); else run(std::false_type(), aux2()); }
}

template <int KPT>
__global__ void __launch_bounds__(DFF_THREADS, DFF_THREADS / 256) dataflow_kernel(const int32_t* __restrict__ plan, DfArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef DfSlot<KPT> Slot;
    constexpr int NS = 16 * KPT / DFF_JS;
    const int tid = threadIdx.x;
    if (S.status && S.status[0] != 0) {   // the batch violates the plan contract: the schedule is garbage - do not walk it
        if (tid == 0) __hip_atomic_fetch_or(S.err, 4 | ((S.status[0] & 0xff) << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (S.sched[0] != S.groups || S.sched[1] != DF_MAGIC || S.sched[2] != DF_RB) {   // a schedule built for another group count
        if (tid == 0) __hip_atomic_fetch_or(S.err, 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (or not built at all) would
        return;                                                                                      // index gtab / grec out of bounds
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup serves NLS groups ("streams"): by the placement table, or workgroup ids pair-major, then cell, then slice
    int pair, c, sl;
    if (S.nroles > 0) {
        const unsigned role = S.role[blockIdx.x];
        if (role == DF_IDLE_ROLE) return;
        pair = (int)(role >> 10); c = (int)((role >> 5) & 31u); sl = (int)(role & 31u);
    } else {
        const int per_pair = S.ncell * NS;
        pair = blockIdx.x / per_pair;
        const int rem = blockIdx.x - pair * per_pair;
        c = rem / NS; sl = rem - c * NS;
    }
    const DfCell& C = S.cell[c];
    DfLds lds;
    lds.ring = smem;
    lds.giring = lds.ring + DFF_NLS * DF_NSLOT * Slot::words;
    lds.rec = reinterpret_cast<int*>(lds.giring + DFF_NLS * DF_GIRING * DF_RB * 3 * DFF_JS);
    int* flags = lds.rec + DFF_NLS * DF_RB * 8 * 16;
    lds.rdy = flags;
    lds.dn = flags + DFF_NLW;
    lds.local = flags + DFF_NLW + DFF_NLS * DFF_NCW;
    lds.bias = reinterpret_cast<float*>(flags + 32);
    if (tid < DFF_NLW + DFF_NLS * DFF_NCW + 1) flags[tid] = 0;
    static_assert(DFF_NLW + DFF_NLS * DFF_NCW + 1 <= 32, "flag words");
    if (tid >= 64 && tid < 64 + 3 * DFF_JS) lds.bias[tid - 64] = C.bias[((tid - 64) / DFF_JS) * (16 * KPT) + sl * DFF_JS + (tid - 64) % DFF_JS];
    if (S.nroles > 0 && wave == 0) {
        // where does this workgroup really run?  Publish it, and - recurrent cells - look where the readers of this cell's
        // state rows run: its own slices and the slices of the projection cell above it, same workgroup set
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        if ((tid & 63) == 0)
            __hip_atomic_store(S.xcc_tab + blockIdx.x, gran_pack(S.epoch, __uint_as_float(xcc)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (C.kind == DF_RECURRENT) {
            bool same = true;
            for (int b0 = 0; b0 < S.nroles; b0 += 64) {
                const int b = b0 + (tid & 63);
                const unsigned r = b < S.nroles ? S.role[b] : DF_IDLE_ROLE;
                const int rc = (int)((r >> 5) & 31u);
                const bool member = r != DF_IDLE_ROLE && (int)(r >> 10) == pair && (rc == c || rc == C.partner);
                unsigned spins = 0;
                bool have = !member;
                unsigned theirs = xcc;
                for (;;) {
                    if (!have) {
                        const gran_t g = gran_ld(S.xcc_tab + b);
                        if ((unsigned)(g >> 32) == S.epoch) { theirs = (unsigned)g; have = true; }
                    }
                    if (__all(have) || !df_retry(spins, S.err, S.spin_limit)) break;
                }
                same = same && have && theirs == xcc;
            }
            if (__all(same) && (tid & 63) == 0) lds.local[0] = 1;
        }
    }
    if (DF_PROF && S.dbg && tid == 0) S.dbg[2 * blockIdx.x] = wall_clock64();
    if (DF_PROF && S.dbg && (int)blockIdx.x == S.dbg_wg && (tid & 63) == 0)   // where the waves of the stamped workgroup run (HW_REG_HW_ID)
        if (wave < 8) S.dbg[2 * gridDim.x + 8 * (int64_t)wave + 7] = 0x100000000ull | __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    __syncthreads();
    const int variant = C.variant;
    if (wave < DFF_NCW) {
        const int cw = wave;
        switch ((variant >> 2) & 3) {
            case DFK_REC0: df_compute<KPT, DFK_REC0>(S, C, sl, pair, lds, cw); break;
            case DFK_RECP: df_compute<KPT, DFK_RECP>(S, C, sl, pair, lds, cw); break;
            default: df_compute<KPT, DFK_PROJ>(S, C, sl, pair, lds, cw); break;
        }
    } else {
        const int set = (wave - DFF_NCW) / DFF_WPS;
        const int grp = df_stream_group(pair, set, S.groups);
        const int w = (wave - DFF_NCW) % DFF_WPS;   // the row of its stream's blocks this wave serves
        if (grp >= 0) {
#define DF_LOADER_CASE(K, RR, EX) case (K) * 4 + ((RR) == 2 ? 2 : 0) + ((EX) ? 1 : 0): df_loader<KPT, K, RR, EX>(plan, S, C, sl, grp, lds, w, set); break;
            if (variant == DFK_REC0 * 4 + 2) { df_loader_fast<KPT, DFK_REC0>(plan, S, C, sl, grp, lds, w, set); }
            else if (variant == DFK_RECP * 4 + 2) { df_loader_fast<KPT, DFK_RECP>(plan, S, C, sl, grp, lds, w, set); }
            else if ((variant >> 2) == DFK_PROJ) { df_loader_fast<KPT, DFK_PROJ>(plan, S, C, sl, grp, lds, w, set); }
            else if constexpr (KPT <= 16) {
                if (variant >= 16) {   // plain aggregators (add / max / none)
                    if (((variant >> 2) & 3) == DFK_REC0) df_loader<KPT, DFK_REC0, -1, false, true>(plan, S, C, sl, grp, lds, w, set);
                    else df_loader<KPT, DFK_RECP, -1, false, true>(plan, S, C, sl, grp, lds, w, set);
                } else
                switch (variant) {
                    DF_LOADER_CASE(DFK_REC0, 2, false) DF_LOADER_CASE(DFK_REC0, 2, true)
                    DF_LOADER_CASE(DFK_REC0, -1, false) DF_LOADER_CASE(DFK_REC0, -1, true)
                    DF_LOADER_CASE(DFK_RECP, 2, false) DF_LOADER_CASE(DFK_RECP, 2, true)
                    DF_LOADER_CASE(DFK_RECP, -1, false) DF_LOADER_CASE(DFK_RECP, -1, true)
                    default: df_loader<KPT, DFK_PROJ, -1, false>(plan, S, C, sl, grp, lds, w, set); break;
                }
            } else {   // H = 320 exists for the benchmarked cell variants only (the host checks: dagnn_dataflow_run)
                if (tid % 64 == 0) __hip_atomic_fetch_or(S.err, 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#undef DF_LOADER_CASE
        }
    }
    if (DF_PROF && S.dbg && tid == 0) S.dbg[2 * blockIdx.x + 1] = wall_clock64();   // compute wave 0 is done
}

template <int KPT> size_t df_lds_bytes() {
    return (size_t)DFF_NLS * (DF_NSLOT * DfSlot<KPT>::words + DF_GIRING * DF_RB * 3 * DFF_JS + DF_RB * 8 * 16) * 4 + 128 + 3 * DFF_JS * 4 + 128;
}

// Which element of W the idx-th float4 of a packed matrix holds: thread tc = (compute wave tc >> 6, unit quad (tc >> 5) & 1, K slice
// (tc >> 2) & 7, unit of the quad tc & 3) of the 256 compute threads.
struct DfPackPos { int sl, g, k4, unit, ks, kp; };
__device__ __forceinline__ DfPackPos df_pack_pos(int64_t idx, int H, bool transposed) {
    (void)transposed;
    DfPackPos P;
    P.kp = H >> 3;
    const int nk4 = P.kp >> 2, nq = 3 * nk4;
    const int tc = (int)(idx & 255);
    const int64_t rest = idx >> 8;
    const int q = (int)(rest % nq);
    P.sl = (int)(rest / nq);
    P.g = q / nk4; P.k4 = q - P.g * nk4;
    P.unit = P.sl * DF_JS + 8 * (tc >> 6) + 4 * ((tc >> 5) & 1) + (tc & 3); P.ks = (tc >> 2) & 7;
    return P;
}

// Pack W [3H, K = H] (torch GRUCell layout) for the dataflow kernel: out[(sl * NQ + q) * 256 + tc] (float4), NQ = 3 H/32,
// q = gate * (H/32) + k4, thread tc = (compute wave w = tc >> 6, unit quad = (tc >> 5) & 1, K slice ks = (tc >> 2) & 7, unit of
// the quad x = tc & 3):  W[gate * H + 32 sl + 8 w + 4 quad + x][ks * H/8 + 4 k4 .. + 3]  (the A operands of df_compute).
// `transposed`: pack the GATE-WISE TRANSPOSED matrix W'[g H + j][u] = W[g H + u][j] instead (the A operands of the reverse
// sweep's products W^T dg, bwd_dataflow.hip) - four strided reads per element instead of one float4.
__global__ void __launch_bounds__(256) df_pack_kernel(const float* __restrict__ W, float4* __restrict__ out, int H, int64_t total,
                                                       int transposed) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const DfPackPos P = df_pack_pos(idx, H, transposed != 0);
        const int sl = P.sl, g = P.g, k4 = P.k4, unit = P.unit, ks = P.ks, kp8 = P.kp;
        (void)sl;
        if (!transposed) {
            out[idx] = *reinterpret_cast<const float4*>(W + (int64_t)(g * H + unit) * H + ks * kp8 + 4 * k4);
        } else {
            const float* src = W + (int64_t)(g * H + ks * kp8 + 4 * k4) * H + unit;
            out[idx] = make_float4(src[0], src[H], src[2 * (int64_t)H], src[3 * (int64_t)H]);
        }
    }
}

// several matrices (and their transposes) in one launch: a training step re-packs every cell's weights, 6 + 6 launches at L = 2
struct DfPackJobs { dagnn_df_pack_job j[DAGNN_MAX_PACK_JOBS]; };
__global__ void __launch_bounds__(256) df_pack_batch_kernel(DfPackJobs P, int H, int64_t total) {
    const dagnn_df_pack_job& J = P.j[blockIdx.y];
    const float* __restrict__ W = J.w;
    if (J.transposed == 2) {   // edge gain of a cell: out[r] = sum_j edge_w[j, r] key[j] (= W_e^T w_key, dagnn.py:363,370), j ascending
        if (blockIdx.x == 0 && (int)threadIdx.x < J.cols) {
            float acc = 0.f;
            for (int j = 0; j < J.rows; ++j) acc = fmaf(W[(int64_t)j * J.cols + threadIdx.x], J.aux[j], acc);
            J.out[threadIdx.x] = acc;
        }
        return;
    }
    float4* __restrict__ out = reinterpret_cast<float4*>(J.out);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const DfPackPos P = df_pack_pos(idx, H, J.transposed != 0);
        const int g = P.g, k4 = P.k4, unit = P.unit, ks = P.ks, kp8 = P.kp;
        if (!J.transposed) {
            out[idx] = *reinterpret_cast<const float4*>(W + (int64_t)(g * H + unit) * H + ks * kp8 + 4 * k4);
        } else {
            const float* src = W + (int64_t)(g * H + ks * kp8 + 4 * k4) * H + unit;
            out[idx] = make_float4(src[0], src[H], src[2 * (int64_t)H], src[3 * (int64_t)H]);
        }
    }
}

// partial attention scores behind the state rows (the format the backward pass reads): part q of row v =
// w_key[16q : 16q + 16] . h[v, 16q : 16q + 16].  One wave per row.
struct DfScoreJobs { float* h[DAGNN_MAX_PACK_JOBS]; const float* wkey[DAGNN_MAX_PACK_JOBS]; };
__global__ void __launch_bounds__(256) df_score_parts_kernel(DfScoreJobs J, int ld_h, int H, int64_t N) {
    float* __restrict__ h = J.h[blockIdx.y];
    const float* __restrict__ wkey = J.wkey[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;
    float* row = h + v * ld_h;
    for (int l4 = lane; 4 * l4 < ((H + 255) & ~255); l4 += 64) {   // (wave-uniform trip count: the shuffles below see every lane; H = 320: two rounds)
        float s = 0.f;
        if (4 * l4 < H) {
            const float4 x = reinterpret_cast<const float4*>(row)[l4];
            const float4 w = reinterpret_cast<const float4*>(wkey)[l4];
            s = x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if ((l4 & 3) == 0 && 4 * l4 < H) row[H + (l4 >> 2)] = s;
    }
}

}  // namespace

#if defined(DF_WIDE_TU) || defined(DF_X_TU)
constexpr int DF_TU_MAX_H = 320;   // this translation unit: csrc/dataflow_w.hip (H = 320) or csrc/dataflow_x.hip (64-unit slices, H = 256 / 320)
#else
constexpr int DF_TU_MAX_H = 256;
extern "C" size_t dagnn_dataflow_bytes(int64_t N, int64_t B, int groups) {
    if (N < 0 || B < 0 || groups < 1 || groups > DF_MAX_GROUPS) return 0;
    return (size_t)df_layout_words(N, B, groups).total * sizeof(int32_t);
}

extern "C" int dagnn_dataflow_layout(int64_t N, int64_t B, int groups, int64_t* o) {
    if (!o || groups < 1 || groups > DF_MAX_GROUPS) return DAGNN_EINVAL;
    DfLayout S = df_layout_words(N, B, groups);
    const int64_t w[13] = {S.grp_of, S.gdepth, S.gload, S.loff, S.gtab[0], S.gtab[1], S.lcnt[0], S.lcnt[1],
                           S.glbase[0], S.glbase[1], S.grec[0], S.grec[1], S.total};
    for (int i = 0; i < 13; ++i) o[i] = w[i] * 4;
    return DAGNN_OK;
}

extern "C" int dagnn_dataflow_groups(int num_cus, int num_dirs, int num_stacked, int H, int64_t B) {
    if (num_cus <= 0 || num_dirs < 1 || num_dirs > DAGNN_MAX_DIRS || num_stacked < 1 || H <= 0 || (H % 64) || H > 320 || B <= 0)
        return 0;
    const int kcells = num_dirs * (2 * num_stacked - 1);   // one projection cell per stacked layer above the first
    if (kcells > DF_MAX_KCELLS) return 0;
    int64_t g = (int64_t)DFF_NLS * (num_cus / (kcells * (H / DFF_JS)));   // NLS groups per workgroup set
    if (g > DF_MAX_GROUPS) g = DF_MAX_GROUPS;
    if (g > B) g = B;
    return (int)g;
}

extern "C" int dagnn_dataflow_schedule(const dagnn_plan* pl, void* ws_, size_t ws_bytes, int groups, int cost_layer,
                                       int cost_row, const int32_t* status, void* stream_) {
    if (!pl || !pl->data || !ws_ || groups < 1 || groups > DF_MAX_GROUPS || cost_layer < 0 || cost_row < 0)
        return DAGNN_EINVAL;
    const int64_t N = pl->N, B = pl->B;
    DfLayout S = df_layout_words(N, B, groups);
    if ((size_t)S.total * 4 > ws_bytes) return DAGNN_ENOSPC;
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipStream_t st = (hipStream_t)stream_;
    int32_t* ws = (int32_t*)ws_;
    const int32_t* plan = (const int32_t*)pl->data;
    if (B == 0 || N == 0) {   // an empty batch: the initial state only (header, tables, counters = 0; records: node = -1)
        hipError_t e = hipMemsetAsync(ws, 0, (size_t)S.grec[0] * 4, st);
        if (e != hipSuccess) return DAGNN_EHIP(e);
        e = hipMemsetAsync(ws + S.grec[0], 0xff, (size_t)(S.total - S.grec[0]) * 4, st);
        return e == hipSuccess ? DAGNN_OK : DAGNN_EHIP(e);
    }
    if (!(pl->flags & DAGNN_PLAN_GENERAL_BUILD) && dagnn_plan_is_small(N, 0, B))   // one workgroup, everything in LDS (small.hip)
        return dagnn_dataflow_schedule_small(pl, ws, groups, cost_layer, cost_row, status, st);
    // workgroup 0: the LPT assignment; the others initialise tables and records next to it
    hipLaunchKernelGGL(df_assign_kernel, dim3(1 + 512), dim3(256), 0, st, plan, L, ws, S, (int)B, groups, cost_layer, cost_row, status);
    DAGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(df_count_kernel, dim3((unsigned)B, 2), dim3(256), 0, st, plan, L, ws, S, status);
    DAGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(df_prefix_kernel, dim3((unsigned)groups, 2), dim3(256), 0, st, ws, S, groups, status);
    DAGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(df_base_kernel, dim3(2), dim3(64), 0, st, ws, S, groups, status);
    DAGNN_CHECK_LAUNCH();
    int64_t lb = (N + groups + 3) / 4;
    if (lb > 2048) lb = 2048;
    hipLaunchKernelGGL(df_lbase_kernel, dim3((unsigned)lb, 2), dim3(256), 0, st, plan, L, ws, S, (int)B, groups, status);
    DAGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(df_records_kernel, dim3((unsigned)((N + 255) / 256), 2), dim3(256), 0, st, plan, L, ws, S, (int)N, status);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

static int df_pack(const float* w, float* out, int H, int transposed, void* stream) {
    if (!w || !out || H <= 0 || (H % 64) || H > 320) return DAGNN_EINVAL;
    const int64_t total = (int64_t)3 * H * H / 4;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(df_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(out), H, total, transposed);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_pack_dataflow(const float* w, float* out, int H, void* stream) { return df_pack(w, out, H, 0, stream); }

extern "C" int dagnn_pack_dataflow_transposed(const float* w, float* out, int H, void* stream) { return df_pack(w, out, H, 1, stream); }

extern "C" int dagnn_pack_dataflow_batch(const dagnn_df_pack_job* jobs, int njob, int H, void* stream) {
    if (!jobs || njob <= 0 || njob > DAGNN_MAX_PACK_JOBS || H <= 0 || (H % 64) || H > 320) return DAGNN_EINVAL;
    DfPackJobs P;
    for (int q = 0; q < njob; ++q) {
        if (!jobs[q].w || !jobs[q].out) return DAGNN_EINVAL;
        if (jobs[q].transposed == 2 && (!jobs[q].aux || jobs[q].rows <= 0 || jobs[q].cols <= 0 || jobs[q].cols > 256)) return DAGNN_EINVAL;
        P.j[q] = jobs[q];
    }
    const int64_t total = (int64_t)3 * H * H / 4;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(df_pack_batch_kernel, dim3((unsigned)blocks, (unsigned)njob), dim3(256), 0, (hipStream_t)stream, P, H, total);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_score_parts_batch(float* const* h, const float* const* w_key, int n, int ld_h, int H, int64_t N, void* stream) {
    if (!h || !w_key || n <= 0 || n > DAGNN_MAX_PACK_JOBS || H <= 0 || (H % 16) || ld_h < H + H / 16 || N < 0) return DAGNN_EINVAL;
    if (N == 0) return DAGNN_OK;
    DfScoreJobs J;
    for (int q = 0; q < n; ++q) {
        if (!h[q] || !w_key[q]) return DAGNN_EINVAL;
        J.h[q] = h[q]; J.wkey[q] = w_key[q];
    }
    hipLaunchKernelGGL(df_score_parts_kernel, dim3((unsigned)((N + 3) / 4), (unsigned)n), dim3(256), 0, (hipStream_t)stream, J, ld_h, H, N);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_score_parts(float* h, int ld_h, int H, const float* w_key, int64_t N, void* stream) {
    return dagnn_score_parts_batch(&h, &w_key, 1, ld_h, H, N, stream);
}

#endif   // !DF_WIDE_TU && !DF_X_TU

#if defined(DF_X_TU)
extern "C" int dagnn_dataflow_run_x(const dagnn_plan* pl, const dagnn_dataflow_args* a, void* stream) {
    if (!pl || !pl->data || !a || !a->schedule) return DAGNN_EINVAL;
    if (a->H != 256 && a->H != 320) return DAGNN_EINVAL;
    if (a->H > 256) {   // (the lean loaders only, as in dagnn_dataflow_run_wide)
        if (pl->num_edge_feats != 2 || a->vid_mod > 0) return DAGNN_EINVAL;
        for (int d = 0; d < 2; ++d)
            for (int i = 0; i < a->num_stacked && i < DAGNN_MAX_STACKED; ++i)
                if (((a->dir_mask >> d) & 1) && (a->cell[d][i].static_score || !a->cell[d][i].edge_gain || a->cell[d][i].agg != 0)) return DAGNN_EINVAL;
    }
#elif defined(DF_WIDE_TU)
extern "C" int dagnn_dataflow_run_wide(const dagnn_plan* pl, const dagnn_dataflow_args* a, void* stream) {
    if (!pl || !pl->data || !a || !a->schedule) return DAGNN_EINVAL;
    if (a->H <= 256) return DAGNN_EINVAL;
    // (the wide shape has the lean loaders only: two edge features folded inline, no static scores / vertex-id key biases)
    if (pl->num_edge_feats != 2 || a->vid_mod > 0) return DAGNN_EINVAL;
    for (int d = 0; d < 2; ++d)
        for (int i = 0; i < a->num_stacked && i < DAGNN_MAX_STACKED; ++i)
            if (((a->dir_mask >> d) & 1) && (a->cell[d][i].static_score || !a->cell[d][i].edge_gain)) return DAGNN_EINVAL;
#else
extern "C" int dagnn_dataflow_run(const dagnn_plan* pl, const dagnn_dataflow_args* a, void* stream) {
    if (!pl || !pl->data || !a || !a->schedule) return DAGNN_EINVAL;
    if (a->slices64 && (a->H == 256 || a->H == 320)) return dagnn_dataflow_run_x(pl, a, stream);   // 64-unit slices: csrc/dataflow_x.hip
    if (a->H > 256) return dagnn_dataflow_run_wide(pl, a, stream);   // H = 320: csrc/dataflow_w.hip
#endif
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3, G = a->groups;
    if (H <= 0 || (H % 64) || H > DF_TU_MAX_H || Ls <= 0 || Ls > DAGNN_MAX_STACKED || !dir_mask || a->ld_h < H || a->gld < H ||
        (Ls > 1 && a->pld < H) || G < 1 || G > DF_MAX_GROUPS || a->epoch == 0 || !a->err)
        return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    // row offsets inside the granule buffers are 32-bit in the kernel (granules: 8 bytes each)
    if ((int64_t)pl->N * a->gld >= (1ll << 31) || (int64_t)pl->N * a->pld >= (1ll << 31)) return DAGNN_EINVAL;
    // ... and the compute waves index every row-major buffer with a 24 x 24 -> 32-bit multiply-add (df_idx)
    if (pl->N >= (1 << 24) || (int64_t)pl->N * a->ld_h >= (1ll << 31) || (int64_t)pl->N * 3 * H >= (1ll << 31) || a->ld_h >= (1 << 24) ||
        a->gld >= (1 << 24) || a->pld >= (1 << 24))
        return DAGNN_EINVAL;
    DfArgs S;
    int nc = 0;
    for (int d = 0; d < 2; ++d) {
        if (!((dir_mask >> d) & 1)) continue;
        for (int i = 0; i < Ls; ++i) {
            const dagnn_dataflow_cell& c = a->cell[d][i];
            if (c.agg < 0 || c.agg > 3 || (c.agg != 0 && (pl->num_edge_feats > 2 || a->vid_mod > 0 || c.static_score))) return DAGNN_EINVAL;
            if (!c.w_hh || !c.b_hh || (c.agg == 0 && !c.w_key && !c.static_score) || !c.h_out || !c.granules) return DAGNN_EINVAL;
            if (i == 0 ? !c.gi0 : (!c.w_ih || !c.b_ih || !c.proj_granules)) return DAGNN_EINVAL;
            if (nc + (i > 0 ? 2 : 1) > DF_MAX_KCELLS) return DAGNN_EINVAL;
            if (a->stat_rows && (!c.gh_out || c.gi_out)) return DAGNN_EINVAL;
            if (i > 0) {   // projection cell: W_ih x (states of layer i - 1) + b_ih
                DfCell& P = S.cell[nc++];
                P.w = (const float4*)c.w_ih; P.bias = c.b_ih;
                P.wkey = nullptr; P.sscore = nullptr; P.gain = nullptr; P.vid = nullptr; P.gi0 = nullptr; P.p_in = nullptr;
                P.agg_w = nullptr; P.agg_b = nullptr; P.agg = 0;
                P.h_out = nullptr;
                if ((uintptr_t)c.proj_granules & 15) return DAGNN_EINVAL;
                P.g_out = (gran_t*)c.proj_granules;
                P.g_in = (const gran_t*)a->cell[d][i - 1].granules;
                P.aux_out = c.gi_out;
                P.dir = d; P.kind = DF_PROJECTION; P.variant = DFK_PROJ * 4; P.partner = -1;
                S.cell[nc - 2].partner = nc - 1;   // (the cell before it in the table is the recurrent cell whose rows it reads)
            }
            DfCell& K = S.cell[nc++];
            K.w = (const float4*)c.w_hh; K.bias = c.b_hh;
            K.wkey = c.static_score ? nullptr : c.w_key;
            K.sscore = c.static_score;
            K.gain = pl->num_edge_feats > 0 ? c.edge_gain : nullptr;
            K.vid = a->vid_mod > 0 ? c.vid_bias : nullptr;
            K.gi0 = i == 0 ? c.gi0 : nullptr;
            K.p_in = i > 0 ? (const pgran_t*)c.proj_granules : nullptr;
            K.h_out = c.h_out;
            K.g_out = (gran_t*)c.granules;
            K.g_in = nullptr;
            K.aux_out = c.gh_out;
            K.dir = d; K.kind = DF_RECURRENT;
            K.variant = (i == 0 ? DFK_REC0 : DFK_RECP) * 4 + ((K.gain && pl->num_edge_feats == 2) ? 2 : 0) + ((K.sscore || K.vid) ? 1 : 0);
            K.partner = -1;
            K.agg = c.agg; K.agg_w = c.agg != 0 ? c.agg_edge_w : nullptr; K.agg_b = c.agg != 0 ? c.agg_edge_b : nullptr;
            if (c.agg != 0) { K.variant = 16 * c.agg + (i == 0 ? DFK_REC0 : DFK_RECP) * 4; K.gain = nullptr; K.wkey = nullptr; }
        }
    }
    const DfLayout SL = df_layout_words(pl->N, pl->B, G);
    S.sched = (const int32_t*)a->schedule;
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    for (int d = 0; d < 2; ++d) { S.gtab[d] = SL.gtab[d]; S.grec[d] = SL.grec[d]; S.col[d] = L.col[d]; S.eattr[d] = L.eattr[d]; }
    S.spin_limit = a->spin_limit ? a->spin_limit : (1u << 22);
    S.N = (int)pl->N;
    S.aux_stat = a->stat_rows ? 1 : 0;
    S.ncell = nc; S.H = H; S.ld_h = a->ld_h; S.gld = a->gld; S.pld = a->pld; S.R = pl->num_edge_feats;
    S.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
    S.groups = G; S.epoch = a->epoch; S.err = (int*)a->err;
    S.dbg = (unsigned long long*)a->debug_timing;
    S.dbg_wg = a->debug_wg;
    S.status = (const int32_t*)a->plan_status;
    const int32_t* plan = (const int32_t*)pl->data;
    unsigned grid = (unsigned)(df_sets_for(G) * nc * (H / DFF_JS));
    // XCD-aware placement (optional: the caller names the CU count and lends a tagged table): units = a recurrent cell's
    // slices + the slices of the projection cell reading its rows; bins = the 8 XCDs (num_cus / 8 CUs each, one workgroup
    // per CU); largest units first, first fit; workgroup (slot k of XCD x) = k * 8 + x under the observed dispatch rule.
    // If the units do not fit, the linear mapping stays (and every hand-off store is write-through, as before).
    S.nroles = 0; S.xcc_tab = (gran_t*)a->xcc_table;
    if (a->num_cus >= 8 && a->num_cus <= DF_MAX_WGS && a->xcc_table && df_sets_for(G) < 64 && nc <= 31) {
        const int NS = H / DFF_JS, sets = df_sets_for(G), cap = a->num_cus / 8;
        int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int xrot = (a->xcd_first % 8 + 8) % 8;   // bin x of the packing = XCD (x + xcd_first) mod 8
        for (int b = 0; b < DF_MAX_WGS; ++b) S.role[b] = DF_IDLE_ROLE;
        bool ok = true;
        int top = 0;
        for (int pass = 0; pass < 2 && ok; ++pass)       // pass 0: units with a partner (2 NS workgroups), pass 1: without
            for (int set = 0; set < sets && ok; ++set)
                for (int c = 0; c < nc && ok; ++c) {
                    if (S.cell[c].kind != DF_RECURRENT || (S.cell[c].partner >= 0) != (pass == 0)) continue;
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
    hipStream_t st = (hipStream_t)stream;
#define DF_LAUNCH(KPT)                                                                                                   \
    do {                                                                                                                 \
        const void* fn = reinterpret_cast<const void*>(dataflow_kernel<KPT>);                                            \
        const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)df_lds_bytes<KPT>()); \
        if (ea != hipSuccess) return DAGNN_EHIP(ea);                                                                     \
        hipLaunchKernelGGL((dataflow_kernel<KPT>), dim3(grid), dim3(DFF_THREADS), df_lds_bytes<KPT>(), st, plan, S);      \
    } while (0)
#if defined(DF_X_TU)
    // (a persistent kernel: every workgroup must be resident)
    if (a->num_cus > 0 && (int64_t)grid > a->num_cus && S.nroles == 0) return DAGNN_EINVAL;
    if (H == 320) DF_LAUNCH(20); else DF_LAUNCH(16);
#elif defined(DF_WIDE_TU)
    DF_LAUNCH(20);
#else
    switch (H / 16) {
        case 4: DF_LAUNCH(4); break;
        case 8: DF_LAUNCH(8); break;
        case 12: DF_LAUNCH(12); break;
        default: DF_LAUNCH(16); break;
    }
#endif
#undef DF_LAUNCH
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}
