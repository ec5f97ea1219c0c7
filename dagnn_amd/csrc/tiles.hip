// tiles.hip - weight-stationary tile schedule of the recurrence for WIDE hidden states (H = 512): the path of
// BASELINE.json's cfg 5 (batch 256, hidden 512, 5 stacked layers, bidirectional).
//
// Reference path replaced: the loop nest of ogbg-code/model/dagnn.py:144-182
//   for d in dirs: for l_idx in range(T): [edge scan :151-157] for i, cell in cells_d:
//       ps_h = AttnConv(...)[layer] (:175-179, message :366-373);  inp = GRUCell(inp, ps_h) (:181);  h[d][i][layer] += inp (:182)
// in the same legal re-ordering as frontier.hip (stacked layer i of a node needs layer i of its predecessors and layer
// i-1 of the node itself, nothing else).
//
// Why another schedule.  At H = 512 the ten cells' GRU matrices are 56.6 MB: a launch per topological layer streams them
// from the infinity cache 375 times (frontier.hip: 25 ms per batch, of which 9 ms in 250 thin launches that do nothing
// else), and the 4-row blocks of the dataflow kernel (dataflow.hip) cannot feed 977 GFLOP of products.  Here the weights
// never move: ONE persistent launch per chunk of stacked layers, every workgroup = (cell, 16-unit slice[, replica]) keeps
// its 48 gate rows x K of W_ih | W_hh in the registers of its 6 compute waves (128 VGPRs each at K = 1024) for the whole
// pass, and the frontier rows stream past it in tiles of 16.  Roles follow the SIMD a wave runs on (wave w -> SIMD w % 4):
// an fp32 MFMA runs at the vector rate and keeps its SIMD's vector ALU busy, so a loader wave next to two waves of
// back-to-back MFMAs gets no issue slot until they are done (measured) -
//   * SIMD 3: three loader waves build the tile's operand rows in LDS, one tile AHEAD of the products: the node's
//     lower-layer state row and the row of its first predecessor by LDS-DMA straight into the operand tile (most rows
//     have one predecessor: that row IS the aggregate), the second predecessor and the partial scores into registers,
//     rows with more predecessors folded in place (online segment soft-max, PyG's exp(x - max) / (sum + 1e-16); scores
//     from the 16-unit partial dots the producers store behind every state row);
//   * SIMDs 0-2: one gate each on two compute waves (the K halves - with an input side: W_ih on the lower-layer row and
//     W_hh on the aggregate, which the GRU's n gate needs apart anyway), [16 rows x K/2] x [K/2 x 16] per wave on
//     v_mfma_f32_16x16x4_f32 (exact fp32), partial tiles to LDS; waves 8-10 have no role and exit;
//   * loader wave 0 adds the two partials of each gate, evaluates the gates for the tile BEHIND the products, stores its
//     16-unit state slice (write-through) and the slice's partial attention score, and publishes a progress counter.
// Hand-off between workgroups (cdna_hip_programming.md Guideline 16, form R1): payload by write-through (sc1) stores,
// `s_waitcnt vmcnt(0)`, then ONE 8-byte {epoch, next own tile} counter per (cell, replica, slice); a consumer polls the
// 32 counters of its own cell (all earlier LAYERS complete) and of the cell below (the same TILE complete) with relaxed
// agent-scope loads.  Rows are then read with ordinary cached loads: a row is only ever read after the counters say it
// is complete, so no cache of the reader's XCD can hold an older copy.  Nothing depends on placement or timing; every
// wait is bounded and raises `err`.
// Tiles follow the plan's batch-level layers (blptr / rowrec of plan.hip): tile k of layer t covers record slots
// [blptr[t] + 16 (k - first tile of t), +16); replica r of a cell takes the tiles whose position in their layer is r mod R.
// Chunks: stacked layer 0 (hidden side only, gi0 from the batched GEMM) is one launch with as many replicas as fit;
// the layers above run together while 32 workgroups per cell fit the device (4 layers x 2 directions = 256 CUs).
// `first_layer`: the walk may start behind the wide first layers (done by dagnn_frontier_run): DESIGN.md 4f.
#include "df_common.h"   // (df_wave_umin)

namespace {

constexpr int TH = 512;                          // hidden size
constexpr int TU = 16;                           // hidden units per workgroup
constexpr int TNS = TH / TU;                     // slices per cell (32)
constexpr int TR = 16;                           // rows per tile
constexpr int TNCW = 6;                          // compute waves: (gate, K half), two on each of the SIMDs 0-2
constexpr int TNLW = 3;                          // loader waves: the three waves of SIMD 3
constexpr int TNQ = 6;                           // rows of a tile per loader wave (rows lw', lw' + 3, ...: 6 + 5 + 5)
constexpr int TTHREADS = 64 * 12;                // 768: 3 waves per SIMD (wave w runs on SIMD w % 4), <= 168 VGPRs
constexpr int TMAXCELL = 16;                     // cells of one launch
constexpr int TMAXREP = 8;
constexpr int TMAXCID = 2 * DAGNN_MAX_STACKED;   // counters: [cell id][replica][slice]
#ifndef T_THIN_ROWS
#define T_THIN_ROWS 8                             // tiles of up to this many live rows: v_mfma_f32_4x4x1 passes of 4 rows (0, 4 or 8)
#endif
#ifndef T_REP0
#define T_REP0 4                                 // replicas of the stacked-layer-0 launch (at most)
#endif
#ifndef T_SLACK
#define T_SLACK 3                                // tiles a cell stays behind the cell below (see wait_low)
#endif
#ifndef T_CHUNK
#define T_CHUNK 2                                // predecessors per further trip of a row with more than two
#endif

typedef float tf4 __attribute__((ext_vector_type(4)));
typedef unsigned tu4 __attribute__((ext_vector_type(4)));

struct TCell {
    const float* whh;    // [3H, H] torch layout
    const float* wih;    // [3H, H] torch layout, or null (stacked layer 0)
    const float* bhh;    // [3H]
    const float* bih;    // [3H] (with wih)
    const float* wkey;   // [H]
    const float* gain;   // [R] or null
    const float* vid;    // [vid_mod] key bias by vertex id (the NA encoder, dvae/dagnn.py:130-134) or null
    const float* gi0;    // [N, 3H] (stacked layer 0) or null
    const float* h_in;   // [N, ld_h] the stacked layer below (with wih)
    float* h_out;        // [N, ld_h]: H states + H/16 partial scores per row
    int dir;
    int cid;             // counter row of this cell
    int low;             // counter row of the cell below when it runs in THIS launch, else -1
};

struct TArgs {
    TCell cell[TMAXCELL];
    int ncell, R, ld_h, nfeat, vid_mod;
    int first_layer[2];       // per direction: the walk starts at this batch-level layer (the layers before it are complete)
    unsigned epoch, spin_limit;
    gran_t* prog;             // [TMAXCID][TMAXREP][TNS] {epoch, P}: every tile k < P of this (cell, replica, slice) is published
    int* err;
    const int32_t* status;    // plan status word or null
    unsigned long long* dbg;  // -DT_STAMPS builds: [workgroups][32] phase sums in 100 MHz ticks (scripts/tiles_stamps.py)
};

#ifdef T_STAMPS
#ifndef T_TRACE_WG
#define T_TRACE_WG 8
#define T_TRACE_IT 400
#endif
__device__ __forceinline__ unsigned long long t_now() {   // pinned in program order (volatile asm + memory clobber)
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define T_CLK(x) const unsigned long long x = t_now()
#define T_ACC(sum, a, b) sum += (b) - (a)
#else
#define T_CLK(x)
#define T_ACC(sum, a, b)
#endif

template <bool HAS_IN> struct TShape {
    static constexpr int K = HAS_IN ? 2 * TH : TH;   // operand row: [lower-layer state |] aggregate
    static constexpr int PITCH = K + 4;              // LDS row pitch (floats)
    static constexpr int KW = K / 2;                 // k range of one compute wave (a K half of its gate)
    static constexpr int KJ = KW / 16;               // groups of 4 MFMAs (one ds_read_b128 of B each)
    static constexpr int BT = 2 * TR * PITCH;        // floats: two operand tiles
    static constexpr int RED = 2 * TNCW * 64 * 4;    // partial tiles of the 6 compute waves (thin tiles: two passes of 4 rows each)
    static constexpr int ASV = 3 * TR * TU;          // the slice's aggregate values of three tiles in flight
    static constexpr int GSV = HAS_IN ? 0 : 3 * TR * 3 * TU;   // stacked layer 0: the slice's gi0 values likewise
    static constexpr int RING = 4 * TR * 16;         // row records (64 B each) of four tiles in flight
    static constexpr int CST = 5 * TU;               // the slice's biases (r, z, n input side, n hidden side) and key weights
    static constexpr int PSV = 2 * TR * 64;          // partial attention scores of the rows' first two predecessors (32 + 32 per row)
    static constexpr size_t lds_bytes = (size_t)(BT + RED + ASV + GSV + RING + CST + PSV + 8) * 4;
};
static_assert(TShape<true>::lds_bytes <= 160 * 1024 && TShape<false>::lds_bytes <= 160 * 1024, "LDS budget");

struct Tile { int k, t, tb, te, slot0, nr, valid; };   // te: first tile index of the next layer

// Walks the batch-level layers of one direction.  Tile k of layer t (first tile tb) belongs to replica (k - tb) % R: the
// FIRST tile of every layer - in the thin tail the only one - is replica 0's, so the dependent chain of the thin layers
// stays inside one replica's 32 workgroups (one XCD) instead of hopping across replicas with every layer (measured on the
// stacked-layer-0 launch, 4 replicas: 20 us per layer with tiles dealt k % R).
struct TileWalk {
    const int32_t* bl;
    int T, R, rep, t, tb, r0, r1;
    __device__ __forceinline__ int ntl() const { return (r1 - r0 + TR - 1) / TR; }
    __device__ __forceinline__ void init(const int32_t* bl_, int T_, int R_, int rep_, int t0) {   // t0: first layer to walk (tile 0)
        bl = bl_; T = T_; R = R_; rep = rep_; t = t0 < T_ ? t0 : T_; tb = 0; r0 = bl[t]; r1 = t < T ? bl[t + 1] : r0;
    }
    __device__ __forceinline__ void advance() { tb += ntl(); ++t; r0 = r1; r1 = t < T ? bl[t + 1] : r1; }
    __device__ __forceinline__ Tile make(int k) const {
        Tile x;
        x.valid = t < T ? 1 : 0; x.k = k; x.t = t; x.tb = tb; x.te = tb + ntl(); x.slot0 = r0 + (k - tb) * TR; x.nr = min(TR, r1 - x.slot0);
        return x;
    }
    // this replica's first tile
    __device__ __forceinline__ Tile first() {
        while (t < T && ntl() <= rep) advance();
        return make(tb + rep);
    }
    // this replica's next tile after its tile k (k must be the tile returned last)
    __device__ __forceinline__ Tile next(int k) {
        if (t < T && k + R < tb + ntl()) return make(k + R);
        if (t < T) advance();
        while (t < T && ntl() <= rep) advance();
        return make(tb + rep);
    }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t t_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffffc, 0x00020000);
}
__device__ __forceinline__ void t_st16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    tu4 x;
    x.x = __float_as_uint(v.x); x.y = __float_as_uint(v.y); x.z = __float_as_uint(v.z); x.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(x, r, byte_off, 0, 16);
}

// sum over the 64 lanes, the same in every lane (row scans, then row 0 -> 1, 2 -> 3 and the first half into the second)
__device__ __forceinline__ float t_wave_total(float v) {
#define T_DPP_ADD(ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    T_DPP_ADD(0x111, 0xf); T_DPP_ADD(0x112, 0xf); T_DPP_ADD(0x114, 0xf); T_DPP_ADD(0x118, 0xf);
    T_DPP_ADD(0x142, 0xa); T_DPP_ADD(0x143, 0xc);
#undef T_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// gate non-linearities on the hardware exp / rcp (as dataflow.hip): sigma(x) = 1 / (1 + e^-x), tanh(x) = 1 - 2 / (1 + e^2x)
__device__ __forceinline__ float t_sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float t_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// passes of 4 rows a tile of nr live rows takes on the 4x4x1 products (0: the 16-column products)
__device__ __forceinline__ int t_thin_passes(int nr) { return nr <= T_THIN_ROWS ? (nr + 3) >> 2 : 0; }

__device__ __forceinline__ void t_fma(float4& acc, float w, const float4& v) {
    acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
__device__ __forceinline__ void t_scale(float4& a, float s) { a.x *= s; a.y *= s; a.z *= s; a.w *= s; }
__device__ __forceinline__ float4 t_add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// LDS flags between the waves of a workgroup, as inline asm: this kernel uses LDS-DMA, and around a compiler-visible LDS
// access hipcc drains EVERY outstanding vector-memory operation of the wave first (s_waitcnt vmcnt(0): the access might
// alias a DMA in flight) - a round trip to memory per flag read.  LDS operations of one wave execute in order, so a flag
// store behind `s_waitcnt lgkmcnt(0)` is ordered after the wave's earlier LDS reads and writes.
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned*)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)p), "v"(v) : "memory");
}

struct TWait {
    int* err; unsigned limit;
    // bounded spin step: false when the budget is gone (the pass is lost; it still ends)
    __device__ __forceinline__ bool again(unsigned& spins, int code) const {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 63u) == 0u && spins > 256u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (spins > limit) { __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        return true;
    }
};

template <bool HAS_IN, bool VID>
__device__ __forceinline__ void tile_body(const int32_t* __restrict__ plan, const PlanLayout& L, const TArgs& S, const TCell& C,
                                          const int slice, const int rep, float* smem) {
    using SH = TShape<HAS_IN>;
    constexpr int PITCH = SH::PITCH, KJ = SH::KJ, KW = SH::KW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = C.dir, R = S.R, ld_h = S.ld_h;
    float* Bt = smem;
    float* red = Bt + SH::BT;
    float* asv = red + SH::RED;
    float* gsv = asv + SH::ASV;
    int* ring = reinterpret_cast<int*>(gsv + SH::GSV);
    float* cst = reinterpret_cast<float*>(ring + SH::RING);
    float* psv = cst + SH::CST;
    unsigned* flags = reinterpret_cast<unsigned*>(psv + SH::PSV);   // [0] / [2] tiles whose lower-layer / own-cell inputs are
                                                                    // there, [1] partial tiles the gate stage has read

    const int32_t* bl = plan + L.blptr[d];
    const int Nn = plan[PH_N];
    const int T = bl[(int64_t)Nn + 1];
    TileWalk W;
    W.init(bl, T, S.R, rep, S.first_layer[d]);
    Tile cur = W.first();
    if (threadIdx.x < 8) flags[threadIdx.x] = 0u;
    if (threadIdx.x < SH::CST) {   // gate constants of the slice: (b_ir + b_hr, b_iz + b_hz, b_in, b_hn, w_key)
        const int k = threadIdx.x >> 4, j = slice * TU + (threadIdx.x & 15);
        float c;
        if (k == 0) c = C.bhh[j] + (HAS_IN ? C.bih[j] : 0.f);
        else if (k == 1) c = C.bhh[TH + j] + (HAS_IN ? C.bih[TH + j] : 0.f);
        else if (k == 2) c = HAS_IN ? C.bih[2 * TH + j] : 0.f;
        else if (k == 3) c = C.bhh[2 * TH + j];
        else c = C.wkey[j];
        cst[threadIdx.x] = c;
    }
    __syncthreads();
    if (!cur.valid) return;
    Tile nxt = W.next(cur.k);

    // Roles by SIMD (wave w runs on SIMD w % 4): an fp32 MFMA runs at the VECTOR rate - it keeps the SIMD's vector ALU busy, and a
    // loader wave sharing a SIMD with two waves of back-to-back MFMAs gets no issue slot until they are done (measured:
    // its phase started 2.6 us late, when the products ended).  So SIMD 3 belongs to the loaders alone, SIMDs 0-2 each
    // run ONE gate of the slice on two waves (the K halves); waves 8-10 have no role.
    const TWait wt{S.err, S.spin_limit};
    const int simd = wave & 3;
    if (simd != 3 && wave >= 8) return;
    const bool is_compute = simd != 3;
    const int lw = wave >> 2;   // loader wave 0..2
    int it = 0;

    if (is_compute) {
        // ------------------------------------------------------------------ compute waves
        // this wave's KW/4 weights, the A operands: lane l holds W[gate row 16 s + (l & 15)][k0 + k(j, l >> 4) + e] (K order: below)
        const int m = lane & 15, kq = lane >> 4;
        const int g = simd, hh = wave >> 2;   // gate; K half: with an input side 0 = W_ih on the lower-layer row, 1 = W_hh on the aggregate
        const float* Wm = (HAS_IN && hh == 0) ? C.wih : C.whh;
        const int kb = HAS_IN ? 0 : hh * KW;
        float wr[KJ][4];
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const float4 w4 = *reinterpret_cast<const float4*>(Wm + (int64_t)(g * TH + slice * TU + m) * TH + kb + 256 * (j >> 4) + 64 * kq + 4 * (j & 15));
            wr[j][0] = w4.x; wr[j][1] = w4.y; wr[j][2] = w4.z; wr[j][3] = w4.w;
        }
        // K order of the products: step j of lane group kq takes k = 256 (j / 16) + 64 kq + 4 (j % 16) .. + 3 (any order is as good
        // for the sum, both operands follow it).  The 16 lanes a ds_read_b128 serves together then read node rows PITCH = 4
        // (mod 64) dwords apart at the same in-row offset mod 64: their 16-byte windows tile the 64 banks, no conflict
        // (with k = 16 j + 4 kq the kq = 0 and kq = 1 lanes of a group collided: 35 % of the LDS cycles, profiles/r03_pmc_tiles.json)
        const int b_off = m * PITCH + hh * KW + 64 * kq;
#ifdef T_STAMPS
        unsigned long long c_mfma = 0, c_red = 0, c_bar = 0;
#endif
        __syncthreads();   // (the loaders: row records of the first two tiles)
        __syncthreads();   // the first tile's operand rows are in LDS
        for (;;) {
            const bool pipelined = nxt.valid && nxt.t == cur.t;
            const float* bp = Bt + (it & 1) * TR * PITCH + b_off;
            T_CLK(c0);
            tf4 acc = tf4{0.f, 0.f, 0.f, 0.f};
            const int thin = t_thin_passes(cur.nr);   // (wave-uniform)
            if (thin > 0) {
                // A tile of <= 8 live rows (most layers of the thin tail hold one or two): v_mfma_f32_4x4x1 on the SAME resident
                // weights - 16 independent 4 x 4 x 1 products per instruction, block b = l >> 2 = (unit quad m >> 2) + 4 kq: lane
                // l supplies W[unit m][k(j, kq, e)] as before (its row of the quad is m & 3 = l & 3) and node (l & 3)'s operand
                // at the same k; the block keeps [4 units x 4 nodes] sums over ITS k values, the gate stage adds the four kq
                // blocks.  8 cycles per instruction instead of 32: 0.9 us per SIMD and pass instead of 3.6 for the 16 columns.
                const float* xb = Bt + (it & 1) * TR * PITCH + (lane & 3) * PITCH + hh * KW + 64 * kq;
                // (two independent accumulators and the operand quads three steps ahead: four MFMAs of 8 cycles hide neither the
                // latency of a dependent chain nor that of the LDS read in front of them)
                auto pass = [&](const float* xp) -> tf4 {
                    tf4 a0 = tf4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
                    float4 bq[4];
#pragma unroll
                    for (int j = 0; j < 3; ++j) bq[j] = *reinterpret_cast<const float4*>(xp + 256 * (j >> 4) + 4 * (j & 15));
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        if (j + 3 < KJ) bq[(j + 3) & 3] = *reinterpret_cast<const float4*>(xp + 256 * ((j + 3) >> 4) + 4 * ((j + 3) & 15));
                        __builtin_amdgcn_sched_barrier(0);   // (the read stays three steps ahead: hipcc otherwise sinks it to one)
                        const float4 b4 = bq[j & 3];
                        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][0], b4.x, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][1], b4.y, a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][2], b4.z, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[j][3], b4.w, a1, 0, 0, 0);
                    }
                    return a0 + a1;
                };
                // (a pass goes straight to its partial tile - wait for the gate stage to be done with the previous tile's first)
                unsigned spins = 0;
                while (lds_ld(&flags[1]) < (unsigned)it) if (!wt.again(spins, 2)) break;
                asm volatile("" ::: "memory");
                {
                    const tf4 r = pass(xb);
                    *reinterpret_cast<float4*>(red + ((hh * 3 + g) * 64 + lane) * 4) = make_float4(r[0], r[1], r[2], r[3]);
                }
                if (thin > 1) {
                    const tf4 r = pass(xb + 4 * PITCH);
                    *reinterpret_cast<float4*>(red + ((TNCW + hh * 3 + g) * 64 + lane) * 4) = make_float4(r[0], r[1], r[2], r[3]);
                }
            } else {
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const float4 b4 = *reinterpret_cast<const float4*>(bp + 256 * (j >> 4) + 4 * (j & 15));
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][0], b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][1], b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][2], b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][3], b4.w, acc, 0, 0, 0);
            }
            }
#ifdef T_STAMPS
            asm volatile("s_nop 0" : "+v"(acc));   // the stamp below waits for the products
#endif
            T_CLK(c1);
            unsigned spins = 0;
            while (lds_ld(&flags[1]) < (unsigned)it) if (!wt.again(spins, 2)) break;   // the previous tile's partials are read
            asm volatile("" ::: "memory");
            T_CLK(c2);
            if (thin == 0) *reinterpret_cast<float4*>(red + ((hh * 3 + g) * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            __syncthreads();
            if (!pipelined) __syncthreads();   // (the loaders: gates of this tile, operand rows of the next)
            T_CLK(c3);
            T_ACC(c_mfma, c0, c1); T_ACC(c_red, c1, c2); T_ACC(c_bar, c2, c3);
#ifdef T_STAMPS
            if (S.dbg && blockIdx.x == T_TRACE_WG && it >= T_TRACE_IT && it < T_TRACE_IT + 256 && lane == 0 && (wave == 0 || wave == 4)) {
                unsigned long long* o = S.dbg + 32 * (int64_t)(512 + it - T_TRACE_IT) + (wave == 0 ? 0 : 4);
                o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
            }
#endif
            if (!nxt.valid) break;
            cur = nxt; nxt = W.next(cur.k); ++it;
        }
#ifdef T_STAMPS
        if (S.dbg && (wave == 0 || wave == 4) && lane == 0) {
            unsigned long long* o = S.dbg + 32 * (int64_t)blockIdx.x;
            if (wave == 0) { o[5] = c_mfma; o[6] = c_red; o[7] = c_bar; } else o[14] = c_mfma;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- loader waves
    __builtin_amdgcn_s_setprio(3);   // ahead of the compute waves' back-to-back MFMAs: a loader instruction never waits behind them
    gran_t* own = S.prog + (int64_t)C.cid * (TMAXREP * TNS);
    const gran_t* low = C.low >= 0 ? S.prog + (int64_t)C.low * (TMAXREP * TNS) : nullptr;
    const __amdgpu_buffer_rsrc_t rs_own = t_rsrc(C.h_out);
    const int4* __restrict__ recs = reinterpret_cast<const int4*>(plan + L.rowrec[d]);
    const int32_t* __restrict__ col = plan + L.col[d];
    const float* __restrict__ eattr = reinterpret_cast<const float*>(plan + L.eattr[d]);
    const int nfeat = C.gain ? S.nfeat : 0;
    const float gain0 = nfeat >= 1 ? C.gain[0] : 0.f, gain1 = nfeat >= 2 ? C.gain[1] : 0.f;
    unsigned own_ok_tb = 0u, low_min = 0u;   // the polling wave's memory of what it has seen (progress only grows)
    // first tile of every replica of a cell (0x7fffffff: it has none): replica r only owes a counter to layers behind it
    int f0 = 0x7fffffff, f1 = 0x7fffffff, f2 = 0x7fffffff, f3 = 0x7fffffff;
    {
        TileWalk V;
        V.init(bl, T, R, 0, S.first_layer[d]);
        int found = 0;
        while (V.t < T && found < R) {
            const int n = V.ntl();
            if (found < 1 && n > 0) f0 = V.tb;
            if (found < 2 && n > 1) f1 = V.tb + 1;
            if (found < 3 && n > 2) f2 = V.tb + 2;
            if (found < 4 && n > 3) f3 = V.tb + 3;
            found = max(found, min(n, 4));
            V.advance();
        }
    }
#ifdef T_STAMPS
    unsigned long long n_block = 0;
#endif

    // the 16 row records of a tile: ONE LDS-DMA instruction (64 lanes x 16 bytes), no registers, no wait here
    auto dma_records = [&](const Tile& x, int ord) {
        int rec = x.slot0 + min(lane >> 2, x.nr - 1);
        rec = min(max(rec, 0), Nn - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(recs + 4 * (int64_t)rec + (lane & 3)),
                                         (__attribute__((address_space(3))) void*)(ring + (ord & 3) * (TR * 16)), 16, 0, 0);
    };
    // wait (one wave polls memory, the others its LDS flag) until the stacked layer below has published tile x ...
    auto wait_low = [&](const Tile& x, unsigned seq) {
        if (lw == TNLW - 1) {
            // the true dependence is tile x.k of the cell below; asking for T_SLACK more tiles of the same layer (which that
            // cell finishes without us) keeps this cell far enough behind that the next tiles find the answer in `low_min`,
            // refreshed without waiting, instead of paying a round trip to the other XCD each
            const unsigned need = (unsigned)min(x.k + 1 + T_SLACK, x.te);
            if (low != nullptr && !(R == 1 && low_min >= (unsigned)(x.k + 1))) {
                unsigned spins = 0;
#ifdef T_STAMPS
                ++n_block;
#endif
                for (;;) {
                    unsigned val = 0xffffffffu;
                    if (lane < TNS) {
                        const gran_t g = gran_ld(low + ((x.k - x.tb) % R) * TNS + lane);
                        val = (unsigned)(g >> 32) == S.epoch ? (unsigned)g : 0u;
                    }
                    const unsigned mn = df_wave_umin(val);
                    if (mn >= need) { low_min = mn; break; }
                    if (!wt.again(spins, 1)) break;
                }
            }
            if (lane == 0) lds_st(&flags[0], seq);
        } else {
            unsigned spins = 0;
            while (lds_ld(&flags[0]) < seq) if (!wt.again(spins, 2)) break;
        }
        asm volatile("" ::: "memory");
    };
    // ... and until every workgroup of this cell has published all tiles of the earlier layers
    auto wait_own = [&](const Tile& x, unsigned seq) {
        if (lw == TNLW - 1) {
            if ((unsigned)x.tb > own_ok_tb) {
                unsigned spins = 0;
                for (;;) {
                    unsigned val = 0xffffffffu;
                    for (int e = lane; e < R * TNS; e += 64) {
                        const int r = e / TNS;
                        if ((r == 0 ? f0 : r == 1 ? f1 : r == 2 ? f2 : f3) < x.tb) {   // replica r owns a tile before this layer
                            const gran_t g = gran_ld(own + e);
                            val = min(val, (unsigned)(g >> 32) == S.epoch ? (unsigned)g : 0u);
                        }
                    }
                    if (df_wave_umin(val) >= (unsigned)x.tb) { own_ok_tb = (unsigned)x.tb; break; }
                    if (!wt.again(spins, 1)) break;
                }
            }
            if (lane == 0) lds_st(&flags[2], seq);
        } else {
            unsigned spins = 0;
            while (lds_ld(&flags[2]) < seq) if (!wt.again(spins, 2)) break;
        }
        asm volatile("" ::: "memory");
    };

    // both waits in ONE polling loop (the first tile of a layer: the two sets of counters arrive at about the same time, and
    // one after the other they cost a round trip each)
    auto wait_both = [&](const Tile& x, unsigned seq) {
        if (lw == TNLW - 1) {
            const bool need_low = low != nullptr && !(R == 1 && low_min >= (unsigned)(x.k + 1));
            const bool need_own = (unsigned)x.tb > own_ok_tb;
            if (need_low || need_own) {
                unsigned spins = 0;
                for (;;) {
                    unsigned vl = 0xffffffffu, vo = 0xffffffffu;
                    if (need_low && lane < TNS) {
                        const gran_t g = gran_ld(low + ((x.k - x.tb) % R) * TNS + lane);
                        vl = (unsigned)(g >> 32) == S.epoch ? (unsigned)g : 0u;
                    }
                    if (need_own)
                        for (int e = lane; e < R * TNS; e += 64) {
                            const int r = e / TNS;
                            if ((r == 0 ? f0 : r == 1 ? f1 : r == 2 ? f2 : f3) < x.tb) {
                                const gran_t g = gran_ld(own + e);
                                vo = min(vo, (unsigned)(g >> 32) == S.epoch ? (unsigned)g : 0u);
                            }
                        }
                    const unsigned ml = df_wave_umin(vl), mo = df_wave_umin(vo);
                    if (ml >= (unsigned)(x.k + 1) && mo >= (unsigned)x.tb) {
                        if (need_low) low_min = ml;
                        if (need_own) own_ok_tb = (unsigned)x.tb;
                        break;
                    }
                    if (!wt.again(spins, 1)) break;
                }
            }
            if (lane == 0) { lds_st(&flags[0], seq); lds_st(&flags[2], seq); }
        } else {
            unsigned spins = 0;
            while (lds_ld(&flags[2]) < seq) if (!wt.again(spins, 2)) break;
        }
        asm volatile("" ::: "memory");
    };

    // the rows this wave builds: 4 of the tile's 16.  Everything of a row that needs no arithmetic goes by LDS-DMA
    // straight into the operand tile (the node's lower-layer row; stacked layer 0: the slice's gi0 values).
    auto issue_direct = [&](const Tile& x, int ord) {
        int lwo = lw;
        asm volatile("" : "+s"(lwo));
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (opaque per call: addresses derived from it are recomputed, not hoisted out of the tile loop and spilled)
        const int* rr = ring + (ord & 3) * (TR * 16);
#pragma unroll
        for (int q = 0; q < TNQ; ++q) {
            const int n = (TNLW - 1 - lwo) + TNLW * q;   // (the wave with the gate stage takes 5 rows, the polling wave 6)
            if (n >= x.nr) break;                        // (nothing behind the tile's last row: a thin tile issues a row or two per wave)
            const int v = rr[n * 16];
            if (HAS_IN) {
                const float* src = C.h_in + (int64_t)v * ld_h + 4 * ln;
                float* dst = Bt + ((ord & 1) * TR + n) * PITCH;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 256),
                                                 (__attribute__((address_space(3))) void*)(dst + 256), 16, 0, 0);
            } else if (ln < 12) {
                const float* src = C.gi0 + (int64_t)v * (3 * TH) + (ln >> 2) * TH + slice * TU + 4 * (ln & 3);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(gsv + ((ord % 3) * TR + n) * (3 * TU)), 16, 0, 0);
            }
        }
    };
    // first two predecessors of the four rows, one trip to memory: the first one's row by LDS-DMA into the aggregate half of
    // the operand tile (most rows have ONE predecessor: that row IS the aggregate), the second one's into registers
    struct Preds { float4 p1[TNQ][2]; int deg[TNQ], eb[TNQ]; };
    // Every load of a live row is UNCONDITIONAL: a load inside a branch ends in a register copy at the join, and the copy waits for
    // the load - four dependent round trips per tile instead of one.  A row without a first / second predecessor reads a
    // row that is certainly complete instead (the node's own lower-layer row; stacked layer 0: its gi0 row) and ignores it.
    // Rows are read with ordinary CACHED loads: a row is only ever read after the counters say it is complete, so no
    // cache of this XCD can hold an older copy, and the 32 workgroups of a cell (one XCD when the dispatch rule holds)
    // fetch a remote row ONCE into their shared L2 instead of 32 times across the fabric (sc1 loads: 40 GB per forward).
    auto issue_preds = [&](const Tile& x, int ord, Preds& P) {
        int lwo = lw;
        asm volatile("" : "+s"(lwo));
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (opaque per call: addresses derived from it are recomputed, not hoisted out of the tile loop and spilled)
        const int4* rr = reinterpret_cast<const int4*>(ring + (ord & 3) * (TR * 16));
#pragma unroll
        for (int q = 0; q < TNQ; ++q) {
            const int n = (TNLW - 1 - lwo) + TNLW * q;
            if (n >= x.nr) break;   // (the slots behind stay undefined and unused: no value to merge at the exit, so no copy that would wait)
            const bool live = x.t > 0;
            const int4 a0 = rr[n * 4], a1 = rr[n * 4 + 1];
            const int dg = a0.z - a0.y;
            P.deg[q] = live ? __builtin_amdgcn_readfirstlane(dg) : 0;
            P.eb[q] = __builtin_amdgcn_readfirstlane(a0.y);
            const float* safe = HAS_IN ? C.h_in + (int64_t)a0.x * ld_h : C.gi0 + (int64_t)a0.x * (3 * TH);
            const float* r0 = (x.t > 0 && dg > 0) ? C.h_out + (int64_t)a1.x * ld_h : safe;
            const float* r1 = (x.t > 0 && dg > 1) ? C.h_out + (int64_t)a1.y * ld_h : safe;
            float* dst = Bt + ((ord & 1) * TR + n) * PITCH + (HAS_IN ? TH : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r0 + 4 * ln),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r0 + 256 + 4 * ln),
                                             (__attribute__((address_space(3))) void*)(dst + 256), 16, 0, 0);
            P.p1[q][0] = *reinterpret_cast<const float4*>(r1 + 4 * ln);
            P.p1[q][1] = *reinterpret_cast<const float4*>(r1 + 256 + 4 * ln);
            // the 32 + 32 partial scores of the two: ONE 4-byte LDS-DMA (lanes 0-31 the first predecessor's, 32-63 the second's)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((ln < TNS ? r0 : r1) + TH + (ln & (TNS - 1))),
                                             (__attribute__((address_space(3))) void*)(psv + ((ord & 1) * TR + n) * 64), 4, 0, 0);
        }
    };
    // attention aggregate of the four rows -> aggregate half of the operand tile
    auto consume = [&](const Tile& x, int ord, Preds& P) {
        int lwo = lw;
        asm volatile("" : "+s"(lwo));
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (opaque per call: addresses derived from it are recomputed, not hoisted out of the tile loop and spilled)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA rows are in LDS
#pragma unroll
        for (int q = 0; q < TNQ; ++q) {
            const int n = (TNLW - 1 - lwo) + TNLW * q;
            if (n >= TR) continue;
            if (n >= x.nr) continue;
            float* arow = Bt + ((ord & 1) * TR + n) * PITCH + (HAS_IN ? TH : 0);
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            if (P.deg[q] >= 1) {
                a0 = *reinterpret_cast<const float4*>(arow + 4 * ln);
                a1 = *reinterpret_cast<const float4*>(arow + 256 + 4 * ln);
            }
            if (P.deg[q] >= 2) {
                const int4 ft = reinterpret_cast<const int4*>(ring + (ord & 3) * (TR * 16))[n * 4 + 2];   // edge features of the two
                const float pv = psv[((ord & 1) * TR + n) * 64 + ln];
                float s0 = fmaf(gain1, __int_as_float(ft.y), fmaf(gain0, __int_as_float(ft.x), t_wave_total(ln < TNS ? pv : 0.f)));
                float s1 = fmaf(gain1, __int_as_float(ft.w), fmaf(gain0, __int_as_float(ft.z), t_wave_total(ln < TNS ? 0.f : pv)));
                if (VID) {   // the key's vertex id scores too (one-hot `vids` of dvae/dagnn.py:130-134 folded into a table)
                    const int4 pr = reinterpret_cast<const int4*>(ring + (ord & 3) * (TR * 16))[n * 4 + 1];
                    s0 += C.vid[(unsigned)pr.x % (unsigned)S.vid_mod];
                    s1 += C.vid[(unsigned)pr.y % (unsigned)S.vid_mod];
                }
                float mx = fmaxf(s0, s1);
                const float w0 = __expf(s0 - mx), w1 = __expf(s1 - mx);
                float ssum = w0 + w1;
                t_scale(a0, w0); t_scale(a1, w0);
                t_fma(a0, w1, P.p1[q][0]); t_fma(a1, w1, P.p1[q][1]);
                // ---- further trips: T_CHUNK more predecessors at a time (ids and edge features from the plan's CSR)
                for (int e0 = 2; e0 < P.deg[q]; e0 += T_CHUNK) {
                    float4 c0[T_CHUNK], c1[T_CHUNK];
                    float cs[T_CHUNK], cf[T_CHUNK];
#pragma unroll
                    for (int c = 0; c < T_CHUNK; ++c) {
                        const bool on = e0 + c < P.deg[q];
                        const int ei = P.eb[q] + (on ? e0 + c : e0);
                        const unsigned pj = (unsigned)col[ei];
                        cf[c] = 0.f;
                        if (nfeat >= 1) cf[c] = gain0 * eattr[(int64_t)ei * nfeat];
                        if (nfeat >= 2) cf[c] = fmaf(gain1, eattr[(int64_t)ei * nfeat + 1], cf[c]);
                        c0[c] = c1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                        cs[c] = 0.f;
                        if (on) {
                            const float* rc = C.h_out + (int64_t)pj * ld_h;
                            c0[c] = *reinterpret_cast<const float4*>(rc + 4 * ln);
                            c1[c] = *reinterpret_cast<const float4*>(rc + 256 + 4 * ln);
                            if (ln < TNS) cs[c] = rc[TH + ln];
                        }
                    }
                    float sc[T_CHUNK], cm = mx;
#pragma unroll
                    for (int c = 0; c < T_CHUNK; ++c) {
                        sc[c] = t_wave_total(cs[c]) + cf[c];
                        if (VID) sc[c] += C.vid[(unsigned)col[P.eb[q] + (e0 + c < P.deg[q] ? e0 + c : e0)] % (unsigned)S.vid_mod];
                        if (e0 + c < P.deg[q]) cm = fmaxf(cm, sc[c]);
                    }
                    const float rs = __expf(mx - cm);
                    ssum *= rs; t_scale(a0, rs); t_scale(a1, rs);
#pragma unroll
                    for (int c = 0; c < T_CHUNK; ++c)
                        if (e0 + c < P.deg[q]) {
                            const float w = __expf(sc[c] - cm);
                            ssum += w; t_fma(a0, w, c0[c]); t_fma(a1, w, c1[c]);
                        }
                    mx = cm;
                }
                const float inv = 1.0f / (ssum + 1e-16f);
                t_scale(a0, inv); t_scale(a1, inv);
            }
            if (P.deg[q] != 1) {   // (one predecessor: the DMA row is the aggregate)
                *reinterpret_cast<float4*>(arow + 4 * ln) = a0;
                *reinterpret_cast<float4*>(arow + 256 + 4 * ln) = a1;
            }
            // the slice's own 16 aggregate values: the gate stage reads them up to two tiles later
            if ((ln >> 2) == (slice & 15)) {
                const bool lo = slice < 16;
                *reinterpret_cast<float4*>(asv + ((ord % 3) * TR + n) * TU + 4 * (ln & 3)) =
                    make_float4(lo ? a0.x : a1.x, lo ? a0.y : a1.y, lo ? a0.z : a1.z, lo ? a0.w : a1.w);
            }
        }
    };

    // ------------------------------------------------------------------ loader wave 0: gates of a finished tile
    auto epilogue = [&](const int x_k, const int x_nr, int ord) {
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (opaque per call: addresses derived from it are recomputed, not hoisted out of the tile loop and spilled)
        const int n = ln & 15, q = ln >> 4, slot3 = ord % 3;
        // gate by gate (r, z, then n): few values live at a time - this wave carries four rows of the next tile meanwhile
        const int thin = t_thin_passes(x_nr);
        auto psum = [&](int g, int half) -> float4 {   // the slice's sums of gate g over K half `half`: units 4 q .. 4 q + 3 of node n
            const float4* rp = reinterpret_cast<const float4*>(red) + (half * 3 + g) * 64;
            if (thin == 0) return rp[ln];
            // thin tile (4x4x1 products): pass n >> 2, the four kq blocks of unit quad q, column n & 3
            const float4* tp = rp + (n >> 2) * (TNCW * 64) + 4 * q + (n & 3);
            return t_add(t_add(tp[0], tp[16]), t_add(tp[32], tp[48]));
        };
        auto sums = [&](int g, float4& gi, float4& gh) {
            if (HAS_IN) {
                gi = psum(g, 0);
                gh = psum(g, 1);
            } else {
                gh = t_add(psum(g, 0), psum(g, 1));
                gi = *reinterpret_cast<const float4*>(gsv + (slot3 * TR + n) * (3 * TU) + g * TU + 4 * q);
            }
        };
        float4 gi, gh, rg, zg, hv;
        sums(0, gi, gh);
        {
            const float4 c = *reinterpret_cast<const float4*>(cst + 4 * q);
            rg = make_float4(t_sigm((gi.x + gh.x) + c.x), t_sigm((gi.y + gh.y) + c.y), t_sigm((gi.z + gh.z) + c.z), t_sigm((gi.w + gh.w) + c.w));
        }
        sums(1, gi, gh);
        {
            const float4 c = *reinterpret_cast<const float4*>(cst + TU + 4 * q);
            zg = make_float4(t_sigm((gi.x + gh.x) + c.x), t_sigm((gi.y + gh.y) + c.y), t_sigm((gi.z + gh.z) + c.z), t_sigm((gi.w + gh.w) + c.w));
        }
        sums(2, gi, gh);
        asm volatile("" : "+v"(gi.x), "+v"(gi.y), "+v"(gi.z), "+v"(gi.w), "+v"(gh.x), "+v"(gh.y), "+v"(gh.z), "+v"(gh.w));   // (the sums above are complete)
        if (ln == 0) lds_st(&flags[1], (unsigned)(ord + 1));   // the compute waves may overwrite the partial tiles
        float part;
        {
            const float4 ci = *reinterpret_cast<const float4*>(cst + 2 * TU + 4 * q), ch = *reinterpret_cast<const float4*>(cst + 3 * TU + 4 * q);
            const float4 a4 = *reinterpret_cast<const float4*>(asv + (slot3 * TR + n) * TU + 4 * q);
            const float4 wk = *reinterpret_cast<const float4*>(cst + 4 * TU + 4 * q);
            const float n0 = t_tanh(fmaf(rg.x, gh.x + ch.x, gi.x + ci.x)), n1 = t_tanh(fmaf(rg.y, gh.y + ch.y, gi.y + ci.y)),
                        n2 = t_tanh(fmaf(rg.z, gh.z + ch.z, gi.z + ci.z)), n3 = t_tanh(fmaf(rg.w, gh.w + ch.w, gi.w + ci.w));
            hv = make_float4(fmaf(zg.x, a4.x - n0, n0), fmaf(zg.y, a4.y - n1, n1), fmaf(zg.z, a4.z - n2, n2), fmaf(zg.w, a4.w - n3, n3));
            part = fmaf(wk.w, hv.w, fmaf(wk.z, hv.z, fmaf(wk.y, hv.y, wk.x * hv.x)));
        }
        const int v = ring[(ord & 3) * (TR * 16) + n * 16];
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (n < x_nr) {
            t_st16(rs_own, ((unsigned)v * (unsigned)ld_h + slice * TU + 4u * q) * 4u, hv);
            if (q == 0) __hip_atomic_store(C.h_out + (int64_t)v * ld_h + TH + slice, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto publish = [&](const int p_next) {   // p_next: this replica's next tile (all its tiles before that one are out)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have landed
        if (lane == 0)
            __hip_atomic_store(own + rep * TNS + slice, ((gran_t)S.epoch << 32) | (gran_t)(unsigned)p_next, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    };

    // ------------------------------------------------------------------ the walk (the same workgroup barriers as above)
    bool pend = false;
    int pend_k = 0, pend_nr = 0;
#ifdef T_STAMPS
    unsigned long long l_epi = 0, l_wait = 0, l_agg = 0, l_bar = 0, l_serial = 0, s_low = 0, s_own = 0, s_issue = 0, s_epi = 0, s_cons = 0, s_bar = 0;
    const unsigned long long l_begin = wall_clock64();
#endif
    if (lw == 1) {
        dma_records(cur, 0);
        if (nxt.valid) dma_records(nxt, 1);
    }
    __syncthreads();
    {
        Preds P;
        wait_both(cur, 1u);
        issue_direct(cur, 0);
        issue_preds(cur, 0, P);
        consume(cur, 0, P);
    }
    __syncthreads();
    for (;;) {
        const bool pipelined = nxt.valid && nxt.t == cur.t;   // the next tile does not wait for this one
        T_CLK(a0);
        Tile nn = nxt;
        {
            Preds P;
            // the polling wave looks at the lower cell's counters once per tile WITHOUT waiting for the answer (it is used
            // behind the rows' own wait below): in the steady state its memory is always ahead of the tile it needs
            const bool refresh = lw == TNLW - 1 && low != nullptr && R == 1 && pipelined;
            const gran_t fresh_g = gran_ld((refresh ? low : own) + (lane & (TNS - 1)));   // (unconditional: see issue_preds)
            if (pipelined) {
                wait_low(nxt, (unsigned)(it + 2));
                wait_own(nxt, (unsigned)(it + 2));
                issue_direct(nxt, it + 1);
                issue_preds(nxt, it + 1, P);
            }
            T_CLK(a1);
            // (off the rows' path: the tile after the next and its row records)
            if (nxt.valid) nn = W.next(nxt.k);
            if (lw == 1 && nn.valid) dma_records(nn, it + 2);
            if (pend && lw == 0) epilogue(pend_k, pend_nr, it - 1);   // while the rows are on their way
            T_CLK(a2);
            if (pipelined) consume(nxt, it + 1, P);
            if (refresh) low_min = max(low_min, df_wave_umin((unsigned)(fresh_g >> 32) == S.epoch ? (unsigned)fresh_g : 0u));
            if (pend && lw == 0) publish(cur.k);
            T_CLK(a3);
            __syncthreads();
            T_CLK(a4);
            T_ACC(l_wait, a0, a1); T_ACC(l_epi, a1, a2); T_ACC(l_agg, a2, a3); T_ACC(l_bar, a3, a4);
#ifdef T_STAMPS
            if (S.dbg && blockIdx.x == T_TRACE_WG && it >= T_TRACE_IT && it < T_TRACE_IT + 256 && lane == 0 && (lw == 0 || lw == TNLW - 1)) {
                unsigned long long* o = S.dbg + 32 * (int64_t)(512 + it - T_TRACE_IT) + (lw == 0 ? 8 : 16);
                o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = pipelined; o[6] = cur.k; o[7] = cur.t;
            }
#endif
        }
        if (pipelined) {
            pend = true; pend_k = cur.k; pend_nr = cur.nr;
        } else {
            T_CLK(b0);
#ifdef T_STAMPS
            unsigned long long b2x = b0;
#endif
            if (lw == 0) { epilogue(cur.k, cur.nr, it); publish(nxt.valid ? nxt.k : 0x7fffffff); }
            T_CLK(b1);
            if (nxt.valid) {
                Preds P;
                wait_both(nxt, (unsigned)(it + 2));
                T_CLK(w1);
                T_CLK(w2);
                issue_direct(nxt, it + 1);
                issue_preds(nxt, it + 1, P);
                T_CLK(b2);
#ifdef T_STAMPS
                b2x = b2;
#endif
                consume(nxt, it + 1, P);
                T_ACC(s_low, b1, w1); T_ACC(s_own, w1, w2); T_ACC(s_issue, w2, b2);
            }
            T_CLK(b3);
            __syncthreads();
            T_CLK(b4);
            T_ACC(s_epi, b0, b1); T_ACC(s_cons, b2x, b3); T_ACC(s_bar, b3, b4);
#ifdef T_STAMPS
            ++l_serial;
#endif
            pend = false;
        }
        if (!nxt.valid) break;
        cur = nxt; nxt = nn; ++it;
    }
#ifdef T_STAMPS
    if (S.dbg && lane == 0 && (lw == TNLW - 1 || lw == 0)) {
        unsigned long long* o = S.dbg + 32 * (int64_t)blockIdx.x;
        if (lw == TNLW - 1) {
            o[0] = wall_clock64() - l_begin; o[1] = l_wait; o[2] = l_agg; o[3] = l_bar; o[4] = it + 1; o[9] = l_serial; o[10] = l_begin;
            o[15] = n_block; o[16] = s_low; o[17] = s_own; o[18] = s_issue; o[19] = s_cons; o[20] = s_bar;
        } else { o[8] = l_epi; o[11] = l_wait; o[12] = l_agg; o[13] = l_bar; o[21] = s_epi; o[22] = s_low; o[23] = s_own; }
    }
#endif
}

template <bool VID>
__global__ void __launch_bounds__(TTHREADS, 1) tiles_kernel(const int32_t* __restrict__ plan, PlanLayout L, TArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (S.status != nullptr) {
        const int st = S.status[0];
        if (st != 0) {   // the batch violates the plan contract: walk nothing, report it like the dataflow kernels do
            if (blockIdx.x == 0 && threadIdx.x == 0)
                __hip_atomic_fetch_or(S.err, 4 | ((st & 0xff) << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    // workgroup b = (unit b % U, slice b / U), unit = (cell, replica): with 8 units the observed dispatch rule (workgroup
    // b -> XCD b % 8) keeps the 32 slices of a unit on one XCD - a speed hint only
    const int units = S.ncell * S.R;
    const int unit = blockIdx.x % units, slice = blockIdx.x / units;
    const TCell& C = S.cell[unit / S.R];
    const int rep = unit % S.R;
    if (C.wih != nullptr) tile_body<true, VID>(plan, L, S, C, slice, rep, smem);
    else tile_body<false, VID>(plan, L, S, C, slice, rep, smem);
}

int tiles_chunks(int num_cus, int ndir, int Ls, int* first, int* count, int* reps) {
    // chunk 0: stacked layer 0 alone, replicated; then as many layers at a time as the device hosts at 32 workgroups a cell
    if (ndir <= 0 || Ls <= 0 || num_cus < TNS * ndir) return 0;
    const int per = num_cus / (TNS * ndir);   // cells of one direction the device hosts
    int n = 0;
    if (Ls > 1 && Ls <= per) {   // every cell fits at once (L <= 4 with two directions on 256 CUs): ONE launch, ONE dependent chain
        int r = per / Ls; if (r > 4) r = 4;
        first[0] = 0; count[0] = Ls; reps[0] = r;
        return 1;
    }
    first[n] = 0; count[n] = 1; reps[n] = per < TMAXREP ? per : TMAXREP; if (reps[n] > T_REP0) reps[n] = T_REP0; ++n;
    for (int i = 1; i < Ls;) {
        const int c = (Ls - i) < per ? (Ls - i) : per;
        int r = per / c; if (r > 4) r = 4;
        first[n] = i; count[n] = c; reps[n] = r; ++n;
        i += c;
    }
    return n;
}

}  // namespace

extern "C" int dagnn_tiles_launches(int num_cus, int num_dirs, int num_stacked, int H, int num_edge_feats) {
    if (H != TH || num_edge_feats < 0 || num_edge_feats > 2 || num_stacked > DAGNN_MAX_STACKED || num_dirs > DAGNN_MAX_DIRS) return 0;
    int first[DAGNN_MAX_STACKED + 1], count[DAGNN_MAX_STACKED + 1], reps[DAGNN_MAX_STACKED + 1];
    return tiles_chunks(num_cus, num_dirs, num_stacked, first, count, reps);
}

extern "C" int dagnn_tiles_run(const dagnn_plan* pl, const dagnn_tiles_args* a, void* stream) {
    if (!pl || !pl->data || !a) return DAGNN_EINVAL;
    const int Ls = a->num_stacked, dir_mask = a->dir_mask & 3;
    if (a->H != TH || Ls <= 0 || Ls > DAGNN_MAX_STACKED || !dir_mask || a->ld_h < TH + TNS || (a->ld_h & 3) || a->num_cus <= 0 ||
        !a->counters || !a->err || a->epoch == 0 || pl->num_edge_feats > 2)
        return DAGNN_EINVAL;
    if ((int64_t)pl->N * a->ld_h * 4 >= (int64_t)0x7ffffff0) return DAGNN_EINVAL;   // 32-bit byte offsets into the state rows
    int ndir = 0, dirs[2];
    for (int d = 0; d < 2; ++d) if ((dir_mask >> d) & 1) dirs[ndir++] = d;
    for (int q = 0; q < ndir; ++q)
        for (int i = 0; i < Ls; ++i) {
            const dagnn_tiles_cell& c = a->cell[dirs[q]][i];
            if (!c.w_hh || !c.b_hh || !c.w_key || !c.h_out) return DAGNN_EINVAL;
            if (i == 0 ? !c.gi0 : (!c.w_ih || !c.b_ih)) return DAGNN_EINVAL;
            if (pl->num_edge_feats > 0 && !c.edge_gain) return DAGNN_EINVAL;
            if (a->vid_mod > 0 && !c.vid_bias) return DAGNN_EINVAL;
        }
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    int first[DAGNN_MAX_STACKED + 1], count[DAGNN_MAX_STACKED + 1], reps[DAGNN_MAX_STACKED + 1];
    const int nchunk = tiles_chunks(a->num_cus, ndir, Ls, first, count, reps);
    if (nchunk <= 0) return DAGNN_EINVAL;
    bool tail = true;   // only the thin tail is walked (first_layer > 0 everywhere): a layer holds one or two tiles, more replicas
    for (int q = 0; q < ndir; ++q) tail = tail && a->first_layer[dirs[q]] > 0;   // only add counters to poll (4 -> 2: cfg 5 21.56 -> 21.14 ms)
    if (tail && reps[0] > 2 && count[0] == 1) reps[0] = 2;
    const PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    const int32_t* plan = (const int32_t*)pl->data;
    const bool vid = a->vid_mod > 0;   // (its own instantiation: the plain kernel's code is the one without the table look-ups)
    const void* fn = vid ? reinterpret_cast<const void*>(tiles_kernel<true>) : reinterpret_cast<const void*>(tiles_kernel<false>);
    const size_t lds_max = TShape<true>::lds_bytes > TShape<false>::lds_bytes ? TShape<true>::lds_bytes : TShape<false>::lds_bytes;
    const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (ea != hipSuccess) return DAGNN_EHIP(ea);
    for (int ch = 0; ch < nchunk; ++ch) {
        TArgs S;
        int nc = 0;
        for (int q = 0; q < ndir; ++q)
            for (int i = first[ch]; i < first[ch] + count[ch]; ++i) {
                const int d = dirs[q];
                const dagnn_tiles_cell& c = a->cell[d][i];
                TCell& K = S.cell[nc++];
                K.whh = c.w_hh; K.wih = i > 0 ? c.w_ih : nullptr; K.bhh = c.b_hh; K.bih = i > 0 ? c.b_ih : nullptr;
                K.wkey = c.w_key; K.gain = pl->num_edge_feats > 0 ? c.edge_gain : nullptr; K.vid = vid ? c.vid_bias : nullptr;
                K.gi0 = i == 0 ? c.gi0 : nullptr;
                K.h_in = i > 0 ? a->cell[d][i - 1].h_out : nullptr;
                K.h_out = c.h_out;
                K.dir = d;
                K.cid = d * DAGNN_MAX_STACKED + i;
                K.low = i > first[ch] ? d * DAGNN_MAX_STACKED + i - 1 : -1;
            }
        S.ncell = nc; S.R = reps[ch]; S.ld_h = a->ld_h; S.nfeat = pl->num_edge_feats; S.vid_mod = vid ? a->vid_mod : 1;
        for (int d = 0; d < 2; ++d) S.first_layer[d] = a->first_layer[d] > 0 ? a->first_layer[d] : 0;
        S.epoch = a->epoch; S.spin_limit = a->spin_limit ? a->spin_limit : (1u << 22);
        S.prog = (gran_t*)a->counters; S.err = (int*)a->err; S.status = (const int32_t*)a->plan_status;
        S.dbg = a->debug_timing ? (unsigned long long*)a->debug_timing + (size_t)ch * 32 * 1024 : nullptr;
        const size_t lds = first[ch] + count[ch] > 1 ? TShape<true>::lds_bytes : TShape<false>::lds_bytes;   // (any cell with an input side)
        if (vid) hipLaunchKernelGGL(tiles_kernel<true>, dim3((unsigned)(nc * reps[ch] * TNS)), dim3(TTHREADS), lds, (hipStream_t)stream, plan, L, S);
        else hipLaunchKernelGGL(tiles_kernel<false>, dim3((unsigned)(nc * reps[ch] * TNS)), dim3(TTHREADS), lds, (hipStream_t)stream, plan, L, S);
        DAGNN_CHECK_LAUNCH();
    }
    return DAGNN_OK;
}
