// frontier.hip - lock-step schedule of the recurrence: ONE launch per batch-level topological
// layer, every (direction, stacked layer) cell in the same launch.
//
// Reference path replaced: the three nested loops of ogbg-code/model/dagnn.py:144-182
//   for d in dirs: for l_idx in range(T): [edge scan :151-157] for i, cell in cells_d:
//       ps_h = AttnConv(...)[layer]  (:175-179, message :366-373);  inp = GRUCell(inp, ps_h) (:181)
//       h[d][i][layer] += inp (:182)
// re-ordered legally: launch s processes layer t = s - i of stacked layer i in both directions
// (cell (d,i) at layer t needs h[d][i] of its predecessors - layers < t, finished in earlier
// launches - and h[d][i-1] of the same node, finished in launch s-1).  T + L - 1 dependent
// launches replace D*L*T dependent micro-steps.
//
// gfx950 design.  This path is bound by its dependent chain (T = 374 layers deep at a median
// frontier of 4-7 nodes), so every launch is built to be SHORT:
//  * its own chain is two memory round trips: 64-byte row record (scalar load; node id, edge
//    range and the first four predecessors + their edge features inline) -> predecessor rows.
//    The per-slice partial attention scores travel in 16 trailing floats of each state row, so
//    the same round trip brings them;
//  * the GRU weights never sit on that chain: a workgroup owns a slice of JS hidden units
//    (3 gates x JS weight columns, K = H) of one cell for a block of <= RB frontier rows, and
//    issues the loads of its whole slice - pre-packed in exactly the order its lanes consume it,
//    every load instruction one contiguous 1 KiB - before it starts the aggregate.  A CU moves
//    64 B/clk, so a slice is sized by what one CU can pull in ~1 us: thin launches use JS=16
//    slices spread over more CUs, fat ones JS=32;
//  * K is split over the 16 lanes of a DPP row, so the K reduction is 4 v_add_dpp per value
//    instead of LDS-crossbar shuffles.
//   A  one wave per row: softmax over the in-edges, alpha-weighted float4 gather -> LDS; for
//      stacked layers > 0 also the node's own lower-layer row;
//   B  fp32 FMA GEMV on the slice, hidden- and input-side, RB rows register-blocked;
//   C  RB x JS threads: gates, h' store, partial score store.
// fp32 VALU FMA: at <= 8 rows per weight pass the fp32 MFMA has the same per-row rate.
#include "frontier_dev.h"

namespace {

// acc[r] += W-slice x operand row r for the first NRW rows of the block (NRW <= RBT is a compile-time
// bound so that blocks with few live rows do not pay for the padding rows).
template <int RBT, int KW, int NRW>
__device__ __forceinline__ void fma_rows(float4 (&acc)[RBT], const float4 (&w)[KW], const float* op, int op_ld,
                                         int koff, int n) {
#pragma unroll
    for (int q = 0; q < KW / 4; ++q) {
        if (4 * q < n) {
#pragma unroll
            for (int r = 0; r < NRW; ++r) {
                const float4 a = *reinterpret_cast<const float4*>(op + r * op_ld + koff + 4 * q);
                fma4(acc[r], a.x, w[4 * q]); fma4(acc[r], a.y, w[4 * q + 1]);
                fma4(acc[r], a.z, w[4 * q + 2]); fma4(acc[r], a.w, w[4 * q + 3]);
            }
        }
    }
}


// -----------------------------------------------------------------------------------------------
// One row block (<= RBT frontier rows of one cell) x one slice of JS hidden units.
//   JS   hidden units per slice (3*JS weight columns = 3*JS/4 float4 column groups, 4 per wave)
//   KW   k values per lane held in registers at a time.  KW = 16 keeps a whole K = 256 slice in
//        registers (issued ahead of the dependent chain in the launch-per-layer kernel, resident
//        across steps in the persistent tail kernel); KW = 4 streams it in chunks so that two
//        workgroups fit a CU and hide each other's latency (fat launches).
//   RESIDENT  the first weight chunk is already in wh/wi (persistent kernel); rows are published
//        with write-through (sc1) stores for the other workgroups of the same launch.
// -----------------------------------------------------------------------------------------------
// Threads per workgroup: 6 waves own the weight columns of a 32-unit slice; 8-row blocks get 8 waves so that
// phase A is one row per wave (a second row per wave is a second dependent memory chain).
template <int RBT> struct WgShape { static constexpr int threads = RBT > 6 ? 512 : FT; };

template <int JS, int RBT, int KW, bool RESIDENT>
__device__ __forceinline__ void process_block(const int32_t* __restrict__ plan, int64_t rowrec_off, int64_t col_off,
                                              int64_t eattr_off, const Cell& C, bool has_pred, int slot0, int nr,
                                              int sl, int H, int ld_h, int Rfeat, int vid_mod, float* smem,
                                              float4 (&wh)[KW], float4 (&wi)[KW], unsigned long long* stamp,
                                              const GranCtx& G) {
    constexpr int NCW = 3 * JS / 16;   // waves that own weight columns (4 column groups each)
    constexpr int SW = 3 * JS;         // slice width in columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool has_in = C.wih != nullptr;
    const int kpt = H >> 4;                 // k values per K-lane (H % 64 == 0)
    const int op_ld = H + 4 * 16;           // padded operand row
    float* a_s = smem;                      // [RBT][op_ld]  aggregates
    float* u_s = a_s + RBT * op_ld;         // [RBT][op_ld]  own lower-layer rows (has_in)
    float* gh_s = u_s + RBT * op_ld;        // [RBT][SW]     hidden-side pre-activations of the slice
    float* gi_s = gh_s + RBT * SW;          // [RBT][SW]     input-side
    int* v_s = reinterpret_cast<int*>(gi_s + RBT * SW);  // [RBT] node ids

    // first row record of this wave: a scalar load (own counter), issued ahead of the weight loads
    const int4* __restrict__ recs = reinterpret_cast<const int4*>(plan + rowrec_off);
    int4 rec_first = make_int4(0, 0, 0, 0);
    if (wave < nr) rec_first = recs[4 * (int64_t)(slot0 + wave)];

    // ---- weights of this slice: issued first, consumed after the aggregate (phase B)
    const int ksl = lane & 15;              // K-lane within the DPP row
    const int nchunk = (kpt + KW - 1) / KW;
    const int64_t wstride = (int64_t)NCW * 64;  // float4 per kk plane of one slice
    const bool owns_cols = wave < NCW;
    const float4* whh = C.whh + (int64_t)sl * kpt * wstride + tid;
    const float4* wih = has_in ? C.wih + (int64_t)sl * kpt * wstride + tid : nullptr;
    if (!RESIDENT) {
        const int n0k = min(KW, kpt);
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            wh[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
            wi[kk] = wh[kk];
            if (owns_cols && kk < n0k) {
                if (has_pred) wh[kk] = whh[kk * wstride];
                if (has_in) wi[kk] = wih[kk * wstride];
            }
        }
    }
    if (stamp) stamp[1] = wall_clock64();

    // ---- phase A: one wave per row (rows wave, wave+6)
    const int32_t* col = plan + col_off;
    const float* eattr = reinterpret_cast<const float*>(plan + eattr_off);
    const int R = C.gain ? Rfeat : 0;
    const int H4 = H >> 2;
    for (int r = wave; r < RBT; r += WgShape<RBT>::threads / 64) {
        float* a_row = a_s + r * op_ld;
        float* u_row = u_s + r * op_ld;
        if (r < nr) {
            const int4* rp = recs + 4 * (int64_t)(slot0 + r);  // wave-uniform address: scalar loads
            const int4 rec0 = r == wave ? rec_first : rp[0];
            if (lane == 0) v_s[r] = rec0.x;
            if (has_in) {
                if (RESIDENT) {  // produced one step ago by another workgroup of this launch: wait on its granules
                    const float4 uv = gran_row_chunk(C.g_in + (int64_t)rec0.x * (H + H / PU), lane, H4, G);
                    if (lane < H4) *reinterpret_cast<float4*>(u_row + apad(4 * lane, kpt)) = uv;
                } else {
                    const float4* ur = reinterpret_cast<const float4*>(C.h_in + (int64_t)rec0.x * ld_h);
                    for (int cc = lane; cc < H4; cc += 64)
                        *reinterpret_cast<float4*>(u_row + apad(4 * cc, kpt)) = ur[cc];
                }
            }
            if (has_pred && rec0.z > rec0.y) {
                aggregate<RESIDENT>(C, col, eattr, rec0.y, rec0.z, rp[1], rp[2], rp[3], H, ld_h, R, vid_mod, kpt, a_row,
                                    lane, G);
            } else {
                for (int cc = lane; cc < H4; cc += 64)
                    *reinterpret_cast<float4*>(a_row + apad(4 * cc, kpt)) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {  // padding rows: finite zeros so the shared FMA pass stays NaN-free
            for (int cc = lane; cc < H4; cc += 64) {
                *reinterpret_cast<float4*>(a_row + apad(4 * cc, kpt)) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has_in) *reinterpret_cast<float4*>(u_row + apad(4 * cc, kpt)) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __syncthreads();
    if (stamp) stamp[2] = wall_clock64();

    // gate inputs that do not depend on the GEMV: issue now, consume in phase C
    const int gr_ = tid / JS, gj_ = tid - gr_ * JS;   // gate thread -> (row, unit of the slice)
    const bool gate_thread = tid < RBT * JS && gr_ < nr;
    const int gv = gate_thread ? v_s[gr_] : 0;
    const int j = sl * JS + gj_;
    float pre_r = 0.f, pre_z = 0.f, pre_n = 0.f, bh_r = 0.f, bh_z = 0.f, bh_n = 0.f, wk = 0.f;
    if (gate_thread) {
        if (has_in) { pre_r = C.bih[j]; pre_z = C.bih[H + j]; pre_n = C.bih[2 * H + j]; }
        else { const float* g0 = C.gi0 + (int64_t)gv * 3 * H; pre_r = g0[j]; pre_z = g0[H + j]; pre_n = g0[2 * H + j]; }
        bh_r = C.bhh[j]; bh_z = C.bhh[H + j]; bh_n = C.bhh[2 * H + j];
        wk = C.wkey ? C.wkey[j] : 0.f;
    }

    // ---- phase B: slice GEMV, K over the 16 lanes of each DPP row
    if (owns_cols) {
        float4 acc_h[RBT], acc_i[RBT];
#pragma unroll
        for (int r = 0; r < RBT; ++r) { acc_h[r] = make_float4(0.f, 0.f, 0.f, 0.f); acc_i[r] = acc_h[r]; }
        const int kbase = ksl * kpt + 4 * ksl;  // == apad(ksl * kpt, kpt)
        // wave-uniform: blocks with <= 2 live rows (the thin tail); only in the latency-bound shapes - the
        // second code copy costs the streamed shapes registers and instruction-cache room
        const bool few = RBT == 4 && KW == 16 && nr <= 2;
        const bool one = RESIDENT && few && nr == 1;   // the persistent tail spreads thin layers: one row per block is the common case
        for (int ch = 0; ch < nchunk; ++ch) {
            const int n = min(KW, kpt - ch * KW);
            if (ch > 0) {  // never taken when RESIDENT (the host only uses it for kpt <= KW)
#pragma unroll
                for (int kk = 0; kk < KW; ++kk) {
                    if (kk < n) {
                        if (has_pred) wh[kk] = whh[(int64_t)(ch * KW + kk) * wstride];
                        if (has_in) wi[kk] = wih[(int64_t)(ch * KW + kk) * wstride];
                    }
                }
            }
            if (one) {
                if (has_pred) fma_rows<RBT, KW, 1>(acc_h, wh, a_s, op_ld, kbase + ch * KW, n);
                if (has_in) fma_rows<RBT, KW, 1>(acc_i, wi, u_s, op_ld, kbase + ch * KW, n);
            } else if (few) {
                if (has_pred) fma_rows<RBT, KW, 2>(acc_h, wh, a_s, op_ld, kbase + ch * KW, n);
                if (has_in) fma_rows<RBT, KW, 2>(acc_i, wi, u_s, op_ld, kbase + ch * KW, n);
            } else {
                if (has_pred) fma_rows<RBT, KW, RBT>(acc_h, wh, a_s, op_ld, kbase + ch * KW, n);
                if (has_in) fma_rows<RBT, KW, RBT>(acc_i, wi, u_s, op_ld, kbase + ch * KW, n);
            }
        }
        if (stamp) stamp[3] = wall_clock64();
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
            if ((few && r >= 2) || (one && r >= 1)) continue;
            if (has_pred) { acc_h[r].x = dpp_row_sum16(acc_h[r].x); acc_h[r].y = dpp_row_sum16(acc_h[r].y);
                            acc_h[r].z = dpp_row_sum16(acc_h[r].z); acc_h[r].w = dpp_row_sum16(acc_h[r].w); }
            if (has_in) { acc_i[r].x = dpp_row_sum16(acc_i[r].x); acc_i[r].y = dpp_row_sum16(acc_i[r].y);
                          acc_i[r].z = dpp_row_sum16(acc_i[r].z); acc_i[r].w = dpp_row_sum16(acc_i[r].w); }
        }
        if (ksl == 15) {  // last lane of each DPP row holds the totals of column group 4*wave + row
            const int cg = 4 * wave + (lane >> 4);
#pragma unroll
            for (int r = 0; r < RBT; ++r) {
                *reinterpret_cast<float4*>(gh_s + r * SW + 4 * cg) = acc_h[r];
                *reinterpret_cast<float4*>(gi_s + r * SW + 4 * cg) = acc_i[r];
            }
        }
    }
    __syncthreads();
    if (stamp) stamp[4] = wall_clock64();

    // ---- phase C: gates for RBT rows x JS units
    if (tid < RBT * JS) {
        float sp = 0.f, hv = 0.f;
        if (gate_thread) {
            const float gr = pre_r + (has_in ? gi_s[gr_ * SW + gj_] : 0.f);
            const float gz = pre_z + (has_in ? gi_s[gr_ * SW + JS + gj_] : 0.f);
            const float gn = pre_n + (has_in ? gi_s[gr_ * SW + 2 * JS + gj_] : 0.f);
            const float hr = gh_s[gr_ * SW + gj_] + bh_r;
            const float hz = gh_s[gr_ * SW + JS + gj_] + bh_z;
            const float hn = gh_s[gr_ * SW + 2 * JS + gj_] + bh_n;
            const float a = a_s[gr_ * op_ld + apad(j, kpt)];
            const float rg = sigm(gr + hr);
            const float zg = sigm(gz + hz);
            const float ng = tanhf(fmaf(rg, hn, gn));
            hv = fmaf(zg, a - ng, ng);  // n + z * (a - n)
            sp = wk * hv;
        }
        // partial score of every 16-unit group (16 = one DPP row): summed by the consumers
        sp = dpp_row_sum16(sp);
        if (gate_thread) {
            float* po = C.h_out + (int64_t)gv * ld_h;
            po[j] = hv;
            if ((tid & 15) == 15) po[H + (j >> 4)] = sp;
            if (C.g_out) {  // granule copy: the hand-off format of the persistent tail kernel
                gran_t* pg = C.g_out + (int64_t)gv * (H + H / PU);
                if (RESIDENT) {  // consumers run in this very launch: one write-through 8-byte store each
                    __hip_atomic_store(pg + j, gran_pack(G.epoch, hv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((tid & 15) == 15)
                        __hip_atomic_store(pg + H + (j >> 4), gran_pack(G.epoch, sp), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                } else {         // consumers run in a later launch: the launch boundary publishes
                    pg[j] = gran_pack(G.epoch, hv);
                    if ((tid & 15) == 15) pg[H + (j >> 4)] = gran_pack(G.epoch, sp);
                }
            }
        }
    }
    if (stamp) stamp[5] = wall_clock64();
}

// Pack W [3H, K] (torch layout) into MFMA B-fragment order for 32-unit slices (the fat launches, csrc/fat.hip): a slice is
// six 16-column blocks n = 2 * gate + half;
// out[((sl * 6 + n) * (K/16) + k16) * 64 + lane] (float4): element q = W[gate*H + sl*32 + half*16 + (lane & 15)][16*k16 + 4*(lane >> 4) + q],
// i.e. the B operand of the q-th of four consecutive v_mfma_f32_16x16x4_f32 over k16's sixteen k values: instruction q
// multiplies k = 16 k16 + 4 kq + q on the lanes of k quarter kq = lane >> 4, so that the matching A fragment of a lane is four
// CONSECUTIVE floats of its row (one ds_read_b128) and every load of a wave is one contiguous 1 KiB.
__device__ __forceinline__ void pack_mfma_range(const float* __restrict__ W, float4* __restrict__ out, int H, int K,
                                                int64_t total, int64_t first, int64_t stride) {
    const int k16n = K >> 4;
    for (int64_t idx = first; idx < total; idx += stride) {
        const int lane = (int)(idx & 63);
        int64_t rest = idx >> 6;
        const int k16 = (int)(rest % k16n); rest /= k16n;
        const int n = (int)(rest % 6);
        const int sl = (int)(rest / 6);
        const int64_t row = (int64_t)(n >> 1) * H + sl * 32 + (n & 1) * 16 + (lane & 15);
        const int k = 16 * k16 + 4 * (lane >> 4);
        out[idx] = *reinterpret_cast<const float4*>(W + row * K + k);
    }
}

__global__ void __launch_bounds__(256) pack_mfma_kernel(const float* __restrict__ W, float4* __restrict__ out, int H,
                                                         int K, int64_t total) {
    pack_mfma_range(W, out, H, K, total, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// ---- launch-per-layer kernel: one launch = one batch-level topological layer, all cells
template <int JS, int RBT, int KW, int MINW>
__global__ void __launch_bounds__(WgShape<RBT>::threads, MINW) frontier_step_kernel(const int32_t* __restrict__ plan, PlanLayout L, StepArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool prof = S.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long* stamp = prof ? S.dbg + 8 * (int64_t)S.step : nullptr;
    if (prof) stamp[0] = wall_clock64();
    const int NS = S.H / JS;
    const int sl = blockIdx.x % NS;
    const int gb = blockIdx.x / NS;
    int c = 0;
    while (c + 1 < S.ncell && gb >= S.blk_start[c + 1]) ++c;
    const Cell& C = S.cell[c];
    const int slot0 = C.row_base + (gb - S.blk_start[c]) * RBT;
    const int nr = min(RBT, C.row_end - slot0);
    const int d = C.dir;
    float4 wh[KW], wi[KW];
    GranCtx G;
    G.epoch = S.epoch; G.err = nullptr;
    process_block<JS, RBT, KW, false>(plan, L.rowrec[d], L.col[d], L.eattr[d], C, C.has_pred != 0, slot0, nr, sl, S.H,
                                      S.ld_h, S.R, S.vid_mod, smem, wh, wi, stamp, G);
    if (prof) stamp[6] = gridDim.x;
}

// ---- persistent tail kernel: ONE launch walks all remaining layers, as a dataflow.
// The tail of the schedule is hundreds of dependent layers with a handful of rows each; a launch
// boundary (~3 us) + kernarg fetch + weight reload + cross-die row fetch per layer is most of their
// cost, and a grid barrier would cost as much (measured: 5.3 us per layer for 64 workgroups).
// Here every workgroup owns (cell, slice, replica) for the whole tail, keeps its weight slice in
// registers, and walks its row blocks in schedule order WITHOUT any barrier: a row block starts as
// soon as the granules (tagged 8-byte copies, see above) of the rows it reads carry this pass's
// epoch.  Dependencies only point to earlier layers / the lower stacked layer, producers never wait
// on consumers and the grid is far below the CU count (all workgroups resident), so it cannot
// deadlock; spins are bounded anyway and report through err_flag.  Nothing depends on placement.
struct TailArgs {
    Cell cell[DAGNN_MAX_CELLS];
    int stacked_idx[DAGNN_MAX_CELLS];  // i of each cell (layer processed at step s is s - i)
    int ncell, nrep, H, ld_h, R, vid_mod;
    int s_begin, s_end;                // steps [s_begin, s_end)
    int use_split;                     // rows of a layer start at blsplit[t] (the deep graphs only) instead of blptr[t]
    unsigned epoch;
    int* err_flag;
    unsigned long long* dbg;
};

template <int JS, int RBT, int KW>
__global__ void __launch_bounds__(FT, 1) frontier_tail_kernel(const int32_t* __restrict__ plan, PlanLayout L, TailArgs S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NCW = 3 * JS / 16;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NS = S.H / JS;
    const int sl = blockIdx.x % NS;
    const int rep = (blockIdx.x / NS) % S.nrep;
    const int c = blockIdx.x / (NS * S.nrep);
    const Cell& C = S.cell[c];
    const int d = C.dir, si = S.stacked_idx[c];
    const int N = plan[PH_N];
    const int32_t* __restrict__ blptr = plan + L.blptr[d];
    const int T = blptr[N + 1];
    const bool has_in = C.wih != nullptr;
    const int kpt = S.H >> 4;
    const bool prof = S.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    GranCtx G;
    G.epoch = S.epoch; G.err = S.err_flag;

    // resident weights: the whole slice (kpt <= KW, checked by the host)
    float4 wh[KW], wi[KW];
    {
        const int64_t wstride = (int64_t)NCW * 64;
        const float4* whh = C.whh + (int64_t)sl * kpt * wstride + tid;
        const float4* wih = has_in ? C.wih + (int64_t)sl * kpt * wstride + tid : nullptr;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            wh[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
            wi[kk] = wh[kk];
            if (wave < NCW && kk < kpt) {
                wh[kk] = whh[kk * wstride];
                if (has_in) wi[kk] = wih[kk * wstride];
            }
        }
    }
    for (int s = S.s_begin; s < S.s_end; ++s) {
        unsigned long long* stamp = prof ? S.dbg + 8 * (int64_t)s : nullptr;
        if (prof) { stamp[0] = wall_clock64(); stamp[6] = gridDim.x; }
        const int t = s - si;
        if (t < 0 || t >= T) continue;
        const int r0 = S.use_split ? plan[L.blsplit[d] + t] : blptr[t], r1 = blptr[t + 1];
        // rows of a thin layer are spread over the replicas (blocks of ceil(rows / nrep) <= RBT rows): a block's
        // latency grows with its live rows, and the layer is as slow as its slowest replica
        const int rbs = min(max((r1 - r0 + S.nrep - 1) / S.nrep, 1), RBT);
        const int nblk = (r1 - r0 + rbs - 1) / rbs;
        for (int rb = rep; rb < nblk; rb += S.nrep) {
            const int slot0 = r0 + rb * rbs;
            process_block<JS, RBT, KW, true>(plan, L.rowrec[d], L.col[d], L.eattr[d], C, t > 0, slot0,
                                             min(rbs, r1 - slot0), sl, S.H, S.ld_h, S.R, S.vid_mod, smem, wh, wi,
                                             stamp, G);
            __syncthreads();  // LDS is reused by the next block
        }
    }
}

// Pack W [3H, K] (torch GRUCell layout: row g*H + j, K contiguous) into slice / lane order for
// slices of JS units: out[((sl * kpt + kk) * NCW + w) * 64 + lane] (float4) = the 4 columns of column
// group cg = 4w + (lane >> 4) of slice sl at k = (lane & 15) * kpt + kk; local column lc = 4cg + q
// -> gate lc / JS, unit sl*JS + lc % JS.
__device__ __forceinline__ void pack_slices_range(const float* __restrict__ W, float4* __restrict__ out, int H, int K,
                                                  int JS, int64_t total, int64_t first, int64_t stride) {
    const int kpt = K >> 4;
    const int NCW = 3 * JS / 16;
    for (int64_t idx = first; idx < total; idx += stride) {
        const int lane = (int)(idx & 63);
        int64_t rest = idx >> 6;
        const int w = (int)(rest % NCW); rest /= NCW;
        const int kk = (int)(rest % kpt);
        const int sl = (int)(rest / kpt);
        const int cg = 4 * w + (lane >> 4);
        const int k = (lane & 15) * kpt + kk;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lc = 4 * cg + q;
            const int row = (lc / JS) * H + sl * JS + (lc % JS);
            v[q] = W[(int64_t)row * K + k];
        }
        out[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ void __launch_bounds__(256) pack_slices_kernel(const float* __restrict__ W, float4* __restrict__ out, int H,
                                                           int K, int JS, int64_t total) {
    pack_slices_range(W, out, H, K, JS, total, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                      (int64_t)gridDim.x * blockDim.x);
}

// All three layouts of up to DAGNN_MAX_PACK_JOBS matrices in ONE launch (training re-packs every cell every step:
// 18 small launches otherwise).  blockIdx.y = layout (16-unit slices, 32-unit slices, MFMA fragments), blockIdx.z = job.
struct PackJobs { dagnn_pack_job j[DAGNN_MAX_PACK_JOBS]; };
__global__ void __launch_bounds__(256) pack_batch_kernel(PackJobs P) {
    const dagnn_pack_job& J = P.j[blockIdx.z];
    const int64_t total = (int64_t)3 * J.H * J.K / 4;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if (blockIdx.y == 0) { if (J.out_slices16) pack_slices_range(J.w, reinterpret_cast<float4*>(J.out_slices16), J.H, J.K, 16, total, first, stride); }
    else if (blockIdx.y == 1) { if (J.out_slices32) pack_slices_range(J.w, reinterpret_cast<float4*>(J.out_slices32), J.H, J.K, 32, total, first, stride); }
    else if (J.out_mfma) pack_mfma_range(J.w, reinterpret_cast<float4*>(J.out_mfma), J.H, J.K, total, first, stride);
}

template <int JS, int RBT, int KW, int MINW>
hipError_t launch_step(int blocks, int H, hipStream_t st, const int32_t* plan, const PlanLayout& L, const StepArgs& S) {
    const int op_ld = H + 64;
    const size_t lds = (size_t)(2 * RBT * op_ld + 2 * RBT * 3 * JS) * sizeof(float) + RBT * sizeof(int);
    hipLaunchKernelGGL((frontier_step_kernel<JS, RBT, KW, MINW>), dim3((unsigned)(blocks * (H / JS))),
                       dim3(WgShape<RBT>::threads), lds, st, plan, L, S);
    return hipGetLastError();
}

}  // namespace

extern "C" int dagnn_pack_slices(const float* w, float* out, int H, int K, int slice_units, void* stream) {
    if (!w || !out || H <= 0 || K <= 0 || (slice_units != 16 && slice_units != 32) || (H % 32) || (K % 64))
        return DAGNN_EINVAL;
    const int64_t total = (int64_t)3 * H * K / 4;  // float4 elements
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_slices_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(out), H, K, slice_units, total);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_pack_batch(const dagnn_pack_job* jobs, int num_jobs, void* stream) {
    if (!jobs || num_jobs < 0 || num_jobs > DAGNN_MAX_PACK_JOBS) return DAGNN_EINVAL;
    if (num_jobs == 0) return DAGNN_OK;
    PackJobs P;
    int64_t most = 0;
    for (int i = 0; i < num_jobs; ++i) {
        const dagnn_pack_job& J = jobs[i];
        if (!J.w || J.H <= 0 || J.K <= 0 || (J.H % 32) || (J.K % 64)) return DAGNN_EINVAL;
        P.j[i] = J;
        const int64_t total = (int64_t)3 * J.H * J.K / 4;
        most = total > most ? total : most;
    }
    int64_t blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)blocks, 3, (unsigned)num_jobs), dim3(256), 0, (hipStream_t)stream, P);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

static void fill_cell(Cell& K, const dagnn_frontier_args* a, const dagnn_plan* pl, int d, int i, int js) {
    const dagnn_frontier_cell& c = a->cell[d][i];
    K.whh = (const float4*)(js == 16 ? c.w_hh_pk16 : c.w_hh_pk32);
    K.wih = i > 0 ? (const float4*)(js == 16 ? c.w_ih_pk16 : c.w_ih_pk32) : nullptr;
    K.whh_m = (const float4*)c.w_hh_mfma;
    K.wih_m = i > 0 ? (const float4*)c.w_ih_mfma : nullptr;
    K.bhh = c.b_hh; K.bih = c.b_ih; K.wkey = c.w_key; K.sscore = c.static_score;
    K.gain = pl->num_edge_feats > 0 ? c.edge_gain : nullptr;
    K.vid = a->vid_mod > 0 ? c.vid_bias : nullptr;
    K.gi0 = i == 0 ? c.gi0 : nullptr;
    K.h_in = i > 0 ? a->cell[d][i - 1].h_out : nullptr;
    K.h_out = c.h_out;
    K.g_out = (gran_t*)c.granules;
    K.g_in = i > 0 ? (const gran_t*)a->cell[d][i - 1].granules : nullptr;
    K.a_pre = nullptr;
    K.dir = d; K.row_base = 0; K.row_end = 0; K.has_pred = 0;
}

extern "C" int dagnn_pack_mfma(const float* w, float* out, int H, int K, void* stream) {
    if (!w || !out || H <= 0 || K <= 0 || (H % 32) || (K % 16)) return DAGNN_EINVAL;
    const int64_t total = (int64_t)3 * H * K / 4;  // float4 elements
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(out), H, K, total);
    DAGNN_CHECK_LAUNCH();
    return DAGNN_OK;
}

extern "C" int dagnn_frontier_run(const dagnn_plan* pl, const dagnn_frontier_args* a, const int32_t* const* layer_ptr,
                                  const int32_t* num_layers, void* stream) {
    if (!pl || !pl->data || !a || !layer_ptr || !num_layers) return DAGNN_EINVAL;
    const int H = a->H, Ls = a->num_stacked, dir_mask = a->dir_mask & 3;
    if (H <= 0 || (H % 64) || Ls <= 0 || !dir_mask || a->ld_h < H + H / PU || (a->ld_h & 3) || a->num_cus <= 0)
        return DAGNN_EINVAL;
    int ndir = 0, dirs[2];
    for (int d = 0; d < 2; ++d) if ((dir_mask >> d) & 1) dirs[ndir++] = d;
    if (ndir * Ls > DAGNN_MAX_CELLS || Ls > DAGNN_MAX_STACKED) return DAGNN_EINVAL;
    if (pl->B == 0 || pl->N == 0) return DAGNN_OK;
    int Tmax = 0;
    for (int q = 0; q < ndir; ++q) {
        const int d = dirs[q];
        if (!layer_ptr[d] || num_layers[d] < 0) return DAGNN_EINVAL;
        if (num_layers[d] > Tmax) Tmax = num_layers[d];
        for (int i = 0; i < Ls; ++i) {
            const dagnn_frontier_cell& c = a->cell[d][i];
            if (!c.w_hh_pk16 || !c.w_hh_pk32 || !c.b_hh || (!c.w_key && !c.static_score) || !c.h_out) return DAGNN_EINVAL;
            if (i == 0 ? !c.gi0 : (!c.w_ih_pk16 || !c.w_ih_pk32 || !c.b_ih)) return DAGNN_EINVAL;
        }
    }
    PlanLayout L = dagnn_plan_layout_words(pl->N, pl->E, pl->B, pl->num_edge_feats);
    hipStream_t st = (hipStream_t)stream;
    const int32_t* plan = (const int32_t*)pl->data;
    const int nsteps = Tmax + Ls - 1;

    // split mode (side stream + per-layer split pointers): the persistent kernel walks the DEEP graphs from layer 0
    // on the side stream while the launches below handle the shallow graphs' rows [ptr[t], split[t]) - the two
    // sets of graphs share nothing, so the deepest chains no longer wait behind the fat layers
    bool split = a->side_stream != nullptr && a->debug_timing == nullptr && a->fork_event != nullptr && a->join_event != nullptr;
    for (int q = 0; q < ndir && split; ++q) split = a->layer_split[dirs[q]] != nullptr;
    auto rows_all = [&](int d, int i, int s) {
        const int t = s - i;
        return (t < 0 || t >= num_layers[d]) ? 0 : layer_ptr[d][t + 1] - layer_ptr[d][t];
    };

    // ---- where the persistent tail starts: the first step after which no cell ever has more rows
    // than its replicas cover in `tail_max_blocks` blocks of 8.  Needs the whole slice in registers
    // (H <= 256), line-aligned state rows (ld_h % 32 == 0) and a sync workspace.
    int s_tail = nsteps;
    const int tail_js = a->tail_slice_units == 16 ? 16 : 32, tail_rb = 4;
    int nrep = a->tail_replicas > 0 ? a->tail_replicas : 0;
    const int tail_wgs = ndir * Ls * (H / tail_js) * (nrep > 0 ? nrep : 1);
    bool tail_ok = nrep > 0 && a->tail_err && a->epoch != 0 && H <= 256 && tail_wgs <= a->num_cus / 2;
    for (int q = 0; q < ndir && tail_ok; ++q)
        for (int i = 0; i < Ls; ++i) tail_ok = tail_ok && a->cell[dirs[q]][i].granules != nullptr;
    if (tail_ok) {
        const int cap = tail_rb * nrep * (a->tail_max_blocks > 0 ? a->tail_max_blocks : 1);
        s_tail = 0;
        for (int s = nsteps - 1; s >= 0; --s) {
            int mx = 0;
            for (int q = 0; q < ndir; ++q)
                for (int i = 0; i < Ls; ++i) mx = mx > rows_all(dirs[q], i, s) ? mx : rows_all(dirs[q], i, s);
            if (mx > cap) { s_tail = s + 1; break; }
        }
        if (nsteps - s_tail < 8) s_tail = nsteps;  // not worth a second kernel
    }
    split = split && tail_ok && s_tail < nsteps;
    auto row_lo = [&](int d, int t) { return layer_ptr[d][t]; };
    auto row_hi = [&](int d, int t) { return split ? a->layer_split[d][t] : layer_ptr[d][t + 1]; };
    auto rows_of = [&](int d, int i, int s) {
        const int t = s - i;
        return (t < 0 || t >= num_layers[d]) ? 0 : row_hi(d, t) - row_lo(d, t);
    };
    const int s_eager = split ? nsteps : s_tail;   // split mode: every step that still has shallow rows

    // ---- the persistent dataflow kernel (launched first in split mode, after the per-layer launches otherwise)
    auto launch_tail = [&](hipStream_t ts, int s_begin) -> int {
        TailArgs T;
        int nc = 0;
        for (int q = 0; q < ndir; ++q)
            for (int i = 0; i < Ls; ++i) {
                fill_cell(T.cell[nc], a, pl, dirs[q], i, tail_js);
                T.stacked_idx[nc++] = i;
            }
        T.ncell = nc; T.nrep = nrep; T.H = H; T.ld_h = a->ld_h; T.R = pl->num_edge_feats;
        T.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
        T.s_begin = s_begin; T.s_end = nsteps;
        T.use_split = split ? 1 : 0;
        T.epoch = a->epoch;
        T.err_flag = (int*)a->tail_err;
        T.dbg = (unsigned long long*)a->debug_timing;
        const int op_ld = H + 64;
        const size_t lds = (size_t)(2 * tail_rb * op_ld + 2 * tail_rb * 3 * tail_js) * sizeof(float) + tail_rb * sizeof(int);
        if (tail_js == 16)
            hipLaunchKernelGGL((frontier_tail_kernel<16, 4, 16>), dim3((unsigned)tail_wgs), dim3(FT), lds, ts, plan, L, T);
        else
            hipLaunchKernelGGL((frontier_tail_kernel<32, 4, 16>), dim3((unsigned)tail_wgs), dim3(FT), lds, ts, plan, L, T);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? DAGNN_OK : DAGNN_EHIP(e);
    };
    DagnnForkJoin fj;   // joins and releases its events on every return path
    bool forked = false;
    if (split) {   // no shallow rows at all (small batches: every graph is "deep"): nothing to overlap, no fork
        int64_t shallow = 0, deep = 0;
        for (int q = 0; q < ndir; ++q)
            for (int t = 0; t < num_layers[dirs[q]]; ++t) {
                shallow += a->layer_split[dirs[q]][t] - layer_ptr[dirs[q]][t];
                deep += layer_ptr[dirs[q]][t + 1] - a->layer_split[dirs[q]][t];
            }
        forked = shallow > 0 && deep > 0;
        if (!forked && deep > 0) {   // only deep graphs: the persistent kernel alone, on the caller's stream
            const int rc = launch_tail(st, 0);
            if (rc != DAGNN_OK) return rc;
        }                            // only shallow graphs: the per-layer launches alone
    }
    if (forked) {
        hipStream_t side = (hipStream_t)a->side_stream;
        const hipError_t ef = fj.begin(st, side, a->fork_event, a->join_event);
        if (ef != hipSuccess) return DAGNN_EHIP(ef);
        const int rc = launch_tail(side, 0);
        fj.mark();
        if (rc != DAGNN_OK) return rc;
    }

    // ---- two chains: the directions share nothing, so with a second stream (and no persistent tail in this call) each
    // direction's launches run on a stream of their own - the launches of a chain are 1-3 rounds of workgroups long, and the
    // other chain's workgroups fill the CUs a launch leaves idle while its last round drains (cfg 5: the fat part of the
    // forward 13.6 -> see DESIGN 4f).  The side stream forks from / joins into the caller's stream on the caller's events.
    const bool dual = !split && s_tail >= nsteps && ndir == 2 && a->side_stream != nullptr && a->debug_timing == nullptr &&
                      a->fork_event != nullptr && a->join_event != nullptr;
    hipStream_t chain_st[2] = {st, st};
    int scratch_off[2] = {0, 0};
    if (dual) {
        const hipError_t ef = fj.begin(st, (hipStream_t)a->side_stream, a->fork_event, a->join_event);
        if (ef != hipSuccess) return DAGNN_EHIP(ef);
        chain_st[1] = (hipStream_t)a->side_stream;
        for (int s = 0; s < s_eager; ++s) {   // the scratch rows of chain 1 sit behind the most chain 0 ever needs at once
            int n0 = 0;
            for (int i = 0; i < Ls; ++i) n0 += rows_of(dirs[0], i, s);
            if (n0 > scratch_off[1]) scratch_off[1] = n0;
        }
    }
    StepArgs S;
    S.H = H; S.ld_h = a->ld_h; S.R = pl->num_edge_feats; S.vid_mod = a->vid_mod > 0 ? a->vid_mod : 1;
    S.dbg = (unsigned long long*)a->debug_timing;
    S.epoch = a->epoch;
    // one launch: the cells of directions dirs[q0 .. q1) at step s, on stream `cs`
    auto launch_group = [&](int s, int q0, int q1, hipStream_t cs, int scr_off) -> int {
        // geometry of this launch: thin launches use 16-unit slices (and 4-row blocks when that
        // still fits one round of workgroups), fat ones 32-unit slices, 8-row blocks, 2 per CU
        int rows_total = 0, blocks8 = 0, blocks4 = 0;
        for (int q = q0; q < q1; ++q)
            for (int i = 0; i < Ls; ++i) {
                const int n = rows_of(dirs[q], i, s);
                rows_total += n; blocks8 += (n + 7) / 8; blocks4 += (n + 3) / 4;
            }
        if (rows_total == 0) return DAGNN_OK;
        // Launch shape {slice units, rows per block}: thin launches prefetch a whole 16-unit slice per
        // workgroup (1 workgroup per CU) and must fit ONE round of CUs; otherwise 32-unit slices with
        // streamed weights: 4-row blocks (96 VGPRs: three 6-wave workgroups per CU) while they fit one
        // round of slots, else 8-row blocks (125 VGPRs, two per CU).
        int js, rb;
        if (blocks4 * (H / 16) <= a->num_cus) { js = 16; rb = 4; }
        else if (blocks4 * (H / 32) <= (a->rb4_max_wgs > 0 ? a->rb4_max_wgs : 3 * a->num_cus / 2)) { js = 32; rb = 4; }
        else if (blocks8 * (H / 16) <= a->num_cus) { js = 16; rb = 8; }
        else { js = 32; rb = 8; }
        int nc = 0, blocks = 0;
        S.blk_start[0] = 0;
        for (int q = q0; q < q1; ++q) {
            const int d = dirs[q];
            for (int i = 0; i < Ls; ++i) {
                const int n = rows_of(d, i, s);
                if (n <= 0) continue;
                Cell& K = S.cell[nc];
                fill_cell(K, a, pl, d, i, js);
                K.row_base = row_lo(d, s - i); K.row_end = K.row_base + n; K.has_pred = (s - i) > 0;
                if (split) K.g_out = nullptr;   // nothing of the shallow graphs is read through granules
                blocks += (n + rb - 1) / rb;
                S.blk_start[++nc] = blocks;
            }
        }
        S.ncell = nc;
        S.step = s;
        // fat launches: 64-row MFMA tiles, gather and soft-max fused into the A staging (csrc/fat.hip).  The threshold
        // is stated for a launch over every direction: a chain of one direction takes its share of it.
        bool mfma_ok = a->agg_scratch != nullptr && scr_off + rows_total <= a->agg_scratch_rows && a->mfma_min_rows > 0 &&
                       rows_total * ndir >= a->mfma_min_rows * (q1 - q0);
        for (int k = 0; k < nc && mfma_ok; ++k)
            mfma_ok = S.cell[k].whh_m != nullptr && (S.cell[k].wih == nullptr || S.cell[k].wih_m != nullptr);
        if (mfma_ok)
            return dagnn_fat_launch(plan, L, S.cell, nc, H, a->ld_h, pl->num_edge_feats, a->vid_mod, a->epoch,
                                    (float*)a->agg_scratch + (int64_t)scr_off * H, a->num_cus, cs);
        hipError_t e;
        if (js == 32) e = rb == 8 ? launch_step<32, 8, 4, 4>(blocks, H, cs, plan, L, S)
                                  : launch_step<32, 4, 4, 3>(blocks, H, cs, plan, L, S);
        else if (rb == 4) e = launch_step<16, 4, 16, 1>(blocks, H, cs, plan, L, S);
        else e = launch_step<16, 8, 16, 1>(blocks, H, cs, plan, L, S);
        return e == hipSuccess ? DAGNN_OK : DAGNN_EHIP(e);
    };
    for (int s = 0; s < s_eager; ++s) {
        if (dual) {
            for (int q = 0; q < 2; ++q) {
                const int rc = launch_group(s, q, q + 1, chain_st[q], scratch_off[q]);
                if (rc != DAGNN_OK) return rc;
            }
        } else {
            const int rc = launch_group(s, 0, ndir, st, 0);
            if (rc != DAGNN_OK) return rc;
        }
    }
    if (dual) fj.mark();
    if (forked) {   // join (fj's destructor): the caller's stream continues only when the deep graphs are finished too
    } else if (!split && s_tail < nsteps) {
        const int rc = launch_tail(st, s_tail);
        if (rc != DAGNN_OK) return rc;
    }
    return DAGNN_OK;
}
