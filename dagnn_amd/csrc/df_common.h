// Definitions shared by the two persistent dataflow kernels: dataflow.hip (forward recurrence) and bwd_dataflow.hip (its
// reverse-mode sweep).  Both walk the same schedule workspace (dagnn_dataflow_schedule) and use the same workgroup
// shape: 4 compute waves + 2 loader sets x 4 loader waves, 32 hidden units per slice, blocks of 4 rows.
#pragma once
#include "common.h"

namespace {

constexpr int DF_JS = 32;      // hidden units per slice
constexpr int DF_RB = 4;       // rows per block = loader waves
constexpr int DF_NCW = 4;      // compute waves
#ifndef DF_NLS_V
#define DF_NLS_V 2
#endif
constexpr int DF_NLS = DF_NLS_V;   // streams per workgroup = loader sets: set s serves group NLS * pair + s - a block costs a loader wave one trip to
                               // memory plus ~1 us of scalar work, twice what the compute waves need for it
#ifndef DF_NSLOT_V
#define DF_NSLOT_V (DF_NLS_V <= 2 ? 4 : 2)
#endif
constexpr int DF_NSLOT = DF_NSLOT_V;    // LDS ring depth (blocks the loaders may run ahead)
// (Measured and removed: two chunks per trip for rows with > 4 in-edges - the second sweep's 32 registers spilled the
// loader at 3 waves per SIMD.)
#ifndef DF_TEAMS_V
#define DF_TEAMS_V 1
#endif
constexpr int DF_TEAMS = DF_TEAMS_V;          // compute teams (of DF_NCW waves): 1 = one team takes the blocks of every stream
                                              // as they become ready; DF_NLS = a team per stream (two waves per SIMD)
static_assert(DF_TEAMS == 1 || DF_TEAMS == DF_NLS, "compute teams");
constexpr int DF_NLW = 12 - DF_NCW * DF_TEAMS;   // loader waves per workgroup (12 waves = 3 per SIMD at <= 168 VGPRs)
constexpr int DF_WPS = DF_NLW / DF_NLS;     // ... per stream
constexpr int DF_RPW = DF_RB / DF_WPS;      // rows of a block per loader wave (one after the other)
static_assert(DF_NLS == 2 || DF_NLS == 4 || DF_NLS == 8, "streams per workgroup");
constexpr int DF_THREADS = 64 * (DF_NCW * DF_TEAMS + DF_NLW);
constexpr int DF_MAX_GROUPS = 64;
constexpr int DF_MAGIC = 0x44463031;   // "DF01"

// ---------------------------------------------------------------- schedule workspace (int32 words)
struct DfLayout {
    int64_t grp_of;        // [B]     group of every graph
    int64_t gdepth;        // [G]     deepest graph of every group (layers)
    int64_t gload;         // [G]     LPT load of every group (cost units; diagnostics)
    int64_t loff;          // [G+1]   offset of group k's per-layer table in lcnt (both directions: depths are equal)
    int64_t gtab[2];       // [2G]    per group {first record, number of blocks}
    int64_t lcnt[2];       // [N+G+1] per (group, layer): rows, then the padded exclusive prefix (records)
    int64_t glbase[2];     // [N+B]   per (graph, layer) (indexed like lstart): first record inside its group
    int64_t grec[2];       // [16 * (4N + 4)] records in (group, layer, graph, node) order, group-layers padded to
                           //         whole blocks; padding records have node = -1
    int64_t total;
};

__host__ __device__ inline DfLayout df_layout_words(int64_t N, int64_t B, int G) {
    DfLayout L;
    int64_t o = 16;
    auto take = [&](int64_t n) { int64_t r = o; o = dagnn_align4(o + n); return r; };
    L.grp_of = take(B);
    L.gdepth = take(G);
    L.gload = take(G);
    L.loff = take(G + 1);
    for (int d = 0; d < 2; ++d) L.gtab[d] = take(2 * (int64_t)G);
    for (int d = 0; d < 2; ++d) L.lcnt[d] = take(N + G + 1);
    for (int d = 0; d < 2; ++d) L.glbase[d] = take(N + B);
    for (int d = 0; d < 2; ++d) L.grec[d] = take(16 * (4 * N + 4));
    L.total = o;
    return L;
}

}  // namespace
