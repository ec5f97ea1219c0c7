// Shared device/host helpers for libdagnn_hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dagnn_hip.h"

#define DAGNN_WAVE 64

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: done once per device and kernel (a bit per device ordinal in a
// process-wide atomic mask; a device beyond 63 sets it on every call).  Returns hipSuccess or the call's error.
#include <atomic>
static inline hipError_t dagnn_lds_attr_once(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// Fork / join of a side stream around a host-side launch loop, on two events the CALLER owns (the library creates and
// destroys nothing).  The destructor is the single exit path: whatever return statement leaves the function, the
// caller's stream is ordered behind the work already queued on the side stream (which still reads and writes buffers
// torch may recycle).
struct DagnnForkJoin {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool joined = true, marked = false;
    hipError_t begin(hipStream_t main_, hipStream_t side_, void* fork_, void* join_) {   // side waits for everything queued on main so far
        main = main_; side = side_; fork = (hipEvent_t)fork_; join = (hipEvent_t)join_;
        if (!fork || !join) return hipErrorInvalidValue;
        hipError_t e = hipEventRecord(fork, main);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, fork, 0);
        if (e == hipSuccess) joined = false;
        return e;
    }
    void mark() { if (join && !joined) { (void)hipEventRecord(join, side); marked = true; } }   // the side stream's work up to here is what main joins
    ~DagnnForkJoin() {
        if (!joined && join) {
            if (!marked) (void)hipEventRecord(join, side);
            (void)hipStreamWaitEvent(main, join, 0);
        }
    }
};

#define DAGNN_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e_ = hipGetLastError();                     \
        if (e_ != hipSuccess) return DAGNN_EHIP(e_);           \
    } while (0)

// ------------------------------------------------------------------ plan layout (int32 words)
// One allocation, offsets fixed by (N, E, B, R) so host and device agree without a header read.
struct PlanLayout {
    int64_t node_ptr, edge_ptr;      // [B+1] each
    int64_t depth[2];                // [B]
    int64_t order[2];                // [N]   node ids sorted by (graph, layer)
    int64_t lstart[2];               // [N+B] per graph: depth_g+1 absolute positions into order
    int64_t rowptr[2];               // [N+B] per graph: n_g+1 absolute offsets into col
    int64_t col[2];                  // [E]   predecessor node id per CSR slot
    int64_t eattr[2];                // [E*R] fp32 edge features in CSR order
    int64_t items;                   // [2B]  (g*2+d) sorted by depth, deepest first
    int64_t pos[2];                  // [N]   during the build: sorted position of each node; final: its rowrec slot
    int64_t cursor[2];               // [N+B] scratch: fill cursors
    int64_t eidx[2];                 // [E]   original edge id per CSR slot
    int64_t blptr[2];                // [N+2] batch-level layer offsets (counts, then scanned); [N+1] = T_d
    int64_t lbase[2];                // [N+B] per (graph, layer): first slot of its rows in rowrec
    int64_t rowrec[2];               // [16N] per batch-level slot, 64 B: {node, e_begin, e_end, graph,
                                     //        pred[0..3], edge feats of the first 4 edges (2 floats each)}
    int64_t brec[2];                 // [16N] backward pass: per slot {node, succ CSR range, first 4 successors, their
                                     //        edge ids} (written by dagnn_backward_prepare)
    int64_t blsplit[2];              // [N+2] per batch-level layer: first slot of the rows of DEEP graphs (depth > thr_d,
                                     //        header word PH_THR0 + d); inside a layer the slots of the shallow graphs
                                     //        come first.  Lets the dataflow kernel walk the deep graphs from layer 0
                                     //        while the per-layer launches handle the shallow ones.
    int64_t total;                   // words
};

__host__ __device__ inline int64_t dagnn_align4(int64_t w) { return (w + 3) & ~int64_t(3); }

__host__ __device__ inline PlanLayout dagnn_plan_layout_words(int64_t N, int64_t E, int64_t B, int R) {
    PlanLayout L;
    int64_t o = 16;  // header words
    auto take = [&](int64_t n) { int64_t r = o; o = dagnn_align4(o + n); return r; };
    L.node_ptr = take(B + 1);
    L.edge_ptr = take(B + 1);
    for (int d = 0; d < 2; ++d) L.depth[d] = take(B);
    for (int d = 0; d < 2; ++d) L.order[d] = take(N);
    for (int d = 0; d < 2; ++d) L.lstart[d] = take(N + B);
    for (int d = 0; d < 2; ++d) L.rowptr[d] = take(N + B);
    for (int d = 0; d < 2; ++d) L.col[d] = take(E);
    for (int d = 0; d < 2; ++d) L.eattr[d] = take(E * (int64_t)(R > 0 ? R : 0));
    L.items = take(2 * B);
    for (int d = 0; d < 2; ++d) L.pos[d] = take(N);
    for (int d = 0; d < 2; ++d) L.cursor[d] = take(N + B);
    for (int d = 0; d < 2; ++d) L.eidx[d] = take(E);
    for (int d = 0; d < 2; ++d) L.blptr[d] = take(N + 2);
    for (int d = 0; d < 2; ++d) L.blsplit[d] = take(N + 2);   // right behind blptr: one device->host read gets both
    for (int d = 0; d < 2; ++d) L.lbase[d] = take(N + B);
    for (int d = 0; d < 2; ++d) L.rowrec[d] = take(16 * N);
    for (int d = 0; d < 2; ++d) L.brec[d] = take(16 * N);
    L.total = o;
    return L;
}

// Plan header words
enum { PH_N = 0, PH_E = 1, PH_B = 2, PH_R = 3, PH_MAGIC = 4, PH_THR0 = 5, PH_THR1 = 6 };
// A batch-level layer with more rows than this is "fat"; thr_d = 1 + the last fat layer of direction d
// (0 if none): graphs deeper than thr_d are the DEEP graphs of direction d.  About one pass of the persistent
// kernel's replicas (4 replicas x 4-row blocks = 16 rows); measured on the headline batch (recurrence per forward,
// MFMA tiles from 300 rows: 10: 3.30 ms, 12: 3.09, 13: 3.02, 14: 3.02, 15: 3.05, 16: 3.11, 24: 3.99, 32: 4.29).
#define DAGNN_PLAN_THIN_ROWS 14
#define DAGNN_PLAN_MAGIC 0x44414731  // "DAG1"

// ------------------------------------------------------------------ small batches (small.hip)
// One-workgroup builds of plan and dataflow schedule, word for word the arrays of the general kernels; taken by
// dagnn_plan_build / dagnn_dataflow_schedule when dagnn_plan_is_small(N, E, B) and the plan's flags allow it.
extern "C" int dagnn_plan_is_small(int64_t N, int64_t E, int64_t B);
int dagnn_plan_build_small(const dagnn_plan* pl, const int64_t* edge_index, const int64_t* layer_fwd, const int64_t* layer_bwd,
                           const int64_t* batch, const float* edge_attr, int32_t* status, hipStream_t stream);
int dagnn_dataflow_schedule_small(const dagnn_plan* pl, int32_t* ws, int groups, int cost_layer, int cost_row,
                                  const int32_t* status, hipStream_t stream);

// ------------------------------------------------------------------ wave-level reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- granules: the hand-off format inside the persistent tail kernel ---------------------------
// Every state float (and partial score) also exists as an 8-byte {tag = epoch of this forward pass,
// value} granule written by ONE aligned 8-byte store.  A consumer re-reads the granules it needs
// with relaxed agent-scope loads (they bypass the non-coherent caches) until every tag matches:
// the data is its own flag, there is no barrier, no fence and no dependence on placement
// (cdna_hip_programming.md Guideline 16, form R2).  Old tags never equal the current epoch because
// the buffers are zero-initialised once and the epoch only grows.
typedef unsigned long long gran_t;
static __device__ __forceinline__ gran_t gran_pack(unsigned epoch, float v) {
    return ((gran_t)epoch << 32) | (gran_t)__float_as_uint(v);
}
static __device__ __forceinline__ gran_t gran_ld(const gran_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct GranCtx { unsigned epoch; int* err; };

// bounded spin helper: returns false (and raises the error flag) when the budget is exhausted
static __device__ __forceinline__ bool gran_retry(unsigned& spins, const GranCtx& G) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 22)) {
        __hip_atomic_store(G.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    return true;
}

