// dataflow_w.hip - the forward dataflow kernel at H = 320 (hidden sizes 257..320: the reference's own training width is 300,
// scripts/ogb_tok.sh:17 `--emb_dim=300`), as a translation unit of its own: only dataflow_kernel<20> and its entry point
// dagnn_dataflow_run_wide are built here (dagnn_dataflow_run forwards H > 256 to it).  Same source as dataflow.hip (kernel
// cells, schedule, protocol, arithmetic); what differs at this width:
//   * a lane of a compute wave keeps 3 H / 8 = 120 weight registers (96 at H = 256).  Until round 6 that put the compute path
//     beyond the 168 registers a wave may hold at three waves per SIMD, and the workgroup was 8 waves (4 compute + 2 x 2 loader
//     waves, two rows of a block per loader wave, one after the other).  The lean compute loop of round 6 fits 168 once the
//     operand reads are left to stream two registers ahead of the products (df_compute: no early-read barrier above H = 256),
//     so the kernel has the 12-wave shape of H <= 256 again - one row per loader wave: recurrence 2.18 -> 1.75 ms on the
//     emb_dim-300 bench batch (B = 160), bitwise the same results.  -DDF_NLW_V=4 still builds the 8-wave shape;
//   * a loader lane carries five column blocks of a row (five 8-byte loads per polled row).
#define DF_WIDE_TU 1
#ifndef DF_NLW_V
#define DF_NLW_V 8
#endif
#include "dataflow.hip"
