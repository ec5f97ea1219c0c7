// dataflow_x.hip - the forward dataflow kernel with 64 hidden units per workgroup, as a translation unit of its own: only
// dataflow_kernel<16> / <20> of this shape and their entry point dagnn_dataflow_run_x are built here (dagnn_dataflow_run forwards
// to it when dagnn_dataflow_args.slices64 is set and H is 256 or 320).  Same source as dataflow.hip - kernel cells, schedule,
// protocol, arithmetic, packed weights - and bitwise the same results; what differs is who shares a compute unit:
//   * a workgroup is a 64-unit slice of a kernel cell: 8 compute waves (8 units each, as before - waves 4..7 take the odd 32-unit
//     slice of the packed matrix), TWO per SIMD, and ONE stream of 4 loader waves; still 12 waves at <= 168 registers;
//   * a cell has H / 64 slices, so the device holds twice the workgroup sets and every set serves ONE group: the group count of
//     a pass is unchanged (the schedule and the reverse sweep are the same), a workgroup walks half the blocks;
//   * a block's rows are gathered once per 64 units instead of once per 32: half the loader work and half the granule traffic of
//     a pass; the second compute wave of a SIMD issues into the gaps of the first (one wave issues every 5-8 cycles).
#define DF_X_TU 1
#define DFF_JS_V 64
#define DFF_NLS_V 1
#define dataflow_kernel dataflow64_kernel
#include "dataflow.hip"
