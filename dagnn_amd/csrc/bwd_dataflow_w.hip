// bwd_dataflow_w.hip - the reverse dataflow sweep at H = 320 (hidden sizes 257..320), the counterpart of dataflow_w.hip: a
// lane of a compute wave keeps 120 weight registers and a loader lane five column blocks of every polled / static row - at
// three waves per SIMD (168 registers) the kernel spilled 47 of them (backward_run 5.6 ms at B = 160) - so the workgroup is 8
// waves (4 compute + 2 x 2 loader waves, two rows of a block per loader wave).  Same source as bwd_dataflow.hip; only
// bwd_dataflow_kernel<20> and its entry point dagnn_bwd_dataflow_run_wide are built here.
#define BD_WIDE_TU 1
#define BD_WPS_V 2
#include "bwd_dataflow.hip"
