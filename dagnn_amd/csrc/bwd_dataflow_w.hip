// bwd_dataflow_w.hip - the reverse dataflow sweep at H = 320 (hidden sizes 257..320), the counterpart of dataflow_w.hip: a
// lane of a compute wave keeps 120 weight registers and a loader lane five column blocks of every polled / static row.
// Until the loader rows were taken off scratch memory and spilled invariants (round 4) the kernel did not fit
// three waves per SIMD - 47 spilled VGPRs at 168, backward_run 5.6 ms at B = 160 - and ran as 8 waves (4 compute + 2 x 2
// loader waves, two rows of a block per loader wave: 4.0 ms, 2.94 ms after that work).  It now needs 157 registers, so it
// has the 12-wave shape of H <= 256 (BD_WPS_V = 4, one row per loader wave: 2.53 ms; BD_WPS_V = 2 still builds the 8-wave
// shape, with its own choice of who stores a row's outputs).  Same source as bwd_dataflow.hip; only bwd_dataflow_kernel<20>
// and its entry point dagnn_bwd_dataflow_run_wide are built here.
#define BD_WIDE_TU 1
#ifndef BD_WPS_V
#define BD_WPS_V 4
#endif
#include "bwd_dataflow.hip"
