// dataflow_w.hip - the forward dataflow kernel at H = 320 (hidden sizes 257..320: the reference's own training width is 300,
// scripts/ogb_tok.sh:17 `--emb_dim=300`), as a translation unit of its own because the workgroup shape differs:
//   * a lane of a compute wave keeps 3 H / 8 = 120 weight registers (96 at H = 256): beyond the 168 a wave may hold at three
//     waves per SIMD, so the workgroup is 8 waves (4 compute + 2 x 2 loader waves, two rows of a block per loader wave) and a
//     wave may use 256 registers - measured at H = 256: 4 instead of 8 loader waves cost 3 % of the pass since the loaders
//     got lean (round 4);
//   * a loader lane carries five column blocks of a row (five 8-byte loads per polled row).
// Same source as dataflow.hip (kernel cells, schedule, protocol, arithmetic); only dataflow_kernel<20> and its entry point
// dagnn_dataflow_run_wide are built here (dagnn_dataflow_run forwards H > 256 to it).
#define DF_WIDE_TU 1
#define DF_NLW_V 4
#include "dataflow.hip"
