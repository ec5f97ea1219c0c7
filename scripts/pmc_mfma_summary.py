#!/usr/bin/env python
"""Fold a rocprofv3 matrix-core counter pass into profiles/pmc_mfma.json.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
      --output-format csv -d gpurun_out/pmc_mfma -o m -- python bench.py --steps 2 --warmup 1 --cpu-passes 0 \
      --no-kernel-timer --train-steps 0
  python scripts/pmc_mfma_summary.py gpurun_out/pmc_mfma/m_counter_collection.csv

SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the matrix pipe over all SIMDs (calibration: the batched input GEMM
issues 13.0 GF / 4096 flop = 3.17 M v_mfma_f32_32x32x2_f32 of 64 cycles each per launch = 203 M, the counter reads
204.5 M); GRBM_GUI_ACTIVE sums the active cycles over the 8 XCDs.  MfmaUtil = busy / (GUI_ACTIVE / 8 * 1024 SIMDs).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8


def main():
    tot, cnt, dur, seen = defaultdict(lambda: defaultdict(float)), defaultdict(int), defaultdict(float), set()
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                cnt[k] += 1
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = {}
    for k, t in tot.items():
        busy, gui = t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), t.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        per[k] = {"launches": cnt[k], "avg_us": round(dur[k] / cnt[k] / 1e3, 2),
                  "SQ_VALU_MFMA_BUSY_CYCLES": int(busy), "GRBM_GUI_ACTIVE": int(gui),
                  "SQ_INSTS_VALU_MFMA_MOPS_F32": int(t.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)),
                  "mfma_util": round(busy / (gui / XCDS * SIMDS), 4)}
    out = {"note": __doc__.split("\n\n")[2].replace("\n", " "), "kernels_using_the_matrix_cores": per,
           "kernels_without_mfma": sorted(k for k in tot if k not in per)}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_mfma.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
