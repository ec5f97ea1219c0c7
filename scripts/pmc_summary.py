#!/usr/bin/env python
"""Fold two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --cpu-passes 0 --no-kernel-timer
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --cpu-passes 0 --no-kernel-timer
  python scripts/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv 6 [out.json]

Counter unit is KiB. FETCH_SIZE is doubled before use (MI355X_MICROARCH.md, HBM section: gfx950 reports
wide coalesced reads at half size); WRITE_SIZE is used as reported.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

RECURRENCE = ("dataflow_kernel", "dataflow64_kernel", "frontier_step_kernel", "frontier_tail_kernel", "frontier_mfma_kernel", "aggregate_rows_kernel")


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def fold(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def sources_sha16():
    """Same fingerprint as bench.kernel_sources_sha16: the counters are only valid for these kernel sources."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in ("dataflow.hip", "df_common.h", "common.h"):
        with open(os.path.join(root, "dagnn_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    out_name = sys.argv[4] if len(sys.argv) > 4 else "pmc_traffic.json"
    ft, fc = fold(fetch_csv)
    wt, _ = fold(write_csv)
    # "auto": every forward of the command launches the recurrence kernel exactly once, whatever legs bench.py runs
    forwards = int(sum(fc[k] for k in fc if k.startswith(RECURRENCE))) if sys.argv[3] == "auto" else int(sys.argv[3])
    per = {}
    rec_bytes = rec_fetch = rec_write = 0.0
    for k in sorted(ft, key=lambda k: -ft[k]):
        if not k.endswith("_kernel") and "_kernel<" not in k:
            continue
        f_kib, w_kib = ft[k] / forwards, wt.get(k, 0.0) / forwards
        per[k] = {"launches_per_forward": round(fc[k] / forwards, 2), "FETCH_SIZE_KiB": round(f_kib, 1),
                  "WRITE_SIZE_KiB": round(w_kib, 1), "hbm_bytes_corrected": int((2 * f_kib + w_kib) * 1024)}
        if k.startswith(RECURRENCE):
            rec_bytes += (2 * f_kib + w_kib) * 1024
            rec_fetch += 2 * f_kib * 1024
            rec_write += w_kib * 1024
    out = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over `python bench.py "
                "--steps 2 --warmup 1 --cpu-passes 0 --no-kernel-timer --train-steps 0 --other-configs 0` (%d forwards: the headline leg, the loader-side-plan leg and the separate-calls leg; the recurrence kernel is the same in all three). Counter "
                "unit is KiB (calibrated: encode_ast_kernel WRITE_SIZE = N*H*4 bytes per launch). FETCH_SIZE is "
                "doubled before use, as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on "
                "gfx950; WRITE_SIZE is used as reported. Produced by scripts/pmc_summary.py." % forwards,
        "per_forward": per,
        "recurrence_kernels": list(RECURRENCE),
        "kernel_sources_sha16": sources_sha16(),
        "recurrence_hbm_bytes_per_forward": int(rec_bytes),
        "fetch_bytes_per_forward": int(rec_fetch),    # (FETCH_SIZE doubled)
        "write_bytes_per_forward": int(rec_write),
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", out_name), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
