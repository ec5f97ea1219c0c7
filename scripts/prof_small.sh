#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/small_cfgs.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -o tr -- python scripts/small_cfgs.py > gpurun_out/prof_small.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_small/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    print("%-90s %6s %12s %10s" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"]))
PY
