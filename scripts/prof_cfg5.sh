#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5b -o tr -- python bench.py --batch 256 --hidden 512 --layers 5 --steps 3 --warmup 1 --cpu-passes 0 --other-configs 0 --train-steps 0 --no-kernel-timer > gpurun_out/prof_cfg5b.log 2>&1
tail -c 600 gpurun_out/prof_cfg5b.log | head -c 300; echo
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_cfg5b/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-100s %6s %12s %10s %6s" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
