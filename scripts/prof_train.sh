#!/bin/bash
# rocprofv3 kernel stats of the training step (and the forward) of the headline configuration -> gpurun_out/<name>/
name=${1:-prof_train}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -o tr -- python bench.py --steps 5 --warmup 2 --cpu-passes 0 --other-configs 0 --train-steps 10 --no-kernel-timer > gpurun_out/$name.log 2>&1
python - "$name" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:50]:
    print("%-100s %6s %12s %10s %6s" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
