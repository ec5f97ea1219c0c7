#!/bin/bash
# kernel stats of the cfg 1 and cfg 4 forwards (210 forwards each): what the GPU does in a host-bound 0.13 / 0.2 ms pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 1 4; do
  CFG=$c timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small_$c -o tr -- python scripts/small_time.py > gpurun_out/prof_small_$c.log 2>&1
  grep "us per" gpurun_out/prof_small_$c.log
  f=$(find gpurun_out/prof_small_$c -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/${R:-r04}_cfg${c}_kernel_stats.csv
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time per forward: %.1f us over %d kernels" % (tot / 210 / 1e3, len(rows)))
for r in rows[:12]:
    print("%-84s %6s %9.1f us/fwd %8s ns avg" % (r["Name"][:84], r["Calls"], int(r["TotalDurationNs"]) / 210 / 1e3, r["AverageNs"].split(".")[0]))
PY
done
