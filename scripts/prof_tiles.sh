#!/bin/bash
# kernel stats of the cfg-5 forward: default path (DAGNN_AMD_TILES=1: per-layer launches for the wide layers + tile kernel for the
# thin tail), the tile kernel alone (=2) and the per-layer launches alone (=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 1 2 0; do
  DAGNN_AMD_TILES=$m timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tiles_$m -o tr -- python scripts/tiles_time.py > gpurun_out/prof_tiles_$m.log 2>&1
  grep "ms per" gpurun_out/prof_tiles_$m.log
  f=$(find gpurun_out/prof_tiles_$m -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/r03_cfg5_tiles${m}_kernel_stats.csv
  python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-90s %6s %12s %10s %6s" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
done
