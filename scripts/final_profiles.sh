#!/bin/bash
# round-end artefacts: kernel stats of the forward and of the training leg, ISA metadata, the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r04}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_fwd -o tr -- python bench.py --steps 5 --warmup 2 --cpu-passes 0 --train-steps 0 --other-configs 0 > gpurun_out/${R}_fwd.log 2>&1
cp $(find gpurun_out/${R}_fwd -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_train -o tr -- python bench.py --steps 2 --warmup 1 --cpu-passes 0 --train-steps 5 --other-configs 0 > gpurun_out/${R}_train.log 2>&1
cp $(find gpurun_out/${R}_train -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_train_kernel_stats.csv
head -6 gpurun_out/${R}_kernel_stats.csv | cut -c1-150
head -8 gpurun_out/${R}_train_kernel_stats.csv | cut -c1-150
python bench.py > gpurun_out/${R}_final_bench.json 2> gpurun_out/${R}_final_bench.err
python - "$R" <<'PY'
import json, sys
d = json.load(open("gpurun_out/%s_final_bench.json" % sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["recurrence_ms_per_forward"], d["roofline"]["traffic"])
print(d["loader_side_plan"]["ms_per_step"], {k: v for k, v in d["training_step"].items() if k in ("ms_per_step", "ms_per_step_median", "kernels_ms_per_step")})
print({k: v.get("ms_per_batch", v) for k, v in d["other_configs"].items()}, d["cpu_baseline"]["value"], d["cpu_baseline"]["vectorised"]["value"])
PY
