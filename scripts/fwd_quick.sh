#!/bin/bash
# quick forward measurement of the headline configuration (no CPU leg, no other configs, no training leg)
python bench.py --steps 20 --warmup 5 --cpu-passes 0 --other-configs 0 --train-steps 0 "$@" > gpurun_out/_fq.json 2> gpurun_out/_fq.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/_fq.json"))
print("fwd ms_per_step", d["ms_per_step"], "recurrence", d["roofline"]["recurrence_ms_per_forward"], "frac", d["roofline"]["frac"], d.get("kernels_ms_per_step"))
PY
