#!/bin/bash
# quick training-step measurement (no CPU leg, no other configs): prints ms_per_step and the training-step kernel spans
python bench.py --steps 10 --warmup 3 --cpu-passes 0 --other-configs 0 --train-steps 10 "$@" > gpurun_out/_tq.json 2> gpurun_out/_tq.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/_tq.json"))
t = d["training_step"]
print("fwd ms_per_step", d["ms_per_step"], "recurrence", d["roofline"]["recurrence_ms_per_forward"])
print("train ms_per_step", t["ms_per_step"], "median", t["ms_per_step_median"], t["kernels_ms_per_step"])
PY
