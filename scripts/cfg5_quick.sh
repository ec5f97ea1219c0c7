#!/bin/bash
python bench.py --batch 256 --hidden 512 --layers 5 --steps 6 --warmup 2 --cpu-passes 0 --other-configs 0 --train-steps 0 "$@" > gpurun_out/_c5.json 2> gpurun_out/_c5.err
python -c "
import json; d=json.load(open('gpurun_out/_c5.json')); print('cfg5 ms_per_step', d['ms_per_step'], d.get('kernels_ms_per_step'))"
