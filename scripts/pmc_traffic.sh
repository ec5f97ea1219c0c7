#!/bin/bash
# HBM traffic of the forward pass from the PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no
# trace domains next to them), folded by scripts/pmc_summary.py into profiles/<name>.json
name=${1:-r04_pmc_traffic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 2 --warmup 1 --cpu-passes 0 --no-kernel-timer --train-steps 0 --other-configs 0"
rm -rf gpurun_out/${name}_fetch gpurun_out/${name}_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${name}_fetch -o f -- $CMD > gpurun_out/${name}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${name}_write -o w -- $CMD > gpurun_out/${name}_write.log 2>&1
F=$(find gpurun_out/${name}_fetch -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/${name}_write -name "*counter_collection.csv" | head -1)
python scripts/pmc_summary.py $F $W auto $name.json | head -12
cp profiles/$name.json gpurun_out/$name.json
