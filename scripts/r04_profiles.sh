#!/bin/bash
# round-4 artefacts in one GPU call: kernel stats (forward, training, cfg 1 / cfg 4, the h = 300 training shape), PMC traffic and
# instruction counters of the dominant kernel, the default bench line.  Everything lands in gpurun_out/; copy what is judged
# into profiles/.
cd $GRAFT_REPO_ROOT
bash scripts/pmc_traffic.sh r04_pmc_traffic > gpurun_out/r04_pmc_traffic.out 2>&1
bash scripts/pmc_instructions.sh r04_pmc_instructions > gpurun_out/r04_pmc_instructions.out 2>&1
bash scripts/prof_small.sh > gpurun_out/r04_prof_small.out 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_h300 -o tr -- env HS=300 python scripts/h300_probe.py > gpurun_out/r04_h300.log 2>&1
cp $(find gpurun_out/r04_h300 -name "*kernel_stats.csv" | head -1) gpurun_out/r04_ogb_tok_h300_kernel_stats.csv
bash scripts/final_profiles.sh r04 > gpurun_out/r04_final.out 2>&1
tail -n 12 gpurun_out/r04_pmc_traffic.out gpurun_out/r04_final.out 2>/dev/null | cut -c1-220
