#!/bin/bash
# round-6 artefacts in one GPU call: PMC traffic / instruction / MFMA-busy counters, kernel stats (forward, training, cfg 1 / cfg 4,
# cfg 5, the h = 300 training shape), the default bench line.  Everything lands in gpurun_out/; copy what is judged into profiles/.
cd $GRAFT_REPO_ROOT
bash scripts/pmc_traffic.sh r06_pmc_traffic > gpurun_out/r06_pmc_traffic.out 2>&1
bash scripts/pmc_instructions.sh r06_pmc_instructions > gpurun_out/r06_pmc_instructions.out 2>&1
bash scripts/pmc_mfma.sh r06_pmc_mfma > gpurun_out/r06_pmc_mfma.out 2>&1
R=r06 bash scripts/prof_small.sh > gpurun_out/r06_prof_small.out 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_h300 -o tr -- env HS=300 python scripts/h300_probe.py > gpurun_out/r06_h300.log 2>&1
cp $(find gpurun_out/r06_h300 -name "*kernel_stats.csv" | head -1) gpurun_out/r06_ogb_tok_h300_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_cfg5 -o tr -- python bench.py --batch 256 --hidden 512 --layers 5 --steps 3 --warmup 1 --cpu-passes 0 --other-configs 0 --train-steps 0 --no-kernel-timer > gpurun_out/r06_cfg5.log 2>&1
cp $(find gpurun_out/r06_cfg5 -name "*kernel_stats.csv" | head -1) gpurun_out/r06_cfg5_kernel_stats.csv
bash scripts/final_profiles.sh r06 > gpurun_out/r06_final.out 2>&1
tail -n 8 gpurun_out/r06_pmc_traffic.out gpurun_out/r06_final.out 2>/dev/null | cut -c1-260
head -5 gpurun_out/r06_cfg5_kernel_stats.csv | cut -c1-160
