#!/bin/bash
# GPU call 1 (round 4): hop decomposition of the forward dataflow kernel + baseline timings
mkdir -p gpurun_out
export DAGNN_AMD_LIB=scripts/tmp/lib_stamps.so
DAGNN_AMD_DEBUG_WG=0 DIR=0 python scripts/df_hops.py > gpurun_out/hops_d0.txt 2>&1
DAGNN_AMD_DEBUG_WG=128 DIR=1 python scripts/df_hops.py > gpurun_out/hops_d1.txt 2>&1
unset DAGNN_AMD_LIB
python scripts/df_probe.py > gpurun_out/probe_base.txt 2>&1
bash scripts/fwd_quick.sh > gpurun_out/fq_base.txt 2>&1
tail -30 gpurun_out/hops_d0.txt gpurun_out/hops_d1.txt gpurun_out/probe_base.txt gpurun_out/fq_base.txt
