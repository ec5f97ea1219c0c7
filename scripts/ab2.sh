#!/bin/bash
# scripts/ab2.sh <variant> ...   ("base" = the shipped library; others: dagnn_amd/lib/variants/libdagnn_hip_<variant>.so from
# scripts/build_variant.sh): recurrence time of each on the headline batch (scripts/df_time.py), one line each.
# Environment: B / H / L pick the batch, ITERS the repeats.
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then unset DAGNN_AMD_LIB; else export DAGNN_AMD_LIB=$PWD/dagnn_amd/lib/variants/libdagnn_hip_$v.so; fi
  TAG=$v python scripts/df_time.py 2>&1 | grep -v "amdgpu.ids" | tail -n 2
done | tee -a gpurun_out/ab.txt
