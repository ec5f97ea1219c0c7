#!/bin/bash
# scripts/abt.sh <variant> ...   ("base" = the shipped library): spans of the training pass on the headline batch, one line each
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then unset DAGNN_AMD_LIB; else export DAGNN_AMD_LIB=scripts/tmp/lib_$v.so; fi
  TAG=$v python scripts/bd_time.py 2>&1 | grep -v "amdgpu.ids" | tail -n 1
done | tee -a gpurun_out/abt.txt
