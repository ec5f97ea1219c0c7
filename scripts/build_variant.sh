#!/bin/bash
# Experiment build of the library: scripts/build_variant.sh <name> [SRC=<file>.hip] [-DFLAG ...]  ->  scripts/tmp/lib_<name>.so
# (one source - dataflow.hip by default - recompiled with the flags, the other objects re-used); run with DAGNN_AMD_LIB=<that path>.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=dataflow
if [[ "$1" == SRC=* ]]; then src=${1#SRC=}; src=${src%.hip}; shift; fi
mkdir -p scripts/tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c -o scripts/tmp/${src}_$name.o dagnn_amd/csrc/$src.hip "$@" -Rpass-analysis=kernel-resource-usage 2> scripts/tmp/build_$name.log || { tail -30 scripts/tmp/build_$name.log; exit 1; }
objs=$(ls dagnn_amd/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/tmp/lib_$name.so $objs scripts/tmp/${src}_$name.o
grep -A12 "_kernelILi16" scripts/tmp/build_$name.log | grep -i "VGPRs:\|Spill\|Occupancy" | head -8
echo scripts/tmp/lib_$name.so
