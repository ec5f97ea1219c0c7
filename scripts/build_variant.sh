#!/bin/bash
# build a variant of the library with extra hipcc flags for ONE source: scripts/build_variant.sh <name> <source.hip> <flags...>
# -> dagnn_amd/lib/variants/libdagnn_hip_<name>.so (run with DAGNN_AMD_LIB=...); the other objects come from the normal build
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/dagnn_amd/lib/variants
obj=$root/dagnn_amd/lib/variants/$(basename $src .hip)_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c -o $obj $root/dagnn_amd/csrc/$src "$@"
objs=$(ls $root/dagnn_amd/lib/obj/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/dagnn_amd/lib/variants/libdagnn_hip_$name.so $objs $obj
echo built $root/dagnn_amd/lib/variants/libdagnn_hip_$name.so
