#!/bin/bash
# quick check of the fat launches (csrc/fat.hip): golden fixtures forced onto the 64-row MFMA tiles, the cfg-5 oracle
# tests, then the cfg-5 forward time.  Output: gpurun_out/fat_quick.log
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_shape_variants or wide_deep_config or batched_weight_packing" 2>&1 | tail -15
bash scripts/cfg5_quick.sh
cat gpurun_out/_c5.err | tail -5
} > gpurun_out/fat_quick.log 2>&1
tail -30 gpurun_out/fat_quick.log
