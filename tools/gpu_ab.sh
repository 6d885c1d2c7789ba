#!/bin/bash
# One gpurun call of a development round: gemm_split A/B timing (product vs tools/_ab builds).
O=gpurun_out/ab; mkdir -p $O
for lib in "" tools/_ab/librqhip_ring3.so; do
  timeout 100 python tools/gemm_probe.py $lib 2>&1 | grep -v "amdgpu.ids"
done | tee $O/gemm_probe.log
