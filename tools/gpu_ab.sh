#!/bin/bash
# One gpurun call of a development round: A/B timings of gemm_split builds (tools/_ab).
O=gpurun_out/ab; mkdir -p $O; R=$GRAFT_REPO_ROOT
for lib in "" tools/_ab/librqhip_w4.so; do
  timeout 200 python tools/bench_gemm_split.py 100000 $lib 2>&1 | grep -v "Warn\|amdgpu.ids" | head -6
done | tee $O/bench_gemm.log
timeout 100 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/bench_gemm.log
# correctness of the 4-wave build (error vs fp64 against the library's)
import os, sys, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "rq-vae-recommender_amd")]
from rqhip import _lib, ops
_lib.load(os.path.abspath("tools/_ab/librqhip_w4.so"))
g = torch.Generator().manual_seed(1)
for (Nc, R, relu) in [(512, 768, False), (768, 512, True), (256, 512, False)]:
    a = torch.randn(100_000, R, generator=g).cuda(); w = (torch.randn(Nc, R, generator=g) / R ** 0.5).cuda()
    c = ops.gemm_split(a, ops.weight_planes(w), Nc, relu=relu)
    ref = a.double() @ w.double().t(); lib = a @ w.t()
    if relu: ref, lib = torch.relu(ref), torch.relu(lib)
    sc = ref.abs().max().item()
    print("w4", Nc, R, relu, "err", (c.double() - ref).abs().max().item() / sc, "lib", (lib.double() - ref).abs().max().item() / sc)
PY
