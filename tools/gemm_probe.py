#!/usr/bin/env python3
"""Times gemm_split (768 -> 512 with ReLU and 512 -> 768 plain, 100 000 rows) in the library build given as argv[1]
(default: the product build); one process per build (tools/ab_build.sh makes the phase-skipping variants)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.load(os.path.join(ROOT, sys.argv[1]))
from rqhip import ops  # noqa: E402

M = 100_000
res = []
for N, K, relu in ((512, 768, True), (768, 512, False), (256, 512, True)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    p = ops.weight_planes(w)
    for _ in range(5):
        ops.gemm_split(x, p, N, relu=relu)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(30):
        ops.gemm_split(x, p, N, relu=relu)
    b.record()
    torch.cuda.synchronize()
    res.append(f"{K}->{N}: {a.elapsed_time(b) / 30 * 1e3:7.1f} us")
print(f"{(sys.argv[1] if len(sys.argv) > 1 else 'product'):36s}", "  ".join(res), flush=True)
