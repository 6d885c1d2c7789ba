#!/usr/bin/env python3
"""Driver for tools/pmc_gemm.sh: ten launches of gemm_split (768 -> 512 with ReLU, 100 000 rows) and of the 512 -> 768 one."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib  # noqa: E402

if os.environ.get("GS_LIB"):     # a tools/ab_build.sh probe build instead of the product library
    _lib.load(os.path.join(ROOT, os.environ["GS_LIB"]))
from rqhip import ops  # noqa: E402

M = 100_000
for N, K, relu in ((512, 768, True), (768, 512, False)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    p = ops.weight_planes(w)
    for _ in range(10):
        ops.gemm_split(x, p, N, relu=relu)
    torch.cuda.synchronize()
