#!/usr/bin/env python3
"""Developer probe: the forward kernel on tiny batches (the reference's own training shapes), per scan form and mode.
Usage (GPU box): python tools/fwd_small_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator().manual_seed(0)
for B, D, K, L in ((64, 64, 256, 3), (640, 64, 256, 3), (64, 32, 256, 3), (640, 32, 256, 3), (4096, 64, 256, 3)):
    for data in ("spread", "tiny-codes"):
        x = (torch.randn(B, D, generator=g) * 0.5).cuda()
        if data == "spread":
            cb = torch.stack([torch.randn(K, D, generator=g).cuda() * 0.5 / (l + 1) for l in range(L)]).contiguous()
        else:   # tools/bench_small_batch.py's codebooks: small against the rows
            cb = torch.stack([torch.randn(K, D, generator=g).cuda() * 0.05 / (l + 1) for l in range(L)]).contiguous()
        row = []
        for mode, mname in ((ops.MODE_EVAL, "eval"), (ops.MODE_STE, "ste"), (ops.MODE_ROTATION, "rot")):
            for scan in ("auto", "fp32"):
                t = timeit(lambda: ops.rq_forward(x, cb, mode, 0.25, want_embs=False, want_residuals=False, scan=scan))
                row.append(f"{mname}/{scan} {t:7.1f}")
        print(f"B={B:5d} D={D} {data:10s}: " + "  ".join(row), flush=True)
