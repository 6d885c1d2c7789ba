#!/usr/bin/env python3
"""Developer tool: weight-gradient launch time (wgrad_split_kernel + wgrad_reduce_kernel) by row count, for the per-launch fixed cost
(prologue, partial-block flush, reduce) against the per-row rate.  Usage (GPU box): python tools/wgrad_fixed_cost.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, ROOT + "/rq-vae-recommender_amd"]
from rqhip import ops


def timed(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, K in ((512, 768), (256, 512), (128, 256)):
    row = []
    for M in (4096, 12500, 25000, 50000, 100000, 200000):
        g = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
        gm = ops.maxima(g, rows=False)[1]; xm = ops.maxima(x, rows=False)[1]
        out = torch.empty(N, K, device="cuda")
        row.append((M, timed(lambda: ops.linear_wgrad(g, None, x, out=out, g_col_max=gm, x_col_max=xm))))
    print(f"dW[{N},{K}]: " + "  ".join(f"M={m}: {t:.1f} us" for m, t in row), flush=True)
    (m1, t1), (m2, t2) = row[-2], row[-1]
    rate = (t2 - t1) / (m2 - m1)
    print(f"    per-row rate {rate * 1e3:.3f} us / 1000 rows; extrapolated fixed cost at M=100000: {row[-2][1] - rate * m1:.1f} us")
