#!/usr/bin/env python3
"""Developer tool: what the leftover round of the product GEMM costs -- the launch at 100 000 rows against the launch at the largest row count
that is whole rounds of 128-row tiles (98 304 rows at two column tiles, 65 536 at one), per layer shape of a configuration-2 step.
Usage (GPU box): python tools/gemm_tail_cost.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, ROOT + "/rq-vae-recommender_amd"]
from rqhip import _lib, ops


def timed(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, Nc, epi in ((768, 512, 1), (512, 256, 1), (128, 256, 1), (256, 512, 1), (512, 768, 0), (768, 512, 3), (512, 256, 3), (128, 256, 3), (256, 512, 3)):
    nct = Nc // 256
    whole = (512 // nct) * 128 * ((100000 * nct) // (512 * 128))          # rows of the whole rounds
    out = []
    for M in (whole, 100000):
        a = torch.randn(M, K, device="cuda"); w = torch.randn(Nc, K, device="cuda") / K ** 0.5
        aux = torch.randn(M, Nc, device="cuda")
        img = ops.weight_images([(w, False)])[0]; rm = ops.maxima(a, cols=False)[0]
        col = torch.zeros(Nc, dtype=torch.int32, device="cuda")
        out.append(timed(lambda: ops.gemm_split_ex(a, img, Nc, epilogue=epi, aux=aux if epi == 3 else None, a_row_max=rm, want_row_max=True, col_max_out=col)))
    print(f"K {K:4d} Nc {Nc:4d} epi {epi}: {whole} rows {out[0]:7.1f} us ({out[0] / whole * 1e3:.3f} us / 1000 rows)   100000 rows {out[1]:7.1f} us   "
          f"the last {100000 - whole} rows ({(100000 - whole) / 1000:.1f} %) cost {out[1] - out[0]:6.1f} us = {(out[1] - out[0]) / out[1] * 100:.1f} % of the launch", flush=True)
