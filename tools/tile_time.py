#!/usr/bin/env python3
"""Developer tool: the time of ONE round of the product GEMM's tiles by tile height (128 / 64 / 32 rows; tile_rows 0 / -10 / -11) --
M = 512 slots x height rows, one column tile (Nc = 256) and two (Nc = 512) -- for the launch plan's leftover model (csrc/gemm_split.hip).
Usage (GPU box): python tools/tile_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib, ops  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'K':>5} {'Nc':>5} {'epi':>4} | one round of 128-row tiles | 64-row | 32-row   (us; rows = slots x height / column tiles)")
for Nc in (256, 512):
    for K in (128, 256, 512, 768):
        for epi in (_lib.EPI_RELU, _lib.EPI_MASK):
            out = []
            for h, code in ((128, 0), (64, -10), (32, -11)):
                M = 512 * h // (Nc // 256)
                a = torch.randn(M, K, device="cuda")
                w = torch.randn(Nc, K, device="cuda") / K ** 0.5
                aux = torch.randn(M, Nc, device="cuda")
                img = ops.weight_images([(w, False)])[0]
                rm = ops.maxima(a, cols=False)[0]
                col = torch.zeros(Nc, dtype=torch.int32, device="cuda")
                out.append(timed(lambda: ops.gemm_split_ex(a, img, Nc, epilogue=epi, aux=aux if epi == _lib.EPI_MASK else None, a_row_max=rm,
                                                           want_row_max=True, col_max_out=col, tile_rows=code)))
            print(f"{K:5d} {Nc:5d} {epi:4d} | {out[0]:8.1f} | {out[1]:8.1f} ({out[1] / out[0]:.2f}) | {out[2]:8.1f} ({out[2] / out[0]:.2f})", flush=True)
