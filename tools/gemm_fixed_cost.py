#!/usr/bin/env python3
"""Developer tool: what a product-GEMM launch costs beyond its rows -- K = 128 (8 stages), ReLU epilogue, by row count (16 / 128 / 512
workgroups of one tile each) and by which maxima are emitted.  What pointed at the tile dispenser (profiles/r06_step_ab.txt).
Usage (GPU box): python tools/gemm_fixed_cost.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, ROOT + "/rq-vae-recommender_amd"]
from rqhip import _lib, ops
def timed(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (2048, 16384, 65536):
  for Nc in (256, 768):
    K = 128
    a = torch.randn(M, K, device="cuda"); w = torch.randn(Nc, K, device="cuda") / K ** 0.5
    img = ops.weight_images([(w, False)])[0]; rm = ops.maxima(a, cols=False)[0]
    col = torch.zeros(Nc, dtype=torch.int32, device="cuda")
    t_both = timed(lambda: ops.gemm_split_ex(a, img, Nc, epilogue=1, a_row_max=rm, want_row_max=True, col_max_out=col))
    t_rows = timed(lambda: ops.gemm_split_ex(a, img, Nc, epilogue=1, a_row_max=rm, want_row_max=True))
    t_none = timed(lambda: ops.gemm_split_ex(a, img, Nc, epilogue=1, a_row_max=rm))
    x = torch.empty(M, Nc, device="cuda")
    t_copy = timed(lambda: x.copy_(x))
    print(f"M {M:6d} Nc {Nc}: both maxima {t_both:6.1f} us, rows only {t_rows:6.1f}, none {t_none:6.1f}; a copy_ of C {t_copy:6.1f}", flush=True)
