#!/usr/bin/env python3
"""A/B of the tile heights of the 128-column layers (gemm_split_kernel<EPI, 128, 2, 4>: 128 x 128 tiles of 4 waves, two workgroups per CU):
default (whole rounds of 128-row tiles, leftover rule of rqhip_gemm_split_ex), all 128-row tiles, all 64-row tiles.  100 000 rows: 782 tiles of
128 rows on 512 workgroup slots are 1.53 rounds.  usage (GPU box): python tools/gemm_narrow_ab.py [rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ARMS = ((0, "default"), (128, "all 128-row"), (64, "all 64-row"))
print(f"{'R -> Nc':>12} {'epilogue':>8} | " + " ".join(f"{n:>12}" for _, n in ARMS) + " | bits equal")
torch.manual_seed(0)
for R, Nc in [(256, 128), (512, 128), (128, 128)]:
    x = torch.relu(torch.randn(M, R, device="cuda"))
    w = torch.randn(Nc, R, device="cuda") / R ** 0.5
    y = torch.relu(torch.randn(M, Nc, device="cuda"))
    img = ops.weight_planes(w, arith=ops.F16X2)
    rows = ops.maxima(x, cols=False)[0]
    for name, epi, aux in (("relu", _lib.EPI_RELU, None), ("store", _lib.EPI_STORE, None), ("mask", _lib.EPI_MASK, y)):
        def run(tr):
            cm = torch.zeros(Nc, dtype=torch.int32, device="cuda")
            return ops.gemm_split_ex(x, img, Nc, epilogue=epi, aux=aux, a_row_max=rows, want_row_max=True, col_max_out=cm, tile_rows=tr)[0]
        outs = [run(tr) for tr, _ in ARMS]
        ts = {tr: [] for tr, _ in ARMS}
        for rep in range(5):
            for tr, _ in ARMS:
                ts[tr].append(timeit(lambda: run(tr)))
        med = [sorted(ts[tr])[2] for tr, _ in ARMS]
        print(f"{R:5d} -> {Nc:4d} {name:>8} | " + " ".join(f"{t:12.1f}" for t in med) + f" | {[torch.equal(outs[0], o) for o in outs[1:]]}")
