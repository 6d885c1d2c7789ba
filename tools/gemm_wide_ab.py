#!/usr/bin/env python3
"""A/B of the product GEMM tile (128 x 256, 4 waves, two workgroups per CU) against the full-output-width tile of the 512-column layers
(128 x 512, 8 waves, ONE workgroup per CU: every strip of A is fetched and split once per row tile) -- VERDICT r5 item 1(a).
f16x2 arithmetic, row + column maxima emitted as in the training step; 100 000 rows; times back to back (HIP events) and bits compared.
usage (GPU box): python tools/gemm_wide_ab.py [rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ARMS = ((0, "128x256 4w x2"), (-5, "128x512 8w x1"), (-6, "128x256 8w x2"))
print(f"{'R -> Nc':>12} {'epilogue':>8} | " + " ".join(f"{n:>14}" for _, n in ARMS) + " | bits equal to the product tile (C, row max, col max)")
torch.manual_seed(0)
for R, Nc in [(768, 512), (512, 768), (256, 512), (512, 256), (512, 512)]:
    x = torch.relu(torch.randn(M, R, device="cuda"))
    w = torch.randn(Nc, R, device="cuda") / R ** 0.5
    y = torch.relu(torch.randn(M, Nc, device="cuda"))
    img = ops.weight_planes(w, arith=ops.F16X2)
    rows = ops.maxima(x, cols=False)[0]
    for name, epi, aux in (("relu", _lib.EPI_RELU, None), ("mask", _lib.EPI_MASK, y)):
        def run(tr):
            cm = torch.zeros(Nc, dtype=torch.int32, device="cuda")
            c, _, crm = ops.gemm_split_ex(x, img, Nc, epilogue=epi, aux=aux, a_row_max=rows, want_row_max=True, col_max_out=cm, tile_rows=tr)
            return c, crm, cm
        outs = [run(tr) for tr, _ in ARMS]
        ts = {tr: [] for tr, _ in ARMS}
        for rep in range(5):                      # arms alternate: clocks / power state are shared
            for tr, _ in ARMS:
                ts[tr].append(timeit(lambda: run(tr), n=10))
        med = [sorted(ts[tr])[2] for tr, _ in ARMS]
        c0, r0, m0 = outs[0]
        same = [(torch.equal(c0, c), torch.equal(r0.max(dim=0).values, r.max(dim=0).values), torch.equal(m0, m)) for c, r, m in outs[1:]]
        print(f"{R:5d} -> {Nc:4d} {name:>8} | " + " ".join(f"{t:14.1f}" for t in med) + f" | {same}")
