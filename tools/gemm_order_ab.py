#!/usr/bin/env python3
"""A/B of the tile order of gemm_f16_kernel at 100 000 rows (f16x2 arithmetic): column tile fastest with one dispenser (product), row tile
fastest (tile_rows = -2), one dispenser per XCD (tile_rows = -8).  Alternating runs, HIP events, 20 launches each, 3 rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
import torch  # noqa: E402
from rqhip import _lib, ops  # noqa: E402

M = 100_000


def timeit(fn, n=20):
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


for Nc, R, epi in ((512, 768, _lib.EPI_RELU), (768, 512, _lib.EPI_STORE), (256, 512, _lib.EPI_RELU), (512, 256, _lib.EPI_RELU), (768, 512, _lib.EPI_RECON)):
    a = torch.relu(torch.randn(M, R, device="cuda"))
    w = torch.randn(Nc, R, device="cuda") / R ** 0.5
    x = torch.randn(M, Nc, device="cuda")
    img = ops.weight_planes(w, arith=ops.F16X2)
    rows = ops.maxima(a, cols=False)[0]
    kw = dict(epilogue=epi, a_row_max=rows)
    if epi == _lib.EPI_RECON:
        kw.update(aux=x, row_scale=1.0 / M)
    base = ops.gemm_split_ex(a, img, Nc, tile_rows=0, **kw)[0]
    res = {0: [], -2: [], -8: []}
    for _ in range(3):
        for tr in res:
            res[tr].append(timeit(lambda: ops.gemm_split_ex(a, img, Nc, tile_rows=tr, **kw)))
    same = all(torch.equal(base, ops.gemm_split_ex(a, img, Nc, tile_rows=tr, **kw)[0]) for tr in (-2, -8))
    print(f"{R:4d} -> {Nc:4d} epi {epi}: column-tile fastest {min(res[0]):7.1f} us | row-tile fastest {min(res[-2]):7.1f} us | per-XCD {min(res[-8]):7.1f} us | same bits {same}")
