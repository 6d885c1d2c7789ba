#!/usr/bin/env python3
"""A/B of the split GEMM's workgroup shapes on the product step's shapes (100 000 rows, f16x2): 8 waves / one workgroup per
CU (round 3) against 4 waves / two workgroups per CU, every epilogue, results compared bit for bit.
Usage (GPU box): python tools/gemm_wg_ab.py [rows]"""
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


g = torch.Generator().manual_seed(3)
print(f"{'R -> Nc  epilogue':>26} | " + " ".join(f"{c:>9}" for c in ("w4 staged", "direct", "direct+stagger")) + " | bits equal")
for R, Nc in ((768, 512), (512, 768), (512, 256), (256, 512), (128, 256), (256, 768)):
    a = torch.randn(M, R, generator=g).cuda()
    w = (torch.randn(Nc, R, generator=g) / R ** 0.5).cuda()
    aux = torch.randn(M, Nc, generator=g).cuda()
    img = ops.weight_images([(w, False)])[0]
    rm = ops.maxima(a, cols=False)[0]
    for epi, name in ((_lib.EPI_STORE, "store"), (_lib.EPI_RELU, "relu"), (_lib.EPI_RECON, "recon"), (_lib.EPI_MASK, "mask")):
        if epi == _lib.EPI_RECON and Nc % 256:
            continue
        outs, ts = [], []
        for tr in (4, 3, 2):
            def run(tr=tr):
                cm = torch.zeros((Nc,), dtype=torch.int32, device="cuda")
                return ops.gemm_split_ex(a, img, Nc, epilogue=epi, aux=aux if epi >= _lib.EPI_RECON else None, row_scale=1e-5,
                                         a_row_max=rm, want_row_max=True, col_max_out=cm, tile_rows=tr) + (cm,)
            c, rows, crm, cm = run()
            outs.append((c, rows, crm.max(dim=0).values, cm))
            ts.append(timeit(run))
        same = all(all((x is None and y is None) or torch.equal(x, y) for x, y in zip(outs[0], o)) for o in outs[1:])
        print(f"{R:>5} -> {Nc:<4} {name:>12} | " + " ".join(f"{t:9.1f}" for t in ts) + f" | {same}", flush=True)
