#!/usr/bin/env python3
"""Round-5 diagnostics run on the GPU box (prints, does not assert): the Nc = 2048 data gradient in both dispenser modes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rqhip import ops  # noqa: E402

g = torch.Generator().manual_seed(5)
for M, Nc, R in ((4099, 2048, 256), (4099, 1536, 256), (4099, 2048, 512), (100000, 2048, 256), (4099, 1024, 256), (4099, 1280, 256)):
    a = torch.randn(M, R, generator=g).cuda()
    w = (torch.randn(R, Nc, generator=g) / R ** 0.5).cuda()          # B = w^T: [Nc, R]
    img = ops.weight_planes(w, transpose=True, arith=ops.F16X2)
    rows = ops.maxima(a, cols=False)[0]
    ref = a.double() @ w.double()
    for tr in (0, -8):
        c = ops.gemm_split_ex(a, img, Nc, a_row_max=rows, tile_rows=tr)[0]
        err = (c.double() - ref).abs()
        bad = (err > 1e-4 * ref.abs().max()).nonzero()
        print(f"M={M} Nc={Nc} R={R} queues={'8' if tr else '1'}: max err {err.max().item():.3e} (ref max {ref.abs().max().item():.3f}); "
              f"bad elements {bad.shape[0]}" + (f", first {bad[0].tolist()}, rows {bad[:, 0].min().item()}..{bad[:, 0].max().item()}, "
                                               f"cols {bad[:, 1].min().item()}..{bad[:, 1].max().item()}" if bad.shape[0] else ""))
