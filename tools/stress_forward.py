#!/usr/bin/env python3
"""Developer stress (GPU box): many back-to-back rq_forward launches on the shapes that use cooperative tiles (LDS
spin-wait synchronisation), checking every result against the first one.  Run under a short `timeout`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rqhip import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for B, D, K, L, mode in ((100000, 32, 256, 3, 1), (640, 32, 256, 3, 1), (20000, 32, 256, 3, 0), (30000, 16, 32, 16, 2),
                         (77, 64, 256, 3, 2), (3000, 128, 64, 2, 1)):
    g = torch.Generator().manual_seed(B)
    x = (torch.randn(B, D, generator=g) * 0.5).cuda()
    cb = (torch.randn(L, K, D, generator=g) * 0.3).cuda()
    ref = ops.rq_forward(x, cb, mode, 0.25, want_embs=False, want_residuals=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    bad = 0
    for i in range(n):
        out = ops.rq_forward(x, cb, mode, 0.25, want_embs=False, want_residuals=False)
        if i % 250 == 0:
            bad += int(not (torch.equal(out.ids, ref.ids) and torch.equal(out.loss, ref.loss)
                            and torch.equal(out.emb_sum, ref.emb_sum)))
    torch.cuda.synchronize()
    print(f"B={B} D={D} K={K} L={L} mode={mode}: {n} launches in {time.perf_counter() - t:.2f} s, mismatching checks: {bad}",
          flush=True)
