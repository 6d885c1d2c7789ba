#!/usr/bin/env python3
"""Times the weight-gradient kernels of the large layers at 100 000 rows in the f16x2 arithmetic (column maxima given) and in the
bf16x3 one (mask fused); RQ_LIB=<path> loads another build of librqhip.so (A/B).   Usage (GPU box): [RQ_LIB=..] python tools/wgrad_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
if os.environ.get("RQ_LIB"):
    from rqhip import _lib
    _lib.load(os.path.abspath(os.environ["RQ_LIB"]))
from rqhip import ops  # noqa: E402

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


print(f"library: {os.environ.get('RQ_LIB') or 'product'}")
for N, K in [(512, 768), (768, 512), (256, 512), (512, 256), (128, 256), (256, 128)]:
    gy = torch.randn(M, N, device="cuda") * 1e-5
    y = torch.relu(torch.randn(M, N, device="cuda"))
    x = torch.randn(M, K, device="cuda")
    _, gc, _ = ops.maxima(gy, rows=False)
    _, xc, _ = ops.maxima(x, rows=False)
    t16 = timeit(lambda: ops.linear_wgrad(gy, None, x, g_col_max=gc, x_col_max=xc))
    t3 = timeit(lambda: ops.linear_wgrad(gy, y, x))
    print(f"  dW [{N:3d},{K:3d}]: f16x2 {t16:7.1f} us   bf16x3 with mask {t3:7.1f} us", flush=True)
