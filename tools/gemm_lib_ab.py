#!/usr/bin/env python3
"""Times the product GEMM (f16x2, gemm_f16_kernel) at 100 000 rows with the library given as argv[1] (a tools/ab_build.sh build) or the in-tree one.
Run once per build in the same gpurun call; HIP events, best of 3 x 20 launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
import torch  # noqa: E402
from rqhip import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.load(os.path.abspath(sys.argv[1]))
from rqhip import ops  # noqa: E402

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


out = []
for Nc, R, epi in ((512, 768, _lib.EPI_RELU), (768, 512, _lib.EPI_STORE), (256, 512, _lib.EPI_RELU), (512, 256, _lib.EPI_RELU), (768, 512, _lib.EPI_RECON)):
    torch.manual_seed(0)
    a = torch.relu(torch.randn(M, R, device="cuda"))
    w = torch.randn(Nc, R, device="cuda") / R ** 0.5
    x = torch.randn(M, Nc, device="cuda")
    img = ops.weight_planes(w, arith=ops.F16X2)
    rows = ops.maxima(a, cols=False)[0]
    kw = dict(epilogue=epi, a_row_max=rows)
    if epi == _lib.EPI_RECON:
        kw.update(aux=x, row_scale=1.0 / M)
    c = ops.gemm_split_ex(a, img, Nc, **kw)[0]
    t = min(timeit(lambda: ops.gemm_split_ex(a, img, Nc, **kw)) for _ in range(3))
    out.append(f"{R}->{Nc}/{epi}: {t:6.1f} us (sum {float(c.double().sum()):.6e})")
print((sys.argv[1] if len(sys.argv) > 1 else "in-tree") + " | " + " | ".join(out))
