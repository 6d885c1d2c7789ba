"""Where does the masked weight-gradient kernel lose its 15 % against the unmasked one?  (developer probe)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import ops  # noqa: E402

M = 100_000


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


for N, K in [(512, 768), (256, 512), (512, 256)]:
    gy = torch.randn(M, N, device="cuda")
    y = torch.relu(torch.randn(M, N, device="cuda"))
    x = torch.randn(M, K, device="cuda")
    print(f"dW [{N},{K}]: plain {timeit(lambda: ops.linear_wgrad(gy, None, x)):.0f} us | masked+writeback "
          f"{timeit(lambda: ops.linear_wgrad(gy, y, x)):.0f} | masked, no writeback "
          f"{timeit(lambda: ops.linear_wgrad(gy, y, x, want_masked=False)):.0f} | mask read from gy itself (no extra "
          f"stream), no writeback {timeit(lambda: ops.linear_wgrad(gy, gy, x, want_masked=False)):.0f}")
