#!/usr/bin/env python3
"""Developer probe (GPU box): Linear+ReLU as two kernels vs torch._addmm_activation (hipBLASLt ReLU epilogue) for
the RQ-VAE MLP shapes at 100 000 rows, with TunableOp tuning enabled for both.
usage: python tools/relu_epilogue_probe.py   (tunes the probed shapes into gpurun_out/tune_probe.csv)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rqhip import tuning  # noqa: E402

torch.set_float32_matmul_precision("highest")
print("tunable:", tuning.enable_tuned_gemms(verbose=True))


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


B = 100000
for n_in, n_out in ((768, 512), (512, 256), (256, 128), (32, 128), (128, 256), (256, 512)):
    x = torch.randn(B, n_in, device="cuda")
    w = torch.randn(n_out, n_in, device="cuda") * 0.05
    zb = torch.zeros(n_out, device="cuda")
    a = timed(lambda: torch.relu(torch.nn.functional.linear(x, w)))
    g = timed(lambda: torch.nn.functional.linear(x, w))
    b = timed(lambda: torch._addmm_activation(zb, x, w.t()))
    y0 = torch.relu(torch.nn.functional.linear(x, w))
    y1 = torch._addmm_activation(zb, x, w.t())
    fl = 2.0 * B * n_in * n_out
    print(f"{n_in:4d}->{n_out:4d}: linear {g:7.1f} us ({fl / g / 1e6:6.1f} TF)  linear+relu {a:7.1f} us   fused epilogue {b:7.1f} us "
          f"({fl / b / 1e6:6.1f} TF)   max|diff| {float((y0 - y1).abs().max()):.2e}", flush=True)
