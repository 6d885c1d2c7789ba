#!/usr/bin/env python3
"""Developer tool: socket power and shader clock (rocm-smi samples) while gemm_split runs back to back for a few seconds.
usage (GPU box): python tools/gemm_power.py [lib.so] [seconds]   -- is the kernel bound by the power limit (clock drops as the
matrix pipe gets busier) or by instruction issue?"""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
if lib:
    _lib.load(os.path.abspath(lib))
from rqhip import ops  # noqa: E402

samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip().splitlines()[-1] if out.strip() else "")
        except Exception as e:  # noqa: BLE001
            samples.append(f"err {e}")
        time.sleep(0.4)


x = torch.randn(100_000, 768, device="cuda")
w = torch.randn(512, 768, device="cuda") / 768 ** 0.5
ARITH = ops.BF16X3 if os.environ.get("RQ_POWER_ARITH") == "bf16x3" else ops.F16X2     # (round 4: the product arithmetic by default)
p = ops.weight_planes(w, arith=ARITH)
_rm = ops.maxima(x, cols=False)[0] if ARITH == ops.F16X2 else None
_c = torch.empty((100_000, 512), device="cuda")


def _gemm():
    ops.gemm_split_ex(x, p, 512, arith=ARITH, epilogue=_lib.EPI_RELU, a_row_max=_rm)


for _ in range(5):
    _gemm()
torch.cuda.synchronize()
hdr = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print("header:", hdr[0] if hdr else "?")
print("idle  :", hdr[-1] if hdr else "?")
th = threading.Thread(target=sampler)
th.start()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time()
n = 0
a.record()
while time.time() - t0 < secs:
    for _ in range(50):
        _gemm()
    n += 50
    torch.cuda.synchronize()
b.record()
torch.cuda.synchronize()
stop = True
th.join()
print(f"{lib or 'product'}: {a.elapsed_time(b) / n * 1e3:.1f} us per 768 -> 512 GEMM over {n} launches")
for s in samples[2:]:
    print("  ", s)
