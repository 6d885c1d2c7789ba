#!/usr/bin/env python3
"""Developer tool for the two-piece fp16 weight-gradient prototype (csrc/wgrad_split.hip, WS_F16=1): error of
dW = g^T x against fp64 next to the library's fp32 GEMM and the shipped three-piece bf16 kernel, with and without the exact
power-of-two column scales, and timing.  NOT run on a GPU in round 3.
Build HERE:  bash tools/ab_build.sh wsf16 wgrad_split.hip -DWS_F16=1        Run on the GPU box:  python tools/wgrad_f16_check.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
MODE = sys.argv[1] if len(sys.argv) > 1 else "driver"

if MODE == "driver":     # one process per build: the two libraries export the same symbols
    for m in ("ship", "f16"):
        subprocess.run([sys.executable, __file__, m], check=False)
    sys.exit(0)

from rqhip import _lib  # noqa: E402

if MODE == "f16":
    h = _lib.load(os.path.join(ROOT, "tools", "_ab", "librqhip_wsf16.so"))
from rqhip import ops  # noqa: E402

vp = C.c_void_p


def timeit(fn, n=10):
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


torch.manual_seed(0)
M = 100_000
x = torch.nn.functional.normalize(torch.randn(M, 768), dim=-1).cuda()
w1 = (torch.randn(512, 768) / 768 ** 0.5).cuda()
y = torch.relu(x @ w1.t())
cases = {
    "layer 1: masked 1/B-scale gradient x unit-norm input  [512, 768]": (torch.randn(M, 512, device="cuda") / M, y, x),
    "layer 2: gradient rows over three decades x post-ReLU [256, 512]":
        (torch.randn(M, 256, device="cuda") / M * torch.pow(10.0, torch.randint(-3, 1, (M, 1), device="cuda").float()), None, y),
}
for name, (g, yy, xx) in cases.items():
    gm = g * (yy > 0) if yy is not None else g
    ref = gm.double().t() @ xx.double()
    sc = ref.abs().max().item()
    lib_e = ((gm.t() @ xx).double() - ref).abs().max().item() / sc
    line = [f"{MODE:4s} {name}: library {lib_e:.2e}"]
    for scaled in ((False, True) if MODE == "f16" else (False,)):
        if MODE == "f16":
            if scaled:
                gmx = torch.zeros(g.shape[1], dtype=torch.int32, device="cuda")
                xmx = torch.zeros(xx.shape[1], dtype=torch.int32, device="cuda")
                assert h.rqhip_col_maxima(vp(gm.data_ptr()), C.c_int64(M), g.shape[1], vp(gmx.data_ptr()), None) == 0   # (the masked gradient's maxima)
                assert h.rqhip_col_maxima(vp(xx.data_ptr()), C.c_int64(M), xx.shape[1], vp(xmx.data_ptr()), None) == 0
                h.rqhip_wgrad_split_set_maxima(vp(gmx.data_ptr()), vp(xmx.data_ptr()))
            else:
                h.rqhip_wgrad_split_set_maxima(None, None)
        run = lambda: ops.linear_wgrad(g, yy, xx)   # noqa: E731
        out = run()
        dw = out[0] if isinstance(out, tuple) else out
        line.append(f"{'kernel' if MODE == 'ship' else ('f16 column-scaled' if scaled else 'f16 unscaled')}: "
                    f"{(dw.double() - ref).abs().max().item() / sc:.2e}, {timeit(run):.1f} us")
    print("  |  ".join(line), flush=True)
