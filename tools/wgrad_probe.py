#!/usr/bin/env python3
"""Developer tool: times the f16x2 weight-gradient kernel in the in-tree build and in every phase-skipping probe build under
tools/_ab/librqhip_w<bits>.so (tools/ab_build.sh w<bits> wgrad_split.hip -DWS_PROBE=<bits>; bits: 1 no split / LDS writes,
2 no global loads, 4 no matrix instructions, 8 no LDS reads, 16 no stage barriers).  One subprocess per library.
Usage (GPU box): python tools/wgrad_probe.py"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
    from rqhip import _lib
    if sys.argv[2] != "-":
        _lib.load(sys.argv[2])
    from rqhip import ops
    M, out = 100_000, []
    for N, K in ((512, 768), (768, 512), (256, 512), (128, 256)):
        gy = torch.randn(M, N, device="cuda") * 1e-5
        x = torch.randn(M, K, device="cuda")
        gc, xc = ops.maxima(gy, rows=False)[1], ops.maxima(x, rows=False)[1]
        dw = torch.empty((N, K), device="cuda")

        def run():
            ops.linear_wgrad(gy, None, x, out=dw, g_col_max=gc, x_col_max=xc)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(" ".join(f"{t:8.1f}" for t in out))
    sys.exit(0)

names = {1: "no split/LDS writes", 2: "no loads", 4: "no MFMA", 8: "no LDS reads", 16: "no barriers"}
libs = [("-", 0)] + sorted(((f, int(os.path.basename(f)[len("librqhip_w"):-3])) for f in glob.glob(os.path.join(ROOT, "tools", "_ab", "librqhip_w*.so"))
                           if os.path.basename(f)[len("librqhip_w"):-3].isdigit()), key=lambda t: t[1])
print(f"{'build':>58} | dW [512,768] [768,512] [256,512] [128,256]   (us, 100 000 rows, kernel + reduce)")
for path, bits in libs * 2:
    what = "product" if bits == 0 else " + ".join(v for k, v in names.items() if bits & k)
    r = subprocess.run([sys.executable, __file__, "--one", path], capture_output=True, text=True)
    print(f"{what:>58} | {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
