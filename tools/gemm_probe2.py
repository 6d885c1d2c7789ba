#!/usr/bin/env python3
"""Developer tool: times the product GEMM loop (gs_tile2) in the in-tree build and in every phase-skipping probe build under
tools/_ab/librqhip_p<bits>.so (tools/ab_build.sh p<bits> gemm_split.hip -DGS_PROBE=<bits>; bits: 1 no stage barriers, 2 no
split / LDS writes, 4 no A loads, 8 no B loads, 16 no matrix instructions, 32 no LDS reads of A).  One subprocess per library.
Usage (GPU box): python tools/gemm_probe2.py"""
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
    M = 100_000
    out = []
    for R, Nc, epi in ((768, 512, _lib.EPI_RELU), (512, 768, _lib.EPI_STORE), (512, 768, _lib.EPI_RECON), (512, 256, _lib.EPI_RELU)):
        a = torch.randn(M, R, device="cuda")
        w = torch.randn(Nc, R, device="cuda") / R ** 0.5
        aux = torch.randn(M, Nc, device="cuda")
        img = ops.weight_images([(w, False)])[0]
        rm = ops.maxima(a, cols=False)[0]
        c = torch.empty((M, Nc), device="cuda")

        def run():
            ops.gemm_split_ex(a, img, Nc, epilogue=epi, aux=aux if epi >= _lib.EPI_RECON else None, row_scale=1e-5, a_row_max=rm)
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

names = {1: "no barriers", 2: "no split/LDS writes", 4: "no A loads", 8: "no B loads", 16: "no MFMA", 32: "no LDS reads", 64: "A loads contiguous", 128: "epilogue contiguous"}
extra = [(a, -1) for a in sys.argv[1:] if a.endswith(".so")]      # other builds to time next to the product (any name)
libs = [("-", 0)] + extra + sorted(((f, int(os.path.basename(f)[len("librqhip_p"):-3])) for f in glob.glob(os.path.join(ROOT, "tools", "_ab", "librqhip_p*.so"))
                           if os.path.basename(f)[len("librqhip_p"):-3].isdigit()), key=lambda t: t[1])
print(f"{'build':>44} | 768->512 relu  512->768 store  512->768 recon  512->256 relu   (us, 100 000 rows)")
for path, bits in libs * 2:
    what = "product" if bits == 0 else os.path.basename(path) if bits < 0 else " + ".join(v for k, v in names.items() if bits & k)
    r = subprocess.run([sys.executable, __file__, "--one", path], capture_output=True, text=True)
    print(f"{what:>44} | {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
