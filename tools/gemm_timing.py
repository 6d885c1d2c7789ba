#!/usr/bin/env python3
"""Developer tool: where a stage of gemm_split goes, per wave of workgroup 0 (s_memtime stamps, csrc/gemm_split.hip GS_TIMING).
Build HERE (no GPU needed):   bash tools/ab_build.sh timing gemm_split.hip -DGS_TIMING      (add -DGS_PC=1 for the staging-wave variant)
Run on the GPU box:           python tools/gemm_timing.py [Nc,R]          (default 512,768; 100 000 rows)
Prints, for every wave, the shader-clock cycles of each phase of the first 14 stages of the workgroup's first tile:
stage = split + LDS writes of the next stage, req = issuing the requests, mult = 48 matrix instructions + their LDS reads,
bar = waiting at the stage barrier.  (The stamps themselves cost ~40 cycles each and an s_waitcnt.)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import _lib  # noqa: E402

h = _lib.load(os.path.join(ROOT, "tools", "_ab", "librqhip_timing.so"))
from rqhip import ops  # noqa: E402

Nc, R = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,768").split(","))
x = torch.randn(100_000, R, device="cuda")
w = torch.randn(Nc, R, device="cuda") / R ** 0.5
p = ops.weight_planes(w)
for _ in range(3):
    ops.gemm_split(x, p, Nc, relu=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 64))()
assert h.rqhip_gs_debug_read(buf) == 0
print(f"gemm_split {R} -> {Nc}, 100000 rows; cycles per phase (rows: stages 0..13 of the first tile)")
for wv in range(8):
    t = [buf[wv * 64 + i] for i in range(64)]
    if not t[0]:
        continue
    print(f"wave {wv}: prologue {t[1] - t[0]} cycles")
    for st in range(14):
        b = 2 + 4 * st
        if b + 3 >= 64 or not t[b + 3]:
            break
        prev = t[b - 1]
        if not t[b]:      # GS_PC build: a tile wave only multiplies and waits
            print(f"   stage {st:2d}: mult {t[b + 2] - prev:5d}  bar {t[b + 3] - t[b + 2]:5d}   total {t[b + 3] - prev:5d}")
            continue
        print(f"   stage {st:2d}: stage {t[b] - prev:5d}  req {t[b + 1] - t[b]:5d}  mult {t[b + 2] - t[b + 1]:5d}  bar {t[b + 3] - t[b + 2]:5d}"
              f"   total {t[b + 3] - prev:5d}")
