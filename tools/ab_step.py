#!/usr/bin/env python3
"""Sustained A/B of the C2 training step with the large activation GEMMs on csrc/gemm_split.hip vs the library fp32 GEMMs
(modules.encoder.use_split_gemms), and the weight gradients on csrc/wgrad_split.hip vs the fp32-MFMA kernel: alternating
blocks of steps in ONE process, so that clocks / power state are shared.  Usage: python tools/ab_step.py [steps per block]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
import bench  # noqa: E402
from data.schemas import SeqBatch  # noqa: E402
from modules import encoder  # noqa: E402
from rqhip import linear  # noqa: E402
from rqhip import dist as rqdist, ops, tuning  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 250
tuning.enable_tuned_gemms()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
X = torch.nn.functional.normalize(torch.randn(100_000, 768, generator=g), dim=-1).to(dev)
model, _ = bench.build_model(dev, X[:20000], 3, 256)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
red = rqdist.FlatGradReducer(model.parameters()).attach(model)
batch = SeqBatch(None, None, None, X, None, None)
_wg = ops.linear_wgrad


def step():
    red.zero_()
    out = model(batch, gumbel_t=0.2)
    out.loss.backward()
    opt.step()


def block(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
rows = []
for rep in range(3):
    for name, split_gemm, split_wgrad in (("split gemm + split wgrad", True, True), ("library gemm + split wgrad", False, True),
                                          ("library gemm + fp32 wgrad", False, False)):
        linear.use_split_gemms(split_gemm)
        ops.linear_wgrad = _wg if split_wgrad else (lambda *a, **k: _wg(*a, **dict(k, exact_fp32=True)))
        encoder.ops.linear_wgrad = ops.linear_wgrad
        ms = block(steps)
        rows.append((name, ms))
        print(f"block {rep}: {name:28s} {ms:7.3f} ms/step  ({steps} steps)", flush=True)
for name in dict(rows):
    v = [m for n, m in rows if n == name]
    print(f"{name:28s} median {np.median(v):.3f} ms/step")
