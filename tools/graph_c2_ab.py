#!/usr/bin/env python3
"""Developer A/B: the configuration-2 training step (100 000 rows) launched eagerly vs captured once into a hipGraph and replayed
(train_rqvae._GraphedStep), alternating blocks in ONE process.  Usage (GPU box): python tools/graph_c2_ab.py [steps per block] [rows]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
import bench  # noqa: E402
import train_rqvae  # noqa: E402
from data.schemas import SeqBatch  # noqa: E402
from rqhip import dist as rqdist, tuning  # noqa: E402
from rqhip.optim import FlatAdamW  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
tuning.enable_tuned_gemms()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
X = torch.nn.functional.normalize(torch.randn(rows, 768, generator=g), dim=-1).to(dev)
model, _ = bench.build_model(dev, X[:20000], 3, 256)
red = rqdist.FlatGradReducer(model.parameters()).attach(model)
opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
batch = SeqBatch(None, None, None, X, None, None)


def eager_step():
    red.zero_()
    out = model(batch, gumbel_t=0.2)
    out.loss.backward()
    opt.step()
    return out


def block(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    eager_step()
gs = train_rqvae._GraphedStep(model, opt, red, rows, 768, dev, 0.2)
gs.capture(X)
loss_g = float(gs.run(X).loss)
res = {"eager": [], "graph": []}
for rep in range(3):
    res["eager"].append(block(eager_step, steps))
    res["graph"].append(block(lambda: gs.graph.replay(), steps))
    print(f"block {rep}: eager {res['eager'][-1]:.4f} ms  graph replay {res['graph'][-1]:.4f} ms", flush=True)
print(f"rows {rows}: eager median {np.median(res['eager']):.4f} ms/step, graph median {np.median(res['graph']):.4f} ms/step, loss after replay {loss_g:.6f}")
