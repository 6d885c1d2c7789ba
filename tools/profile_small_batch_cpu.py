#!/usr/bin/env python3
"""Where the HOST time of an eager small-batch step goes (the step is launch-bound: 47 device activities, ~1.2 ms at batch 640):
cProfile over 300 steps, top functions by own time and by cumulative time.   Usage (GPU box): python tools/profile_small_batch_cpu.py [c3]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from data.schemas import SeqBatch  # noqa: E402
from modules.quantize import QuantizeForwardMode  # noqa: E402
from modules.rqvae import RqVae  # noqa: E402
from rqhip import dist as rqdist  # noqa: E402
from rqhip import tuning  # noqa: E402

tuning.enable_tuned_gemms()
C3 = len(sys.argv) > 1 and sys.argv[1] == "c3"
B = 64 if C3 else 640
torch.manual_seed(0)
m = RqVae(input_dim=768, embed_dim=64 if C3 else 32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3, n_cat_features=0,
          codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.ROTATION_TRICK if C3 else QuantizeForwardMode.STE).cuda()
with torch.no_grad():
    for l, layer in enumerate(m.layers):
        layer.embedding.weight.copy_(torch.randn_like(layer.embedding.weight) * (0.3 / (l + 1)))
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
reducer = rqdist.FlatGradReducer(m.parameters()).attach(m)
x = torch.nn.functional.normalize(torch.randn(B, 768, device="cuda"), dim=-1)
batch = SeqBatch(None, None, None, x, None, None)


def step():
    reducer.zero_()
    out = m(batch, 0.2)
    out.loss.backward()
    reducer.allreduce_mean()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print(f"eager step with the flat reducer, B={B}: {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(28)
    print("\n".join(l[:160] for l in buf.getvalue().splitlines()[4:44]))
