#!/usr/bin/env python3
"""Where the HOST time of an eager small-batch step goes (batch 640 of configs/rqvae_amazon.gin: ~40 launches of a few microseconds, the
device is idle most of the 1.2 ms): cProfile over N eager steps, top functions by own and by cumulative time.
  python tools/eager_host_profile.py [c3] [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from data.schemas import SeqBatch  # noqa: E402
from modules.quantize import QuantizeForwardMode  # noqa: E402
from modules.rqvae import RqVae  # noqa: E402
from rqhip import tuning  # noqa: E402
from rqhip.optim import FlatAdamW  # noqa: E402

if "--no-tunable" not in sys.argv:
    tuning.enable_tuned_gemms()
if "--no-small" in sys.argv:
    from rqhip import linear as _linear
    _linear.use_small_kernels(False)
c3 = "c3" in sys.argv
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
N = nums[0] if nums else 400
B = 64 if c3 else 640
torch.manual_seed(0)
m = RqVae(input_dim=768, embed_dim=64 if c3 else 32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3, n_cat_features=0,
          codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.ROTATION_TRICK if c3 else QuantizeForwardMode.STE).cuda()
with torch.no_grad():
    for l, layer in enumerate(m.layers):
        layer.embedding.weight.copy_(torch.randn_like(layer.embedding.weight) * (0.05 / (l + 1)))
opt = FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
x = torch.nn.functional.normalize(torch.randn(B, 768, device="cuda"), dim=-1)
batch = SeqBatch(None, None, None, x, None, None)


def step():
    for p in m.parameters():
        p.grad = None
    out = m(batch, 0.2)
    out.loss.backward()
    opt.step()
    return out.loss


def fwd_only():
    return m(batch, 0.2).loss


for _ in range(20):
    step()
torch.cuda.synchronize()
for name, fn in (("step", step), ("forward only", fwd_only)):
    t = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / N * 1e3:.3f} ms (host-bound wall time per call)")
# phases by wall time, each ended by a synchronize (upper bounds: the sync adds the device tail)
tf = tb = to = 0.0
for _ in range(N):
    for p in m.parameters():
        p.grad = None
    t0 = time.perf_counter()
    out = m(batch, 0.2)
    t1 = time.perf_counter()
    out.loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    tf, tb, to = tf + t1 - t0, tb + t2 - t1, to + t3 - t2
torch.cuda.synchronize()
print(f"host time per phase (no syncs inside): forward {tf / N * 1e3:.3f} ms, backward {tb / N * 1e3:.3f} ms, optimizer {to / N * 1e3:.3f} ms")
# (the autograd engine runs a ROCm backward on its own thread, where cProfile does not look: keep it on this thread for the profile)
torch.autograd.set_multithreading_enabled(False)
t = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
print(f"step with the backward on the calling thread: {(time.perf_counter() - t) / N * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(70)
    print(f"==== by {key} ({N} steps) ====")
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[6:]))
