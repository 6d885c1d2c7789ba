#!/usr/bin/env python3
"""The reference's real training shapes are launch-bound on a GPU: eager step time, kernel launches per step, and the
same step replayed from a hipGraph (torch.cuda.CUDAGraph).
  python tools/bench_small_batch.py            # rqvae_amazon.gin: batch 640, D = 32, STE
  python tools/bench_small_batch.py c3         # rqvae_ml32m.gin (BASELINE config 3): batch 64, D = 64, rotation trick, lr 1e-4"""
import os
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

tuning.enable_tuned_gemms()
C3 = len(sys.argv) > 1 and sys.argv[1] == "c3"
B = 64 if C3 else (int(sys.argv[1]) if len(sys.argv) > 1 else 640)
torch.manual_seed(0)
m = RqVae(input_dim=768, embed_dim=64 if C3 else 32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
          n_cat_features=0, codebook_kmeans_init=False,
          codebook_mode=QuantizeForwardMode.ROTATION_TRICK if C3 else QuantizeForwardMode.STE).cuda()
with torch.no_grad():
    for l, layer in enumerate(m.layers):
        layer.embedding.weight.copy_(torch.randn_like(layer.embedding.weight) * (0.05 / (l + 1)))
opt = torch.optim.AdamW(m.parameters(), lr=1e-4 if C3 else 1e-3, weight_decay=0.01 if C3 else 1e-4, fused=True,
                        capturable=True)
x = torch.nn.functional.normalize(torch.randn(B, 768, device="cuda"), dim=-1)
batch = SeqBatch(None, None, None, x, None, None)


def step():
    for p in m.parameters():
        p.grad = None
    out = m(batch, 0.2)
    out.loss.backward()
    opt.step()
    return out.loss


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


ms = timeit(step)
print(f"eager  B={B}: {ms:.3f} ms/step  {B / ms * 1e3:,.0f} items/s")
try:   # kernel launches of one eager step
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    n = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
    ours = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "rqhip" in e.name)
    print(f"launches per step: {n} device activities ({ours} hand-written kernels)")
    if os.environ.get("RQ_LIST_LAUNCHES"):
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CUDA:
                print(f"    {e.name[:110]:110s} {e.device_time:8.1f} us")
except Exception as e:  # noqa
    print("launch count unavailable:", repr(e)[:200])
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    for p in m.parameters():
        p.grad = None
    with torch.cuda.graph(g):
        loss = step()
    ms = timeit(g.replay)
    print(f"graph  B={B}: {ms:.3f} ms/step  {B / ms * 1e3:,.0f} items/s   (loss {float(loss):.5f})")
except Exception as e:  # noqa
    print("graph capture failed:", repr(e)[:500])
