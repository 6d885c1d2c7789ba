#!/usr/bin/env python3
"""Developer probe: why is the forward kernel slow inside tools/bench_small_batch.py c3 (batch 64, D = 64, rotation trick)?
Times rqhip_rq_forward on the model's own latents / codebooks before and after training steps, per scan form, and prints the
margin statistics (rows the filtered scan has to re-decide exactly).   Usage (GPU box): python tools/fwd_c3_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from data.schemas import SeqBatch  # noqa: E402
from modules.quantize import QuantizeForwardMode  # noqa: E402
from modules.rqvae import RqVae  # noqa: E402
from rqhip import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
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
m = RqVae(input_dim=768, embed_dim=64, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3, n_cat_features=0,
          codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.ROTATION_TRICK).cuda()
with torch.no_grad():
    for l, layer in enumerate(m.layers):
        layer.embedding.weight.copy_(torch.randn_like(layer.embedding.weight) * (0.05 / (l + 1)))
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, fused=True)
x = torch.nn.functional.normalize(torch.randn(64, 768, device="cuda"), dim=-1)
batch = SeqBatch(None, None, None, x, None, None)
for phase, steps in (("at init", 0), ("after 50 steps", 50), ("after 250 steps", 200)):
    m.train()
    for _ in range(steps):
        for p in m.parameters():
            p.grad = None
        m(batch, 0.2).loss.backward()
        opt.step()
    with torch.no_grad():
        res0 = m.encode(x).contiguous()
        cbs = torch.stack([l.weight for l in m.layers]).detach().contiguous()
    k = ops.rq_forward(res0, cbs, ops.MODE_ROTATION, 0.25, want_margin=True, want_embs=False, want_residuals=False)
    xn = res0.norm(dim=1)
    cn = cbs.norm(dim=2)
    t = {scan: timeit(lambda: ops.rq_forward(res0, cbs, ops.MODE_ROTATION, 0.25, want_embs=False, want_residuals=False, scan=scan))
         for scan in ("auto", "fp32")}
    distinct = [int(torch.unique(k.ids[l]).numel()) for l in range(3)]
    print(f"{phase:16s}: |res0| {xn.min().item():.3g}..{xn.max().item():.3g}  |code| {cn.min().item():.3g}..{cn.max().item():.3g}  "
          f"tie_margin min {k.tie_margin.min().item():.3g} median {k.tie_margin.median().item():.3g}  distinct ids per level {distinct}  "
          f"forward auto {t['auto']:.1f} us  fp32 {t['fp32']:.1f} us", flush=True)
