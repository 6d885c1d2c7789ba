#!/usr/bin/env python3
"""Does the small-batch training trajectory depend on which weight-gradient kernels run?  The same model, seed and fixed batch
(rqvae_amazon.gin's shape: batch 640, D = 32, STE) trained for 400 steps with (a) the job-table kernel (csrc/wgrad_jobs.hip, the
default), (b) round 4's per-layer split kernels, (c) strict fp32 (library GEMMs, oracle-ordered weight gradients); the loss at
fixed steps.  The arg-min assignments make the trajectory chaotic in the last bits of the gradients: the three arms separate
after some tens of steps by amounts of the same size -- the check that no arm is systematically off."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from data.schemas import SeqBatch  # noqa: E402
from modules.quantize import QuantizeForwardMode  # noqa: E402
from modules.rqvae import RqVae  # noqa: E402
from rqhip import linear  # noqa: E402
from rqhip.optim import FlatAdamW  # noqa: E402


def run(name):
    jobs = linear.use_wgrad_jobs(name == "jobs")
    arith = linear.use_arith("fp32" if name == "fp32" else "f16x2")
    try:
        torch.manual_seed(0)
        m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3, n_cat_features=0,
                  codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda()
        with torch.no_grad():
            for l, layer in enumerate(m.layers):
                layer.embedding.weight.copy_(torch.randn_like(layer.embedding.weight) * (0.05 / (l + 1)))
        opt = FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
        x = torch.nn.functional.normalize(torch.randn(640, 768, device="cuda"), dim=-1)
        batch = SeqBatch(None, None, None, x, None, None)
        out = []
        for step in range(1, 401):
            for p in m.parameters():
                p.grad = None
            loss = m(batch, 0.2).loss
            loss.backward()
            opt.step()
            if step in (1, 2, 5, 10, 20, 50, 100, 200, 400):
                out.append(f"{step}: {float(loss):.6f}")
        print(f"{name:8s} " + "  ".join(out))
    finally:
        linear.use_wgrad_jobs(jobs)
        linear.use_arith(arith)


for arm in ("jobs", "per-layer", "fp32"):
    run(arm)
