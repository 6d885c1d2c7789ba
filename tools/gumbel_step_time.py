#!/usr/bin/env python3
"""Developer probe (GPU box): one RQ-VAE training step (forward + backward, no optimizer) at 100 000 rows in each
quantiser mode, same model shape as bench.py."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
x = torch.nn.functional.normalize(torch.randn(B, 768, device="cuda"), dim=-1)
for mode in (QuantizeForwardMode.STE, QuantizeForwardMode.ROTATION_TRICK, QuantizeForwardMode.GUMBEL_SOFTMAX):
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
              n_cat_features=0, codebook_kmeans_init=False, codebook_mode=mode).cuda().train()

    def step():
        m.zero_grad(set_to_none=True)
        out = m(SeqBatch(None, None, None, x, None, None), 0.2)
        out.loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"{mode.name:16s} B={B}: {(time.perf_counter() - t) / 10 * 1e3:7.3f} ms per forward+backward", flush=True)
