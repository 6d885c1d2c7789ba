"""Wall time of the codebook initialisation: Kmeans.run on the first 20 000 rows' residuals, level by level, as the
first training forward does (reference train_rqvae.py:178-183).  Usage (GPU box): python tools/bench_kmeans.py [K] [L]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from init.kmeans import Kmeans  # noqa: E402
from rqhip import ops, tuning  # noqa: E402
from modules.rqvae import RqVae  # noqa: E402
from modules.quantize import QuantizeForwardMode  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tuning.enable_tuned_gemms()
torch.manual_seed(0)
model = RqVae(768, 32, [512, 256, 128], K, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE, n_layers=L,
              n_cat_features=0).cuda()
g = torch.Generator().manual_seed(1234)
X = torch.nn.functional.normalize(torch.randn(20000, 768, generator=g), dim=-1).cuda()
with torch.no_grad():
    res0 = model.encode(X)
for rep in range(2):   # the first pass pays module loading; the second is the number
    np.random.seed(0)
    torch.manual_seed(0)
    res = res0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = []
    for l in range(L):
        km = Kmeans(k=K)
        calls = [0]
        real = ops.kmeans_lloyd

        def counting(*a, **k):
            calls[0] += 1
            return real(*a, **k)

        ops.kmeans_lloyd = counting
        out = km.run(res)
        ops.kmeans_lloyd = real
        iters.append(calls[0])
        cb = out.centroids
        o = ops.rq_forward(res, cb[None], ops.MODE_EVAL, 0.25, want_residuals=False, want_norm=False)
        res = res - o.embs[0]
    torch.cuda.synchronize()
    print(f"pass {rep}: k-means of {L} levels (20000 x 32, K={K}): {time.perf_counter() - t0:.4f} s, batches per level {iters}")
