#!/usr/bin/env python3
"""Generate TunableOp selections for the RQ-VAE MLP GEMMs at the shipped batch sizes (run on the GPU box):
    python tools/tune_gemms.py --out gpurun_out/tunableop_gfx950.csv 100000 20000 640
then copy the CSV to rq-vae-recommender_amd/tuning/tunableop_gfx950.csv."""
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

argv = sys.argv[1:]
out_file = None
if argv[:1] == ["--out"]:
    out_file, argv = argv[1], argv[2:]
assert tuning.enable_tuned_gemms(verbose=True, tune=True, out_file=out_file)
sizes = [int(v) for v in argv] or [100000]
for embed, mode in ((32, QuantizeForwardMode.STE), (64, QuantizeForwardMode.ROTATION_TRICK)):
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=embed, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
              n_cat_features=0, codebook_kmeans_init=False, codebook_mode=mode).cuda()
    for B in sizes:
        x = torch.nn.functional.normalize(torch.randn(B, 768, device="cuda"), dim=-1)
        t0 = time.perf_counter()
        for train in (True, False):
            m.train(train)
            out = m(SeqBatch(None, None, None, x, None, None), 0.2)
            if train:
                out.loss.backward()
        torch.cuda.synchronize()
        print(f"D={embed} B={B}: tuned in {time.perf_counter() - t0:.1f} s", flush=True)
import torch.cuda.tunable as tunable  # noqa: E402
pass  # results are flushed to the file at process exit
print("wrote", tunable.get_filename())
