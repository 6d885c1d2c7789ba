#!/usr/bin/env python3
"""A/B of round 6's second captured step shape (the short batch that ends every epoch) on the reference's configs/rqvae_amazon.gin at its
own corpus size (12 101 synthetic items, 95 % in the training split: 17 full batches of 640 + one short batch per epoch): iterations per
second of train_rqvae.train with both shapes replayed (graph_epoch_tail=True) against round 5's form (tail eager + a re-capture per epoch),
and eager.   usage (GPU box): python tools/epoch_tail_ab.py [iterations]      (tools/bench_small_batch.py --json calls `measure`)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rq-vae-recommender_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

ARMS = (("graph_both_shapes", "both shapes replayed", dict(use_hip_graph=True, graph_epoch_tail=True)),
        ("graph_full_batches_only", "full batches replayed, tail eager + re-capture (round 5)", dict(use_hip_graph=True, graph_epoch_tail=False)),
        ("eager", "eager", dict(use_hip_graph=False)))


def measure(iters: int = 2000, say=print, arms=ARMS) -> dict:
    """{arm: iterations / s} of the gin-driven training loop on a 12 101-item synthetic corpus (a short run first: process warm-up)."""
    import io
    from contextlib import redirect_stdout

    import numpy as np
    import torch
    import train_rqvae
    from rqhip import ginlite
    out = {"gin": "configs/rqvae_amazon.gin", "corpus_items": 12101, "batch": 640, "iterations": iters}
    for key, name, kw in (("warm_up", "(process warm-up)", dict(use_hip_graph=True, graph_epoch_tail=True)),) + tuple(arms):
        ginlite.clear_config()
        ginlite.parse_config_file(os.path.join(PKG, "configs", "rqvae_amazon.gin"))
        torch.manual_seed(0)
        np.random.seed(0)
        n = 200 if key == "warm_up" else iters
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            res = train_rqvae.train(iterations=n, eval_every=10 ** 9, save_model_every=10 ** 9, save_dir_root="/tmp/epoch_tail_ab/",
                                    wandb_logging=False, dataset_folder="synthetic:12101", log_every=10 ** 9, do_eval=False, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        say(f"{name:62s} {n / dt:8.1f} it/s  ({dt:.2f} s for {n} iterations; captures {res['graph_captures']})")
        if key != "warm_up":
            out[key + "_it_per_s"] = round(n / dt, 1)
            out[key + "_captures"] = {str(k): v for k, v in res["graph_captures"].items()}
    ginlite.clear_config()
    return out


if __name__ == "__main__":
    measure(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
