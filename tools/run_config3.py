#!/usr/bin/env python3
"""BASELINE configuration 3 end to end: `train_rqvae.train` with the bindings of configs/rqvae_ml32m.gin (768 -> [512,256,128]
-> 64, 3 x 256 codes, rotation trick, batch 64, AdamW 1e-4 / 0.01, HIP k-means init) on a synthetic ML-32M-sized item
matrix (dataset_folder="synthetic:<n>", default 87 585 items as in MovieLens-32M), for a bounded number of iterations, eager and with the
hipGraph step.  Prints iterations/s of the whole loop (k-means warm-up, steps, eval, id-diversity pass, checkpoint).
Usage (GPU box): python tools/run_config3.py [iterations]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
N_ITEMS = int(os.environ.get("RQ_C3_ITEMS", "87585"))   # (tool parameter, not a product switch)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
if os.environ.get("RQ_C3_MODE"):   # (the parent process stays light: no torch, no GPU context)
    import numpy as np  # noqa: E402
    import torch  # noqa: E402
    import train_rqvae  # noqa: E402
    from rqhip import ginlite  # noqa: E402
def run(graph: bool) -> None:
    ginlite.clear_config()
    ginlite.parse_config_file(os.path.join(ROOT, "rq-vae-recommender_amd", "configs", "rqvae_ml32m.gin"))
    torch.manual_seed(0)
    np.random.seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = train_rqvae.train(iterations=iters, eval_every=iters, save_model_every=iters, save_dir_root=tmp + "/",
                                wandb_logging=False, dataset_folder=f"synthetic:{N_ITEMS}",
                                log_every=int(os.environ.get("RQ_C3_LOG_EVERY", 10 ** 9)), use_hip_graph=graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"config 3 (rqvae_ml32m.gin bindings, {N_ITEMS} synthetic items), {iters + 1} iterations, "
          f"{'hipGraph step' if graph else 'eager step'}: {dt:.2f} s whole loop = {(iters + 1) / dt:,.0f} it/s "
          f"({64 * (iters + 1) / dt:,.0f} items/s), final loss {res['loss']:.5f}", flush=True)


if os.environ.get("RQ_C3_MODE"):          # child: one mode
    run(os.environ["RQ_C3_MODE"] == "graph")
else:                                     # parent: one child per mode, so that a fault in one is reported, not fatal
    import subprocess
    for mode in ("eager", "graph"):
        env = dict(os.environ, RQ_C3_MODE=mode)
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), str(iters)], env=env, timeout=150,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            lines = [l for l in p.stdout.splitlines() if l.startswith("config 3") or "fault" in l.lower()]
            print("\n".join(lines) if lines else f"config 3, {mode}: no result (exit code {p.returncode})", flush=True)
            if p.returncode != 0:
                print(f"config 3, {mode} step: FAILED with exit code {p.returncode}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"config 3, {mode} step: TIMED OUT after 150 s", flush=True)
