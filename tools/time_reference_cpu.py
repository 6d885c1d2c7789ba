#!/usr/bin/env python3
"""Time THE REFERENCE'S OWN MODULES on this host's CPU cores (SURVEY.md section 8d "CPU baseline").

Needs /root/reference (so it runs in the build container, not on the GPU box); the numbers are committed to
BASELINE.md section 2 and `profiles/r02_reference_cpu_timing.json`.  The reference is imported in place (stub gin,
see oracle/gen_golden.py); eager forward (`RqVae.forward._torchdynamo_orig_callable`), fp32, synthetic unit-norm
768-d items, codebooks spread like trained ones, 3 warm-up + timed iterations, wall clock.

Scopes: S-full = RqVae.forward + backward + AdamW step (B = 640, the reference's own batch, and B = 8192);
S-rq = the quantize stack alone (3 levels on 32-d latents, fwd + bwd); tokenisation = get_semantic_ids (eval);
Kmeans.run seconds per iteration (20 000 x 32, k = 256).
Usage: python tools/time_reference_cpu.py [--threads 8] [--out profiles/r02_reference_cpu_timing.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden  # noqa: E402


def wall(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    q, r, km, _sem, sch = gen_golden.import_reference()
    fwd = r.RqVae.forward._torchdynamo_orig_callable
    res = {"host": {"cpu_count": os.cpu_count(), "threads": args.threads, "torch": torch.__version__},
           "what": "reference modules imported from /root/reference, eager, fp32"}

    def model_for(mode):
        torch.manual_seed(0)
        m = r.RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256,
                    codebook_kmeans_init=False, codebook_mode=mode, n_layers=3, commitment_weight=0.25,
                    n_cat_features=0)
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for l, layer in enumerate(m.layers):
                layer.embedding.weight.copy_(torch.randn(256, 32, generator=g) * (0.05 / (l + 1)))
        return m

    g = torch.Generator().manual_seed(1234)
    X = torch.nn.functional.normalize(torch.randn(100_000, 768, generator=g), dim=-1)

    for mode_name, mode in (("ste", q.QuantizeForwardMode.STE), ("gumbel", q.QuantizeForwardMode.GUMBEL_SOFTMAX)):
        for B in (640, 8192):
            if mode_name == "gumbel" and B != 640:
                continue
            m = model_for(mode)
            m.train()
            opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
            batch = sch.SeqBatch(user_ids=None, ids=None, ids_fut=None, x=X[:B], x_fut=None, seq_mask=None)

            def step():
                opt.zero_grad()
                out = fwd(m, batch, 0.2)
                out.loss.backward()
                opt.step()

            t = wall(step, 3, 20 if B == 640 else 5)
            res[f"s_full_{mode_name}_B{B}"] = {"ms": round(t * 1e3, 3), "items_per_s": round(B / t, 1)}
            print(f"S-full {mode_name} B={B}: {t * 1e3:.2f} ms  {B / t:,.0f} items/s", flush=True)

    # S-rq: the quantize stack alone on 32-d latents
    m = model_for(q.QuantizeForwardMode.STE)
    m.train()
    for B in (640, 8192):
        lat = (torch.randn(B, 32, generator=g) * 0.05).requires_grad_(True)

        def rq():
            res_, loss = lat, 0
            for layer in m.layers:
                o = layer(res_, temperature=0.2)
                loss = loss + o.loss
                res_ = res_ - o.embeddings
            for p in m.layers.parameters():
                p.grad = None
            lat.grad = None
            (loss.mean() + res_.sum() * 0).backward()

        t = wall(rq, 3, 50 if B == 640 else 20)
        res[f"s_rq_ste_B{B}"] = {"ms": round(t * 1e3, 3), "items_per_s": round(B / t, 1)}
        print(f"S-rq STE B={B}: {t * 1e3:.3f} ms  {B / t:,.0f} items/s", flush=True)

    # tokenisation only
    m.eval()
    with torch.no_grad():
        t = wall(lambda: m.get_semantic_ids(X), 1, 3)
    res["tokenize_100000"] = {"ms": round(t * 1e3, 1), "items_per_s": round(100_000 / t, 1)}
    print(f"tokenise 100000 rows: {t:.3f} s  {100_000 / t:,.0f} items/s", flush=True)

    # k-means: seconds per Lloyd iteration
    with torch.no_grad():
        lat = m.encode(X[:20000])
    iters = 10
    np.random.seed(0)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    km.Kmeans(k=256, max_iters=iters).run(lat.clone())
    t = (time.perf_counter() - t0) / iters
    res["kmeans_20000x32_k256"] = {"s_per_iter": round(t, 4)}
    print(f"Kmeans.run 20000x32 k=256: {t * 1e3:.1f} ms/iteration", flush=True)

    # the port bench.py's `cpu_baseline` times on the GPU host (oracle/torch_port.py; /root/reference does not exist there), on
    # the same host, threads, rows and model shape as the reference's S-full numbers above: how far the port is from what it stands for
    from oracle import torch_port
    for B in (640, 8192):
        r_port = torch_port.time_training_steps(X[:B].contiguous(), steps=20 if B == 640 else 5, warmup=3, hidden=[512, 256, 128],
                                                embed_dim=32, n_levels=3, codebook_size=256, beta=0.25)
        ref_ips = res[f"s_full_ste_B{B}"]["items_per_s"]
        res[f"port_s_full_ste_B{B}"] = {"items_per_s": round(r_port["items_per_s"], 1),
                                        "port_over_reference": round(r_port["items_per_s"] / ref_ips, 3)}
        print(f"port S-full STE B={B}: {r_port['items_per_s']:,.0f} items/s = {r_port['items_per_s'] / ref_ips:.2f} x the reference's own modules", flush=True)

    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
