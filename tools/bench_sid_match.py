#!/usr/bin/env python3
"""Micro-benchmark of the decoder-side id matching kernels (csrc/sid_match.hip) on one MI355X.

  prefix index build (once per corpus), prefix lookup per beam step (P = batch x beams x candidates),
  first-match rank for TopKAccumulator -- each against the reference's formulation of the same step run with
  torch ops on the same GPU (modules/model.py:169-182, evaluate/metrics.py:16-19), where that fits.

usage: python tools/bench_sid_match.py [N_items ...]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rq-vae-recommender_amd"))
from rqhip import ops  # noqa: E402


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def torch_check_valid_prefix(codebooks, prefix, batch_size=100000):
    """The reference's formulation (model.py:176-182), for timing on the same device."""
    trimmed = codebooks[:, : prefix.shape[1]]
    out = []
    for i in range(0, prefix.shape[0], batch_size):
        batch = prefix[i:i + batch_size]
        out.append((trimmed.unsqueeze(1) == batch.unsqueeze(0)).all(dim=2).any(dim=0))
    return torch.cat(out)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [12_101, 1_000_000, 10_000_000]
    H, K = 4, 1024
    P = 256 * 10 * 64  # eval batch 256 x top-10 beams x 64 sampled candidates (model.py:316-317)
    g = torch.Generator(device="cuda").manual_seed(0)
    for N in sizes:
        corpus = torch.randint(0, K if N > 100_000 else 256, (N, H), device="cuda", generator=g)
        t_build = timed(lambda: ops.prefix_index_build(corpus), reps=5, warm=1)
        index = ops.prefix_index_build(corpus)
        print(f"N={N:>10,} H={H}: index {index.numel() / 2**20:8.1f} MiB, build {t_build * 1e3:8.3f} ms "
              f"({N * H / t_build / 1e9:.2f} G prefixes/s)")
        for h in (1, 2, 3, 4):
            rows = torch.randint(0, N, (P,), device="cuda", generator=g)
            prefix = corpus[rows, :h].clone()
            prefix[::2, h - 1] = torch.randint(0, K, (prefix[::2].shape[0],), device="cuda", generator=g)
            t = timed(lambda: ops.prefix_lookup(index, corpus, prefix))
            line = f"    lookup h={h} P={P:,}: {t * 1e6:8.1f} us  ({P / t / 1e9:.2f} G prefixes/s)"
            if N * P <= 4e9:
                ref = torch_check_valid_prefix(corpus, prefix)
                assert torch.equal(ref, ops.prefix_lookup(index, corpus, prefix))
                t_ref = timed(lambda: torch_check_valid_prefix(corpus, prefix), reps=3, warm=1)
                line += f"   torch formulation {t_ref * 1e3:8.2f} ms  (x{t_ref / t:,.0f})"
            print(line)
    for B, Kc, D in ((256, 10, 3), (100_000, 10, 3), (1_000_000, 10, 4)):
        actual = torch.randint(0, 8, (B, D), device="cuda", generator=g)
        top_k = torch.randint(0, 8, (B, Kc, D), device="cuda", generator=g)
        t = timed(lambda: ops.topk_first_match(actual, top_k))

        def ref():
            pos = (actual[:, None, :] == top_k).all(-1)
            return pos.max(-1)
        found, rank = ref()
        got = ops.topk_first_match(actual, top_k)
        assert torch.equal(torch.where(found, rank, torch.full_like(rank, -1)), got)
        t_ref = timed(ref)
        nbytes = (B * Kc * D + B * D + B) * 8
        print(f"topk_first_match B={B:>9,} K={Kc} D={D}: {t * 1e6:8.1f} us ({nbytes / t / 1e9:7.1f} GB/s)   "
              f"torch formulation {t_ref * 1e6:8.1f} us")


if __name__ == "__main__":
    main()
