#!/usr/bin/env python3
"""A/B of the three forms of the forward scan (BASELINE.json north_star "MFMA only if it wins
over the LDS path"): filtered bf16-split scan (the product default at D = 32 / 64), all-fp32 MFMA scan, LDS / VALU scan.
Times each with HIP events around batches of launches; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel
table (tools/profile_scan_ab.sh).  Prints one JSON line per shape."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
if os.environ.get("RQ_LIB"):   # another build of librqhip.so (tools/ab_build.sh), loaded before the first op
    from rqhip import _lib  # noqa: E402
    _lib.load(os.path.abspath(os.environ["RQ_LIB"]))
from rqhip import ops  # noqa: E402

PEAK_F32 = 157.3


def time_scan(x, cb, mode, scan, reps=30, warm=5, **kw):
    f = lambda: ops.rq_forward(x, cb, mode, 0.25, want_embs=False, want_residuals=False, scan=scan, **kw)  # noqa: E731
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best   # us per call (csq prologue + main kernel)


def main():
    g = torch.Generator().manual_seed(0)
    shapes = [("c2", 100_000, 32, 256, 3), ("c2_1M", 1_000_000, 32, 256, 3), ("c4_micro", 125_000, 32, 1024, 4),
              ("batch640", 640, 32, 256, 3), ("d64", 100_000, 64, 256, 3)]
    only = set(sys.argv[1:])
    for name, B, D, K, L in shapes:
        if only and name not in only:
            continue
        x = (torch.randn(B, D, generator=g) * 0.5).cuda()
        cb = torch.stack([x[torch.randperm(B, generator=g)[:K].cuda()] / (l + 1) + 0.02 * torch.randn(K, D, generator=g).cuda()
                          for l in range(L)]).contiguous()
        flops = B * L * (2 * D * K + 5 * D)
        row = {"shape": name, "B": B, "D": D, "K": K, "L": L}
        ref = None
        for scan in ("auto", "fp32", "valu"):
            if scan == "valu" and D != 32:
                continue
            us = time_scan(x, cb, ops.MODE_STE, scan)
            out = ops.rq_forward(x, cb, ops.MODE_STE, 0.25, want_embs=False, want_residuals=False, scan=scan)
            if ref is None:
                ref = out
            same = bool(torch.equal(out.ids, ref.ids) and torch.equal(out.loss.view(torch.int32), ref.loss.view(torch.int32)))
            row[scan] = {"us": round(us, 2), "tflops_algorithmic": round(flops / us / 1e6, 1),
                         "frac_fp32_peak": round(flops / us / 1e6 / PEAK_F32, 3), "same_bits_as_auto": same}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
