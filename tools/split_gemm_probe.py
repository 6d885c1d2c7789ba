#!/usr/bin/env python3
"""Ceiling probe for VERDICT r2 item 5 (leave the fp32 matrix pipe for the MLP GEMMs): how fast does the LIBRARY run the
six-term bf16-split product, and how exact is it?

X = xh + xm + xl, W = wh + wm + wl (bf16 pieces, 24 mantissa bits kept), X W^T ~ hh + hm + mh + hl + lh + mm: the six
products are concatenated along K so that ONE bf16 GEMM with fp32 accumulation ([M, 6K] x [6K, N]) forms the sum inside
the matrix pipe.  A hand-written kernel can at best approach the rate the library reaches on this shape; the fp32 library
GEMM of the same layer is timed beside it.  Error is measured against fp64 on a row sample."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import tuning  # noqa: E402


def split3(v):
    h = v.bfloat16()
    r = v - h.float()
    m = r.bfloat16()
    l = (r - m.float()).bfloat16()
    return h, m, l


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    tuning.enable_tuned_gemms()
    torch.manual_seed(0)
    M = 100_000
    for N, K in ((512, 768), (768, 512), (256, 512)):
        X = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") / K ** 0.5
        xh, xm, xl = split3(X)
        wh, wm, wl = split3(W)
        A6 = torch.cat([xh, xh, xm, xh, xl, xm], dim=1).contiguous()
        B6 = torch.cat([wh, wm, wh, wl, wh, wm], dim=1).contiguous()
        A3 = torch.cat([xh, xh, xm], dim=1).contiguous()
        B3 = torch.cat([wh, wm, wh], dim=1).contiguous()
        row = {"M": M, "N": N, "K": K}
        t32 = timeit(lambda: X @ W.t())
        row["fp32_library_us"] = round(t32, 1)
        row["fp32_tflops"] = round(2 * M * N * K / t32 / 1e6, 1)
        try:
            t6 = timeit(lambda: torch.mm(A6, B6.t(), out_dtype=torch.float32))
            t3 = timeit(lambda: torch.mm(A3, B3.t(), out_dtype=torch.float32))
            t1 = timeit(lambda: torch.mm(xh, wh.t(), out_dtype=torch.float32))
            row.update({"split6_library_us": round(t6, 1), "split6_bf16_tflops": round(2 * M * N * 6 * K / t6 / 1e6, 1),
                        "split3_library_us": round(t3, 1), "bf16_1term_us": round(t1, 1),
                        "bf16_1term_tflops": round(2 * M * N * K / t1 / 1e6, 1),
                        "split6_speedup_vs_fp32": round(t32 / t6, 3)})
            # split cost if it were a separate pass (three bf16 planes from fp32)
            row["split_pass_us"] = round(timeit(lambda: split3(X)), 1)
            idx = torch.randperm(M, device="cuda")[:2048]
            ref = X[idx].double() @ W.double().t()
            e6 = (torch.mm(A6[idx], B6.t(), out_dtype=torch.float32).double() - ref).abs().max().item()
            e32 = ((X[idx] @ W.t()).double() - ref).abs().max().item()
            row.update({"max_abs_err_split6": float(f"{e6:.3e}"), "max_abs_err_fp32_library": float(f"{e32:.3e}"),
                        "ref_scale": round(ref.abs().max().item(), 3)})
        except Exception as exc:  # noqa: BLE001
            row["error"] = f"{type(exc).__name__}: {str(exc)[:200]}"
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
