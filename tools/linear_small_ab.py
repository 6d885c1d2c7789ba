#!/usr/bin/env python3
"""rqhip_linear_small (csrc/mlp_small.hip) against the library GEMM, layer by layer at the reference's batch sizes: device time per
launch from hipGraph replays of 20 back-to-back launches (what a replayed training step pays), for every launch plan.
  python tools/linear_small_ab.py [rows ...]      (default 640 64)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rqhip import _lib, ops, tuning  # noqa: E402

LIBS = [a for a in sys.argv[1:] if a.endswith(".so")]
if LIBS:
    _lib.load(LIBS[0])      # an A/B build (tools/ab_build.sh)
tuning.enable_tuned_gemms()
LAYERS = [(512, 768), (256, 512), (128, 256), (32, 128), (128, 32), (256, 128), (512, 256), (768, 512)]
REP = 20


def graph_us(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 50
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (n * REP) * 1e6


def main():
    rows = [int(a) for a in sys.argv[1:] if a.isdigit()] or [640, 64]
    plans = [(1, 4), (1, 8), (1, 16), (2, 4), (2, 8)]
    for M in rows:
        print(f"== {M} rows: us per launch (graph replay of {REP} launches); plan = (col_blocks, waves), * = the library's choice")
        tot = {"lib": 0.0, "auto": 0.0}
        for kind in ("forward", "dgrad"):
            for n_out, n_in in LAYERS:
                if kind == "dgrad" and n_in == 768:
                    continue
                w = torch.randn(n_out, n_in, device="cuda") * 0.05
                if kind == "forward":
                    a = torch.randn(M, n_in, device="cuda")
                    zb = torch.zeros(n_out, device="cuda")
                    lib = graph_us(lambda: torch._addmm_activation(zb, a, w.t()))
                    N, Kr, kn, epi, aux = n_out, n_in, False, _lib.EPI_RELU, None
                else:
                    a = torch.randn(M, n_out, device="cuda")
                    below = torch.randn(M, n_in, device="cuda")
                    lib = graph_us(lambda: torch.ops.aten.threshold_backward(a.mm(w), below, 0.0))
                    N, Kr, kn, epi, aux = n_in, n_out, True, _lib.EPI_MASK, below
                out = torch.empty(M, N, device="cuda")
                auto = ops.linear_small_plan(M, N, Kr)
                cells = []
                for cb, ks in plans:
                    if cb == 2 and N % 64:
                        cells.append("     -")
                        continue
                    us = graph_us(lambda: ops.linear_small(a, w, w_kn=kn, epilogue=epi, aux=aux, col_blocks=cb, waves=ks, out=out))
                    cells.append(f"{us:6.2f}" + ("*" if (cb, ks) == auto else " "))
                    if (cb, ks) == auto:
                        tot["auto"] += us
                tot["lib"] += lib
                print(f"  {kind:7s} {Kr:4d} -> {N:4d}: library{' + mask' if kind == 'dgrad' else ' (relu)'} {lib:6.2f} | " +
                      " ".join(f"{p}:{c}" for p, c in zip(plans, cells)))
        print(f"  sum over the 15 launches of a step: library {tot['lib']:.1f} us, rqhip_linear_small (auto plan) {tot['auto']:.1f} us")


if __name__ == "__main__":
    main()
