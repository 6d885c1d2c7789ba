#!/usr/bin/env python3
"""Times csrc/gemm_split.hip against the tuned library fp32 GEMMs of the same MLP layers at 100 000 rows
(forward with / without the ReLU epilogue, data gradient).  Usage (GPU box): python tools/bench_gemm_split.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import ops, tuning  # noqa: E402

tuning.enable_tuned_gemms()
torch.set_float32_matmul_precision("highest")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
if len(sys.argv) > 2:   # another build of librqhip.so (tools/ab_build.sh)
    from rqhip import _lib
    _lib.load(os.path.abspath(sys.argv[2]))
    print(f"# library: {sys.argv[2]}")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"{'layer W [N,K]':>14} {'GFLOP':>6} | {'fwd lib(relu)':>13} {'fwd split 256/auto':>17} | {'dgrad lib':>9} {'dgrad split 256/auto':>19} | planes us")
for N, K in [(512, 768), (256, 512), (512, 256), (768, 512), (128, 256), (256, 128), (128, 32)]:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    g = torch.randn(M, N, device="cuda")
    zb = torch.zeros(N, device="cuda")
    gf = 2.0 * M * N * K / 1e9
    t_lib_f = timeit(lambda: torch._addmm_activation(zb, x, w.t()))
    pb = None
    if ops.gemm_split_supported(N, K):
        pf = ops.weight_planes(w)
        t_planes = timeit(lambda: ops.weight_planes(w))
        t_spl_f = timeit(lambda: ops.gemm_split(x, pf, N, relu=True, tile_rows=256))
        t_spl_f2 = timeit(lambda: ops.gemm_split(x, pf, N, relu=True))
    else:
        t_planes = t_spl_f = t_spl_f2 = float("nan")
    t_lib_b = timeit(lambda: g.mm(w))
    if ops.gemm_split_supported(K, N):
        pb = ops.weight_planes(w, transpose=True)
        t_spl_b = timeit(lambda: ops.gemm_split(g, pb, K, tile_rows=256))
        t_spl_b2 = timeit(lambda: ops.gemm_split(g, pb, K))
    else:
        t_spl_b = t_spl_b2 = float("nan")
    print(f"{str((N, K)):>14} {gf:6.1f} | {t_lib_f:13.1f} {t_spl_f:8.1f}/{t_spl_f2:8.1f} | {t_lib_b:9.1f} {t_spl_b:9.1f}/{t_spl_b2:9.1f} | {t_planes:6.1f}")
