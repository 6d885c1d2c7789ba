#!/usr/bin/env python3
"""Kernel-level micro-benchmark (developer tool; not part of the judged bench.py contract).

Times the hand-written kernels in isolation with HIP events for a sweep of shapes and prints achieved
TFLOP/s / GB/s.  Usage:  python tools/bench_kernels.py [fwd|bwd|kmeans|all] [--reps N]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from rqhip import ops  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def fwd(reps, shapes=None):
    shapes = shapes or [(32768, 32, 256, 3), (65536, 32, 256, 3), (100000, 32, 256, 3), (131072, 32, 256, 3),
                        (262144, 32, 256, 3), (1048576, 32, 256, 3), (100000, 32, 256, 1), (100000, 64, 256, 3),
                        (262144, 32, 1024, 4), (640, 32, 256, 3), (8192, 32, 256, 3)]
    for B, D, K, L in shapes:
        g = torch.Generator().manual_seed(0)
        x = (torch.randn(B, D, generator=g) * 0.5).cuda()
        cb = (torch.randn(L, K, D, generator=g) * 0.3).cuda()
        for mode, name in ((1, "ste"), (0, "eval")):
            ops.profile_enable(reps + 8)
            ms_call = timeit(lambda: ops.rq_forward(x, cb, mode, 0.25, want_embs=False, want_residuals=False), reps)
            ks = ops.profile_read()
            ops.profile_enable(0)
            ms = sum(ks[-reps:]) / reps
            fl = B * L * (2 * D * K + 5 * D)
            print(f"fwd {name:4s} B={B:8d} D={D:3d} K={K:5d} L={L}: kernel {ms*1e3:9.1f} us  call {ms_call*1e3:9.1f} us  "
                  f"{fl/ms/1e9:7.1f} TFLOP/s  {B/ms/1e3:9.1f} M rows/s", flush=True)


def bwd(reps, shapes=None):
    for B, D, K, L in shapes or [(100000, 32, 256, 3), (1048576, 32, 256, 3), (262144, 32, 1024, 4), (125000, 32, 1024, 4),
                                 (8192, 32, 256, 3), (640, 32, 256, 3), (100000, 64, 256, 3)]:
        g = torch.Generator().manual_seed(0)
        x = (torch.randn(B, D, generator=g) * 0.5).cuda()
        cb = (torch.randn(L, K, D, generator=g) * 0.3).cuda()
        out = ops.rq_forward(x, cb, 1, 0.25, want_embs=False, want_residuals=False)
        ge = torch.randn(B, D, generator=g).cuda()
        gl = torch.full((B,), 1.0 / B).cuda()
        ms = timeit(lambda: ops.rq_backward(x, cb, 1, 0.25, out.ids, g_embsum=ge, g_loss=gl), reps)
        ms_m = timeit(lambda: ops.rq_backward(x, cb, 1, 0.25, out.ids, g_embsum=ge, g_loss=gl, cbgrad="matrix"), reps)
        print(f"    (codebook gradient as a one-hot matrix product: {ms_m * 1e3:.1f} us)")
        by = B * (12 * D + 8 * L)
        print(f"bwd ste  B={B:8d} D={D:3d} K={K:5d} L={L}: call {ms*1e3:9.1f} us  {by/ms/1e6:7.1f} GB/s algorithmic", flush=True)


def kmeans(reps):
    for B, D, K in [(20000, 32, 256), (20000, 64, 256), (20000, 32, 1024)]:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, D, generator=g).cuda()
        c = x[:K].clone()
        ms_a = timeit(lambda: ops.kmeans_assign(x, c), reps)
        a = ops.kmeans_assign(x, c)
        ms_u = timeit(lambda: ops.kmeans_update(x, a, c.clone()), reps)
        print(f"kmeans B={B} D={D} K={K}: assign {ms_a*1e3:8.1f} us  update {ms_u*1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--one", type=str, default=None, help="B,D,K,L for a single forward shape")
    ap.add_argument("--lib", type=str, default=None, help="another build of librqhip.so (tools/ab_build.sh) instead of the in-tree one")
    a = ap.parse_args()
    if a.lib:
        from rqhip import _lib
        _lib.load(os.path.abspath(a.lib))
        print(f"# library: {a.lib}")
    if a.one and a.what == "bwd":
        bwd(a.reps, [tuple(int(v) for v in a.one.split(","))])
    elif a.one:
        fwd(a.reps, [tuple(int(v) for v in a.one.split(","))])
    else:
        if a.what in ("fwd", "all"):
            fwd(a.reps)
        if a.what in ("bwd", "all"):
            bwd(a.reps)
        if a.what in ("kmeans", "all"):
            kmeans(a.reps)

# tie-margin variant of the forward kernel (tokenisation with flags): what the runner-up tournament costs
if __name__ == "__main__" and a.what in ("fwd", "all") and not a.one:
    import torch as _t
    from rqhip import ops as _ops
    _g = _t.Generator().manual_seed(0)
    _x = (_t.randn(100_000, 32, generator=_g) * 0.5).cuda()
    _cb = (_t.randn(3, 256, 32, generator=_g) * 0.3).cuda()

    def _time(fn, n=20):
        for _ in range(3):
            fn()
        a, b = _t.cuda.Event(enable_timing=True), _t.cuda.Event(enable_timing=True)
        _t.cuda.synchronize(); a.record()
        for _ in range(n):
            fn()
        b.record(); _t.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    t0 = _time(lambda: _ops.rq_forward(_x, _cb, 0, 0.25, want_embs=False, want_residuals=False))
    t1 = _time(lambda: _ops.rq_forward(_x, _cb, 0, 0.25, want_embs=False, want_residuals=False, want_margin=True))
    print(f"fwd eval B=100000 3x256: call {t0:.1f} us plain, {t1:.1f} us with tie_margin (+{(t1 / t0 - 1) * 100:.0f} %)")
