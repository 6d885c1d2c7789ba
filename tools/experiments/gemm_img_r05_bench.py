#!/usr/bin/env python3
"""Round-5 kernel timings at 100 000 rows: the image kernels (csrc/gemm_img.hip) beside round 4's (csrc/gemm_split.hip, wgrad_split.hip),
back to back, HIP events.  `python tools/bench_img.py [rows]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rqhip import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
g = torch.Generator().manual_seed(0)


def t(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"rows {M}")
for Nc, R, epi in ((512, 768, 1), (256, 512, 1), (128, 256, 1), (256, 128, 1), (512, 256, 1), (768, 512, 2), (512, 768, 3), (256, 512, 3), (128, 256, 3), (256, 128, 3), (512, 256, 3)):
    a = torch.relu(torch.randn(M, R, generator=g)).cuda()
    w = (torch.randn(Nc, R, generator=g) / R ** 0.5).cuda()
    x = torch.randn(M, Nc, generator=g).cuda()
    img_w = ops.weight_planes(w, arith=ops.F16X2)
    rows = ops.maxima(a, cols=False)[0]
    ia = ops.img_pack(a, want_t=False)[0]
    iy = ops.img_pack(torch.relu(x))[0]
    kw4 = dict(epilogue=epi, a_row_max=rows, want_row_max=True, col_max_out=torch.zeros(Nc, dtype=torch.int32, device="cuda"))
    if epi >= 2:
        kw4["aux"] = x
    if epi == 2:
        kw4["row_scale"] = 1.0 / M
    old = t(lambda: ops.gemm_split_ex(a, img_w, Nc, **kw4))
    kw5 = dict(epilogue=epi)
    if epi == 2:
        kw5.update(aux=x, row_scale=1.0 / M)
    if epi == 3:
        kw5.update(y=iy)
    res = {}
    for name, extra in (("C only", dict(want_c=True)), ("R+T", dict(want_r=True, want_t=True)), ("R only", dict(want_r=True)),
                        ("R+T xcd", dict(want_r=True, want_t=True, xcd_queues=True))):
        res[name] = t(lambda: ops.gemm_img(ia, img_w, Nc, **kw5, **extra))
    print(f"gemm {R:4d} -> {Nc:4d} epi {epi}: round 4 {old:7.1f} us | img: " + " | ".join(f"{k} {v:7.1f}" for k, v in res.items()))

for N, K in ((512, 768), (256, 512), (128, 256), (256, 128), (512, 256), (768, 512)):
    gy = (torch.randn(M, N, generator=g) * (torch.rand(M, N, generator=g) > 0.5)).cuda()
    x = torch.randn(M, K, generator=g).cuda()
    gc, xc = ops.maxima(gy, rows=False)[1], ops.maxima(x, rows=False)[1]
    old = t(lambda: ops.linear_wgrad(gy, None, x, g_col_max=gc, x_col_max=xc))
    ig, ix = ops.img_pack(gy, want_r=False)[0], ops.img_pack(x, want_r=False)[0]
    new = t(lambda: ops.linear_wgrad_img(ig, ix))
    print(f"wgrad dW[{N},{K}]: round 4 {old:7.1f} us | img {new:7.1f} us")

for N in (768, 512, 128):
    a = torch.randn(M, N, generator=g).cuda()
    print(f"pack [{M},{N}]: R+T {t(lambda: ops.img_pack(a)):7.1f} us | R {t(lambda: ops.img_pack(a, want_t=False)):7.1f} | T {t(lambda: ops.img_pack(a, want_r=False)):7.1f} "
          f"| maxima rows+cols {t(lambda: ops.maxima(a)):7.1f}")
