"""Times csrc/wgrad.hip against what it replaces (aten threshold_backward + the library weight-gradient GEMM with the
committed TunableOp selection) for every layer of the 768-512-256-128-32 MLPs at 100 000 rows.
Usage (GPU box): python tools/bench_wgrad.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import ops, tuning  # noqa: E402

tuning.enable_tuned_gemms()
torch.set_float32_matmul_precision("highest")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000


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


print(f"{'layer dW [N,K]':>16} {'GFLOP':>7} | {'mask us':>8} {'lib gemm us':>11} {'lib TF':>7} | {'hip masked us':>13} {'hip TF':>7} {'hip plain us':>12} | {'fp32-kernel masked':>18} {'plain':>8}")
tot_lib = tot_hip = 0.0
for N, K in [(512, 768), (256, 512), (128, 256), (32, 128), (128, 32), (256, 128), (512, 256), (768, 512)]:
    gy = torch.randn(M, N, device="cuda")
    y = torch.relu(torch.randn(M, N, device="cuda"))
    x = torch.randn(M, K, device="cuda")
    gf = 2.0 * M * N * K / 1e9
    t_mask = timeit(lambda: torch.ops.aten.threshold_backward(gy, y, 0.0))
    g = torch.ops.aten.threshold_backward(gy, y, 0.0)
    t_lib = timeit(lambda: g.t().mm(x))
    t_hip = timeit(lambda: ops.linear_wgrad(gy, y, x))
    t_hip_plain = timeit(lambda: ops.linear_wgrad(gy, None, x))
    t_f32 = timeit(lambda: ops.linear_wgrad(gy, y, x, exact_fp32=True))
    t_f32_plain = timeit(lambda: ops.linear_wgrad(gy, None, x, exact_fp32=True))
    masked = (N, K) not in ((32, 128), (768, 512))
    tot_lib += t_lib + (t_mask if masked else 0)
    tot_hip += t_hip if masked else t_hip_plain
    print(f"{str((N, K)):>16} {gf:7.1f} | {t_mask:8.1f} {t_lib:11.1f} {gf / t_lib * 1e3 / 1e3:7.1f} | {t_hip:13.1f} "
          f"{gf / t_hip * 1e3 / 1e3:7.1f} {t_hip_plain:12.1f} | {t_f32:18.1f} {t_f32_plain:8.1f}")
print(f"per training step (6 masked + 2 plain layers): library {tot_lib:.0f} us -> hip {tot_hip:.0f} us")
