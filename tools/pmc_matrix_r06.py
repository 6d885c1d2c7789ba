#!/usr/bin/env python3
"""Driver for tools/pmc_matrix_r06.sh: ten launches each of the product GEMM (768 -> 512, ReLU, f16x2, 100 000 rows) and of the batched
weight gradient dW[512, 768] + dW[256, 512] (rqhip_linear_wgrad_f16_batch), back to back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
import torch  # noqa: E402
from rqhip import _lib, ops  # noqa: E402

M = 100_000
x = torch.relu(torch.randn(M, 768, device="cuda"))
w = torch.randn(512, 768, device="cuda") / 768 ** 0.5
img = ops.weight_planes(w, arith=ops.F16X2)
rows = ops.maxima(x, cols=False)[0]
g = torch.randn(M, 512, device="cuda") * (torch.rand(M, 512, device="cuda") > 0.5)
gc, xc = ops.maxima(g, rows=False)[1], ops.maxima(x, rows=False)[1]
g2 = torch.randn(M, 256, device="cuda") * (torch.rand(M, 256, device="cuda") > 0.5)
x2 = torch.relu(torch.randn(M, 512, device="cuda"))
gc2, xc2 = ops.maxima(g2, rows=False)[1], ops.maxima(x2, rows=False)[1]
for _ in range(10):
    ops.gemm_split_ex(x, img, 512, epilogue=_lib.EPI_RELU, a_row_max=rows, want_row_max=True,
                      col_max_out=torch.zeros(512, dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()
for _ in range(10):
    ops.linear_wgrad_f16_batch([(g, x, gc, xc), (g2, x2, gc2, xc2)])
torch.cuda.synchronize()
