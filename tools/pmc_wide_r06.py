#!/usr/bin/env python3
"""Driver for tools/pmc_wide_r06.sh: ten launches each of the 768 -> 512 ReLU GEMM (f16x2, 100 000 rows) on the product tile (128 x 256, two
workgroups per CU: kernel gemm_f16_kernel<1, 256>) and on the full-width tile (128 x 512, one workgroup per CU: gemm_f16_kernel<1, 512>)."""
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
for tr in (0, -5):
    for _ in range(10):
        ops.gemm_split_ex(x, img, 512, epilogue=_lib.EPI_RELU, a_row_max=rows, want_row_max=True,
                          col_max_out=torch.zeros(512, dtype=torch.int32, device="cuda"), tile_rows=tr)
    torch.cuda.synchronize()
