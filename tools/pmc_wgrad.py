"""Run the bf16-split weight-gradient kernel of the 512 x 768 layer a few times (for rocprofv3 --pmc passes; tools/pmc_wgrad.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import ops  # noqa: E402

M, N, K = 100_000, 512, 768
g = torch.Generator().manual_seed(0)
gy = torch.randn(M, N, generator=g).cuda()
y = torch.relu(torch.randn(M, N, generator=g)).cuda()
x = torch.randn(M, K, generator=g).cuda()
for _ in range(6):
    ops.linear_wgrad(gy, None, x)
    ops.linear_wgrad(gy, y, x)
torch.cuda.synchronize()
