"""Run the C2-shaped rq_forward launch a few times (for rocprofv3 --pmc passes; see tools/pmc_forward.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")]
from rqhip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, 32, generator=g) * 0.5).cuda()
cb = (torch.randn(3, 256, 32, generator=g) * 0.3).cuda()
for _ in range(12):
    ops.rq_forward(x, cb, 1, 0.25, want_embs=False, want_residuals=False)
torch.cuda.synchronize()
