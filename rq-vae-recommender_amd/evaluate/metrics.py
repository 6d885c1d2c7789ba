"""Retrieval metrics over generated semantic ids, API of the reference's evaluate/metrics.py.

`TopKAccumulator.accumulate` needs, per row, the position of the first generated tuple equal to the target
tuple.  The reference builds a [B, K, D] equality tensor, reduces it and synchronises three times per call
(`.item()`, two `len(...)`, metrics.py:16-25).  Here the positions come from one HIP kernel
(csrc/sid_match.hip, rqhip_topk_first_match); the per-batch sums stay on the device and are read once, in
`reduce()`.
"""
from collections import defaultdict
from typing import Dict, List

import torch
from torch import Tensor

from rqhip import ops


class TopKAccumulator:
    def __init__(self, ks: List[int] = [1, 5, 10]) -> None:
        self.ks = ks
        self.reset()

    def reset(self) -> None:
        self.total = 0
        self.metrics = defaultdict(int)  # name -> device scalar (float64 / int64) until reduce()

    def accumulate(self, actual: Tensor, top_k: Tensor) -> None:
        """actual [B, D] int64 target ids, top_k [B, K, D] int64 generated ids, best first."""
        B, D = actual.shape
        rank = ops.topk_first_match(actual, top_k)  # [B]: first matching position or -1
        found = rank >= 0
        # 1 / log2(rank + 2), evaluated in fp32 like the reference (metrics.py:20), summed in fp64
        gain = 1.0 / torch.log2(rank.clamp_min(0).to(torch.float32) + 2.0)
        self.metrics["ndcg"] = self.metrics["ndcg"] + torch.where(found, gain, torch.zeros_like(gain)).double().sum()
        for k in self.ks:
            self.metrics[f"h@{k}"] = self.metrics[f"h@{k}"] + (found & (rank < k)).sum()
        self.total += B

    def reduce(self) -> Dict[str, float]:
        out = {}
        for name, v in self.metrics.items():
            out[name] = (v.item() if isinstance(v, Tensor) else v) / self.total
        return out
