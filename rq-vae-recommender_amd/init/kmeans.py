"""k-means codebook initialisation on the MI355X (API of reference init/kmeans.py:8-72).

The Lloyd loop, the seeding (`np.random.choice`, numpy global RNG) and the reseeding of empty clusters
(`torch.randint`, torch global CPU RNG, one draw per empty cluster in ascending cluster order) stay on the
host exactly where the reference has them, so that the same seeds give the same run.  The two data-parallel
steps of an iteration are HIP kernels (csrc/kmeans.hip):

    assign  (kmeans.py:40-43)  fused distance + argmin, no B x K x D temporary
    update  (kmeans.py:44-59)  per-cluster mean in row order + the convergence statistic of kmeans.py:68

One 12-byte device->host read per iteration (empty-cluster flag + shift) replaces the reference's K-iteration
Python loop of masked means.
"""
from typing import NamedTuple, Optional

import numpy as np
import torch

from rqhip import ops


def kmeans_init_(tensor: torch.Tensor, x: torch.Tensor) -> None:
    """Overwrite `tensor` [K,D] with k-means centroids of `x` [B,D] (in place, no grad)."""
    assert tensor.dim() == 2
    assert x.dim() == 2
    with torch.no_grad():
        out = Kmeans(k=tensor.shape[0]).run(x)
        tensor.data.copy_(out.centroids)


class KmeansOutput(NamedTuple):
    centroids: torch.Tensor
    assignment: torch.Tensor


class Kmeans:
    def __init__(self, k: int, max_iters: Optional[int] = None, stop_threshold: float = 1e-10) -> None:
        self.k = k
        self.iters = max_iters
        self.stop_threshold = stop_threshold
        self.centroids = None
        self.assignment = None

    def _init_centroids(self, x: torch.Tensor) -> None:
        rows = np.random.choice(x.shape[0], self.k, replace=False)
        self.centroids = x[torch.as_tensor(rows, device=x.device)].to(torch.float32).contiguous()
        self.assignment = None

    def _update_centroids(self, x: torch.Tensor) -> float:
        """One Lloyd step (assign + update + reseed of empty clusters); returns the max centroid shift."""
        before = self.centroids.clone()
        assign = ops.kmeans_assign(x, self.centroids)
        counts, shift_sq = ops.kmeans_update(x, assign, self.centroids)
        any_empty, shift_sq = torch.stack([(counts == 0).any().to(torch.float32), shift_sq]).tolist()  # one sync
        shift = float(np.sqrt(np.float32(shift_sq)))
        if any_empty:
            if x.size(0) == 0:
                raise ValueError("Can not choose random element from x, x is empty")
            empty = (counts == 0).nonzero().flatten().tolist()  # ascending cluster order, as the reference's loop
            for cluster in empty:
                pick = int(torch.randint(0, x.size(0), (1,)))
                self.centroids[cluster] = x[pick]
            moved = self.centroids[empty] - before[empty]
            shift = max(shift, float(torch.linalg.vector_norm(moved, dim=1).max()))
        self.assignment = assign
        return shift

    def run(self, x: torch.Tensor) -> KmeansOutput:
        x = x.detach().to(torch.float32).contiguous()
        self._init_centroids(x)
        i = 0
        while self.iters is None or i < self.iters:
            if self._update_centroids(x) < self.stop_threshold:
                break
            i += 1
        return KmeansOutput(centroids=self.centroids, assignment=self.assignment)
