"""k-means codebook initialisation on the MI355X (API of reference init/kmeans.py:8-72).

The Lloyd loop, the seeding (`np.random.choice`, numpy global RNG) and the reseeding of empty clusters
(`torch.randint`, torch global CPU RNG, one draw per empty cluster in ascending cluster order) stay on the
host exactly where the reference has them, so that the same seeds give the same run.  The two data-parallel
steps of an iteration are HIP kernels (csrc/kmeans.hip):

    assign  (kmeans.py:40-43)  fused distance + argmin, no B x K x D temporary
    update  (kmeans.py:44-59)  per-cluster mean in row order + the convergence statistic of kmeans.py:68

`Kmeans.run` drives them in BATCHES: `rqhip_kmeans_lloyd` enqueues up to 16 iterations whose kernels stop by
themselves (device flag) when the run converges or an empty cluster needs the host's RNG; the host reads one
16-byte state per batch.  The per-iteration host loop this replaces cost ~30x the kernels' time (1.05 s for the
three levels of the bench warm-up; now ~0.05 s).
"""
from typing import NamedTuple, Optional

import numpy as np
import torch

import torch.distributed as dist

from rqhip import ops, wide


def kmeans_init_(tensor: torch.Tensor, x: torch.Tensor, rows_sharded: bool = False) -> None:
    """Overwrite `tensor` [K,D] with k-means centroids of `x` [B,D] (in place, no grad).

    rows_sharded (not in the reference's signature; default = its behaviour): `x` is THIS RANK'S BLOCK of the rows and
    the Lloyd loop is the row-sharded one -- one all-reduce of [K, D+1] sums || counts per iteration (SURVEY.md section
    8e).  Quantize passes its `kmeans_rows_sharded` attribute, which train_rqvae sets for the warm-up forward when it
    feeds every rank its own slice of the first 20 000 items."""
    assert tensor.dim() == 2
    assert x.dim() == 2
    with torch.no_grad():
        sharded = rows_sharded and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        out = Kmeans(k=tensor.shape[0]).run(x, sharded=sharded)
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
        if wide.kmeans_covers(x.shape[1]):
            assign = ops.kmeans_assign(x, self.centroids)
            counts, shift_sq = ops.kmeans_update(x, assign, self.centroids)
        else:   # D > 128: the same step as PyTorch-ROCm operators (rqhip/wide.py)
            assign, counts = wide.kmeans_update(x, self.centroids)
            shift_sq = ((self.centroids - before) ** 2).sum(dim=1).max()
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

    BATCH = 16  # Lloyd iterations enqueued per host visit

    def run(self, x: torch.Tensor, sharded: bool = False) -> KmeansOutput:
        if sharded:
            if not wide.kmeans_covers(x.shape[1]):
                return self._run_gathered(x)
            return self._run_sharded(x)
        x = x.detach().to(torch.float32).contiguous()
        self._init_centroids(x)
        if not wide.kmeans_covers(x.shape[1]):
            return self._run_stepwise(x)
        B = x.shape[0]
        dev = x.device
        assign = torch.empty((B,), dtype=torch.int64, device=dev)
        counts = torch.empty((self.k,), dtype=torch.int64, device=dev)
        state = torch.zeros((4,), dtype=torch.int32, device=dev)   # see include/rqhip.h: rqhip_kmeans_lloyd
        done = 0   # iterations executed (the reference's calls of _update_centroids)
        i = 0      # the reference's loop counter: iterations that did not converge (kmeans.py:64-70)
        while self.iters is None or i < self.iters:
            n = self.BATCH if self.iters is None else min(self.BATCH, self.iters - i)
            ops.kmeans_lloyd(x, self.centroids, assign, counts, state, n, self.stop_threshold)
            stop, now_done, shift_bits, _ = state.tolist()            # the one sync of this batch
            ran, done = now_done - done, now_done
            if stop == 1:        # converged: the reference breaks without counting the iteration
                break
            if stop == 2:        # the last executed iteration met empty clusters: reseed them here, in ascending
                i += ran - 1     # cluster order, with the host's torch RNG stream (kmeans.py:48-54)
                shift = float(np.sqrt(np.array([shift_bits], dtype=np.int32).view(np.float32)[0]))
                empty = (counts == 0).nonzero().flatten().tolist()
                before = self.centroids[empty].clone()      # the device leaves empty clusters' centroids untouched
                for cluster in empty:
                    pick = int(torch.randint(0, B, (1,)))
                    self.centroids[cluster] = x[pick]
                moved = self.centroids[empty] - before
                shift = max(shift, float(torch.linalg.vector_norm(moved, dim=1).max()))
                if shift < self.stop_threshold:
                    break
                i += 1
                state[0] = 0
                continue
            i += ran             # the whole batch ran without converging
        self.assignment = assign
        return KmeansOutput(centroids=self.centroids, assignment=self.assignment)

    def _run_stepwise(self, x: torch.Tensor) -> KmeansOutput:
        """The reference's loop (kmeans.py:61-72), one `_update_centroids` per iteration: latent widths the batched kernels do
        not take (D > 128)."""
        wide._note(f"Kmeans on {x.shape[1]}-wide rows")
        if x.shape[0] == 0:
            raise ValueError("Can not choose random element from x, x is empty")
        i = 0
        while self.iters is None or i < self.iters:
            if self._update_centroids(x) < self.stop_threshold:
                break
            i += 1
        return KmeansOutput(centroids=self.centroids, assignment=self.assignment)

    # ---- row-sharded run (one process per GPU, torch.distributed) ---------------------------------------------
    def _rows_from_all_ranks(self, x: torch.Tensor, picks, lo: int) -> torch.Tensor:
        """Global rows `picks` ([n] ints, identical on every rank) -> [n, D] on every rank: the owner of a row puts
        it into a zero buffer, one all-reduce(sum) assembles the set (x + 0 is exact)."""
        buf = torch.zeros((len(picks), x.shape[1]), dtype=torch.float32, device=x.device)
        mine = [(j, p - lo) for j, p in enumerate(picks) if lo <= p < lo + x.shape[0]]
        if mine:
            j, r = zip(*mine)
            buf[torch.as_tensor(j, device=x.device)] = x[torch.as_tensor(r, device=x.device)]
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return buf

    def _run_gathered(self, x: torch.Tensor) -> KmeansOutput:
        """Row-sharded call on latent widths the HIP k-means kernels do not take (D > 128; rqhip/wide.py): the warm-up rows (at most
        20 000) are all-gathered and every rank runs the same stepwise loop on the whole matrix under ONE seed drawn by rank 0 -- same
        data, same draws, same operators: identical codebooks on every rank (this configuration trained on one GPU and
        raised on several)."""
        from rqhip.dist import allgather_rows
        x = x.detach().to(torch.float32).contiguous()
        rank = dist.get_rank()
        sizes = torch.zeros((dist.get_world_size(),), dtype=torch.int64, device=x.device)
        sizes[rank] = x.shape[0]
        dist.all_reduce(sizes)
        lo = int(sizes[:rank].sum())
        box = [int(torch.randint(0, 2 ** 31 - 1, (1,))) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        full = allgather_rows(x)
        np_state = np.random.get_state()
        with torch.random.fork_rng(devices=[x.device] if x.is_cuda else []):
            torch.manual_seed(box[0])
            np.random.seed(box[0] % (2 ** 32))
            try:
                self._init_centroids(full)
                out = self._run_stepwise(full)
            finally:
                np.random.set_state(np_state)
        self.assignment = out.assignment[lo:lo + x.shape[0]]
        return KmeansOutput(centroids=self.centroids, assignment=self.assignment)

    def _run_sharded(self, x: torch.Tensor) -> KmeansOutput:
        """`x` is this rank's block of rows (blocks in rank order form the global matrix).  Rank 0 owns both RNG
        streams -- `np.random.choice` for the seed rows, `torch.randint` for reseeds (kmeans.py:35,53) -- and broadcasts
        the drawn row numbers; per iteration the ranks exchange ONE all-reduce of the [K, D+1] sums || counts, after
        which every rank computes the same centroids, shift and stop flags.  Sums are formed per rank and then across
        ranks, so centroids equal a single-GPU run's to fp32 rounding, not bit for bit."""
        x = x.detach().to(torch.float32).contiguous()
        dev, rank, world = x.device, dist.get_rank(), dist.get_world_size()
        sizes = torch.zeros((world,), dtype=torch.int64, device=dev)
        sizes[rank] = x.shape[0]
        dist.all_reduce(sizes)
        sizes = sizes.tolist()
        lo, B = sum(sizes[:rank]), sum(sizes)

        def draw(fn):  # rank 0 draws, everyone gets the same list
            box = [fn() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        if B == 0:
            raise ValueError("Can not choose random element from x, x is empty")
        seeds = draw(lambda: np.random.choice(B, self.k, replace=False).tolist())
        self.centroids = self._rows_from_all_ranks(x, seeds, lo).contiguous()
        assign = torch.empty((x.shape[0],), dtype=torch.int64, device=dev)
        counts = torch.empty((self.k,), dtype=torch.int64, device=dev)
        sums = torch.empty((self.k, x.shape[1] + 1), dtype=torch.float32, device=dev)
        state = torch.zeros((4,), dtype=torch.int32, device=dev)
        done = i = 0
        while self.iters is None or i < self.iters:
            n = self.BATCH if self.iters is None else min(self.BATCH, self.iters - i)
            for _ in range(n):   # enqueued back to back: the collective runs on the stream too (RCCL)
                ops.kmeans_partial_sums(x, self.centroids, assign, sums, state)
                dist.all_reduce(sums, op=dist.ReduceOp.SUM)
                ops.kmeans_apply_sums(sums, self.centroids, counts, state, self.stop_threshold)
            stop, now_done, shift_bits, _ = state.tolist()
            ran, done = now_done - done, now_done
            if stop == 1:
                break
            if stop == 2:
                i += ran - 1
                shift = float(np.sqrt(np.array([shift_bits], dtype=np.int32).view(np.float32)[0]))
                empty = (counts == 0).nonzero().flatten().tolist()
                picks = draw(lambda: [int(torch.randint(0, B, (1,))) for _ in empty])
                before = self.centroids[empty].clone()
                self.centroids[empty] = self._rows_from_all_ranks(x, picks, lo)
                moved = self.centroids[empty] - before
                shift = max(shift, float(torch.linalg.vector_norm(moved, dim=1).max()))
                if shift < self.stop_threshold:
                    break
                i += 1
                state[0] = 0
                continue
            i += ran
        self.assignment = assign
        return KmeansOutput(centroids=self.centroids, assignment=self.assignment)
