"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL/xGMI ("nccl" backend on ROCm).

The path shards by rows (items are independent through encoder, every RQ level, decoder and losses,
SURVEY.md section 8e).  The only data-path exchange of a training step is the gradient reduction that HF
accelerate's DDP wrap performs implicitly in the reference (train_rqvae.py:153,195).  Here it is explicit
and MI355X-shaped: after backward all gradients are packed into ONE flat fp32 buffer (4.6 MB for the Amazon
config) with a single kernel, so a step issues exactly one in-place all-reduce -- the payload is far below
the xGMI bandwidth regime, what matters is one collective instead of a bucket per layer -- followed by a 1/W
scale (DDP's mean); the optimizer then reads views of that buffer.  The part of the buffer whose gradients are complete before
the encoder's backward starts (decoder weights, codebooks: half of it) is reduced UNDER the encoder's backward: `arm()` before the
step's last backward, a tensor hook on the encoder's output launches those all-reduces asynchronously, `allreduce_mean()` waits
for them and reduces the rest.  The k-means init is row-sharded too: every rank
takes its block of the first <= 20 000 rows through the model, rank 0 draws the seed / reseed row numbers and
broadcasts them, and each Lloyd iteration is one all-reduce of the [K, D+1] sums || counts (init/kmeans.py) -- which
also removes the reference's latent per-rank-divergent init.

On CPU (tests) the same code runs over the gloo backend.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn


def env_world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(device_type: str = "cuda", backend: Optional[str] = None, force: bool = False,
                  device_index: Optional[int] = None) -> tuple[int, int, int]:
    """Initialise the default process group when launched with WORLD_SIZE > 1; returns (rank, local, world).

    backend: default "nccl" (= RCCL) for cuda, "gloo" for cpu.  "gloo" with device tensors is what the tests use to run
    TWO ranks of the product step on ONE GPU (RCCL refuses two ranks on one device; gloo stages through the host).
    force: create the group even for a single rank (exercises the process-group code path).
    device_index: the GPU this rank uses (default LOCAL_RANK)."""
    rank, local_rank, world = env_world()
    dev = local_rank if device_index is None else device_index
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            torch.cuda.set_device(dev)
        if device_type == "cuda" and backend == "nccl":
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, dev, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def shard_bounds(n_rows: int, rank: Optional[int] = None, world: Optional[int] = None) -> tuple[int, int]:
    """Contiguous row shard [lo, hi) of rank r: sizes differ by at most one row."""
    rank = get_rank() if rank is None else rank
    world = world_size() if world is None else world
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@torch.no_grad()
def broadcast_module(module: nn.Module, src: int = 0) -> None:
    """Make every rank's parameters and buffers identical to rank `src`'s (after k-means init / at start)."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


class FlatGradReducer:
    """One collective per step for all gradients, without per-parameter accumulation kernels and without a packing copy.

    Usage per step:  reducer.zero_()  ->  loss.backward() (any number of times)  ->  reducer.allreduce_mean()
    ->  optimizer.step().

    `zero_()` drops the .grad tensors, so autograd ASSIGNS fresh gradients (no `grad += new` kernel per parameter;
    those 11 small adds were 0.37 ms of a 6.8 ms step on MI355X).  `attach(model)` tells the backward functions of
    this package (modules/encoder.py, rqhip/autograd.py) where each parameter's slice of the flat fp32 buffer is: they
    then WRITE the step's first gradient of every weight / codebook straight into it and hand autograd an alias, so
    `.grad` already lives in the buffer when backward ends -- `allreduce_mean()` is the bare all-reduce + 1/W scale.
    Gradients produced by other code (a parameter this package does not know) are packed the old way, with one
    `torch.cat(out=...)`.  With one rank nothing is exchanged.
    """

    def __init__(self, params: Iterable[nn.Parameter]) -> None:
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dtype = self.params[0].device, self.params[0].dtype
        # every slice starts on a 16-byte boundary (the kernels that read the buffer -- csrc/adamw.hip, the weight-gradient kernels'
        # sinks -- use 16-byte accesses): a parameter whose numel is not a multiple of 4 is followed by up to 3 floats of padding,
        # which stay zero and ride along in the all-reduce
        self._padded = [(p.numel() + 3) // 4 * 4 for p in self.params]
        self.flat = torch.zeros(sum(self._padded), device=dev, dtype=dtype)
        self._views = []
        self._offsets = []
        self.epoch = 0      # bumped by zero_(): a slice may be written in place by ONE backward node per epoch
        # overlap of the all-reduce with the encoder's backward (see arm / boundary_hook)
        self._early_params: List[nn.Parameter] = []
        self._early_runs: List[tuple] = []      # [lo, hi) element ranges of the flat buffer complete at the boundary
        self._late_runs: List[tuple] = []       # the rest
        self._armed = False
        self._pending: list = []
        self.overlap_launches = 0               # (tests / logs) early launches so far
        self._timing = None                     # enable_timing(): [(event before, event after, host seconds)] per allreduce_mean()
        offset = 0
        for p, padded in zip(self.params, self._padded):
            n = p.numel()
            self._views.append(self.flat[offset:offset + n].view_as(p))
            self._offsets.append(offset)
            offset += padded

    def attach(self, model: Optional[nn.Module] = None) -> "FlatGradReducer":
        """Publish the slices: `param._rq_grad_view` for every parameter and, for an RqVae whose level codebooks are plain
        embeddings lying back to back in the buffer, `model._rq_cb_grad_sink` for the fused quantiser backward."""
        from types import SimpleNamespace
        for p, v in zip(self.params, self._views):
            p._rq_grad_view = v
            p._rq_grad_owner = self
            p._rq_sink_epoch = -1
        layers = getattr(model, "layers", None)
        if layers is not None and len(layers) > 0 and all(hasattr(l, "embedding") for l in layers):
            cbs = [l.embedding.weight for l in layers]
            idx = [next((i for i, p in enumerate(self.params) if p is c), None) for c in cbs]
            shapes_ok = all(c.shape == cbs[0].shape for c in cbs)
            K, D = cbs[0].shape
            if (None not in idx and shapes_ok and all(b == a + 1 for a, b in zip(idx, idx[1:]))
                    and all(self._offsets[b] == self._offsets[a] + K * D for a, b in zip(idx, idx[1:]))):   # (no padding between them)
                off = self._offsets[idx[0]]
                model._rq_cb_grad_sink = SimpleNamespace(view=self.flat[off:off + len(cbs) * K * D].view(len(cbs), K, D),
                                                         params=cbs, owner=self)
        # what is complete when the gradient of the encoder's OUTPUT exists: everything that is not an encoder parameter
        enc = getattr(model, "encoder", None)
        if enc is not None and model is not None:
            enc_ids = {id(p) for p in enc.parameters()}
            early = [i for i, p in enumerate(self.params) if id(p) not in enc_ids]
            if early and len(early) < len(self.params):
                self._early_params = [self.params[i] for i in early]
                self._early_runs = self._runs(early)
                self._late_runs = self._runs([i for i in range(len(self.params)) if i not in set(early)])
                model._rq_reducer = self
        return self

    def _runs(self, idx: List[int]) -> List[tuple]:
        """Maximal contiguous [lo, hi) ranges of the flat buffer covered by the parameters `idx` (ascending)."""
        runs: List[list] = []
        for i in sorted(idx):
            lo, hi = self._offsets[i], self._offsets[i] + self._padded[i]
            if runs and runs[-1][1] == lo:
                runs[-1][1] = hi
            else:
                runs.append([lo, hi])
        return [tuple(r) for r in runs]

    def arm(self) -> None:
        """Announce that the NEXT backward is the last one before `allreduce_mean()` (every step without gradient accumulation;
        the last micro-batch with it): its boundary hook may start reducing.  Cleared by `zero_()` and by the launch."""
        self._armed = True

    def boundary_hook(self, grad: Tensor) -> None:
        """Tensor hook on the encoder's output (modules/rqvae.py registers it when a reducer is attached): runs when that
        gradient exists, i.e. after the decoder's and the quantiser's backward and before the encoder's.  With several ranks
        and an armed step, the finished part of the buffer goes on the wire now (asynchronously: RCCL runs it on its own
        stream behind an event of this one) while the encoder's backward computes.  Only when every one of those gradients
        was WRITTEN into the buffer by its backward node in this epoch -- a gradient that still lives elsewhere is packed by
        `allreduce_mean()`, which then reduces everything at once, as before."""
        del grad
        if not self._armed or world_size() == 1 or not self._early_runs:
            return None
        self._armed = False
        if not all(getattr(p, "_rq_sink_epoch", -1) == self.epoch for p in self._early_params):
            return None
        self._pending = [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for lo, hi in self._early_runs]
        self.overlap_launches += 1
        return None

    def zero_(self) -> None:
        """Use instead of optimizer.zero_grad().  Starts a new epoch: until the next call, the FIRST backward node that
        produces a parameter's gradient may write it straight into the parameter's slice (`claim_grad_sink`); any further
        producer of the same parameter in the same backward pass -- one module applied twice under one loss -- sees the
        slice taken and returns an ordinary tensor, which autograd accumulates."""
        self.epoch += 1
        self._armed = False
        # early all-reduces of a step that never reached allreduce_mean() (its backward or the caller raised after the hook
        # fired): every rank issued them, so they complete; wait and drop them -- left in place, the NEXT allreduce_mean()
        # would take them for this step's and skip the early part of the buffer
        for work in self._pending:
            work.wait()
        self._pending = []
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def allreduce_mean(self) -> Tensor:
        w = world_size()
        if w == 1 and not (dist.is_available() and dist.is_initialized()):
            return self.flat
        for p, v in zip(self.params, self._views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():   # produced by code that does not know the buffer: one small copy
                v.copy_(p.grad)
            p.grad = v
        timed = self._timing is not None and self.flat.is_cuda
        if timed:
            import time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
        if self._pending:          # the early part is on the wire (boundary_hook): wait for it, reduce the rest
            for work in self._pending:
                work.wait()
            self._pending = []
            for lo, hi in self._late_runs:
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if timed:
            e1.record()
            self._timing.append((e0, e1, time.perf_counter() - t0))
        self.flat.mul_(1.0 / w)
        return self.flat

    def enable_timing(self, on: bool = True) -> None:
        """Bench only: bracket the EXPOSED part of every following reduction -- the wait for the early all-reduces plus the late
        ones, i.e. what is not hidden under the encoder's backward -- with two events on the current stream (and the host clock)."""
        self._timing = [] if on else None

    def exposed_ms(self) -> List[tuple]:
        """[(device ms, host ms)] per allreduce_mean() since enable_timing(); synchronises; clears the list."""
        if not self._timing:
            return []
        torch.cuda.synchronize()
        out = [(a.elapsed_time(b), 1e3 * h) for a, b, h in self._timing]
        self._timing = []
        return out


def claim_grad_sink(w: Tensor):
    """The slice of the flat buffer `w`'s gradient may be written into, or None.  Granted once per parameter and epoch
    (`FlatGradReducer.zero_`) and only while `.grad` is unset: AccumulateGrad runs after ALL producers of a parameter
    have run, so `w.grad is None` alone cannot tell the first producer from the second (two of them writing the same
    slice and autograd then summing the alias with itself was a silent wrong gradient -- ADVICE r2)."""
    view = getattr(w, "_rq_grad_view", None)
    owner = getattr(w, "_rq_grad_owner", None)
    if view is None or owner is None or w.grad is not None or getattr(w, "_rq_sink_epoch", -1) == owner.epoch:
        return None
    w._rq_sink_epoch = owner.epoch
    return view


@torch.no_grad()
def allgather_rows(local: Tensor) -> Tensor:
    """Concatenate per-rank row blocks (possibly of unequal length) along dim 0 -- used to assemble the
    corpus-wide semantic-id table from row shards."""
    w = world_size()
    if w == 1:
        return local
    n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    sizes = [int(s) for s in sizes]
    m = max(sizes)
    pad = local.new_zeros((m,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
