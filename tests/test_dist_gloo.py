"""The N > 1 path on CPU: two gloo processes exercise the sharding helpers, the one-collective flat
gradient reduction (DDP-mean semantics), parameter broadcast and the ragged all-gather of id rows."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from rqhip import dist as rqdist
    r, _, w = rqdist.init_from_env("cpu")
    assert (r, w) == (rank, world) and rqdist.world_size() == world and rqdist.get_rank() == rank

    # row sharding: contiguous, disjoint, covering
    lo, hi = rqdist.shard_bounds(11)
    spans = [rqdist.shard_bounds(11, k, world) for k in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == 11 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert (lo, hi) == spans[rank]

    # broadcast: rank 1 starts from different weights, ends with rank 0's
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(6, 4, bias=False), torch.nn.ReLU(), torch.nn.Linear(4, 2, bias=False))
    rqdist.broadcast_module(model)
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 4, bias=False), torch.nn.ReLU(), torch.nn.Linear(4, 2, bias=False))
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a, b)

    # flat gradient buffer: one all-reduce == mean of per-rank gradients == gradient of the global-batch mean loss
    reducer = rqdist.FlatGradReducer(model.parameters())
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    lo, hi = rqdist.shard_bounds(8)
    reducer.zero_()
    model(X[lo:hi]).pow(2).sum(dim=1).mean().backward()
    flat = reducer.allreduce_mean().clone()
    ref.zero_grad()
    ref(X).pow(2).sum(dim=1).mean().backward()
    want = torch.cat([p.grad.flatten() for p in ref.parameters()])
    assert torch.allclose(flat, want, rtol=1e-5, atol=1e-6), (flat - want).abs().max()
    for p in model.parameters():   # .grad are views of the reduced buffer: the optimizer needs no copy
        assert p.grad.untyped_storage().data_ptr() == reducer.flat.untyped_storage().data_ptr()
    # a second step re-packs fresh gradients (zero_ drops the views, autograd assigns new tensors)
    reducer.zero_()
    assert all(p.grad is None for p in model.parameters())
    model(X[lo:hi]).pow(2).sum(dim=1).mean().backward()
    flat2 = reducer.allreduce_mean()
    assert torch.allclose(flat2, want, rtol=1e-5, atol=1e-6)

    # ---- the overlapped reduction: gradients that are final before the encoder's backward go on the wire under it -----------
    from rqhip.dist import claim_grad_sink

    class SinkLinear(torch.autograd.Function):      # what modules/encoder.py's nodes do: write dW into the flat buffer's slice
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            gw, sink = g.t() @ x, claim_grad_sink(w)
            if sink is not None:
                sink.copy_(gw)
                gw = sink.view_as(sink)
            return g @ w, gw

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = torch.nn.Linear(6, 4, bias=False)
            self.decoder = torch.nn.Linear(4, 6, bias=False)

        def forward(self, x):
            h = torch.relu(SinkLinear.apply(x, self.encoder.weight))
            red = getattr(self, "_rq_reducer", None)
            if red is not None:
                h.register_hook(red.boundary_hook)
            return SinkLinear.apply(h, self.decoder.weight)

    torch.manual_seed(5)
    toy, toy_ref = Toy(), Toy()
    toy_ref.load_state_dict(toy.state_dict())
    red = rqdist.FlatGradReducer(toy.parameters()).attach(toy)
    assert red._early_runs == [(24, 48)] and red._late_runs == [(0, 24)]      # decoder weight early, encoder weight late
    lo, hi = rqdist.shard_bounds(8)
    toy_ref(X).pow(2).sum(dim=1).mean().backward()
    want = torch.cat([p.grad.flatten() for p in toy_ref.parameters()])
    for armed in (False, True):
        red.zero_()
        if armed:
            red.arm()
        toy(X[lo:hi]).pow(2).sum(dim=1).mean().backward()
        assert red.overlap_launches == int(armed) and bool(red._pending) == armed
        got = red.allreduce_mean().clone()
        assert not red._pending and torch.allclose(got, want, rtol=1e-5, atol=1e-6), (armed, (got - want).abs().max())
    # gradient accumulation: only the LAST backward of a step may start reducing (armed by the loop just before it)
    red.zero_()
    half = (hi - lo) // 2
    (toy(X[lo:lo + half]).pow(2).sum(dim=1).sum() / (hi - lo)).backward()
    assert red.overlap_launches == 1
    red.arm()
    (toy(X[lo + half:hi]).pow(2).sum(dim=1).sum() / (hi - lo)).backward()
    assert red.overlap_launches == 2
    got = red.allreduce_mean()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (got - want).abs().max()

    # an aborted step (the hook fired, allreduce_mean() never ran) must not leak its early all-reduces into the next one
    red.zero_()
    red.arm()
    toy(X[lo:hi]).pow(2).sum(dim=1).mean().backward()
    assert red._pending
    red.zero_()                                   # the step is abandoned here
    assert not red._pending
    toy(X[lo:hi]).pow(2).sum(dim=1).mean().backward()    # unarmed: one whole-buffer all-reduce
    got = red.allreduce_mean()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (got - want).abs().max()

    # ragged all-gather of id rows
    local = torch.arange(3 + rank).unsqueeze(1).repeat(1, 2) + 100 * rank
    full = rqdist.allgather_rows(local)
    assert full.shape == (3 + 4, 2) and full[3].tolist() == [100, 100] and full[:3, 0].tolist() == [0, 1, 2]

    rqdist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]


def test_single_process_defaults():
    from rqhip import dist as rqdist
    assert rqdist.world_size() == 1 and rqdist.get_rank() == 0
    assert rqdist.shard_bounds(10) == (0, 10)
    t = torch.arange(4)
    assert rqdist.allgather_rows(t) is t


# ---- row-sharded k-means: the collective choreography on two gloo ranks ------------------------
def _kmeans_worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    from rqhip import dist as rqdist
    from rqhip import ops
    from oracle import rq_oracle as o
    import init.kmeans as km
    rqdist.init_from_env("cpu")

    # The two array steps are HIP kernels (no CPU path in the product); here they are replaced by the oracle's
    # restatements so that the HOST logic -- seeding by rank 0, one all-reduce of sums || counts per iteration, identical
    # stop decisions on every rank, reseed of empty clusters through rank 0's RNG -- runs on CPU tensors over gloo.
    def fake_partial_sums(x, cent, assign, sums, state):
        if int(state[0]) != 0:
            return
        xn, cn = x.numpy(), cent.numpy()
        K, D = cn.shape
        a = o.kmeans_assign(xn, cn) if xn.shape[0] else np.zeros((0,), np.int64)
        assign.copy_(torch.from_numpy(a))
        s = np.zeros((K, D + 1), np.float32)
        for i, k in enumerate(a):
            s[k, :D] = s[k, :D] + xn[i]
            s[k, D] += 1
        sums.copy_(torch.from_numpy(s))
        state[2] = 0

    def fake_apply_sums(sums, cent, counts, state, thr):
        if int(state[0]) != 0:
            return
        s = sums.numpy()
        K, D = cent.shape
        n = s[:, D]
        old = cent.numpy().copy()
        new = old.copy()
        nz = n > 0
        new[nz] = s[nz, :D] / n[nz, None]
        cent.copy_(torch.from_numpy(new))
        counts.copy_(torch.from_numpy(n.astype(np.int64)))
        shift_sq = np.float32(((new[nz] - old[nz]) ** 2).sum(axis=1).max()) if nz.any() else np.float32(0)
        state[2] = int(np.array([shift_sq], np.float32).view(np.int32)[0])
        state[1] += 1
        if (~nz).any():
            state[0] = 2
        elif np.sqrt(shift_sq) < thr:
            state[0] = 1

    ops.kmeans_partial_sums, ops.kmeans_apply_sums = fake_partial_sums, fake_apply_sums
    g = torch.Generator().manual_seed(5)
    K, D, B = 12, 6, 400
    centers = torch.randn(K // 2, D, generator=g) * 3       # fewer modes than codes: empty clusters do occur
    X = centers[torch.randint(0, K // 2, (B,), generator=g)] + 0.2 * torch.randn(B, D, generator=g)
    lo, hi = rqdist.shard_bounds(B)
    np.random.seed(3)
    torch.manual_seed(3)          # only rank 0's streams are consumed
    res = km.Kmeans(k=K).run(X[lo:hi].contiguous(), sharded=True)
    # every rank holds the same centroids ...
    both = [torch.zeros_like(res.centroids) for _ in range(world)]
    dist.all_gather(both, res.centroids)
    assert torch.equal(both[0], both[1])
    # ... and they equal a single-process run of the reference loop on the whole matrix (sums in another order)
    if rank == 0:
        np.random.seed(3)
        torch.manual_seed(3)
        xn = X.numpy()
        draws = iter(lambda: int(torch.randint(0, B, (1,))), None)
        cent, assign, _ = o.kmeans_run(xn, np.random.choice(B, K, replace=False), reseed_draws=lambda: next(draws))
        np.testing.assert_allclose(res.centroids.numpy(), cent, rtol=1e-5, atol=1e-6)
        assert np.array_equal(res.assignment.numpy(), assign[lo:hi])
    rqdist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_two_rank_sharded_kmeans_choreography():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kmeans_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]
