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
