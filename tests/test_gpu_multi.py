"""The multi-GPU path on REAL peers: RCCL over xGMI with one process per GPU (reference train_rqvae.py:153,195,
HF accelerate's DDP wrap).  Every test here needs at least two visible GPUs and skips itself otherwise -- the gpurun boxes of this
pool expose one, so these arm themselves on the first multi-GPU node.  What they pin, in the order a first
SCALE run exercises it:
  (i)   `python bench.py --gpus 2 --steps 5`: the line parses, two RCCL ranks, the early all-reduce went out under the encoder's
        backward, per-rank exposed all-reduce times are present;
  (ii)  a 2-rank RCCL training step (row shards of 2 x B rows) == the single-rank step on the same 2 x B rows (the assertion of
        tests/test_dist_gloo.py / test_gpu_dist2.py, on RCCL), both ranks end with identical parameters;
  (iii) the hipGraph step with a 2-rank RCCL group: captured once, replayed 100 x, equals the eager trajectory;
  (iv)  the row-sharded k-means over RCCL == the single-rank k-means.
The one-GPU rehearsals of the same code (two ranks on cuda:0 over gloo; an RCCL group of one inside a hipGraph) are
tests/test_gpu_dist2.py; the host logic on CPU ranks is tests/test_dist_gloo.py.
"""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

from conftest import PKG, ROOT

N_GPUS = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
needs_peers = pytest.mark.skipif(N_GPUS < 2, reason=f"needs >= 2 GPUs for RCCL peers (this box has {N_GPUS})")

ROWS = 4096        # rows of the global batch: 2048 per rank


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@needs_peers
def test_bench_two_gpus_line():
    """(i) The driver's SCALE command at N = 2 (`python bench.py --gpus 2` relaunches itself under torch.distributed.run)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "16384",
           "--no-cpu-baseline", "--no-parity", "--no-small-batch", "--min-seconds", "0.3"]
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800, env=_clean_env())
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["dry_ranks"] == 0 and j["scaling"] == "weak"
    assert j["steps"] == 5 and j["config"]["rows_per_gpu_per_step"] == 16384
    assert abs(j["value"] - 2 * 16384 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]       # rows of ALL ranks / max-over-ranks time
    ar = j["allreduce"]
    assert ar["backend"].startswith("nccl") and ar["overlap_launches"] > 0
    assert len(ar["exposed_ms_per_rank_device_host"]) == 2 and all(h > 0 for _, h in ar["exposed_ms_per_rank_device_host"])
    assert j["roofline"]["launches"] > 0 and 0 < j["roofline"]["frac"] < 1


def _worker(rank, world, port, tmp):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import traceback
    try:
        _worker_body(rank, world, tmp)
    except Exception:
        with open(os.path.join(tmp, f"error_{rank}.txt"), "w") as fh:
            fh.write(traceback.format_exc())
        raise


def _worker_body(rank, world, tmp):
    import numpy as np
    import torch.distributed as dist
    from init.kmeans import Kmeans
    from rqhip import dist as rqdist
    from rqhip.optim import FlatAdamW
    from test_gpu_dist2 import _make_model, _one_step
    from train_rqvae import _GraphedStep

    torch.cuda.set_device(rank)
    g = torch.Generator().manual_seed(1234)
    X = torch.nn.functional.normalize(torch.randn(ROWS, 768, generator=g), dim=-1).cuda()

    # ---- before any process group exists: the single-rank references on this rank's own GPU -----------------------------------------
    lat = torch.nn.functional.normalize(torch.randn(6000, 32, generator=torch.Generator().manual_seed(5)), dim=-1).cuda()
    np.random.seed(3)
    torch.manual_seed(3)
    plain_km = Kmeans(k=64, max_iters=3).run(lat).centroids.clone()
    ref = _make_model(kmeans_init=False)
    start = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():       # codebooks as k-means of a first forward would leave them: well spread, the same on every rank (same seed)
        for l, layer in enumerate(ref.layers):
            layer.embedding.weight.copy_(torch.randn(256, 32, generator=torch.Generator().manual_seed(40 + l)).cuda() * (0.3 / (l + 1)))
            layer.kmeans_initted = True
    start = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    rflat, _, rloss = _one_step(ref, X, rqdist)          # world_size() == 1 here: the full batch, one process
    rparams = torch.cat([p.detach().flatten() for p in ref.parameters()])

    # ---- RCCL, one rank per GPU ----------------------------------------------------------------------------------------------------
    r, dev, w = rqdist.init_from_env("cuda", force=world == 1)      # (world 1: the rehearsal below; a group of one is still RCCL)
    assert (r, dev, w) == (rank, rank, world) and dist.get_backend() == "nccl" and rqdist.world_size() == world

    # (iv) row-sharded Lloyd iterations over RCCL == the plain run; identical on every rank
    lo, hi = rqdist.shard_bounds(lat.shape[0])
    np.random.seed(3)
    torch.manual_seed(3)
    shard_km = Kmeans(k=64, max_iters=3).run(lat[lo:hi], sharded=True).centroids
    km_err = (shard_km - plain_km).abs().max().item()
    both = [torch.empty_like(shard_km) for _ in range(world)]
    dist.all_gather(both, shard_km.contiguous())
    assert all(torch.equal(both[0], b) for b in both[1:]), "ranks disagree on the centroids"

    # (ii) one product training step on this rank's rows of the global batch
    model = _make_model(kmeans_init=False)
    model.load_state_dict(start)
    for layer in model.layers:
        layer.kmeans_initted = True
    lo, hi = rqdist.shard_bounds(ROWS)
    flat, aliased, loss = _one_step(model, X[lo:hi], rqdist)
    assert aliased == len(list(model.parameters())), f"only {aliased} gradients were written in place"
    params = torch.cat([p.detach().flatten() for p in model.parameters()])
    both = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(both, params)
    assert all(torch.equal(both[0], b) for b in both[1:]), "ranks ended the step with different parameters"
    gscale = max(rflat.abs().max().item(), 1e-6)
    gerr = (flat - rflat).abs().max().item()
    perr = (params - rparams).abs().max().item()

    # (iii) the hipGraph step with real peers: capture once, replay 100 x, against 100 eager steps from the same start
    losses, finals = [], []
    for graphed in (False, True):
        m = _make_model(kmeans_init=False)
        m.load_state_dict(start)
        for layer in m.layers:
            layer.kmeans_initted = True
        opt = FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
        reducer = rqdist.FlatGradReducer(m.parameters()).attach(m)
        xb = X[lo:lo + 640].contiguous()
        step = _GraphedStep(m, opt, reducer, 640, 768, torch.device("cuda", rank), 0.2)
        m.train()
        step.x.copy_(xb)
        for _ in range(3):
            step._step()
        if graphed:
            step.capture(xb)
            for _ in range(100):
                out = step.run(xb)
            assert step.captures == 1
        else:
            for _ in range(100):
                out = step._step()
        torch.cuda.synchronize()
        losses.append(float(out.loss))
        finals.append(torch.cat([p.detach().flatten() for p in m.parameters()]))
        both = [torch.empty_like(finals[-1]) for _ in range(world)]
        dist.all_gather(both, finals[-1])
        assert all(torch.equal(both[0], b) for b in both[1:]), f"ranks diverged ({'graph' if graphed else 'eager'})"
    rqdist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        torch.save({"km_err": km_err, "gerr": gerr, "gscale": gscale, "perr": perr, "loss": loss, "rloss": rloss,
                    "eager": losses[0], "graph": losses[1], "graph_param_err": (finals[0] - finals[1]).abs().max().item()},
                   os.path.join(tmp, "out.pt"))


def _run_ranks(world):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker, args=(r, world, port, tmp)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=800)
        for p in procs:
            if p.is_alive():
                p.kill()
        errs = [open(os.path.join(tmp, f)).read() for f in sorted(os.listdir(tmp)) if f.startswith("error_")]
        assert all(p.exitcode == 0 for p in procs), ([p.exitcode for p in procs], errs)
        return torch.load(os.path.join(tmp, "out.pt"))


def test_the_rccl_worker_with_a_group_of_one():
    """Rehearsal on a one-GPU box: the very worker the two-GPU test spawns, as ONE rank with an RCCL process group of one -- every line
    of it runs (references, sharded k-means, the step, the captured step with its all-reduces inside the graph), only the peers are
    missing; with one rank the "sharded" results must equal the single-rank references to the same tolerances."""
    res = _run_ranks(1)
    print("one RCCL rank vs no process group:", res)
    assert res["km_err"] <= 1e-5 and res["gerr"] <= 1e-5 * max(res["gscale"], 1e-3) + 1e-9 and res["perr"] <= 1e-5, res
    assert res["eager"] == res["eager"] and abs(res["eager"] - res["graph"]) <= 2e-3 * max(1.0, abs(res["eager"])), res


@needs_peers
def test_two_rccl_ranks_step_graph_and_kmeans():
    """(ii) + (iii) + (iv) in one pair of processes (one process group, one import of torch per rank)."""
    res = _run_ranks(2)
    print("two RCCL ranks vs single rank:", res)
    assert res["km_err"] <= 1e-5, res                                           # (iv)
    assert res["gerr"] <= 1e-5 * max(res["gscale"], 1e-3) + 1e-9, res         # (ii) reduced gradients == full-batch gradients
    assert res["perr"] <= 1e-5, res                                             #      parameters after AdamW (lr 1e-3: an update is <= 1e-3)
    assert abs(res["loss"] - res["rloss"]) <= 0.5, res                          #      (each rank's loss is its shard's mean)
    # (iii) 103 AdamW steps from the same start, replayed vs eager: the bound of tests/test_gpu_train.py:test_hip_graph_step_matches_eager
    assert res["eager"] == res["eager"] and abs(res["eager"] - res["graph"]) <= 2e-3 * max(1.0, abs(res["eager"])), res
