"""The PRODUCT training step and the sharded k-means warm-up with world_size = 2 -- on the one GPU a gpurun box has
(VERDICT r2 item 6).  RCCL refuses two ranks on one device, so the two processes share cuda:0 and talk over gloo, which
accepts device tensors (it stages them through the host); everything else is the production code path:
`rqdist.init_from_env(backend="gloo")`, `RqVae.forward` + backward through the HIP kernels writing into the flat
gradient buffer (`FlatGradReducer.attach`, the codebook-gradient sink), `allreduce_mean()` on its process-group branch,
fused AdamW, and the row-sharded Lloyd loop with the real `rqhip_kmeans_partial_sums` / `rqhip_kmeans_apply_sums`.

Asserted: both ranks end with bit-identical parameters; they equal the single-process full-batch step to 1e-5; the
reduced gradients equal the full-batch gradients to 1e-5; sharded k-means == plain k-means (3 Lloyd iterations, same
seeds) to 1e-5 and identical on both ranks; the lazy in-model warm-up with `kmeans_rows_sharded` leaves both ranks with
the same codebooks.
"""
import os
import socket
import sys
import tempfile

import pytest
import torch

from conftest import PKG, ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]   # (spawned ranks import torch again: slow on a cold box)

# (RQ_TEST_ROWS: the second test re-runs the workers at a batch that takes the seam node -- modules/rqvae.py: >= 4096 rows per rank)
ROWS = int(os.environ.get("RQ_TEST_ROWS", "2048"))
WARM = 2048


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_model(kmeans_init):
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    torch.manual_seed(0)
    return RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
                 n_cat_features=0, codebook_kmeans_init=kmeans_init, codebook_mode=QuantizeForwardMode.STE).cuda()


def _items():
    g = torch.Generator().manual_seed(1234)
    return torch.nn.functional.normalize(torch.randn(ROWS, 768, generator=g), dim=-1).cuda()


def _one_step(model, rows, rqdist, micro=1):
    """One product step (micro > 1: the accumulation form of BASELINE config 4 -- `micro` micro-batches, each weighted by its
    share of the rows, ONE reduction): zero_ -> [arm before the last] backward(s) -> allreduce_mean -> AdamW."""
    from data.schemas import SeqBatch
    from rqhip.autograd import loss_scale
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    reducer = rqdist.FlatGradReducer(model.parameters()).attach(model)
    model.train()
    reducer.zero_()
    n = rows.shape[0]
    cuts = [n * i // micro for i in range(micro + 1)]
    for i in range(micro):
        part = rows[cuts[i]:cuts[i + 1]]
        share = part.shape[0] / n
        if i + 1 == micro:
            reducer.arm()          # the last backward: decoder / codebook gradients are reduced under the encoder's backward
        with loss_scale(share):
            out = model(SeqBatch(None, None, None, part, None, None), gumbel_t=0.2)
        (out.loss if micro == 1 else out.loss * share).backward()
    assert reducer.overlap_launches == (1 if rqdist.world_size() > 1 else 0), reducer.overlap_launches
    # the backward kernels wrote every gradient straight into the flat buffer (no packing copy)
    aliased = sum(int(p.grad is not None and p.grad.data_ptr() == v.data_ptr()) for p, v in zip(reducer.params, reducer._views))
    flat = reducer.allreduce_mean().clone()
    opt.step()
    torch.cuda.synchronize()
    return flat, aliased, float(out.loss)


def _worker(rank, world, port, tmp):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    import torch.distributed as dist
    from init.kmeans import Kmeans
    from rqhip import dist as rqdist
    torch.cuda.set_device(0)
    X = _items()

    # ---- before any process group exists: the single-process references --------------------------------------------
    lat = torch.nn.functional.normalize(torch.randn(6000, 32, generator=torch.Generator().manual_seed(5)), dim=-1).cuda()
    np.random.seed(3)
    torch.manual_seed(3)
    plain_km = Kmeans(k=64, max_iters=3).run(lat).centroids.clone()

    # ---- two ranks, one GPU, gloo ---------------------------------------------------------------------------------------
    r, dev, w = rqdist.init_from_env("cuda", backend="gloo", device_index=0)
    assert (r, dev, w) == (rank, 0, world) and rqdist.world_size() == 2

    # (a) row-sharded Lloyd iterations with the real kernels: equal to the plain run, identical on both ranks
    lo, hi = rqdist.shard_bounds(lat.shape[0])
    np.random.seed(3)
    torch.manual_seed(3)      # (rank 0's streams are the ones that count; rank 1's draws are discarded)
    shard_km = Kmeans(k=64, max_iters=3).run(lat[lo:hi], sharded=True).centroids
    err = (shard_km - plain_km).abs().max().item()
    assert err <= 1e-5, f"sharded k-means differs from the plain run by {err}"
    both = [torch.empty_like(shard_km) for _ in range(world)]
    dist.all_gather(both, shard_km.contiguous())
    assert torch.equal(both[0], both[1]), "ranks disagree on the centroids"
    # (a2) latent widths the HIP k-means kernels do not take (D > 128): the sharded call gathers the warm-up rows and every rank runs the
    # same seeded loop (this raised with several ranks) -- identical centroids everywhere, this rank's block of assignments
    import warnings
    gw = torch.Generator().manual_seed(21)
    wide_rows = (torch.randn(8, 160, generator=gw)[torch.randint(0, 8, (600,), generator=gw)] + 0.1 * torch.randn(600, 160, generator=gw)).cuda()
    wlo, whi = rqdist.shard_bounds(600)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wk = Kmeans(k=8, max_iters=5).run(wide_rows[wlo:whi].contiguous(), sharded=True)
    wboth = [torch.empty_like(wk.centroids) for _ in range(world)]
    dist.all_gather(wboth, wk.centroids.contiguous())
    assert torch.equal(wboth[0], wboth[1]) and tuple(wk.assignment.shape) == (whi - wlo,)
    d2 = ((wide_rows[wlo:whi, None, :] - wk.centroids[None]) ** 2).sum(-1)
    assert (d2.argmin(dim=1) == wk.assignment).float().mean().item() > 0.99      # assignments belong to the returned centroids

    # (b) the lazy in-model warm-up, row-sharded: every rank takes its block of the warm rows through the model
    model = _make_model(kmeans_init=True)
    rqdist.broadcast_module(model)
    from data.schemas import SeqBatch
    lo, hi = rqdist.shard_bounds(WARM)
    np.random.seed(0)
    torch.manual_seed(0)
    for layer in model.layers:
        layer.kmeans_rows_sharded = True
    model.train()
    model(SeqBatch(None, None, None, X[lo:hi], None, None), 0.2)
    for layer in model.layers:
        layer.kmeans_rows_sharded = False
        assert layer.kmeans_initted
    cbs = torch.stack([l.embedding.weight.detach() for l in model.layers])
    both = [torch.empty_like(cbs) for _ in range(world)]
    dist.all_gather(both, cbs.contiguous())
    assert torch.equal(both[0], both[1]) and torch.isfinite(cbs).all()
    start = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # (c) one product training step on this rank's half of the batch
    lo, hi = rqdist.shard_bounds(ROWS)
    flat, aliased, loss = _one_step(model, X[lo:hi], rqdist)
    assert aliased == len(list(model.parameters())), f"only {aliased} gradients were written in place"
    params = torch.cat([p.detach().flatten() for p in model.parameters()])
    both = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(both, params)
    assert torch.equal(both[0], both[1]), "ranks ended the step with different parameters"
    gb = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gb, flat)
    assert torch.equal(gb[0], gb[1])

    # (d) the accumulation step of BASELINE config 4: three micro-batches per rank, ONE reduction (armed before the last backward)
    model4 = _make_model(kmeans_init=False)
    model4.load_state_dict(start)
    for layer in model4.layers:
        layer.kmeans_initted = True
    flat4, aliased4, _ = _one_step(model4, X[lo:hi], rqdist, micro=3)
    assert aliased4 == len(list(model4.parameters()))
    params4 = torch.cat([p.detach().flatten() for p in model4.parameters()])
    both = [torch.empty_like(params4) for _ in range(world)]
    dist.all_gather(both, params4)
    assert torch.equal(both[0], both[1]), "ranks ended the accumulation step with different parameters"

    # (e) the row-sharded corpus tokenisation (semids.py:76-110): every rank tokenises its rows, the table is all-gathered,
    # the dedup column is computed on the gathered table -- identical on both ranks
    from data.processed import ItemData, RecDataset
    from modules.tokenizer.semids import SemanticIdTokenizer
    tok = SemanticIdTokenizer(input_dim=768, hidden_dims=[512, 256, 128], output_dim=32, codebook_size=256, n_layers=3,
                              n_cat_feats=0)
    tok.rq_vae = model
    ds = ItemData(root="/nonexistent", dataset=RecDataset.AMAZON, train_test_split="all", item_matrix=X[:1501],
                  is_train=torch.ones(1501, dtype=torch.bool, device="cuda"))
    table = tok.precompute_corpus_ids(ds, sharded=True)
    assert table.shape == (1501, 4)
    both = [torch.empty_like(table) for _ in range(world)]
    dist.all_gather(both, table.contiguous())
    assert torch.equal(both[0], both[1])
    rqdist.barrier()
    dist.destroy_process_group()

    # ---- rank 0, alone again: the same step on the full batch, single process ------------------------------------------
    if rank == 0:
        ref = _make_model(kmeans_init=False)
        ref.load_state_dict(start)
        for layer in ref.layers:
            layer.kmeans_initted = True
        rflat, _, rloss = _one_step(ref, X, rqdist)
        rparams = torch.cat([p.detach().flatten() for p in ref.parameters()])
        gscale = max(rflat.abs().max().item(), 1e-6)
        gerr = (flat - rflat).abs().max().item()
        perr = (params - rparams).abs().max().item()
        # (d): six micro-batches in two ranks == the full-batch step; (e): sharded table == the single-process table
        perr4 = (params4 - rparams).abs().max().item()
        gerr4 = (flat4 - rflat).abs().max().item()
        tok.reset()
        local_table = tok.precompute_corpus_ids(ds, sharded=False)
        ids_differ = int((local_table[:, :3] != table[:, :3]).any(dim=1).sum())
        torch.save({"gerr": gerr, "gscale": gscale, "perr": perr, "loss": loss, "rloss": rloss, "perr4": perr4, "gerr4": gerr4,
                    "ids_differ": ids_differ, "dedup_equal": bool(ids_differ > 0 or torch.equal(local_table, table))},
                   os.path.join(tmp, "out.pt"))


def test_two_ranks_on_one_gpu_product_step_and_sharded_kmeans():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=420)
        for p in procs:
            if p.is_alive():
                p.kill()
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        res = torch.load(os.path.join(tmp, "out.pt"))
    print("two ranks on one GPU vs single process:", res)
    assert res["gerr"] <= 1e-5 * max(res["gscale"], 1e-3) + 1e-9, res       # reduced gradients == full-batch gradients
    assert res["perr"] <= 1e-5, res                                           # parameters after AdamW
    # the accumulation step.  (At the seam-sized batch its micro-batches fall BELOW 4096 rows: they run the small-batch kernel family --
    # library GEMMs, three-piece bf16 job-table weight gradients -- against a full-batch reference on the split-fp16 family; both are
    # fp32-accurate to 1e-8 absolute on gradients of scale 6e-6, but the FIRST AdamW step moves every parameter by lr * g / (|g| + eps): a
    # gradient entry of 1e-8 that differs by 1e-8 moves its parameter by half of lr -- the parameters are compared only where both arms run
    # the same kernel family.)
    loose = int(os.environ.get("RQ_TEST_ROWS", "2048")) >= 8192
    assert res["gerr4"] <= 1e-5 * max(res["gscale"], 1e-3) + 1e-9 and (loose or res["perr4"] <= 1e-5), res
    assert res["ids_differ"] <= 1 and res["dedup_equal"], res                 # sharded tokenisation (a near-tie may flip: tests/parity_gate.py)


def test_two_ranks_at_a_batch_that_takes_the_seam_node(monkeypatch):
    """The same two-rank step at 5120 rows per rank: RqVae.forward then runs the encoder tail / levels / decoder head as ONE node
    (rqhip/autograd.py:RqSeamFunction) and the early all-reduce hangs on the gradient of the hidden activation in front of it -- the
    decoder's, the codebooks' and the seam's own decoder-side weight gradient must be final when it starts (also in the accumulation
    form, where they reach the flat buffer through autograd's accumulation, not in place)."""
    monkeypatch.setenv("RQ_TEST_ROWS", "10240")
    test_two_ranks_on_one_gpu_product_step_and_sharded_kmeans()


def _graph_worker(port, tmp):
    """The hipGraph step with a live RCCL process group (one rank: what a one-GPU box allows): the all-reduces of
    FlatGradReducer.allreduce_mean are captured with the step and replayed."""
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import traceback
    import torch.distributed as dist
    from rqhip import dist as rqdist
    from train_rqvae import _GraphedStep
    try:
        _graph_body(tmp, dist, rqdist, _GraphedStep)
    except Exception:     # (the parent shows it: a bare exit code says nothing)
        torch.save({"error": traceback.format_exc()}, os.path.join(tmp, "graph.pt"))
        raise


def _graph_body(tmp, dist, rqdist, _GraphedStep):
    rqdist.init_from_env("cuda", force=True)                   # backend "nccl" == RCCL
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    X = _items()[:640]
    losses = []
    for graphed in (False, True):
        model = _make_model(kmeans_init=False)
        for layer in model.layers:
            layer.kmeans_initted = True
        from rqhip.optim import FlatAdamW
        opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)          # as train_rqvae.train
        reducer = rqdist.FlatGradReducer(model.parameters()).attach(model)
        step = _GraphedStep(model, opt, reducer, 640, 768, torch.device("cuda", 0), 0.2)
        model.train()
        step.x.copy_(X)
        for _ in range(3):          # as train_rqvae.train: a few eager steps first (the optimizer's state must exist before a
            step._step()            # capture -- its lazy initialisation inside one would be replayed with every step)
        if graphed:
            step.capture(X)
            for _ in range(22):
                out = step.run(X)
            assert step.captures == 1
        else:
            for _ in range(22):
                out = step._step()
        torch.cuda.synchronize()
        losses.append(float(out.loss))
    torch.save({"eager": losses[0], "graph": losses[1]}, os.path.join(tmp, "graph.pt"))
    dist.destroy_process_group()


def test_hip_graph_step_with_rccl_group():
    """`*_graph.gin` with a process group: the captured step contains the RCCL all-reduce (one rank here)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as tmp:
        p = ctx.Process(target=_graph_worker, args=(_free_port(), tmp))
        p.start()
        p.join(timeout=170)
        if p.is_alive():
            p.kill()
        res = torch.load(os.path.join(tmp, "graph.pt")) if os.path.exists(os.path.join(tmp, "graph.pt")) else {}
        assert p.exitcode == 0 and "error" not in res, (p.exitcode, res.get("error"))
    # 25 AdamW steps from the same start: the two trajectories agree to rounding noise amplified by the optimiser (the bound of
    # tests/test_gpu_train.py:test_hip_graph_step_matches_eager)
    assert res["eager"] == res["eager"] and abs(res["eager"] - res["graph"]) <= 2e-3 * max(1.0, abs(res["eager"])), res


def test_bench_dry_ranks_line_is_consistent():
    """`python bench.py --dry-ranks 2`: the whole multi-rank path of the bench -- self-relaunch under torch.distributed.run,
    init_from_env, armed early all-reduce, long_run, per-rank exposed all-reduce times, JSON assembly on rank 0 -- with two
    ranks on the one GPU over gloo.  The line must parse and be self-consistent."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dry-ranks", "2", "--batch", "8192", "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--no-parity", "--no-small-batch", "--min-seconds", "0.3"]
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=540, env=env)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_ranks"] == 2 and j["rccl_ranks"] == 0 and j["n_physical_gpus"] == 1
    assert j["steps"] == 4 and j["config"]["rows_per_gpu_per_step"] == 8192
    # value = rows of all ranks / time of the K steps
    assert abs(j["value"] - 2 * 8192 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]
    assert j["long_run"]["steps"] > 4 and abs(j["long_run"]["items_per_s"] - 2 * 8192 / (j["long_run"]["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]
    ar = j["allreduce"]
    assert len(ar["exposed_ms_per_rank_device_host"]) == 2 and all(h > 0 for _, h in ar["exposed_ms_per_rank_device_host"])
    assert ar["overlap_launches"] > 0                      # the early half went on the wire under the encoder's backward
    assert j["roofline"]["bound"] == "mfma" and j["roofline"]["launches"] > 0 and 0 < j["roofline"]["frac"] < 1
