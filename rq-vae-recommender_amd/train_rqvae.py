"""`python train_rqvae.py <config.gin>` -- RQ-VAE tokenizer training on MI355X.

Drop-in for the reference's entry point (train_rqvae.py:24-305): the same gin-configurable `train(...)` with
the same 29 keyword arguments and defaults, the same schedule (k-means warm-up forward at iteration 0,
`iterations + 1` optimiser steps, gradient accumulation, eval / checkpoint / id-diversity cadence) and the
same checkpoint dictionary (`iter`, `model`, `model_config`, `optimizer`; state_dict keys unchanged).

What is MI355X-native about it:
  * the model's quantisation stack, k-means init and id statistics are HIP kernels (modules/, init/);
  * one process per GPU (`torchrun --nproc-per-node N train_rqvae.py cfg.gin`): no accelerate/DDP wrapper --
    gradients live in one flat buffer and a step issues exactly one RCCL all-reduce (rqhip/dist.py); every
    rank draws its own batches from the full dataset, as the un-`prepare`d dataloader of the reference does
    (train_rqvae.py:119-120); the k-means warm-up is row-sharded: every rank takes its block of the first 20 000 items
    through the model, rank 0 draws the seed / reseed rows, each Lloyd iteration all-reduces the [K, D+1] sums || counts;
  * the item-feature matrix is resident in HBM and batches are gathered on the device (data/processed.py),
    there is no host->device copy in the loop;
  * the progress-bar losses are read back every `log_every` steps instead of three `.cpu().item()` syncs
    per step (train_rqvae.py:197-199);
  * at the reference's batch sizes (640 / 64 rows) a step is ~45 kernel launches of a few microseconds each, i.e.
    launch-bound.  So, like the reference -- whose forward is graph-captured by default (`torch.compile(mode="reduce-overhead")`,
    modules/rqvae.py:141) -- the whole step (forward, HIP quantisation kernels, backward, all-reduce, fused AdamW) is captured into a
    hipGraph and replayed on full-size batches BY DEFAULT when the batch is below 4096 rows and there is no gradient accumulation
    (`use_hip_graph=None`, the signature's default: auto; `True` / `False` force it; configs/*_graph.gin bind True explicitly):
    1.24 -> 0.35 ms per step at batch 640 on MI355X (tools/bench_small_batch.py, `secondary.small_batch` of the bench line).  The training
    schedule is the reference's either way: the short batch that ends an epoch (drop_last=False, train_rqvae.py:82-88) has a constant
    size, so it is a SECOND captured shape replayed from its own graph (round 6; `graph_epoch_tail=False` = round 5's form: tail
    eager + one re-capture per epoch -- 1194 vs 3258 iterations/s on the reference's 12 101-item corpus, profiles/r06_epoch_tail_ab.txt);
    both graphs are re-captured (together, all warm-ups first) only after an eval / tokenisation / checkpoint excursion.
wandb is optional (not installed here): with `wandb_logging=True` and no wandb module, metrics are printed.
"""
import os
import time
from typing import List

import numpy as np
import torch
from rqhip.optim import FlatAdamW

from data.processed import ItemData, RecDataset
from modules.quantize import QuantizeForwardMode
from modules.rqvae import RqVae
from modules.tokenizer.semids import SemanticIdTokenizer
from modules.utils import parse_config
from rqhip import dist as rqdist
from rqhip.autograd import loss_scale
from rqhip import tuning

try:
    import gin
except ImportError:  # pragma: no cover
    from rqhip import ginlite as gin

try:
    import wandb  # noqa: F401
    _HAVE_WANDB = True
except ImportError:  # pragma: no cover
    wandb = None
    _HAVE_WANDB = False


class _DeviceBatcher:
    """Endless stream of random batches gathered on the device: a shuffled epoch at a time, without
    replacement, short final batch kept (what BatchSampler(RandomSampler(ds), bs, drop_last=False) wrapped in
    `cycle` yields in the reference, train_rqvae.py:82-89)."""

    def __init__(self, dataset: ItemData, batch_size: int, generator: torch.Generator | None = None) -> None:
        self.dataset, self.batch_size, self.generator = dataset, batch_size, generator
        self._perm, self._pos = None, 0

    def __iter__(self):
        return self

    def __next__(self):
        n = len(self.dataset)
        if self._perm is None or self._pos >= n:
            self._perm = torch.randperm(n, generator=self.generator)
            self._pos = 0
        idx = self._perm[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        return self.dataset[idx]

    def epoch(self):
        """One pass in random order (the eval loop of train_rqvae.py:236-256)."""
        perm = torch.randperm(len(self.dataset), generator=self.generator)
        for start in range(0, len(perm), self.batch_size):
            yield self.dataset[perm[start:start + self.batch_size]]


def _id_diversity(tokenizer: SemanticIdTokenizer, index_dataset: ItemData, n_layers: int, codebook_size: int) -> dict:
    """Entropy / codebook usage / duplicate statistics of train_rqvae.py:272-292."""
    tokenizer.reset()
    # every rank calls this (see the training loop), so the explicit sharded form is safe here
    corpus_ids = tokenizer.precompute_corpus_ids(index_dataset, sharded=rqdist.world_size() > 1)
    n = corpus_ids.shape[0]
    log = {"max_id_duplicates": (corpus_ids[:, -1].max() / n).item()}
    _, counts = torch.unique(corpus_ids[:, :-1], dim=0, return_counts=True)
    p = counts / n
    log["rqvae_entropy"] = (-(p * torch.log(p)).sum()).item()
    for cid in range(n_layers):
        _, counts = torch.unique(corpus_ids[:, cid], return_counts=True)
        log[f"codebook_usage_{cid}"] = len(counts) / codebook_size
    return log


class _GraphedStep:
    """One optimisation step captured into a hipGraph (torch.cuda.CUDAGraph) and replayed on a static batch."""

    def __init__(self, model, optimizer, reducer, batch_size: int, feature_dim: int, device, gumbel_t: float) -> None:
        self.x = torch.zeros((batch_size, feature_dim), device=device)
        self.batch_size = batch_size
        self.graph = None
        self.captures = 0
        self._model, self._opt, self._reducer, self._t = model, optimizer, reducer, gumbel_t
        self._seed = torch.ones((), dtype=torch.float32, device=device)
        self.out = None

    def _step(self):
        from data.schemas import SeqBatch
        self._reducer.zero_()
        out = self._model(SeqBatch(None, None, None, self.x, None, None), gumbel_t=self._t)
        self._reducer.arm()
        out.loss.backward(gradient=self._seed)      # (a cached ones(): `backward()` fills a fresh one every step -- a launch)
        self._reducer.allreduce_mean()     # (one rank: nothing; several: the RCCL all-reduces are part of the captured graph)
        self._opt.step()
        return out

    def warm_up(self, x: torch.Tensor) -> None:
        """The two warm-up steps graph capture needs, on batch `x`; their optimizer updates are rolled back (parameters and optimizer state
        are snapshotted and restored), so a capture -- the first one or a re-capture after an eager excursion -- does not advance training."""
        import copy
        if not self._opt.state:
            raise RuntimeError("_GraphedStep.capture: run a few eager steps first -- the optimizer creates its state lazily in its "
                               "first step, and a creation recorded into the graph would reset the moments at every replay")
        self.x.copy_(x)
        self.captures += 1
        if self.captures in (10, 100, 1000):     # (every eval pass / corpus tokenisation / checkpoint costs one)
            print(f"use_hip_graph: capture #{self.captures} of the {self.batch_size}-row step (two rolled-back warm-up steps + a capture each)",
                  flush=True)
        params = [p.detach().clone() for p in self._model.parameters()]
        opt_state = copy.deepcopy(self._opt.state_dict())
        # the warm-up steps must not advance any random stream either (Gumbel noise, dropout): an eager run and a graphed run
        # of the same seed see the same draws
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):       # warm-up on a side stream, as graph capture requires
                for _ in range(2):
                    self._step()
        finally:                                # (also when a warm-up step raised: the caller falls back to eager from where it was)
            torch.cuda.current_stream().wait_stream(side)
            with torch.no_grad():
                for p, saved in zip(self._model.parameters(), params):
                    p.copy_(saved)
            self._opt.load_state_dict(opt_state)
            torch.set_rng_state(rng_cpu)
            torch.cuda.set_rng_state(rng_dev)

    def capture_only(self) -> None:
        """Record one step on the static batch (nothing executes); `warm_up` must have run since the last eager work."""
        self._reducer.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = self._step()
        # keep the static output buffers but not the captured step's autograd graph: its AccumulateGrad nodes are
        # bound to the capture stream and would poison any later eager step
        self.out = type(out)(*[v.detach() for v in out])
        del out

    def capture(self, x: torch.Tensor) -> None:
        """warm_up + capture_only for one step shape."""
        self.warm_up(x)
        self.capture_only()

    def run(self, x: torch.Tensor):
        self.x.copy_(x)
        self.graph.replay()
        return self.out

    def invalidate(self) -> None:
        """Drop the captured graph; the next full-size step captures a fresh one.  Called after every eager
        excursion (eval pass, corpus tokenisation, checkpoint): library GEMM workspaces (rocBLAS reallocates its
        device memory when a new shape needs more) and other lazily created state may have moved under the
        captured kernels -- replaying across such an excursion faulted or hung on ROCm 7.0."""
        self.graph = None
        self.out = None


def _capture_or_eager(graphs: dict, batches: dict, world: int) -> dict:
    """Capture every step shape of `graphs` ({rows: _GraphedStep}; `batches`: {rows: a batch of that many rows}) -- the full batch and,
    when an epoch ends in a short batch, that size too -- or return {}: a model variant that cannot be captured (a quantiser shape on the
    torch-operator path with a host synchronisation, a library that allocates inside the step) trains eagerly with a message instead of
    ending the run.  ALL warm-up steps (eager launches, rolled back) run before the FIRST capture: eager work between a capture and its
    replays is what made replays fault on ROCm 7.0 (library workspaces move), and the warm-up of one shape is eager work for the other.
    With several ranks the outcome is agreed on (a rank that fell back while the others replay would issue another sequence of
    collectives)."""
    ok = True
    try:
        for rows, g in graphs.items():
            g.warm_up(batches[rows])
        for g in graphs.values():
            g.capture_only()
    except Exception as e:  # noqa: BLE001
        ok = False
        for g in graphs.values():
            g.invalidate()
        print(f"use_hip_graph: capturing the training step failed ({type(e).__name__}: {str(e)[:200]}); falling back to the eager step",
              flush=True)
    if world > 1:
        import torch.distributed as _dist
        flags = [None] * world
        _dist.all_gather_object(flags, ok)
        ok = all(flags)
    return graphs if ok else {}


@gin.configurable
def train(
    iterations=50000,
    batch_size=64,
    learning_rate=0.0001,
    weight_decay=0.01,
    dataset_folder="dataset/ml-1m",
    dataset=RecDataset.ML_1M,
    pretrained_rqvae_path=None,
    save_dir_root="out/",
    use_kmeans_init=True,
    split_batches=True,
    amp=False,
    wandb_logging=False,
    do_eval=True,
    force_dataset_process=False,
    mixed_precision_type="fp16",
    gradient_accumulate_every=1,
    save_model_every=1000000,
    eval_every=50000,
    commitment_weight=0.25,
    vae_n_cat_feats=18,
    vae_input_dim=18,
    vae_embed_dim=16,
    vae_hidden_dims=[18, 18],
    vae_codebook_size=32,
    vae_codebook_normalize=False,
    vae_codebook_mode=QuantizeForwardMode.GUMBEL_SOFTMAX,
    vae_sim_vq=False,
    vae_n_layers=3,
    dataset_split="beauty",
    log_every=100,
    use_hip_graph=None,
    mlp_arith=None,
    graph_epoch_tail=True,
):
    params = dict(locals())
    del split_batches  # every rank always draws its own full batch (reference behaviour with a bare dataloader)
    if amp:
        raise NotImplementedError(
            f"amp=True ({mixed_precision_type}): the HIP quantisation kernels are fp32 only, as the parity "
            "contract (bit-exact ids vs the fp32 reference) requires")
    if not torch.cuda.is_available():
        raise RuntimeError("train_rqvae needs a ROCm GPU: the quantisation path has no CPU implementation")

    tuning.enable_tuned_gemms()  # fp32 library-GEMM selections for the encoder/decoder
    if mlp_arith is not None:
        # the arithmetic of the encoder / decoder GEMMs at batches of 4096 rows and more: "f16x2" (default), "fp32" (library GEMMs at every
        # batch size: results independent of the batching), "bf16x3" (round 3, A/B)
        from rqhip import linear as _lin
        _lin.use_arith(mlp_arith)
    rank, local_rank, world = rqdist.init_from_env("cuda")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    is_main = rank == 0
    print(f"Device: {device} (rank {rank}/{world})")

    def make(split):
        return ItemData(root=dataset_folder, dataset=dataset, force_process=force_dataset_process and split != "eval",
                        train_test_split=split, split=dataset_split).to_device(device)

    train_dataset = make("train" if do_eval else "all")
    train_batches = _DeviceBatcher(train_dataset, batch_size)
    eval_batches = _DeviceBatcher(make("eval"), batch_size) if do_eval else None
    index_dataset = make("all") if do_eval else train_dataset

    model = RqVae(
        input_dim=vae_input_dim, embed_dim=vae_embed_dim, hidden_dims=vae_hidden_dims,
        codebook_size=vae_codebook_size, codebook_kmeans_init=use_kmeans_init and pretrained_rqvae_path is None,
        codebook_normalize=vae_codebook_normalize, codebook_sim_vq=vae_sim_vq, codebook_mode=vae_codebook_mode,
        n_layers=vae_n_layers, n_cat_features=vae_n_cat_feats, commitment_weight=commitment_weight,
    ).to(device)
    # None = auto: launch-bound batches replay the step from a hipGraph (the reference's forward is graph-captured by default too)
    # (auto only where the step's collectives can be captured: one rank, or RCCL -- a gloo group cannot be)
    import torch.distributed as _dist
    capturable_comm = world == 1 or (_dist.is_initialized() and _dist.get_backend() == "nccl")
    graphable = ((batch_size < 4096 and capturable_comm) if use_hip_graph is None else bool(use_hip_graph)) and gradient_accumulate_every == 1
    # the reference's AdamW update (train_rqvae.py:136-138) as ONE kernel over all parameters (rqhip/optim.py, csrc/adamw.hip; the state
    # dict is torch.optim.AdamW's, so checkpoints interchange); graph-safe by construction
    optimizer = FlatAdamW(params=model.parameters(), lr=learning_rate, weight_decay=weight_decay)

    use_wandb = wandb_logging and is_main and _HAVE_WANDB
    if wandb_logging and is_main and not _HAVE_WANDB:
        print("wandb is not installed: metrics go to stdout")
    if use_wandb:
        wandb.login()
        wandb.init(project="rq-vae-training", config=params)

    start_iter = 0
    if pretrained_rqvae_path is not None:
        model.load_pretrained(pretrained_rqvae_path)
        state = torch.load(pretrained_rqvae_path, map_location=device, weights_only=False)
        optimizer.load_state_dict(state["optimizer"])
        start_iter = state["iter"] + 1

    rqdist.broadcast_module(model)
    reducer = rqdist.FlatGradReducer(model.parameters()).attach(model)
    seed_one = None      # the backward's seed, created once on the loss's device

    tokenizer = SemanticIdTokenizer(
        input_dim=vae_input_dim, hidden_dims=vae_hidden_dims, output_dim=vae_embed_dim,
        codebook_size=vae_codebook_size, n_layers=vae_n_layers, n_cat_feats=vae_n_cat_feats,
        rqvae_weights_path=pretrained_rqvae_path, rqvae_codebook_normalize=vae_codebook_normalize,
        rqvae_sim_vq=vae_sim_vq)
    tokenizer.rq_vae = model

    t = 0.2  # the reference's constant gumbel temperature (train_rqvae.py:177)
    if world > 1:
        # A replayed step issues another sequence of collectives than an eager one (early + late runs vs one whole-buffer all-reduce), so
        # every rank must take the same branch on the same iterations.  That follows from identical batchers -- the same number of rows and
        # the same batch size on every rank -- and from the same graph decision; gathered on EVERY rank whatever its own decision (a rank
        # that decided "eager" from its own inputs while the others capture would hang in the first collective).
        mine = (len(train_dataset), int(batch_size), int(gradient_accumulate_every), bool(graphable))
        seen = [None] * world
        _dist.all_gather_object(seen, mine)
        if any(other != mine for other in seen):
            if rank == 0 and any(g for *_x, g in seen):
                print(f"use_hip_graph: ranks disagree on (rows, batch_size, accumulate, graph) = {seen}; every rank takes the eager step")
            graphable = False
    # The step shapes of the run: full batches and, when the corpus is not a multiple of the batch size, the short batch that ends every
    # epoch (always the same size: train_rqvae.py:82-89's BatchSampler(drop_last=False)).  BOTH are captured and replayed (round 6), so an
    # epoch has no eager step and no re-capture: round 5 trained the tail eagerly and re-captured after it -- two rolled-back warm-up
    # steps, a snapshot of the optimizer state and a capture per epoch, more than the 18 replays of an Amazon-Beauty epoch cost together.
    graphs = {}
    if graphable:
        n_rows = len(train_dataset)
        # (graph_epoch_tail=False: round 5's form -- only full batches replayed, the tail eager and a re-capture behind it; A/B)
        sizes = [s for s in ((batch_size, n_rows % batch_size) if graph_epoch_tail or n_rows < batch_size else (batch_size,)) if 0 < s <= n_rows]
        graphs = {s: _GraphedStep(model, optimizer, reducer, s, vae_input_dim, device, t) for s in dict.fromkeys(sizes)}
        # several ranks: the RCCL all-reduces of FlatGradReducer are captured with the step (one graph per rank and shape, replayed in
        # lockstep: every rank runs the same sequence of batch sizes)
    graph_after = start_iter + 3  # a few eager steps first (k-means init, allocator warm-up)
    window: List[torch.Tensor] = []
    shown = (float("nan"),) * 3
    t0 = time.time()
    # One process drives one GPU: the autograd engine's per-device worker thread buys nothing here and costs a thread hand-off per
    # backward (0.13 ms of a 0.9 ms eager batch-640 step, tools/eager_host_profile.py) -- the backward runs on this thread.
    engine_threads = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(False)
    try:
        for it in range(start_iter, start_iter + 1 + iterations):
            model.train()
            if it == 0 and use_kmeans_init:
                # lazy k-means init of every level on its own residuals of the first <= 20 000 items
                # (train_rqvae.py:178-183).  With several ranks each takes its block of those rows through the model and the
                # Lloyd iterations all-reduce the [K, D+1] sums || counts (init/kmeans.py): every rank
                # ends with the same codebooks -- unlike the reference, whose ranks would each seed their own
                n_warm = min(20000, len(train_dataset))
                lo, hi = rqdist.shard_bounds(n_warm)
                for layer in model.layers:
                    layer.kmeans_rows_sharded = world > 1
                try:
                    model(train_dataset[torch.arange(lo, hi)], t)  # output (and its autograd graph) dropped at once
                finally:
                    for layer in model.layers:
                        layer.kmeans_rows_sharded = False

            data = next(train_batches) if gradient_accumulate_every == 1 else None
            g_step = graphs.get(data.x.shape[0]) if (graphs and it >= graph_after and data is not None) else None
            if g_step is not None and g_step.graph is None:
                # (re-)capture every shape at once; a shape whose batch is not at hand warms up on the first rows of the corpus -- the
                # warm-up steps are rolled back and a capture executes nothing, so which rows they see does not matter
                at_hand = {rows: (data.x if rows == data.x.shape[0] else train_dataset[torch.arange(rows)].x) for rows in graphs}
                graphs = _capture_or_eager(graphs, at_hand, world)
                g_step = graphs.get(data.x.shape[0])
            if g_step is not None and g_step.graph is not None:
                model_output = g_step.run(data.x)
                total_loss = model_output.loss.detach()
            else:
                reducer.zero_()
                total_loss = 0
                for micro in range(gradient_accumulate_every):
                    data = data if data is not None else next(train_batches)
                    if micro + 1 == gradient_accumulate_every:
                        reducer.arm()      # the last backward of the step: finished gradients go on the wire under the encoder's
                    with loss_scale(1.0 / gradient_accumulate_every):   # hint for the speculative recon-loss gradient
                        model_output = model(data, gumbel_t=t)
                    loss = model_output.loss / gradient_accumulate_every
                    if seed_one is None or seed_one.device != loss.device:
                        seed_one = torch.ones((), dtype=torch.float32, device=loss.device)
                    loss.backward(gradient=seed_one)     # (a cached ones(): `backward()` fills a fresh one every step -- a launch)
                    total_loss = total_loss + loss.detach()
                    # keep only detached values: a live autograd graph from an eager step would pin AccumulateGrad
                    # nodes to the default stream and break the hipGraph capture of a later step
                    model_output = type(model_output)(*[v.detach() for v in model_output])
                    del loss
                    data = None
                reducer.allreduce_mean()
                optimizer.step()
                for g_any in graphs.values():
                    g_any.invalidate()     # an eager step ran between two replays (only the first iterations and accumulation steps are)

            window.append(torch.stack([total_loss, model_output.reconstruction_loss.detach(),
                                       model_output.rqvae_loss.detach()]))  # stack copies: safe with graph-static outputs
            window = window[-1000:]
            if it % log_every == 0:
                shown = tuple(torch.stack(window).mean(dim=0).tolist())  # the only host sync of a normal step
                if is_main:
                    rate = (it - start_iter + 1) / max(time.time() - t0, 1e-9)
                    print(f"iter {it}: loss: {shown[0]:.4f}, rl: {shown[1]:.4f}, vl: {shown[2]:.4f} ({rate:.1f} it/s)")

            log = {}
            last = it + 1 == iterations
            if use_wandb or (wandb_logging and is_main and it % log_every == 0):
                norms = model_output.embs_norm.mean(dim=0)
                log.update({f"emb_avg_norm_{i}": norms[i].item() for i in range(vae_n_layers)})
                log.update({"learning_rate": optimizer.param_groups[0]["lr"], "total_loss": float(total_loss),
                            "reconstruction_loss": model_output.reconstruction_loss.item(),
                            "rqvae_loss": model_output.rqvae_loss.item(), "temperature": t,
                            "p_unique_ids": model_output.p_unique_ids.item()})

            if do_eval and ((it + 1) % eval_every == 0 or last):
                model.eval()
                rows = []
                with torch.no_grad():
                    for batch in eval_batches.epoch():
                        out = model(batch, gumbel_t=t)
                        rows.append(torch.stack([out.loss, out.reconstruction_loss, out.rqvae_loss]))
                if rows:
                    ev = torch.stack(rows).mean(dim=0).tolist()
                    log.update({"eval_total_loss": ev[0], "eval_reconstruction_loss": ev[1], "eval_rqvae_loss": ev[2]})

            if (it + 1) % eval_every == 0 or last:
                model.eval()
                log.update(_id_diversity(tokenizer, index_dataset, vae_n_layers, vae_codebook_size))  # collective

            if is_main and ((it + 1) % save_model_every == 0 or last):
                os.makedirs(save_dir_root, exist_ok=True)
                state = {"iter": it, "model": model.state_dict(), "model_config": model.config,
                         "optimizer": optimizer.state_dict()}
                if getattr(train_dataset, "synthetic", False):
                    state["data"] = "synthetic"   # extra key: a checkpoint trained on noise says so
                torch.save(state, save_dir_root + f"checkpoint_{it}.pt")

            if graphs and ((do_eval and ((it + 1) % eval_every == 0 or last)) or (it + 1) % eval_every == 0
                           or last or (it + 1) % save_model_every == 0):
                for g_any in graphs.values():
                    g_any.invalidate()

            if is_main and log:
                if use_wandb:
                    wandb.log(log)
                elif wandb_logging:
                    print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in log.items()})

    finally:
        torch.autograd.set_multithreading_enabled(engine_threads)
    if use_wandb:
        wandb.finish()
    rqdist.barrier()
    return {"loss": shown[0], "reconstruction_loss": shown[1], "rqvae_loss": shown[2],
            "graph_captures": {rows: g.captures for rows, g in graphs.items()}}


if __name__ == "__main__":
    parse_config()
    train()
