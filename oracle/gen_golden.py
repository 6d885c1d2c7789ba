#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF on CPU.

The reference (EdoardoBotta/RQ-VAE-Recommender, mounted read-only at /root/reference) has no tests,
golden vectors or checkpoints for this path, so the oracle is pinned against
outputs of the reference's own modules, imported in place -- nothing is copied from it.  The
reference cannot travel to the GPU box, so the (small) vectors are committed together with this
script.  Re-run:  python oracle/gen_golden.py   (needs /root/reference; CPU only; ~1 min)

`gin-config` is not installable here; a 3-function stand-in is registered in sys.modules before the
import (decorators become identities).  `data.processed` (torch_geometric/polars) is stubbed for
modules/tokenizer/semids.py, which only uses the name `ItemData` as a type annotation.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("RQ_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_stubs() -> None:
    gin = types.ModuleType("gin")
    gin.constants_from_enum = lambda cls=None, **kw: cls
    gin.configurable = lambda f=None, **kw: f if callable(f) else (lambda g: g)
    gin.parse_config_file = lambda *a, **k: None
    sys.modules.setdefault("gin", gin)
    dp = types.ModuleType("data.processed")
    dp.ItemData = object
    sys.modules.setdefault("data.processed", dp)


def import_reference():
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from modules import quantize as q  # noqa
    from modules import rqvae as r  # noqa
    from init import kmeans as km  # noqa
    from modules.tokenizer import semids as sem  # noqa
    from data import schemas as sch  # noqa
    return q, r, km, sem, sch


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=True)


def gen_quantize(q):
    """Quantize.forward (quantize.py:104-163), one level, eval / STE / rotation + autograd grads."""
    cases = [
        ("a", 96, 32, 256, 11),
        ("b", 50, 64, 256, 12),   # ml32m width, B not a multiple of anything
        ("c", 33, 16, 32, 13),    # class defaults of train(): D=16, K=32
        ("d", 7, 8, 5, 14),       # K not a multiple of 32, tiny
    ]
    for tag, B, D, K, seed in cases:
        for mode_name, fm, training in (("eval", q.QuantizeForwardMode.STE, False),
                                        ("ste", q.QuantizeForwardMode.STE, True),
                                        ("rotation", q.QuantizeForwardMode.ROTATION_TRICK, True)):
            g = torch.Generator().manual_seed(seed)
            x = torch.randn(B, D, generator=g) * 0.7
            cb = torch.randn(K, D, generator=g) * 0.6
            layer = q.Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, forward_mode=fm,
                               commitment_weight=0.25)
            with torch.no_grad():
                layer.embedding.weight.copy_(cb)
            layer.train(training)
            xr = x.clone().requires_grad_(True)
            out = layer(xr, temperature=0.2)
            g_emb = torch.randn(B, D, generator=g)
            g_loss = torch.rand(B, generator=g)
            (out.embeddings * g_emb).sum().add((out.loss * g_loss).sum()).backward()
            np.savez_compressed(
                os.path.join(OUT, f"quantize_{mode_name}_{tag}.npz"),
                x=np32(x), codebook=np32(cb), beta=np.float32(0.25),
                ids=out.ids.numpy().astype(np.int64), embeddings=np32(out.embeddings), loss=np32(out.loss),
                g_emb=np32(g_emb), g_loss=np32(g_loss), grad_x=np32(xr.grad),
                grad_codebook=np32(layer.embedding.weight.grad))


def gen_cosine(q):
    """QuantizeDistance.COSINE (quantize.py:118-124) -- never selected by RqVae, kept for API parity."""
    for tag, B, D, K, seed, fm, training in (("a", 60, 32, 64, 61, q.QuantizeForwardMode.STE, True),
                                             ("b", 40, 16, 32, 62, q.QuantizeForwardMode.ROTATION_TRICK, True),
                                             ("c", 33, 24, 20, 63, q.QuantizeForwardMode.STE, False)):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, D, generator=g)
        cb = torch.randn(K, D, generator=g) * torch.rand(K, 1, generator=g).add(0.2)   # varied code norms
        layer = q.Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, forward_mode=fm,
                           distance_mode=q.QuantizeDistance.COSINE, commitment_weight=0.25)
        with torch.no_grad():
            layer.embedding.weight.copy_(cb)
        layer.train(training)
        xr = x.clone().requires_grad_(True)
        out = layer(xr, temperature=0.2)
        g_emb = torch.randn(B, D, generator=g)
        g_loss = torch.rand(B, generator=g)
        (out.embeddings * g_emb).sum().add((out.loss * g_loss).sum()).backward()
        np.savez_compressed(
            os.path.join(OUT, f"cosine_{tag}.npz"), x=np32(x), codebook=np32(cb), training=np.bool_(training),
            rotation=np.bool_(fm == q.QuantizeForwardMode.ROTATION_TRICK),
            ids=out.ids.numpy().astype(np.int64), embeddings=np32(out.embeddings), loss=np32(out.loss),
            g_emb=np32(g_emb), g_loss=np32(g_loss), grad_x=np32(xr.grad),
            grad_codebook=np32(layer.embedding.weight.grad))


def gen_gumbel(q):
    """Training-mode GUMBEL_SOFTMAX level (quantize.py:131-136; gumbel.py:8-20).  The uniform noise the
    reference drew is recovered by re-seeding and calling torch.rand with the same shape."""
    for tag, B, D, K, seed, T in (("a", 40, 32, 256, 21, 0.2), ("b", 17, 16, 32, 22, 0.5)):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, D, generator=g) * 0.5
        cb = torch.randn(K, D, generator=g) * 0.5
        layer = q.Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False,
                           forward_mode=q.QuantizeForwardMode.GUMBEL_SOFTMAX, commitment_weight=0.25)
        with torch.no_grad():
            layer.embedding.weight.copy_(cb)
        layer.train()
        xr = x.clone().requires_grad_(True)
        torch.manual_seed(1000 + seed)
        out = layer(xr, temperature=T)
        torch.manual_seed(1000 + seed)
        U = torch.rand(B, K)
        g_emb = torch.randn(B, D, generator=g)
        g_loss = torch.rand(B, generator=g)
        (out.embeddings * g_emb).sum().add((out.loss * g_loss).sum()).backward()
        np.savez_compressed(
            os.path.join(OUT, f"gumbel_{tag}.npz"),
            x=np32(x), codebook=np32(cb), U=np32(U), temperature=np.float32(T), beta=np.float32(0.25),
            ids=out.ids.numpy().astype(np.int64), embeddings=np32(out.embeddings), loss=np32(out.loss),
            g_emb=np32(g_emb), g_loss=np32(g_loss), grad_x=np32(xr.grad),
            grad_codebook=np32(layer.embedding.weight.grad))


def gen_rqvae(q, r, sch):
    """RqVae.get_semantic_ids / RqVae.forward (rqvae.py:118-175) on a small model, all weights stored.
    forward is taken un-compiled (RqVae.forward._torchdynamo_orig_callable) to skip inductor."""
    fwd = getattr(r.RqVae.forward, "_torchdynamo_orig_callable", None)
    assert fwd is not None, "expected torch.compile-wrapped RqVae.forward"
    cfgs = [
        ("small_ste", dict(input_dim=48, embed_dim=16, hidden_dims=[32, 24], codebook_size=32, n_layers=3,
                           n_cat_features=0), q.QuantizeForwardMode.STE, 80, 31),
        ("small_rot", dict(input_dim=48, embed_dim=16, hidden_dims=[32, 24], codebook_size=32, n_layers=3,
                           n_cat_features=0), q.QuantizeForwardMode.ROTATION_TRICK, 80, 32),
        ("wide_ste", dict(input_dim=64, embed_dim=32, hidden_dims=[48], codebook_size=256, n_layers=4,
                          n_cat_features=0), q.QuantizeForwardMode.STE, 200, 33),
        ("cat_ste", dict(input_dim=40, embed_dim=8, hidden_dims=[24], codebook_size=16, n_layers=2,
                         n_cat_features=6), q.QuantizeForwardMode.STE, 64, 34),
        # the two codebook options of train(): vae_codebook_normalize (level-0 codebook + encoder output L2-normalised,
        # rqvae.py:71,83) and vae_sim_vq (a DxD linear on every codebook, quantize.py:76)
        ("norm_ste", dict(input_dim=40, embed_dim=16, hidden_dims=[24], codebook_size=32, n_layers=3,
                          n_cat_features=0, codebook_normalize=True), q.QuantizeForwardMode.STE, 72, 35),
        ("simvq_rot", dict(input_dim=40, embed_dim=16, hidden_dims=[24], codebook_size=32, n_layers=2,
                           n_cat_features=0, codebook_sim_vq=True), q.QuantizeForwardMode.ROTATION_TRICK, 72, 36),
    ]
    for tag, kw, mode, B, seed in cfgs:
        torch.manual_seed(seed)
        model = r.RqVae(codebook_kmeans_init=False, codebook_mode=mode, commitment_weight=0.25, **kw)
        g = torch.Generator().manual_seed(seed)
        # spread the codebooks like trained ones (uniform(0,1) init collapses every row onto few codes)
        with torch.no_grad():
            for l, layer in enumerate(model.layers):
                layer.embedding.weight.copy_(torch.randn(kw["codebook_size"], kw["embed_dim"], generator=g)
                                             * (0.35 / (l + 1)))
        x = torch.nn.functional.normalize(torch.randn(B, kw["input_dim"], generator=g), dim=-1)
        if kw["n_cat_features"]:
            x[:, -kw["n_cat_features"]:] = (torch.rand(B, kw["n_cat_features"], generator=g) > 0.7).float()
        x[B // 2] = x[B // 4]  # a guaranteed duplicate row -> p_unique_ids < 1
        batch = sch.SeqBatch(user_ids=None, ids=None, ids_fut=None, x=x, x_fut=None, seq_mask=None)
        save = {f"param::{k}": np32(v) for k, v in model.state_dict().items()}
        save["x"] = np32(x)
        save["beta"] = np.float32(0.25)
        for training in (True, False):
            model.train(training)
            model.zero_grad()
            sem = model.get_semantic_ids(x, 0.2)
            out = fwd(model, batch, 0.2)
            out.loss.backward()
            p = "train_" if training else "eval_"
            save[p + "res0"] = np32(model.encode(x))
            save[p + "embeddings"] = np32(sem.embeddings)        # [B,D,L]
            save[p + "residuals"] = np32(sem.residuals)          # [B,D,L]
            save[p + "sem_ids"] = sem.sem_ids.numpy().astype(np.int64)  # [B,L]
            save[p + "quantize_loss"] = np32(sem.quantize_loss)
            save[p + "loss"] = np32(out.loss)
            save[p + "reconstruction_loss"] = np32(out.reconstruction_loss)
            save[p + "rqvae_loss"] = np32(out.rqvae_loss)
            save[p + "embs_norm"] = np32(out.embs_norm)
            save[p + "p_unique_ids"] = np32(out.p_unique_ids)
            for k, v in model.named_parameters():
                save[p + "grad::" + k] = np32(v.grad)
        np.savez_compressed(os.path.join(OUT, f"rqvae_{tag}.npz"), **save)


def gen_kmeans(km):
    """Kmeans.run (kmeans.py:61-72).  Records np.random.choice's rows and every torch.randint draw."""
    cases = [("a", 300, 16, 8, 41, None), ("b", 500, 32, 64, 42, 3), ("dup", 64, 8, 6, 43, None)]
    for tag, B, D, K, seed, max_iters in cases:
        g = torch.Generator().manual_seed(seed)
        centers = torch.randn(K, D, generator=g) * 2.0
        x = centers[torch.randint(0, K, (B,), generator=g)] + 0.3 * torch.randn(B, D, generator=g)
        np.random.seed(seed)
        init_idx = np.random.choice(B, K, replace=False)
        if tag == "dup":
            # two identical seed rows -> the later one can never win the argmin -> empty cluster -> reseed
            x[init_idx[3]] = x[init_idx[1]]
        draws = []
        real_randint = torch.randint

        def logging_randint(*a, **k):
            out = real_randint(*a, **k)
            draws.append(int(out.reshape(-1)[0]))
            return out

        np.random.seed(seed)
        torch.manual_seed(seed)
        torch.randint = logging_randint
        try:
            algo = km.Kmeans(k=K, max_iters=max_iters)
            out = algo.run(x.clone())
        finally:
            torch.randint = real_randint
        np.savez_compressed(
            os.path.join(OUT, f"kmeans_{tag}.npz"),
            x=np32(x), k=np.int64(K), seed=np.int64(seed), max_iters=np.int64(-1 if max_iters is None else max_iters),
            init_idx=init_idx.astype(np.int64), reseed_draws=np.asarray(draws, dtype=np.int64),
            centroids=np32(out.centroids), assignment=out.assignment.numpy().astype(np.int64))


def gen_dedup(q, r, sem, sch):
    """SemanticIdTokenizer.precompute_corpus_ids (semids.py:76-110): [N, L+1], last column = number of
    earlier rows with the same tuple.  N > 512 so that the cross-batch branch (:100-104) runs."""
    torch.manual_seed(51)
    tok = sem.SemanticIdTokenizer(input_dim=24, output_dim=8, hidden_dims=[16], codebook_size=4, n_layers=3,
                                  n_cat_feats=0)
    g = torch.Generator().manual_seed(51)
    with torch.no_grad():
        for l, layer in enumerate(tok.rq_vae.layers):
            layer.embedding.weight.copy_(torch.randn(4, 8, generator=g) * (0.4 / (l + 1)))
    N = 700
    X = torch.randn(N, 24, generator=g)

    class DS:
        def __len__(self):
            return N

        def __getitem__(self, idx):
            # like ItemData.__getitem__ (data/processed.py:74-86) for a list index: ids is [1, len(idx)]
            ids = torch.tensor(idx).unsqueeze(0)
            return sch.SeqBatch(user_ids=-torch.ones_like(ids.squeeze(0)), ids=ids,
                                ids_fut=-torch.ones_like(ids.squeeze(0)), x=X[idx],
                                x_fut=-torch.ones_like(ids.squeeze(0)), seq_mask=torch.ones_like(ids, dtype=bool))

    ids = tok.precompute_corpus_ids(DS())
    save = {f"param::{k}": np32(v) for k, v in tok.rq_vae.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "dedup_a.npz"), x=np32(X), corpus_ids=ids.numpy().astype(np.int64), **save)


def gen_sid_match():
    """Decoder-side consumers: EncoderDecoderRetrievalModel._check_valid_prefix
    (modules/model.py:169-182, called unbound on a stand-in `self` that only carries `.codebooks`) and
    TopKAccumulator (evaluate/metrics.py:7-28)."""
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from modules import model as m  # noqa  (pulls in transformers' T5; ~15 s)
    from evaluate.metrics import TopKAccumulator

    def ref_valid(corpus, prefix, batch_size):
        self_ = types.SimpleNamespace(codebooks=torch.from_numpy(corpus))
        return m.EncoderDecoderRetrievalModel._check_valid_prefix(self_, torch.from_numpy(prefix),
                                                                  batch_size=batch_size).numpy()

    for tag, N, H, K, P, seed in (("a", 500, 3, 8, 300, 71), ("b", 2000, 4, 256, 700, 72), ("c", 1, 3, 5, 40, 73)):
        rng = np.random.default_rng(seed)
        corpus = rng.integers(0, K, size=(N, H)).astype(np.int64)
        if tag == "b":
            corpus[:, -1] = rng.integers(0, 3, size=N)     # dedup-like last column
            corpus[7] = [2**40 + 5, -3, 2**33, 0]            # full 64-bit values must be compared, not hashes
        save = {"corpus": corpus}
        for h in range(1, H + 1):
            real = corpus[rng.integers(0, N, size=P // 2), :h]
            rand = rng.integers(-1 if tag == "b" else 0, K + 1, size=(P - P // 2, h)).astype(np.int64)
            prefix = np.concatenate([real, rand], axis=0)
            if tag == "b" and h >= 2:
                prefix[0, :h] = corpus[7, :h]
                prefix[1, :h] = corpus[7, :h]
                prefix[1, 0] = 5                              # same low 32 bits as 2**40+5: must NOT match row 7
            prefix = prefix[rng.permutation(P)]
            save[f"prefix_h{h}"] = prefix
            save[f"valid_h{h}"] = ref_valid(corpus, prefix, batch_size=97)
        np.savez_compressed(os.path.join(OUT, f"prefix_{tag}.npz"), **save)

    for tag, B, K, D, vocab, seed in (("a", 64, 10, 3, 6, 81), ("b", 33, 20, 4, 256, 82)):
        rng = np.random.default_rng(seed)
        acc = TopKAccumulator(ks=[1, 5, 10])
        save = {}
        for part in range(2):
            actual = rng.integers(0, vocab, size=(B, D)).astype(np.int64)
            top_k = rng.integers(0, vocab, size=(B, K, D)).astype(np.int64)
            for b in range(0, B, 2):                          # plant matches at assorted ranks, some twice
                k = int(rng.integers(0, K))
                top_k[b, k] = actual[b]
                if b % 4 == 0:
                    top_k[b, min(K - 1, k + 2)] = actual[b]
            pos = (torch.from_numpy(actual)[:, None, :] == torch.from_numpy(top_k)).all(-1)
            found, rank = pos.max(-1)
            save[f"actual_{part}"] = actual
            save[f"top_k_{part}"] = top_k
            save[f"rank_{part}"] = np.where(found.numpy(), rank.numpy(), -1).astype(np.int64)
            acc.accumulate(actual=torch.from_numpy(actual), top_k=torch.from_numpy(top_k))
        red = acc.reduce()
        save["metric_names"] = np.array(sorted(red))
        save["metric_values"] = np.array([red[k] for k in sorted(red)], dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, f"topk_{tag}.npz"), **save)


def gen_config3_step(q, r, km, sch):
    """BASELINE configuration 3 at its real shape (reference configs/rqvae_ml32m.gin:4-25): 768 -> [512,256,128] -> 64,
    3 x 256 codes, ROTATION_TRICK, batch 64, AdamW lr 1e-4 / weight decay 0.01, k-means initialised codebooks.
    One full training step of the reference (forward, backward, optimizer step).  The MLP weights are regenerable
    (torch.manual_seed(0) construction; a sha256 pins them), so only inputs, codebooks and compact views of the
    1.15 M-parameter gradients / updates are stored: everything for the codebooks, norm + sum + a 8 x 16 corner for
    every MLP weight."""
    import hashlib
    fwd = r.RqVae.forward._torchdynamo_orig_callable
    torch.manual_seed(0)
    model = r.RqVae(input_dim=768, embed_dim=64, hidden_dims=[512, 256, 128], codebook_size=256,
                    codebook_kmeans_init=False, codebook_mode=q.QuantizeForwardMode.ROTATION_TRICK, n_layers=3,
                    commitment_weight=0.25, n_cat_features=0)
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        if "embedding" not in k:
            h.update(k.encode())
            h.update(np.ascontiguousarray(v.numpy()).view(np.uint8).reshape(-1))
    g = torch.Generator().manual_seed(303)
    x_init = torch.nn.functional.normalize(torch.randn(4096, 768, generator=g), dim=-1)
    x = torch.nn.functional.normalize(torch.randn(64, 768, generator=g), dim=-1)
    model.eval()
    with torch.no_grad():                      # codebooks: the reference's k-means, level by level (10 iterations)
        res = model.encode(x_init)
        for l, layer in enumerate(model.layers):
            np.random.seed(300 + l)
            torch.manual_seed(300 + l)
            layer.embedding.weight.copy_(km.Kmeans(k=256, max_iters=10).run(res.clone()).centroids)
            res = res - layer(res, temperature=0.2).embeddings
    save = {"x": np32(x), "weights_sha256": h.hexdigest(), "lr": np.float64(1e-4), "weight_decay": np.float64(0.01),
            "codebooks": np.stack([np32(l.embedding.weight) for l in model.layers])}
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    batch = sch.SeqBatch(user_ids=None, ids=None, ids_fut=None, x=x, x_fut=None, seq_mask=None)
    opt.zero_grad()
    out = fwd(model, batch, 0.2)
    with torch.no_grad():
        save["sem_ids"] = model.get_semantic_ids(x, 0.2).sem_ids.numpy().astype(np.int64)
    out.loss.backward()
    opt.step()
    for name in ("loss", "reconstruction_loss", "rqvae_loss", "p_unique_ids"):
        save[name] = np32(getattr(out, name))
    save["embs_norm"] = np32(out.embs_norm)

    def compact(prefix, k, t):
        if "embedding" in k:
            save[f"{prefix}::{k}"] = np32(t)
        else:
            save[f"{prefix}_stat::{k}"] = np.array([t.norm().item(), t.sum().item()], np.float64)
            save[f"{prefix}_corner::{k}"] = np32(t[:8, :16])

    for k, v in model.named_parameters():
        compact("grad", k, v.grad)
        compact("delta", k, v.detach() - before[k])
    np.savez_compressed(os.path.join(OUT, "config3_step.npz"), **save)


def gen_wide(q, km):
    """Shapes / mode combinations the HIP kernels do not cover (rq-vae-recommender_amd/rqhip/wide.py runs them as PyTorch-ROCm
    operators): COSINE x GUMBEL_SOFTMAX (quantize.py:118-136), embed_dim > 128, Gumbel-softmax with more than 1024 codes,
    k-means on rows wider than 128.  The uniform noise is recorded as in gen_gumbel."""
    cases = [("cosine_gumbel", 24, 16, 20, 71, q.QuantizeForwardMode.GUMBEL_SOFTMAX, q.QuantizeDistance.COSINE, True, 0.5),
             ("d160_ste", 21, 160, 40, 72, q.QuantizeForwardMode.STE, q.QuantizeDistance.L2, True, 0.2),
             ("d160_rotation", 9, 160, 40, 73, q.QuantizeForwardMode.ROTATION_TRICK, q.QuantizeDistance.L2, True, 0.2),
             ("d160_eval", 17, 160, 40, 74, q.QuantizeForwardMode.STE, q.QuantizeDistance.L2, False, 0.2),
             ("gumbel_k1100", 12, 8, 1100, 75, q.QuantizeForwardMode.GUMBEL_SOFTMAX, q.QuantizeDistance.L2, True, 0.3)]
    for tag, B, D, K, seed, fm, dm, training, T in cases:
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, D, generator=g) * 0.5
        cb = torch.randn(K, D, generator=g) * torch.rand(K, 1, generator=g).add(0.3)
        layer = q.Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, forward_mode=fm, distance_mode=dm,
                           commitment_weight=0.25)
        with torch.no_grad():
            layer.embedding.weight.copy_(cb)
        layer.train(training)
        xr = x.clone().requires_grad_(True)
        torch.manual_seed(1000 + seed)
        out = layer(xr, temperature=T)
        torch.manual_seed(1000 + seed)
        U = torch.rand(B, K)
        g_emb = torch.randn(B, D, generator=g)
        g_loss = torch.rand(B, generator=g)
        (out.embeddings * g_emb).sum().add((out.loss * g_loss).sum()).backward()
        np.savez_compressed(
            os.path.join(OUT, f"wide_{tag}.npz"),
            x=np32(x), codebook=np32(cb), U=np32(U), temperature=np.float32(T), training=np.bool_(training),
            mode=np.int64(fm.value), cosine=np.bool_(dm == q.QuantizeDistance.COSINE),
            ids=out.ids.numpy().astype(np.int64), embeddings=np32(out.embeddings), loss=np32(out.loss),
            g_emb=np32(g_emb), g_loss=np32(g_loss), grad_x=np32(xr.grad),
            grad_codebook=np32(layer.embedding.weight.grad))
    # k-means on 160-wide rows (kmeans.py:61-72), seeds as in gen_kmeans
    B, D, K, seed = 240, 160, 6, 76
    g = torch.Generator().manual_seed(seed)
    centers = torch.randn(K, D, generator=g) * 2.0
    x = centers[torch.randint(0, K, (B,), generator=g)] + 0.3 * torch.randn(B, D, generator=g)
    np.random.seed(seed)
    torch.manual_seed(seed)
    out = km.Kmeans(k=K).run(x.clone())
    np.savez_compressed(os.path.join(OUT, "wide_kmeans_d160.npz"), x=np32(x), k=np.int64(K), seed=np.int64(seed),
                        centroids=np32(out.centroids), assignment=out.assignment.numpy().astype(np.int64))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--only-sid-match" in sys.argv:
        gen_sid_match()
        return
    q, r, km, sem, sch = import_reference()
    if "--only-config3" in sys.argv:
        gen_config3_step(q, r, km, sch)
        return
    if "--only-wide" in sys.argv:
        gen_wide(q, km)
        return
    gen_quantize(q)
    gen_gumbel(q)
    gen_cosine(q)
    gen_wide(q, km)
    gen_rqvae(q, r, sch)
    gen_kmeans(km)
    gen_dedup(q, r, sem, sch)
    gen_config3_step(q, r, km, sch)
    gen_sid_match()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(os.listdir(OUT))} fixtures, {total / 1024:.0f} KiB -> {os.path.abspath(OUT)}")


if __name__ == "__main__":
    main()
