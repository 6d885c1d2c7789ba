#!/usr/bin/env python3
"""Reference-generated semantic ids at the BASELINE.json shapes (tests/golden/parity_c2.npz, parity_c4.npz).

Runs THE REFERENCE ITSELF (imported from /root/reference, CPU, stub gin -- see gen_golden.py) on

  c2: the config-2 model 768 -> [512,256,128] -> 32, 3 x 256 codes, ALL 100 000 rows of
      X = normalize(randn(100000, 768, seed 1234))
  c4: the config-4 shape 4 x 1024 codes, D = 32, 300 000 rows of the same generator

with weights from `torch.manual_seed(0)` construction and codebooks from the reference's own k-means
(`init.kmeans.Kmeans`, 15 iterations, level by level on the first 20 000 rows' residuals), in eval mode and in
STE training mode, and commits

  * the ids of every row (uint8 / uint16; the STE ids as a sparse difference against the eval ids),
  * the codebooks, the per-row quantize loss of the first 16 384 rows, the three scalar losses of the
    full-batch training step (decoder + ReconstructionLoss from the reference's modules),
  * sha256 of the inputs, the weights and the encoder output `res0` (so a test can tell whether the host it
    runs on reproduces the reference's encoder bits),
  * the "hard rows": the 2048 rows with the smallest top-2 distance margin, with their `res0` bits,
  * the same level loop run by the reference on exactly regenerable latents Z (numpy PCG64 normal draws scaled
    to the encoder output's spread), so that the HIP kernel can be compared with the reference on ALL rows on
    any host without needing the reference's encoder bits,
  * the evidence of the oracle-vs-reference comparison made here: which rows differ, at which level, with which
    tie margin, and what an fp64 evaluation of the two candidate codes says.

The reference cannot travel to the GPU box; this script and its outputs can.  ~6 min, ~14 GB peak.
Re-run:  python oracle/gen_parity_fixtures.py [c2] [c4]
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import gen_golden  # noqa: E402
from oracle import rq_oracle as o  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
INPUT_DIM, HIDDEN, EMBED, BETA = 768, [512, 256, 128], 32, 0.25
KMEANS_ROWS, KMEANS_ITERS = 20000, 15
N_HARD, N_LOSS = 2048, 16384
TAU = 1e-6          # rows with a relative top-2 margin below this are "flagged near-ties"


def sha(a) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.view(np.uint8).reshape(-1)).hexdigest()


def synthetic_items(n: int, seed: int = 1234) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n, INPUT_DIM, generator=g), dim=-1)


def regenerable_latents(n: int, scale: float, seed: int = 4321) -> np.ndarray:
    """Latents any host reproduces bit for bit: numpy's PCG64 ziggurat normals (integer + scalar libm code),
    scaled in float64 and rounded once to fp32."""
    z = np.random.Generator(np.random.PCG64(seed)).standard_normal((n, EMBED))
    return (z * scale).astype(np.float32)


def weights_sha(model) -> str:
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        if "embedding" in k:
            continue
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().numpy()).view(np.uint8).reshape(-1))
    return h.hexdigest()


def first_divergence(a: np.ndarray, b: np.ndarray):
    """rows where id tuples differ and the first level at which they do (a, b: [N,L])."""
    diff = a != b
    rows = np.nonzero(diff.any(axis=1))[0]
    return rows, diff[rows].argmax(axis=1)


def adjudicate_fp64(resid: np.ndarray, cb_l: np.ndarray, id_a: int, id_b: int):
    """(d_a - d_b) / d_min in float64 for one residual row against two codes of one level."""
    x = resid.astype(np.float64)
    da = ((x - cb_l[id_a].astype(np.float64)) ** 2).sum()
    db = ((x - cb_l[id_b].astype(np.float64)) ** 2).sum()
    return (da - db) / max(min(da, db), 1e-300)


def make_case(tag: str, n_rows: int, K: int, L: int, n_latent_rows: int, q, r, km):
    t0 = time.time()
    torch.manual_seed(0)
    model = r.RqVae(input_dim=INPUT_DIM, embed_dim=EMBED, hidden_dims=HIDDEN, codebook_size=K,
                    codebook_kmeans_init=False, codebook_mode=q.QuantizeForwardMode.STE, n_layers=L,
                    commitment_weight=BETA, n_cat_features=0)
    X = synthetic_items(n_rows)
    # codebooks: the reference's k-means, level by level on the residuals of the first 20 000 rows
    model.eval()
    with torch.no_grad():
        res = model.encode(X[:KMEANS_ROWS])
        for l, layer in enumerate(model.layers):
            np.random.seed(100 + l)
            torch.manual_seed(100 + l)
            out = km.Kmeans(k=K, max_iters=KMEANS_ITERS).run(res.clone())
            layer.embedding.weight.copy_(out.centroids)
            res = res - layer(res, temperature=0.2).embeddings
            print(f"[{tag}] k-means level {l} done ({time.time() - t0:.0f} s)", flush=True)
    cbs = np.stack([l.embedding.weight.detach().numpy() for l in model.layers]).astype(np.float32)

    save = {"n_rows": np.int64(n_rows), "K": np.int64(K), "L": np.int64(L), "beta": np.float32(BETA),
            "tau": np.float32(TAU), "codebooks": cbs, "x_seed": np.int64(1234),
            "x_sha256": sha(X.numpy()), "weights_sha256": weights_sha(model)}
    id_dtype = np.uint8 if K <= 256 else np.uint16
    ref = {}
    with torch.no_grad():
        res0 = model.encode(X)
        for training in (False, True):
            model.train(training)
            sem = model.get_semantic_ids(X, 0.2)
            ref[training] = (sem.sem_ids.numpy().astype(np.int64), sem.quantize_loss.numpy().astype(np.float32),
                             sem.embeddings)
        # full-batch training-step losses from the reference's own decoder / loss modules (RqVae.forward itself
        # builds a B x B x L boolean tensor for p_unique_ids and cannot run at this batch size)
        model.train(True)
        embs = ref[True][2]
        x_hat = model.decode(embs.sum(axis=-1))
        recon = model.reconstruction_loss(x_hat, X)
        qloss = torch.from_numpy(ref[True][1])
        save["train_loss"] = np.float64((recon + qloss).mean().item())
        save["train_reconstruction_loss"] = np.float64(recon.mean().item())
        save["train_rqvae_loss"] = np.float64(qloss.mean().item())
        model.eval()
    res0_np = res0.numpy().astype(np.float32)
    save["res0_sha256"] = sha(res0_np)
    ids_eval, ids_train = ref[False][0], ref[True][0]
    save["ids_eval"] = ids_eval.astype(id_dtype)
    tr_rows = np.nonzero((ids_eval != ids_train).any(axis=1))[0]
    save["ids_train_diff_rows"] = tr_rows.astype(np.int64)
    save["ids_train_diff_vals"] = ids_train[tr_rows].astype(id_dtype)
    save["loss_eval_head"] = ref[False][1][:N_LOSS]
    save["loss_train_head"] = ref[True][1][:N_LOSS]
    print(f"[{tag}] reference done ({time.time() - t0:.0f} s); eval/train id rows differing: {len(tr_rows)}", flush=True)

    # ---- the oracle on the reference's res0: where do they differ, and are those rows near-ties? ----------
    evidence = {}
    min_margin = None
    for training, mode in ((False, o.MODE_EVAL), (True, o.MODE_STE)):
        oc = o.rq_forward(res0_np, cbs, mode, BETA, want_margin=True)
        oid = oc["ids"].T
        rid = ref[training][0]
        rows, lev = first_divergence(oid, rid)
        marg = oc["tie_margin"][lev, rows]
        adj = [adjudicate_fp64(oc["residuals"][l, i], cbs[l], int(oid[i, l]), int(rid[i, l])) for i, l in zip(rows, lev)]
        p = "train" if training else "eval"
        evidence[p] = (rows, lev, marg, adj)
        save[f"oracle_mismatch_rows_{p}"] = rows.astype(np.int64)
        save[f"oracle_mismatch_level_{p}"] = lev.astype(np.int64)
        save[f"oracle_mismatch_margin_{p}"] = marg.astype(np.float32)
        save[f"oracle_mismatch_fp64_gap_{p}"] = np.asarray(adj, dtype=np.float64)   # < 0: the oracle's code is closer
        save[f"oracle_loss_max_abs_err_{p}"] = np.float64(np.abs(oc["loss"] - ref[training][1]).max())
        save[f"oracle_flagged_rows_{p}"] = np.int64((oc["tie_margin"].min(axis=0) < TAU).sum())
        if not training:
            min_margin = oc["tie_margin"].min(axis=0)
        print(f"[{tag}] oracle vs reference ({p}): {len(rows)} rows differ of {n_rows}; margins "
              f"{[float(f'{m:.2e}') for m in marg]}; fp64 gaps {[float(f'{a:.2e}') for a in adj]}; "
              f"loss max abs err {save[f'oracle_loss_max_abs_err_{p}']:.2e}; rows flagged < {TAU:g}: "
              f"{int(save[f'oracle_flagged_rows_{p}'])}", flush=True)
        assert (marg < TAU).all(), "an oracle/reference mismatch is NOT a flagged near-tie"

    # ---- hard rows: smallest eval-mode margins, with their res0 bits ----------------------------------------
    hard = np.sort(np.argsort(min_margin, kind="stable")[:N_HARD])
    save["hard_rows"] = hard.astype(np.int64)
    save["hard_res0"] = res0_np[hard]

    # ---- the reference's level loop on regenerable latents (encoder bypassed) -------------------------------
    scale = float(res0_np.std())
    Z = regenerable_latents(n_latent_rows, scale)
    save["z_seed"] = np.int64(4321)
    save["z_scale"] = np.float64(scale)
    save["z_sha256"] = sha(Z)
    class _Bypass(torch.nn.Module):          # get_semantic_ids reads next(self.encoder.parameters()).dtype
        def __init__(self):
            super().__init__()
            self.dummy = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return x

    enc = model.encoder
    model.encoder = _Bypass()
    with torch.no_grad():
        zsem = model.get_semantic_ids(torch.from_numpy(Z), 0.2)
    model.encoder = enc
    zid = zsem.sem_ids.numpy().astype(np.int64)
    save["z_ids_eval"] = zid.astype(id_dtype)
    save["z_loss_eval_head"] = zsem.quantize_loss.numpy().astype(np.float32)[:N_LOSS]
    oz = o.rq_forward(Z, cbs, o.MODE_EVAL, BETA, want_margin=True)
    rows, lev = first_divergence(oz["ids"].T, zid)
    marg = oz["tie_margin"][lev, rows]
    save["z_oracle_mismatch_rows"] = rows.astype(np.int64)
    save["z_oracle_mismatch_level"] = lev.astype(np.int64)
    save["z_oracle_mismatch_margin"] = marg.astype(np.float32)
    print(f"[{tag}] regenerable latents: oracle vs reference {len(rows)} rows differ of {n_latent_rows}, margins "
          f"{[float(f'{m:.2e}') for m in marg]}", flush=True)
    assert (marg < TAU).all()

    path = os.path.join(OUT, f"parity_{tag}.npz")
    np.savez_compressed(path, **save)
    print(f"[{tag}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB, {time.time() - t0:.0f} s)", flush=True)


def main():
    torch.set_num_threads(8)
    q, r, km, _sem, _sch = gen_golden.import_reference()
    which = [a for a in sys.argv[1:] if a in ("c2", "c4")] or ["c2", "c4"]
    if "c2" in which:
        make_case("c2", 100_000, 256, 3, 100_000, q, r, km)
    if "c4" in which:
        make_case("c4", 300_000, 1024, 4, 100_000, q, r, km)


if __name__ == "__main__":
    main()
