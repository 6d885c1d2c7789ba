"""Pins the CPU oracle (oracle/rq_oracle.c) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by oracle/gen_golden.py running the reference's own
Quantize / RqVae / Kmeans / SemanticIdTokenizer on CPU.  Bars (BASELINE.json north_star): semantic
ids bit-exact, losses within 1e-5 (fp32); other floating outputs within the tolerances written here.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import rq_oracle as o

LOSS_ATOL = 1e-5  # north_star: "losses within 1e-5 fp32"


def _names(pat):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, pat)))


def _rel_close(a, b, rtol, atol):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


MODES = {"eval": o.MODE_EVAL, "ste": o.MODE_STE, "rotation": o.MODE_ROTATION}


@pytest.mark.parametrize("name", _names("quantize_*.npz"))
def test_quantize_level_forward_and_backward(name):
    g = load_golden(name)
    mode = MODES[name.split("_")[1]]
    out = o.rq_forward(g["x"], g["codebook"][None], mode, float(g["beta"]))
    assert np.array_equal(out["ids"][0], g["ids"]), "semantic ids must be bit-exact"
    # losses here are O(10-100) (unit-variance rows); 1e-5 is applied relative to that scale
    _rel_close(out["loss"], g["loss"], rtol=2e-6, atol=LOSS_ATOL)
    _rel_close(out["embs"][0], g["embeddings"], rtol=1e-5, atol=2e-6)
    if mode == o.MODE_EVAL:
        return  # eval fixtures were produced with grad through emb_out; checked in the rq-level test
    g_x, g_cb = o.rq_backward(g["x"], g["codebook"][None], mode, float(g["beta"]), out["ids"],
                              g_embs=g["g_emb"][None], g_loss=g["g_loss"])
    _rel_close(g_x, g["grad_x"], rtol=2e-5, atol=2e-5)
    _rel_close(g_cb[0], g["grad_codebook"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", _names("quantize_eval_*.npz"))
def test_quantize_eval_backward(name):
    g = load_golden(name)
    out = o.rq_forward(g["x"], g["codebook"][None], o.MODE_EVAL, float(g["beta"]))
    g_x, g_cb = o.rq_backward(g["x"], g["codebook"][None], o.MODE_EVAL, float(g["beta"]), out["ids"],
                              g_embs=g["g_emb"][None], g_loss=g["g_loss"])
    _rel_close(g_x, g["grad_x"], rtol=2e-5, atol=2e-5)
    _rel_close(g_cb[0], g["grad_codebook"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", _names("gumbel_*.npz"))
def test_gumbel_level(name):
    g = load_golden(name)
    T, beta = float(g["temperature"]), float(g["beta"])
    out = o.gumbel_forward(g["x"], g["codebook"], g["U"], T, beta)
    assert np.array_equal(out["ids"], g["ids"])
    _rel_close(out["emb"], g["embeddings"], rtol=1e-4, atol=1e-5)
    _rel_close(out["loss"], g["loss"], rtol=1e-5, atol=LOSS_ATOL)
    g_x, g_cb = o.gumbel_backward(g["x"], g["codebook"], g["U"], T, beta, g_emb=g["g_emb"], g_loss=g["g_loss"])
    scale = max(1.0, float(np.abs(g["grad_x"]).max()))
    _rel_close(g_x, g["grad_x"], rtol=1e-4, atol=2e-5 * scale)
    scale = max(1.0, float(np.abs(g["grad_codebook"]).max()))
    _rel_close(g_cb, g["grad_codebook"], rtol=1e-4, atol=2e-5 * scale)


def _codebooks(g):
    L = len([k for k in g if k.startswith("param::layers.") and k.endswith("embedding.weight")])
    return np.stack([g[f"param::layers.{l}.embedding.weight"] for l in range(L)])


def _plain(names):
    """fixtures whose codebooks are the raw embedding weights (no sim_vq / normalisation in front of the kernel)"""
    return [n for n in names if "norm" not in n and "simvq" not in n]


@pytest.mark.parametrize("name", _plain(_names("rqvae_*.npz")))
@pytest.mark.parametrize("phase", ["train", "eval"])
def test_rq_stack_matches_reference_get_semantic_ids(name, phase):
    g = load_golden(name)
    cbs = _codebooks(g)
    mode = o.MODE_EVAL if phase == "eval" else (o.MODE_ROTATION if "rot" in name else o.MODE_STE)
    p = phase + "_"
    out = o.rq_forward(g[p + "res0"], cbs, mode, float(g["beta"]))
    assert np.array_equal(out["ids"].T, g[p + "sem_ids"]), "semantic-id tuples must be bit-exact"
    _rel_close(out["loss"], g[p + "quantize_loss"], rtol=2e-6, atol=LOSS_ATOL)
    _rel_close(out["embs"].transpose(1, 2, 0), g[p + "embeddings"], rtol=1e-5, atol=1e-6)
    _rel_close(out["residuals"].transpose(1, 2, 0), g[p + "residuals"], rtol=1e-5, atol=1e-6)
    _rel_close(out["embs_norm"], g[p + "embs_norm"], rtol=1e-5, atol=1e-6)
    _rel_close(out["emb_sum"], g[p + "embeddings"].sum(-1), rtol=1e-5, atol=1e-6)
    n = o.count_rows_without_later_duplicate(out["ids"])
    assert abs(n / out["ids"].shape[1] - float(g[p + "p_unique_ids"])) < 1e-7
    assert float(g[p + "p_unique_ids"]) < 1.0  # the fixture plants a duplicate row
    _rel_close(out["loss"].mean(), g[p + "rqvae_loss"], rtol=1e-6, atol=LOSS_ATOL)


@pytest.mark.parametrize("name", _plain(_names("rqvae_*.npz")))
def test_rq_stack_codebook_gradients(name):
    """d loss / d codebooks of RqVae.forward: in the STE and rotation modes the codebooks only see the
    quantize loss (gradient 1/B per row), and the decoder gradient reaches the levels through emb_sum."""
    g = load_golden(name)
    cbs = _codebooks(g)
    mode = o.MODE_ROTATION if "rot" in name else o.MODE_STE
    res0 = g["train_res0"]
    B = res0.shape[0]
    out = o.rq_forward(res0, cbs, mode, float(g["beta"]))
    gl = np.full((B,), 1.0 / B, np.float32)
    _, g_cb = o.rq_backward(res0, cbs, mode, float(g["beta"]), out["ids"], g_loss=gl)
    for l in range(cbs.shape[0]):
        _rel_close(g_cb[l], g[f"train_grad::layers.{l}.embedding.weight"], rtol=1e-4, atol=1e-7)


def test_ordered_codebook_gradient_is_a_reordering_of_the_plain_one():
    """rqo_rq_backward_ordered (the restatement of the HIP kernels' fixed summation order, parameterised by the geometry
    rqhip_rq_backward_plan reports): one workgroup with one whole-batch unit IS the plain ascending-row sum, any other
    geometry differs from it by fp32 rounding only, g_res0 never changes, and the result is a function of the geometry."""
    rng = np.random.default_rng(3)
    B, D, K, L = 1000, 32, 64, 3
    x = (rng.standard_normal((B, D)) * 0.7).astype(np.float32)
    cbs = (rng.standard_normal((L, K, D)) * 0.4).astype(np.float32)
    ids = o.rq_forward(x, cbs, o.MODE_STE, 0.25)["ids"]
    g = dict(g_embsum=(rng.standard_normal((B, D)) / B).astype(np.float32), g_loss=rng.random(B).astype(np.float32))
    r_res0, r_cb = o.rq_backward(x, cbs, o.MODE_STE, 0.25, ids, **g)
    one_res0, one_cb = o.rq_backward(x, cbs, o.MODE_STE, 0.25, ids, order=(1, 1, B), **g)
    assert np.array_equal(one_res0, r_res0) and np.array_equal(one_cb, r_cb)
    seen = []
    for order in ((4, 8, 32), (16, 1, 64), (7, 1, 128), (256, 1, 64)):
        o_res0, o_cb = o.rq_backward(x, cbs, o.MODE_STE, 0.25, ids, order=order, **g)
        assert np.array_equal(o_res0, r_res0)
        np.testing.assert_allclose(o_cb, r_cb, rtol=1e-5, atol=1e-6 * float(np.abs(r_cb).max()))
        again = o.rq_backward(x, cbs, o.MODE_STE, 0.25, ids, order=order, **g)[1]
        assert np.array_equal(again, o_cb)
        seen.append(o_cb)
    assert any(not np.array_equal(seen[0], s) for s in seen[1:])   # the order matters at the last bit


@pytest.mark.parametrize("name", _names("kmeans_*.npz"))
def test_kmeans_matches_reference(name):
    g = load_golden(name)
    draws = list(g["reseed_draws"])
    it = iter(draws)
    max_iters = None if int(g["max_iters"]) < 0 else int(g["max_iters"])
    cent, assign, _ = o.kmeans_run(g["x"], g["init_idx"], reseed_draws=lambda: next(it), max_iters=max_iters)
    assert np.array_equal(assign, g["assignment"])
    _rel_close(cent, g["centroids"], rtol=1e-5, atol=1e-6)
    assert next(it, None) is None, "every recorded torch.randint draw must have been consumed"
    if "dup" in name:
        assert len(draws) > 0, "fixture is supposed to exercise the empty-cluster reseed"


def test_dedup_column_matches_precompute_corpus_ids():
    g = load_golden("dedup_a.npz")
    corpus = g["corpus_ids"]            # [N, L+1]
    ids = np.ascontiguousarray(corpus[:, :-1].T)
    assert np.array_equal(o.dedup_rank(ids), corpus[:, -1])
    assert corpus[:, -1].max() > 0
    # and the ids themselves, through the encoder-free part of the path
    W = [g[k] for k in sorted(k for k in g if k.startswith("param::encoder"))]
    h = g["x"]
    for i, w in enumerate(W):
        h = h @ w.T
        if i != len(W) - 1:
            h = np.maximum(h, 0)
    out = o.rq_forward(h.astype(np.float32), _codebooks(g), o.MODE_EVAL, 0.25)
    mism = (out["ids"].T != corpus[:, :-1]).any(axis=1).mean()
    assert mism == 0.0, f"{mism:.4f} of rows differ"


@pytest.mark.parametrize("name", _names("prefix_*.npz"))
def test_prefix_valid_matches_check_valid_prefix(name):
    """modules/model.py:169-182 run by oracle/gen_golden.py -> valid_h*; the C restatement must agree exactly."""
    g = load_golden(name)
    corpus = g["corpus"]
    seen = set()
    for h in range(1, corpus.shape[1] + 1):
        got = o.prefix_valid(corpus, g[f"prefix_h{h}"])
        assert np.array_equal(got, g[f"valid_h{h}"])
        seen.update(np.unique(got).tolist())
    if corpus.shape[0] > 1:
        assert seen == {False, True}  # both outcomes occur in the fixture
    # h == 0: all() over no columns -> every prefix is valid when the corpus is not empty
    assert o.prefix_valid(corpus, np.zeros((3, 0), np.int64)).all()
    assert not o.prefix_valid(np.zeros((0, 3), np.int64), np.zeros((3, 2), np.int64)).any()


@pytest.mark.parametrize("name", _names("topk_*.npz"))
def test_topk_first_match_and_metrics_match_topk_accumulator(name):
    """evaluate/metrics.py:16-28: first-match positions bit-exact, reduced metrics to fp32 rounding."""
    g = load_golden(name)
    ranks = []
    for part in range(2):
        r = o.topk_first_match(g[f"actual_{part}"], g[f"top_k_{part}"])
        assert np.array_equal(r, g[f"rank_{part}"])
        ranks.append(r)
    assert (np.concatenate(ranks) == -1).any() and (np.concatenate(ranks) > 0).any()
    got = o.topk_metrics(np.concatenate(ranks))
    for k, v in zip(g["metric_names"].tolist(), g["metric_values"].tolist()):
        assert abs(got[k] - v) <= 1e-6, (k, got[k], v)


def test_argmin_semantics_ties_and_nan():
    """quantize.py:128 / torch.min: first index on ties, a NaN distance wins."""
    x = np.zeros((3, 4), np.float32)
    x[1] = [1, 2, 3, 4]
    x[2] = np.nan
    cb = np.zeros((1, 6, 4), np.float32)
    cb[0, 2] = cb[0, 4] = [1, 2, 3, 4]       # duplicated code: row 1 must take index 2
    out = o.rq_forward(x, cb, o.MODE_EVAL)
    assert out["ids"][0].tolist() == [0, 2, 0]
    cb[0, 3, 1] = np.nan                       # NaN code: dist[:,3] is NaN for every row -> index 3
    out = o.rq_forward(x[:2], cb, o.MODE_EVAL)
    assert out["ids"][0].tolist() == [3, 3]


def test_empty_batch():
    out = o.rq_forward(np.zeros((0, 8), np.float32), np.ones((2, 4, 8), np.float32), o.MODE_STE)
    assert out["ids"].shape == (2, 0) and out["loss"].shape == (0,)


def test_recon_loss_oracle_matches_definition():
    rng = np.random.default_rng(3)
    for n in (768, 48, 7, 130):
        a = rng.standard_normal((9, n)).astype(np.float32)
        b = rng.standard_normal((9, n)).astype(np.float32)
        ref = ((a.astype(np.float64) - b) ** 2).sum(-1)
        np.testing.assert_allclose(o.recon_loss(a, b), ref, rtol=2e-6)


@pytest.mark.parametrize("name", ["rqvae_small_ste.npz", "rqvae_wide_ste.npz"])
def test_torch_port_cpu_baseline_program_matches_reference(name):
    """oracle/torch_port.py (what bench.py times as cpu_baseline) computes the reference's training loss and
    gradients: same weights and batch as the golden fixture -> same loss, reconstruction part and grads."""
    import torch
    from oracle import torch_port
    g = load_golden(name)
    enc = sorted(k for k in g if k.startswith("param::encoder"))
    dec = sorted(k for k in g if k.startswith("param::decoder"))
    m = torch_port.PortModel(input_dim=g["x"].shape[1], hidden=[g[k].shape[0] for k in enc[:-1]],
                             embed_dim=g[enc[-1]].shape[0], n_levels=_codebooks(g).shape[0],
                             codebook_size=_codebooks(g).shape[1], beta=float(g["beta"]))
    m.enc = [torch.tensor(g[k]).requires_grad_(True) for k in enc]
    m.dec = [torch.tensor(g[k]).requires_grad_(True) for k in dec]
    m.codebooks = [torch.tensor(c).requires_grad_(True) for c in _codebooks(g)]
    loss, p_unique = m.step_loss(torch.tensor(g["x"]), stat_rows=g["x"].shape[0])
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g["train_loss"]), rtol=1e-6, atol=1e-5)
    assert abs(float(p_unique) - float(g["train_p_unique_ids"])) < 1e-7
    for l, c in enumerate(m.codebooks):
        np.testing.assert_allclose(c.grad.numpy(), g[f"train_grad::layers.{l}.embedding.weight"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(m.enc[0].grad.numpy(), g["train_grad::" + enc[0][len("param::"):]], rtol=1e-4, atol=1e-7)


def test_linear_chain_orientation_masks_and_epilogues_on_exact_inputs():
    """rqo_linear_chain (the seam kernel's GEMMs): on small-integer operands every partial sum is exact, so the FMA chain must equal the
    integer matrix product -- for both weight orientations, with the ReLU backward on load (xmask) and the three epilogues; on real-valued
    operands it stays within fp32 accumulation error of the fp64 product."""
    from oracle import rq_oracle as o
    rng = np.random.default_rng(7)
    x = rng.integers(-8, 9, size=(37, 128)).astype(np.float32)
    w = rng.integers(-8, 9, size=(32, 128)).astype(np.float32)
    want = (x.astype(np.int64) @ w.astype(np.int64).T).astype(np.float32)
    assert np.array_equal(o.linear_chain(x, w), want)
    assert np.array_equal(o.linear_chain(x, np.ascontiguousarray(w.T), transposed=True), want)
    assert np.array_equal(o.linear_chain(x, w, epilogue=1), np.maximum(want, 0.0))
    xm = rng.standard_normal((37, 128)).astype(np.float32)
    xm[0, :5] = 0.0                                             # a mask value of exactly 0 drops the entry (threshold_backward: mask <= 0)
    masked = np.where(xm > 0, x, 0.0).astype(np.float32)
    assert np.array_equal(o.linear_chain(x, w, xmask=xm), (masked.astype(np.int64) @ w.astype(np.int64).T).astype(np.float32))
    om = rng.standard_normal((37, 32)).astype(np.float32)
    assert np.array_equal(o.linear_chain(x, w, epilogue=3, omask=om), np.where(om > 0, want, 0.0).astype(np.float32))
    xr, wr = rng.standard_normal((64, 128)).astype(np.float32), (rng.standard_normal((32, 128)) / 11.3).astype(np.float32)
    ref = xr.astype(np.float64) @ wr.astype(np.float64).T
    bound = 129 * 2.0 ** -24 * (np.abs(xr).astype(np.float64) @ np.abs(wr).astype(np.float64).T)
    assert (np.abs(o.linear_chain(xr, wr).astype(np.float64) - ref) <= bound).all()
