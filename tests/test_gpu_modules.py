"""GPU tests of the reference-shaped Python API (modules.quantize / modules.rqvae / init.kmeans /
tokenizer) against golden outputs of the reference itself (tests/golden, see oracle/gen_golden.py):
semantic ids exact, losses within 1e-5, gradients of every parameter within fp32 tolerance."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

RQVAE_CFG = {
    "small_ste": dict(input_dim=48, embed_dim=16, hidden_dims=[32, 24], codebook_size=32, n_layers=3, n_cat_features=0),
    "small_rot": dict(input_dim=48, embed_dim=16, hidden_dims=[32, 24], codebook_size=32, n_layers=3, n_cat_features=0),
    "wide_ste": dict(input_dim=64, embed_dim=32, hidden_dims=[48], codebook_size=256, n_layers=4, n_cat_features=0),
    "cat_ste": dict(input_dim=40, embed_dim=8, hidden_dims=[24], codebook_size=16, n_layers=2, n_cat_features=6),
    "norm_ste": dict(input_dim=40, embed_dim=16, hidden_dims=[24], codebook_size=32, n_layers=3, n_cat_features=0,
                     codebook_normalize=True),
    "simvq_rot": dict(input_dim=40, embed_dim=16, hidden_dims=[24], codebook_size=32, n_layers=2, n_cat_features=0,
                      codebook_sim_vq=True),
}


def _build(tag, g):
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    mode = QuantizeForwardMode.ROTATION_TRICK if "rot" in tag else QuantizeForwardMode.STE
    m = RqVae(codebook_kmeans_init=False, codebook_mode=mode, commitment_weight=0.25, **RQVAE_CFG[tag])
    m.load_state_dict({k[len("param::"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param::")})
    return m.cuda()


@pytest.mark.parametrize("tag", sorted(RQVAE_CFG))
@pytest.mark.parametrize("phase", ["train", "eval"])
def test_rqvae_matches_reference(tag, phase):
    from data.schemas import SeqBatch
    g = load_golden(f"rqvae_{tag}.npz")
    m = _build(tag, g)
    m.train(phase == "train")
    x = torch.from_numpy(g["x"]).cuda()
    p = phase + "_"
    sem = m.get_semantic_ids(x, 0.2)
    assert sem.sem_ids.shape == g[p + "sem_ids"].shape and sem.sem_ids.dtype == torch.int64
    assert sem.sem_ids.stride() == (1, x.shape[0])                       # the reference's [B,L] view layout
    assert np.array_equal(sem.sem_ids.cpu().numpy(), g[p + "sem_ids"]), "semantic ids must be bit-exact"
    np.testing.assert_allclose(sem.quantize_loss.detach().cpu().numpy(), g[p + "quantize_loss"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(sem.embeddings.detach().cpu().numpy(), g[p + "embeddings"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sem.residuals.detach().cpu().numpy(), g[p + "residuals"], rtol=1e-5, atol=5e-6)  # res0 comes from rocBLAS vs MKL GEMMs

    batch = SeqBatch(user_ids=None, ids=None, ids_fut=None, x=x, x_fut=None, seq_mask=None)
    out = m(batch, 0.2)
    for k in ("loss", "reconstruction_loss", "rqvae_loss", "p_unique_ids"):
        np.testing.assert_allclose(getattr(out, k).detach().cpu().numpy(), g[p + k], rtol=2e-6, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(out.embs_norm.cpu().numpy(), g[p + "embs_norm"], rtol=1e-5, atol=1e-6)
    out.loss.backward()
    for name, prm in m.named_parameters():
        ref = g[p + "grad::" + name]
        got = prm.grad.cpu().numpy()
        scale = max(1e-6, float(np.abs(ref).max()))
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5 * scale, err_msg=f"{phase} grad {name}")  # rocBLAS vs MKL GEMM sums


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "quantize_*.npz"))))
def test_quantize_module_matches_reference(name):
    from modules.quantize import Quantize, QuantizeForwardMode
    g = load_golden(name)
    kind = name.split("_")[1]
    fm = QuantizeForwardMode.ROTATION_TRICK if kind == "rotation" else QuantizeForwardMode.STE
    K, D = g["codebook"].shape
    q = Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, forward_mode=fm, commitment_weight=0.25).cuda()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(g["codebook"]))
    q.train(kind != "eval")
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    out = q(x, temperature=0.2)
    assert np.array_equal(out.ids.cpu().numpy(), g["ids"])
    np.testing.assert_allclose(out.loss.detach().cpu().numpy(), g["loss"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(out.embeddings.detach().cpu().numpy(), g["embeddings"], rtol=1e-5, atol=2e-6)
    ge, gl = torch.from_numpy(g["g_emb"]).cuda(), torch.from_numpy(g["g_loss"]).cuda()
    ((out.embeddings * ge).sum() + (out.loss * gl).sum()).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["grad_x"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(q.embedding.weight.grad.cpu().numpy(), g["grad_codebook"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", ["kmeans_a.npz", "kmeans_b.npz", "kmeans_dup.npz"])
def test_kmeans_run_matches_reference(name):
    """Same numpy / torch seeds as the reference run -> same seed rows, same reseed draws, same result."""
    from init.kmeans import Kmeans, kmeans_init_
    g = load_golden(name)
    seed = int(g["seed"])
    max_iters = None if int(g["max_iters"]) < 0 else int(g["max_iters"])
    x = torch.from_numpy(g["x"]).cuda()
    np.random.seed(seed)
    torch.manual_seed(seed)
    out = Kmeans(k=int(g["k"]), max_iters=max_iters).run(x)
    assert np.array_equal(out.assignment.cpu().numpy(), g["assignment"])
    np.testing.assert_allclose(out.centroids.cpu().numpy(), g["centroids"], rtol=1e-5, atol=1e-6)
    w = torch.zeros(int(g["k"]), x.shape[1], device="cuda")
    np.random.seed(seed)
    torch.manual_seed(seed)
    if max_iters is None:
        kmeans_init_(w, x)
        np.testing.assert_allclose(w.cpu().numpy(), g["centroids"], rtol=1e-5, atol=1e-6)


def test_lazy_kmeans_init_inside_rqvae_forward():
    """quantize.py:107-108 via train_rqvae.py:178-183: the first training forward k-means-initialises every
    level on its own residuals; afterwards the fused path takes over and agrees with the level-by-level one."""
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    torch.manual_seed(0)
    np.random.seed(0)
    m = RqVae(input_dim=64, embed_dim=16, hidden_dims=[32], codebook_size=16, n_layers=3, n_cat_features=0,
              codebook_kmeans_init=True, codebook_mode=QuantizeForwardMode.STE).cuda()
    before = [l.weight.detach().clone() for l in m.layers]
    x = torch.nn.functional.normalize(torch.randn(500, 64), dim=-1).cuda()
    batch = SeqBatch(None, None, None, x, None, None)
    assert not m._can_fuse()
    out1 = m(batch, 0.2)
    assert all(l.kmeans_initted for l in m.layers) and m._can_fuse()
    assert all(not torch.equal(b, l.weight) for b, l in zip(before, m.layers))
    out2 = m(batch, 0.2)       # fused kernel, same weights
    assert torch.allclose(out1.loss, out2.loss, rtol=1e-6, atol=1e-6)
    assert torch.equal(out1.p_unique_ids, out2.p_unique_ids)


def test_tokenizer_dedup_column_matches_reference():
    from data.processed import ItemData
    from modules.tokenizer.semids import SemanticIdTokenizer
    g = load_golden("dedup_a.npz")
    tok = SemanticIdTokenizer(input_dim=24, output_dim=8, hidden_dims=[16], codebook_size=4, n_layers=3, n_cat_feats=0)
    tok.rq_vae.load_state_dict({k[len("param::"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param::")})
    tok = tok.cuda()
    X = torch.from_numpy(g["x"])
    ds = ItemData(root="/nonexistent", item_matrix=torch.cat([X, torch.zeros(X.shape[0], 768 - 24)], dim=1),
                  train_test_split="all")
    ds.item_data = ds.item_data[:, :24].contiguous()   # ItemData slices [:768]; the fixture model is 24-d
    ids = tok.precompute_corpus_ids(ds.to_device("cuda"))
    assert ids.shape == g["corpus_ids"].shape
    assert np.array_equal(ids.cpu().numpy(), g["corpus_ids"])
    assert tok.cached_ids is ids


@pytest.mark.parametrize("B,L,K", [(1, 3, 4), (700, 3, 4), (5000, 2, 3), (100_000, 3, 256), (300_000, 4, 6)])
def test_dedup_rank_vs_definition(B, L, K):
    from rqhip import ops
    g = torch.Generator().manual_seed(B + L + K)
    ids = torch.randint(0, K, (L, B), generator=g)
    rank, n = ops.dedup_rank(ids.cuda(), K)
    # definition via a stable sort on the packed key (host)
    key = torch.zeros(B, dtype=torch.int64)
    for l in range(L):
        key = key * K + ids[l]
    order = torch.sort(key, stable=True).indices
    sk = key[order]
    start = torch.ones(B, dtype=torch.bool)
    start[1:] = sk[1:] != sk[:-1]
    pos = torch.arange(B)
    run_start = torch.cummax(torch.where(start, pos, torch.zeros_like(pos)), 0).values
    want = torch.empty(B, dtype=torch.int64)
    want[order] = pos - run_start
    assert torch.equal(rank.cpu(), want)
    assert int(n) == int(start.sum())


@pytest.mark.parametrize("B,L,K", [(1, 3, 4), (64, 3, 4), (640, 3, 256), (700, 3, 4), (100_000, 3, 256), (300_000, 4, 6)])
def test_unique_fraction_is_the_references_statistic(B, L, K):
    """p_unique_ids as reference modules/rqvae.py:159-167 forms it (rows without a later duplicate / B), bit for bit in fp32; twice through the
    same cached workspace and once more under a hipGraph replay (the training step's form)."""
    from rqhip import ops
    g = torch.Generator().manual_seed(7 * B + L + K)
    ids = torch.randint(0, K, (L, B), generator=g)
    if B <= 1000:                                                    # rqvae.py:159-167 on [B, L], as written there
        t = ids.t()
        dup = (t.unsqueeze(1) == t.unsqueeze(0)).all(dim=-1)
        want = (~torch.triu(dup, diagonal=1)).all(dim=1).sum() / B
    else:                                                            # the same count without the B x B matrix
        want = torch.tensor(torch.unique(ids.t(), dim=0).shape[0]) / B
    dev = ids.cuda()
    for _ in range(2):
        got = ops.unique_fraction(dev)
        assert got.dtype == torch.float32 and got.dim() == 0
        assert got.cpu().item() == want.to(torch.float32).item()
    n = ops.dedup_rank(dev, K, want_rank=False)[1]
    assert int(n) == round(want.item() * B)
    if B <= 1000:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.unique_fraction(dev)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = ops.unique_fraction(dev)
        for _ in range(3):
            out.fill_(-1.0)
            graph.replay()
            assert out.cpu().item() == want.to(torch.float32).item()


def test_dedup_all_rows_identical():
    from rqhip import ops
    ids = torch.zeros((3, 70_000), dtype=torch.int64).cuda()
    rank, n = ops.dedup_rank(ids, 256)
    assert int(n) == 1 and torch.equal(rank.cpu(), torch.arange(70_000))


@pytest.mark.parametrize("name", ["gumbel_a.npz", "gumbel_b.npz"])
def test_quantize_module_gumbel_mode(name, monkeypatch):
    """Quantize(forward_mode=GUMBEL_SOFTMAX).train(): draws torch.rand(B, K) on its device like the reference
    (distributions/gumbel.py:10); with that draw replaced by the reference's own noise the outputs match."""
    from modules.quantize import Quantize, QuantizeForwardMode
    g = load_golden(name)
    K, D = g["codebook"].shape
    q = Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, forward_mode=QuantizeForwardMode.GUMBEL_SOFTMAX).cuda()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(g["codebook"]))
    q.train()
    U = torch.from_numpy(g["U"]).cuda()
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *a, **k: U if tuple(a[0]) == tuple(U.shape) else real_rand(*a, **k))
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    out = q(x, temperature=float(g["temperature"]))
    assert np.array_equal(out.ids.cpu().numpy(), g["ids"])
    np.testing.assert_allclose(out.embeddings.detach().cpu().numpy(), g["embeddings"], rtol=2e-4, atol=1e-5)
    ((out.embeddings * torch.from_numpy(g["g_emb"]).cuda()).sum() + (out.loss * torch.from_numpy(g["g_loss"]).cuda()).sum()).backward()
    sx = max(1.0, float(np.abs(g["grad_x"]).max()))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["grad_x"], rtol=2e-3, atol=5e-5 * sx)


def test_rqvae_gumbel_training_step_runs_level_by_level():
    """GUMBEL_SOFTMAX is the class default of RqVae (rqvae.py:47): one optimisation step must work end to end."""
    from data.schemas import SeqBatch
    from modules.rqvae import RqVae
    torch.manual_seed(0)
    m = RqVae(input_dim=64, embed_dim=16, hidden_dims=[32], codebook_size=32, n_layers=3, n_cat_features=0,
              codebook_kmeans_init=False).cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    x = torch.nn.functional.normalize(torch.randn(200, 64), dim=-1).cuda()
    m.train()
    assert not m._can_fuse()
    out = m(SeqBatch(None, None, None, x, None, None), 0.2)
    out.loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    opt.step()
    m.eval()
    assert m._can_fuse()
    sem = m.get_semantic_ids(x)
    assert sem.sem_ids.shape == (200, 3)


@pytest.mark.parametrize("name", ["cosine_a.npz", "cosine_b.npz", "cosine_c.npz"])
def test_quantize_module_cosine_distance(name):
    """QuantizeDistance.COSINE (quantize.py:118-124): ids exact vs the reference, outputs and gradients close."""
    from modules.quantize import Quantize, QuantizeDistance, QuantizeForwardMode
    g = load_golden(name)
    K, D = g["codebook"].shape
    fm = QuantizeForwardMode.ROTATION_TRICK if bool(g["rotation"]) else QuantizeForwardMode.STE
    q = Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, forward_mode=fm, distance_mode=QuantizeDistance.COSINE).cuda()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(g["codebook"]))
    q.train(bool(g["training"]))
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    out = q(x, temperature=0.2)
    assert np.array_equal(out.ids.cpu().numpy(), g["ids"])
    np.testing.assert_allclose(out.embeddings.detach().cpu().numpy(), g["embeddings"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(out.loss.detach().cpu().numpy(), g["loss"], rtol=2e-6, atol=1e-5)
    ((out.embeddings * torch.from_numpy(g["g_emb"]).cuda()).sum() + (out.loss * torch.from_numpy(g["g_loss"]).cuda()).sum()).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["grad_x"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(q.embedding.weight.grad.cpu().numpy(), g["grad_codebook"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B,D,K,max_iters", [(20000, 32, 256, None), (3000, 32, 64, 5), (5000, 16, 1024, None)])
def test_kmeans_batched_run_equals_stepwise_oracle(B, D, K, max_iters):
    """Kmeans.run (batches of device-driven iterations, one host read per batch) against the same loop taken one
    iteration at a time with the oracle's assign / update: same seed rows, same reseeds, bit-identical centroids."""
    from init.kmeans import Kmeans
    from oracle import rq_oracle as o
    g = torch.Generator().manual_seed(B + K)
    centers = torch.randn(K // 2, D, generator=g) * 2.0                 # fewer modes than codes -> empty clusters happen
    x = (centers[torch.randint(0, K // 2, (B,), generator=g)] + 0.3 * torch.randn(B, D, generator=g)).contiguous()
    np.random.seed(7)
    torch.manual_seed(7)
    out = Kmeans(k=K, max_iters=max_iters).run(x.cuda())
    # stepwise restatement (reference init/kmeans.py:33-72 with the oracle's array steps)
    np.random.seed(7)
    torch.manual_seed(7)
    xn = x.numpy()
    cent = xn[np.random.choice(B, K, replace=False)].copy()
    i = 0
    while max_iters is None or i < max_iters:
        old = cent.copy()
        assign = o.kmeans_assign(xn, cent)
        counts = o.kmeans_update(xn, assign, cent)
        for k in np.nonzero(counts == 0)[0]:
            cent[k] = xn[int(torch.randint(0, B, (1,)))]
        if o.kmeans_shift(cent, old) < 1e-10:
            break
        i += 1
    assert np.array_equal(out.assignment.cpu().numpy(), assign)
    assert np.array_equal(out.centroids.cpu().numpy().view(np.uint32), cent.view(np.uint32))


def test_kmeans_sharded_path_on_one_rank_equals_plain_run():
    """The row-sharded Lloyd loop (partial sums -> all-reduce over RCCL -> apply) with a single rank must reproduce the
    plain run bit for bit: same kernels' arithmetic, the collective is the identity."""
    import os
    import torch.distributed as dist
    from init.kmeans import Kmeans
    g = torch.Generator().manual_seed(11)
    centers = torch.randn(40, 32, generator=g) * 2.0
    x = (centers[torch.randint(0, 40, (6000,), generator=g)] + 0.3 * torch.randn(6000, 32, generator=g)).cuda()
    np.random.seed(9)
    torch.manual_seed(9)
    plain = Kmeans(k=64).run(x)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        np.random.seed(9)
        torch.manual_seed(9)
        sharded = Kmeans(k=64).run(x, sharded=True)
    finally:
        dist.destroy_process_group()
    assert torch.equal(plain.assignment, sharded.assignment)
    assert torch.equal(plain.centroids, sharded.centroids)


def test_row_sharded_training_step_equals_full_batch_step():
    """SURVEY.md section 8e, logical W-way sharding on one GPU: the mean over two row shards of the per-shard gradients
    (what the flat all-reduce + 1/W produces) equals the gradient of the full-batch step, every parameter, 1e-5."""
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
              n_cat_features=0, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda()
    with torch.no_grad():
        for l, layer in enumerate(m.layers):
            layer.embedding.weight.copy_(torch.randn(256, 32, device="cuda") * (0.05 / (l + 1)))
    x = torch.nn.functional.normalize(torch.randn(8192, 768, device="cuda"), dim=-1)
    m.train()

    def grads(rows):
        for p in m.parameters():
            p.grad = None
        m(SeqBatch(None, None, None, rows, None, None), 0.2).loss.backward()
        return [p.grad.clone() for p in m.parameters()]

    full = grads(x)
    a, b = grads(x[:4096]), grads(x[4096:])
    for f, ga, gb in zip(full, a, b):
        mean = (ga + gb) / 2
        assert (mean - f).abs().max().item() <= 1e-5 * max(f.abs().max().item(), 1e-3)


def test_config3_real_shape_training_step_matches_reference():
    """BASELINE configuration 3 at its real shape (rqvae_ml32m.gin: D = 64, rotation trick, batch 64, AdamW 1e-4 / 0.01):
    one full training step -- forward, backward, optimizer update -- against the reference's own step
    (tests/golden/config3_step.npz, oracle/gen_golden.py:gen_config3_step)."""
    import hashlib
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    g = load_golden("config3_step.npz")
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=64, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
              n_cat_features=0, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.ROTATION_TRICK,
              commitment_weight=0.25)
    h = hashlib.sha256()
    for k, v in m.state_dict().items():
        if "embedding" not in k:
            h.update(k.encode())
            h.update(np.ascontiguousarray(v.numpy()).view(np.uint8).reshape(-1))
    assert h.hexdigest() == str(g["weights_sha256"])       # same seeded construction as the reference
    with torch.no_grad():
        for l, layer in enumerate(m.layers):
            layer.embedding.weight.copy_(torch.from_numpy(g["codebooks"][l]))
    m = m.cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    x = torch.from_numpy(g["x"]).cuda()
    out = m(SeqBatch(None, None, None, x, None, None), 0.2)
    with torch.no_grad():
        ids = m.get_semantic_ids(x, 0.2).sem_ids.cpu().numpy()
    out.loss.backward()
    opt.step()
    assert np.array_equal(ids, g["sem_ids"])
    for name in ("loss", "reconstruction_loss", "rqvae_loss", "p_unique_ids"):
        assert abs(float(getattr(out, name)) - float(g[name])) < 1e-5, name
    np.testing.assert_allclose(out.embs_norm.cpu().numpy(), g["embs_norm"], rtol=1e-4, atol=1e-6)
    for k, v in m.named_parameters():
        grad, delta = v.grad.float().cpu(), (v.detach() - before[k]).float().cpu()
        gscale = max(float(grad.abs().max()), 1e-12)
        if "embedding" in k:
            ref_g, ref_d = g[f"grad::{k}"], g[f"delta::{k}"]
            got_g, got_d = grad.numpy(), delta.numpy()
        else:
            for prefix, t in (("grad", grad), ("delta", delta)):
                stat = g[f"{prefix}_stat::{k}"]
                assert abs(float(t.norm()) - stat[0]) <= 1e-3 * stat[0] + 1e-9, (prefix, k)
            ref_g, ref_d = g[f"grad_corner::{k}"], g[f"delta_corner::{k}"]
            got_g, got_d = grad[:8, :16].numpy(), delta[:8, :16].numpy()
        np.testing.assert_allclose(got_g, ref_g, rtol=2e-3, atol=2e-5 * gscale, err_msg=f"grad {k}")
        # the first AdamW update is -lr * g / (|g| + 1e-8) - lr * wd * p: ill-conditioned where the gradient is ~0, so
        # the update is compared where the gradient is not negligible (elsewhere only its bound lr * (1 + wd |p|))
        solid = np.abs(ref_g) > 1e-3 * gscale
        np.testing.assert_allclose(got_d[solid], ref_d[solid], rtol=2e-3, atol=1e-9, err_msg=f"delta {k}")
        assert np.abs(got_d).max() <= float(g["lr"]) * 1.05 + 1e-7


def test_flat_reducer_attach_one_model_applied_twice_under_one_loss():
    """a parameter that feeds TWO backward nodes of one backward() -- the same RqVae / MLP applied
    to two batches under one loss -- must get the SUM of both gradients.  AccumulateGrad runs only after both producers,
    so `.grad is None` cannot tell them apart; the slice of the flat buffer is claimed once per epoch (`zero_()`), the
    second producer returns an ordinary tensor and autograd accumulates."""
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    from rqhip.dist import FlatGradReducer

    def make():
        torch.manual_seed(0)
        m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
                  n_cat_features=0, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda()
        with torch.no_grad():
            for l, layer in enumerate(m.layers):
                layer.embedding.weight.copy_(torch.randn(256, 32, device="cuda") * (0.05 / (l + 1)))
        return m.train()

    x = torch.nn.functional.normalize(torch.randn(4096, 768, device="cuda"), dim=-1)
    b1, b2 = SeqBatch(None, None, None, x[:2048], None, None), SeqBatch(None, None, None, x[2048:], None, None)
    plain, attached = make(), make()
    red = FlatGradReducer(attached.parameters()).attach(attached)
    for _ in range(2):      # two steps: the claim must be released by zero_()
        for p in plain.parameters():
            p.grad = None
        red.zero_()
        (plain(b1, 0.2).loss + 0.5 * plain(b2, 0.2).loss).backward()
        (attached(b1, 0.2).loss + 0.5 * attached(b2, 0.2).loss).backward()
        for (name, p), q in zip(attached.named_parameters(), plain.parameters()):
            scale = max(q.grad.abs().max().item(), 1e-6)
            assert (p.grad - q.grad).abs().max().item() <= 1e-6 * scale, name   # (sum order of the two terms may differ)
        # (where autograd summed two producers the result is its own tensor, not the slice: allreduce_mean() packs those)


def test_recon_loss_second_backward_through_a_retained_graph():
    """the speculative reconstruction-loss gradient is handed out once; a second backward through a
    retained graph recomputes instead of returning the first call's (in-place fixed-up) tensor."""
    from rqhip.autograd import ReconLossFunction
    x = torch.randn(512, 768, device="cuda")
    x_hat = torch.randn(512, 768, device="cuda", requires_grad=True)
    out = ReconLossFunction.apply(x_hat, x)
    w1 = torch.rand(512, device="cuda")
    g1, = torch.autograd.grad(out, x_hat, grad_outputs=w1, retain_graph=True)
    g1 = g1.clone()
    g2, = torch.autograd.grad(out, x_hat, grad_outputs=torch.full((512,), 1.0 / 512, device="cuda"))
    assert torch.allclose(g1, 2 * (x_hat.detach() - x) * w1[:, None], rtol=1e-6, atol=1e-7)
    assert torch.allclose(g2, 2 * (x_hat.detach() - x) / 512, rtol=1e-6, atol=1e-9)


def test_flat_reducer_attach_puts_gradients_in_the_flat_buffer_without_copies():
    """rqhip.dist.FlatGradReducer.attach: the backward functions write each weight / codebook gradient straight into its
    slice of the flat all-reduce buffer (first gradient of a step) and autograd accumulates later ones there; values
    equal an un-attached run bit for bit."""
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    from rqhip.dist import FlatGradReducer

    def make():
        torch.manual_seed(0)
        m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
                  n_cat_features=0, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda()
        with torch.no_grad():
            for l, layer in enumerate(m.layers):
                layer.embedding.weight.copy_(torch.randn(256, 32, device="cuda") * (0.05 / (l + 1)))
        return m.train()

    x = torch.nn.functional.normalize(torch.randn(6000, 768, device="cuda"), dim=-1)
    halves = [SeqBatch(None, None, None, x[:3000], None, None), SeqBatch(None, None, None, x[3000:], None, None)]
    plain, attached = make(), make()
    red = FlatGradReducer(attached.parameters()).attach(attached)
    assert hasattr(attached, "_rq_cb_grad_sink")
    for n_micro in (1, 2):
        for p in plain.parameters():
            p.grad = None
        red.zero_()
        for b in halves[:n_micro]:
            plain(b, 0.2).loss.backward()
            attached(b, 0.2).loss.backward()
        base = red.flat.untyped_storage().data_ptr()
        for (name, p), q, v in zip(attached.named_parameters(), plain.parameters(), red._views):
            assert p.grad.untyped_storage().data_ptr() == base and p.grad.data_ptr() == v.data_ptr(), name
            assert torch.equal(p.grad, q.grad), name


@pytest.mark.parametrize("frozen_encoder", [False, True])
def test_batched_weight_gradients_across_the_two_stacks(frozen_encoder, monkeypatch):
    """rqhip/linear.py: at a split-kernel batch the 256 x 256-tiled weight gradients of the decoder wait for the encoder's launch (ONE
    rqhip_linear_wgrad_f16_batch of four layers, gradients in the flat buffer); with the encoder frozen the autograd engine's end-of-backward
    callback launches the decoder's two.  Every gradient agrees with the per-layer launches to rounding of the sums, the loss is the same
    bits, nothing is left waiting."""
    from data.schemas import SeqBatch
    from rqhip import linear, ops
    from rqhip.dist import FlatGradReducer
    x = torch.nn.functional.normalize(torch.randn(8192, 768, device="cuda"), dim=-1)
    batch = SeqBatch(None, None, None, x, None, None)
    calls = []
    real = ops.linear_wgrad_f16_batch

    def counting(jobs, outs=None):
        calls.append(len(jobs))
        return real(jobs, outs=outs)
    monkeypatch.setattr(ops, "linear_wgrad_f16_batch", counting)
    res = {}
    for arm in ("cross", "per_stack", "per_layer"):
        m = _rqvae_768()
        if frozen_encoder:
            for p in m.encoder.parameters():
                p.requires_grad_(False)
        red = FlatGradReducer([p for p in m.parameters() if p.requires_grad]).attach(m)
        b0, b1 = linear.use_wgrad_batch(arm != "per_layer"), linear.use_wgrad_cross_stack(arm == "cross")
        try:
            calls.clear()
            red.zero_()
            out = m(batch, 0.2)
            out.loss.backward()
            torch.cuda.synchronize()
            assert not linear._XSTACK
            res[arm] = (out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad}, list(calls))
        finally:
            linear.use_wgrad_batch(b0)
            linear.use_wgrad_cross_stack(b1)
    # (cross: the four 256 x 256-tiled layers, then the two half-tiled ones -- dW [256, 128] of the decoder, dW [128, 256] of the encoder)
    assert res["cross"][2] == ([2] if frozen_encoder else [4, 2]) and res["per_layer"][2] == []
    assert res["per_stack"][2] == ([2] if frozen_encoder else [2, 2])
    for arm in ("cross", "per_stack"):
        assert torch.equal(res[arm][0], res["per_layer"][0])
        for n, gref in res["per_layer"][1].items():
            got = res[arm][1][n]
            assert (got - gref).abs().max().item() <= 4e-6 * gref.abs().max().item() + 1e-12, (arm, n)


def test_split_gemms_follow_the_optimizer():
    """The bf16-split GEMMs read the weights through an image that is rebuilt at every use: three optimiser steps with the
    FUSED AdamW (whose in-place update does not bump `Parameter._version` -- a version-keyed cache of the images trained on
    stale weights) must give the same losses as the same steps on the library GEMMs."""
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    from rqhip import linear

    def run(split):
        before = linear.use_split_gemms(split)
        try:
            torch.manual_seed(0)
            m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
                      n_cat_features=0, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda().train()
            with torch.no_grad():
                for l, layer in enumerate(m.layers):
                    layer.embedding.weight.copy_(torch.randn(256, 32, device="cuda") * (0.05 / (l + 1)))
            opt = torch.optim.AdamW(m.parameters(), lr=1e-2, fused=True)
            x = torch.nn.functional.normalize(torch.randn(8192, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(1)), dim=-1)
            losses = []
            for _ in range(4):
                opt.zero_grad(set_to_none=True)
                out = m(SeqBatch(None, None, None, x, None, None), 0.2)
                out.loss.backward()
                opt.step()
                losses.append(float(out.reconstruction_loss))
            return losses
        finally:
            linear.use_split_gemms(before)

    a, b = run(True), run(False)
    assert a[0] != a[-1]                                   # the model moved
    for la, lb in zip(a, b):
        assert abs(la - lb) <= 2e-5 * abs(lb), (a, b)


@pytest.mark.parametrize("rows", [9000, 640])
@pytest.mark.parametrize("arith", ["f16x2", "bf16x3"])
def test_mlp_stack_node_equals_the_per_layer_functions_bit_for_bit(arith, rows):
    """modules/encoder.py: the whole-stack autograd node (weight images in one launch, scales handed from epilogue to epilogue,
    the ReLU backward in the data-gradient epilogues) runs the arithmetic of the per-layer Functions (one node per layer, scales
    from maxima passes, mask in its own pass): the same maxima give the same exponents, so outputs and every gradient are
    bit-identical.  rows = 640: the small-batch path -- library GEMMs, and the job-table weight gradients (csrc/wgrad_jobs.hip), one
    launch per stack in the node, one launch per layer in the Functions: same kernel, same bits."""
    from modules.encoder import MLP
    from rqhip import linear
    before = linear.use_arith(arith)
    # (the node's batched weight-gradient launch cuts its layers into another number of row ranges than the per-layer launches: same
    # arithmetic, another balanced tree over the partial blocks -- compared at rounding level below, and off for the bit comparison)
    batch_before = linear.use_wgrad_batch(False)
    try:
        torch.manual_seed(5)
        mlp = MLP(768, [512, 256, 128], 32).cuda()
        x = torch.nn.functional.normalize(torch.randn(rows, 768, device="cuda"), dim=-1).requires_grad_(True)
        gout = torch.randn(rows, 32, device="cuda") * 1e-4

        def run(fn):
            for p in mlp.parameters():
                p.grad = None
            x.grad = None
            y = fn(x, list(mlp.mlp))
            y.backward(gout)
            return [y.detach().clone()] + [p.grad.clone() for p in mlp.parameters()] + [x.grad.clone()]

        a, b = run(mlp._run), run(mlp._run_layerwise)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        dec = MLP(32, [128, 256, 512], 768).cuda()            # the decoder's shape: its input needs a gradient
        z = torch.randn(rows, 32, device="cuda", requires_grad=True)
        g2 = torch.randn(rows, 768, device="cuda") * 1e-5
        outs = []
        for fn in (dec._run, dec._run_layerwise):
            for p in dec.parameters():
                p.grad = None
            z.grad = None
            y = fn(z, list(dec.mlp))
            y.backward(g2)
            outs.append([y.detach().clone()] + [p.grad.clone() for p in dec.parameters()] + [z.grad.clone()])
        for u, v in zip(*outs):
            assert torch.equal(u, v)
        if rows >= 4096 and arith == "f16x2":
            linear.use_wgrad_batch(True)
            for p in dec.parameters():
                p.grad = None
            z.grad = None
            y = dec._run(z, list(dec.mlp))
            y.backward(g2)
            batched = [y.detach().clone()] + [p.grad.clone() for p in dec.parameters()] + [z.grad.clone()]
            for u, v in zip(batched, outs[0]):
                assert (u - v).abs().max().item() <= 4e-6 * v.abs().max().item() + 1e-12
            assert torch.equal(batched[0], outs[0][0]) and torch.equal(batched[-1], outs[0][-1])     # (outputs and input gradient: untouched)
    finally:
        linear.use_arith(before)
        linear.use_wgrad_batch(batch_before)


def _rqvae_768(seed=0):
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    torch.manual_seed(seed)
    m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
              n_cat_features=0, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda().train()
    with torch.no_grad():
        for l, layer in enumerate(m.layers):
            layer.embedding.weight.copy_(torch.randn(256, 32, device="cuda") * (0.05 / (l + 1)))
    return m


@pytest.mark.parametrize("case", ["mean", "half_without_hint", "second_backward"])
def test_fused_last_layer_and_reconstruction_loss_equals_the_composed_pair(case, monkeypatch):
    """RqVae.forward at a large batch runs the last decoder layer and ReconstructionLoss as one kernel (x_hat is never
    stored; modules/encoder.py:_MLPStack with a target).  Losses and every parameter gradient must equal the composed path (decoder,
    then the loss kernel) -- also when the upstream row gradient is not the announced one (rows are rescaled) and on a
    second backward through a retained graph (x_hat is recomputed)."""
    from data.schemas import SeqBatch
    from modules.encoder import MLP, _MLPStack
    x = torch.nn.functional.normalize(torch.randn(8192, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(3)), dim=-1)
    calls = []
    real_apply = _MLPStack.apply

    def run(fused):
        m = _rqvae_768()
        if not fused:
            monkeypatch.setattr(MLP, "reconstruction_rows", lambda self, z, t, first=0: None)
        else:
            monkeypatch.undo()
            monkeypatch.setattr(_MLPStack, "apply", lambda *a: (calls.append(1) if a[1] is not None else None, real_apply(*a))[1])
        out = m(SeqBatch(None, None, None, x, None, None), 0.2)
        loss = out.loss * 0.5 if case == "half_without_hint" else out.loss
        if case == "second_backward":
            loss.backward(retain_graph=True)
            for p in m.parameters():
                p.grad = None
        loss.backward()
        return out, {n: p.grad.clone() for n, p in m.named_parameters()}

    out_f, g_f = run(True)
    out_c, g_c = run(False)
    assert calls, "the fused kernel did not run"
    assert abs(float(out_f.reconstruction_loss) - float(out_c.reconstruction_loss)) <= 2e-6 * abs(float(out_c.reconstruction_loss))
    assert abs(float(out_f.loss) - float(out_c.loss)) <= 2e-6 * abs(float(out_c.loss))
    for n in g_c:
        scale = g_c[n].abs().max().item() + 1e-30
        assert (g_f[n] - g_c[n]).abs().max().item() <= 2e-6 * scale, n


@pytest.mark.parametrize("name", ["wide_cosine_gumbel.npz", "wide_d160_ste.npz", "wide_d160_rotation.npz", "wide_d160_eval.npz",
                                  "wide_gumbel_k1100.npz"])
def test_shapes_outside_the_kernels_match_reference(name, monkeypatch):
    """COSINE x GUMBEL_SOFTMAX, embed_dim > 128 and Gumbel-softmax with more than 1024 codes (reference quantize.py:104-163
    has no limits): rqhip/wide.py runs the reference's expression as PyTorch-ROCm operators on the device tensors; checked
    against reference outputs and autograd gradients (oracle/gen_golden.py:gen_wide), the reference's uniform noise fed in."""
    import warnings
    from modules.quantize import Quantize, QuantizeDistance, QuantizeForwardMode
    g = load_golden(name)
    K, D = g["codebook"].shape
    q = Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, forward_mode=QuantizeForwardMode(int(g["mode"])),
                 distance_mode=QuantizeDistance.COSINE if bool(g["cosine"]) else QuantizeDistance.L2).cuda()
    assert not q.train(bool(g["training"])).kernel_covers()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(g["codebook"]))
    U = torch.from_numpy(g["U"]).cuda()
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *a, **k: U if tuple(a[0]) == tuple(U.shape) else real_rand(*a, **k))
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        out = q(x, temperature=float(g["temperature"]))
    assert out.embeddings.is_cuda and np.array_equal(out.ids.cpu().numpy(), g["ids"])
    np.testing.assert_allclose(out.embeddings.detach().cpu().numpy(), g["embeddings"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out.loss.detach().cpu().numpy(), g["loss"], rtol=2e-5, atol=1e-5)
    ((out.embeddings * torch.from_numpy(g["g_emb"]).cuda()).sum() + (out.loss * torch.from_numpy(g["g_loss"]).cuda()).sum()).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["grad_x"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(q.embedding.weight.grad.cpu().numpy(), g["grad_codebook"], rtol=1e-4, atol=2e-5)


def test_wide_level_runs_in_row_tiles(monkeypatch):
    """rqhip/wide.py processes a level in tiles of 16 384 rows (no [B, K] matrix): a 40 000-row call equals the same call in one tile."""
    import warnings
    from modules.quantize import Quantize, QuantizeForwardMode
    from rqhip import wide
    torch.manual_seed(4)
    x = torch.randn(40_000, 160, device="cuda")
    for mode, training in ((QuantizeForwardMode.STE, True), (QuantizeForwardMode.ROTATION_TRICK, True), (QuantizeForwardMode.STE, False)):
        q = Quantize(embed_dim=160, n_embed=48, do_kmeans_init=False, forward_mode=mode).cuda().train(training)
        res = []
        for tile in (16384, 1 << 30):
            monkeypatch.setattr(wide, "_TILE_ROWS", tile)
            xi = x.clone().requires_grad_(True)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                out = q(xi, temperature=0.2)
            (out.embeddings.sum() + out.loss.sum()).backward()
            res.append((out.ids.clone(), out.embeddings.detach().clone(), out.loss.detach().clone(), xi.grad.clone(), q.embedding.weight.grad.clone()))
            q.embedding.weight.grad = None
        (i1, e1, l1, g1, w1), (i2, e2, l2, g2, w2) = res
        assert tuple(i1.shape) == (40_000,) and (i1 != i2).sum().item() <= 2           # (a GEMM of another height may round a near-tie differently)
        same = i1 == i2
        assert torch.allclose(e1[same], e2[same], rtol=1e-5, atol=1e-6) and torch.allclose(l1[same], l2[same], rtol=1e-5, atol=1e-6)
        assert torch.allclose(g1[same], g2[same], rtol=1e-4, atol=1e-5) and torch.allclose(w1, w2, rtol=1e-3, atol=1e-3)


def test_wide_latents_through_rqvae_and_kmeans():
    """embed_dim = 160: k-means init (reference seeds -> reference centroids), then RqVae forward + backward level by level."""
    import warnings
    from data.schemas import SeqBatch
    from init.kmeans import Kmeans
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    g = load_golden("wide_kmeans_d160.npz")
    x = torch.from_numpy(g["x"]).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        np.random.seed(int(g["seed"]))
        torch.manual_seed(int(g["seed"]))
        out = Kmeans(k=int(g["k"])).run(x)
        assert np.array_equal(out.assignment.cpu().numpy(), g["assignment"])
        np.testing.assert_allclose(out.centroids.cpu().numpy(), g["centroids"], rtol=1e-5, atol=1e-6)
        torch.manual_seed(0)
        np.random.seed(0)
        m = RqVae(input_dim=64, embed_dim=160, hidden_dims=[96], codebook_size=12, n_layers=2, n_cat_features=0,
                  codebook_kmeans_init=True, codebook_mode=QuantizeForwardMode.STE).cuda()
        xb = torch.nn.functional.normalize(torch.randn(300, 64, device="cuda"), dim=-1)
        m.train()
        res = m(SeqBatch(None, None, None, xb, None, None), 0.2)
        res.loss.backward()
        assert not m._can_fuse() and all(l.kmeans_initted for l in m.layers)
        assert torch.isfinite(res.loss).item() and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        m.eval()
        sem = m.get_semantic_ids(xb)
        assert sem.sem_ids.shape == (300, 2) and int(sem.sem_ids.max()) < 12
