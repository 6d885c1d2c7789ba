"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar: bit-exact for semantic ids AND for every fp32 output of the non-transcendental paths (the oracle
fixes the reduction orders the kernels use); committed golden fixtures produced by the reference itself
are checked too (ids exact, losses within 1e-5).
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import rq_oracle as o

pytestmark = pytest.mark.gpu

MODES = {"eval": 0, "ste": 1, "rotation": 2}


def _gpu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _run_forward(x, cbs, mode, beta=0.25, want_margin=False):
    from rqhip import ops
    out = ops.rq_forward(_gpu(x), _gpu(cbs), mode, beta, want_margin=want_margin)
    torch.cuda.synchronize()
    return {k: (None if v is None else v.cpu().numpy()) for k, v in out._asdict().items()}


def _assert_bitexact(got, ref, what):
    assert got.shape == ref.shape, what
    if got.dtype.kind == "f":
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    else:
        same = got == ref
    assert same.all(), f"{what}: {(~same).sum()} of {same.size} elements differ; first at {np.argwhere(~same)[:3].tolist()}"


def _check_forward(x, cbs, mode, beta=0.25):
    """Both kernel variants (plain, and the one that also tracks the runner-up distance for tie_margin) against
    the oracle, every output bit for bit."""
    ref = o.rq_forward(x, cbs, mode, beta, want_margin=True)
    for want_margin in (False, True):
        got = _run_forward(x, cbs, mode, beta, want_margin)
        for k in ("ids", "embs", "residuals", "emb_sum", "loss", "embs_norm") + (("tie_margin",) if want_margin else ()):
            _assert_bitexact(got[k], ref[k], f"{k} (mode {mode}, B={x.shape[0]}, cb={cbs.shape}, margin={want_margin})")


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("B,D,K,L", [(1, 32, 256, 3), (33, 32, 256, 3), (640, 32, 256, 3), (4099, 32, 256, 3),
                                     (64, 64, 256, 3), (100, 16, 32, 3), (50, 8, 5, 2), (77, 24, 100, 4),
                                     (65, 30, 70, 2), (40, 128, 64, 2), (300, 32, 1024, 4), (96, 64, 2048, 2),
                                     # cooperative tiles, several per workgroup x 16 levels: the LDS step counters wrap
                                     (30000, 16, 32, 16), (20000, 32, 256, 3)])
def test_forward_bitexact_vs_oracle(mode, B, D, K, L):
    rng = np.random.default_rng(B * 131 + D * 7 + K + L + mode)
    x = (rng.standard_normal((B, D)) * 0.7).astype(np.float32)
    cbs = (rng.standard_normal((L, K, D)) * np.array([0.6 / (l + 1) for l in range(L)])[:, None, None]).astype(np.float32)
    _check_forward(x, cbs, mode)


def test_forward_ties_duplicates_nan_inf():
    """quantize.py:128: first index on ties; NaN distance wins; Inf handled like torch.min."""
    rng = np.random.default_rng(5)
    B, D, K, L = 70, 32, 96, 2
    x = rng.standard_normal((B, D)).astype(np.float32)
    cbs = rng.standard_normal((L, K, D)).astype(np.float32)
    cbs[0, 40] = cbs[0, 7]          # duplicated codes -> lowest index
    cbs[0, 90] = cbs[0, 7]
    x[3] = cbs[0, 7]                 # exact hit on the duplicated code
    x[10] = 0.0
    x[11] = np.nan
    x[12, 5] = np.inf
    x[13, 0] = 3e19                  # |x|^2 overflows to +Inf
    x[14] = 1e-30                    # subnormal squares
    _check_forward(x, cbs, 0)
    _check_forward(x, cbs, 1)
    cbs2 = cbs.copy()
    cbs2[0, 50, 3] = np.nan          # a NaN code: every row's level-0 id is 50
    _check_forward(x[:40], cbs2, 0)
    cbs3 = cbs.copy()
    cbs3[1, 20, 1] = np.inf
    _check_forward(x[:40], cbs3, 1)


def test_forward_all_equal_rows_and_zero_codebook():
    x = np.ones((130, 32), np.float32)
    cbs = np.zeros((3, 256, 32), np.float32)
    _check_forward(x, cbs, 1)
    got = _run_forward(x, cbs, 0)
    assert (got["ids"] == 0).all()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("B,D,K", [(64, 64, 256), (640, 32, 256), (20000, 32, 256), (5000, 64, 256), (300, 32, 1024)])
def test_forward_zero_and_vanishing_rows(mode, B, D, K):
    """Rows that are exactly zero (a dead encoder output; they stay on the filtered scan: their scores are the exact -|c|^2/2),
    rows of denormal / vanishing magnitude (exact scan), signed zeros, and codebooks with equal norms (zero rows then tie:
    first index) -- every output bit for bit against the oracle, plain and margin variants."""
    rng = np.random.default_rng(B + D + K + mode)
    x = (rng.standard_normal((B, D)) * 0.5).astype(np.float32)
    kind = rng.integers(0, 6, B)
    x[kind == 0] = 0.0
    x[kind == 1] = -0.0
    x[kind == 2] *= np.float32(1e-30)                     # squares underflow to 0, features are not zero
    x[kind == 3] = np.float32(1e-42) * np.sign(x[kind == 3])   # denormal features
    cbs = (rng.standard_normal((3, K, D)) * np.array([0.4, 0.2, 0.1])[:, None, None]).astype(np.float32)
    cbs[1, 7] = cbs[1, 3]                                 # duplicate codes: first index wins
    cbs[0, 5] = -cbs[0, 2]                                # equal norms, different codes: a zero row ties between them
    cbs[0, 2] *= 0.01
    cbs[0, 5] *= 0.01                                     # ... and they are the smallest norms of level 0
    _check_forward(x, cbs, mode)
    # a whole batch of zeros (what a collapsed encoder produces)
    _check_forward(np.zeros((min(B, 700), D), np.float32), cbs, mode)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_forward_rows_not_16_byte_aligned(mode):
    """The full-width kernels move rows as float4s; a contiguous view that starts 4 bytes into its storage must
    take the element-wise path and give the same bits."""
    from rqhip import ops
    rng = np.random.default_rng(21)
    B, D, K, L = 333, 32, 256, 3
    x = (rng.standard_normal((B, D)) * 0.6).astype(np.float32)
    cbs = (rng.standard_normal((L, K, D)) * 0.4).astype(np.float32)
    store = torch.zeros(B * D + 1, dtype=torch.float32, device="cuda")
    xv = store[1:].view(B, D)
    xv.copy_(_gpu(x))
    assert xv.data_ptr() % 16 == 4 and xv.is_contiguous()
    out = ops.rq_forward(xv, _gpu(cbs), mode, 0.25)
    ref = o.rq_forward(x, cbs, mode, 0.25)
    for k in ("ids", "embs", "residuals", "emb_sum", "loss", "embs_norm"):
        _assert_bitexact(getattr(out, k).cpu().numpy(), ref[k], f"{k} (unaligned rows, mode {mode})")


def test_forward_empty_batch():
    got = _run_forward(np.zeros((0, 32), np.float32), np.ones((3, 256, 32), np.float32), 1)
    assert got["ids"].shape == (3, 0) and got["loss"].shape == (0,)


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "quantize_*.npz"))))
def test_forward_vs_reference_golden(name):
    g = load_golden(name)
    mode = MODES[name.split("_")[1]]
    got = _run_forward(g["x"], g["codebook"][None], mode, float(g["beta"]))
    assert np.array_equal(got["ids"][0], g["ids"])
    np.testing.assert_allclose(got["loss"], g["loss"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(got["embs"][0], g["embeddings"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("B,D,K,L,mode", [(100_000, 32, 256, 3, 0), (100_000, 32, 256, 3, 1),
                                          (262_144, 32, 1024, 4, 0), (131_072, 64, 256, 3, 2)])
def test_forward_full_size_properties(B, D, K, L, mode):
    """BASELINE config 2 (100k rows, 3 x 256), a config-4 shard (4 x 1024, LDS streaming mode) and the ml32m width
    at full size: size-independent properties -- ids in range, decode(ids) reproduces the codewords, the residual
    chain closes, emb_sum is the level sum -- and the oracle on EVERY row (ids, loss, norms, tie margins)."""
    from rqhip import ops
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, D, generator=g) * 0.5
    cbs = torch.randn(L, K, D, generator=g) * torch.tensor([0.5 / (l + 1) for l in range(L)])[:, None, None]
    out = ops.rq_forward(x.cuda(), cbs.cuda(), mode, 0.25, want_margin=True)
    ids = out.ids.cpu()
    assert ids.min() >= 0 and ids.max() < K
    embs, res = out.embs.cpu(), out.residuals.cpu()
    assert torch.equal(res[0], x)
    for l in range(L - 1):
        assert torch.equal(res[l + 1], res[l] - embs[l])            # rqvae.py:130, bit for bit
    if mode == 0:
        dec = torch.stack([cbs[l][ids[l]] for l in range(L)])
        assert torch.equal(dec, embs)                               # eval: emb_out is the codeword itself
    acc = embs[0].clone()
    for l in range(1, L):
        acc = acc + embs[l]
    assert torch.equal(acc, out.emb_sum.cpu())
    ref = o.rq_forward(x.numpy(), cbs.numpy(), mode, 0.25, want_margin=True)
    assert np.array_equal(ref["ids"], ids.numpy())
    _assert_bitexact(out.loss.cpu().numpy(), ref["loss"], "loss")
    _assert_bitexact(out.embs_norm.cpu().numpy(), ref["embs_norm"], "norm")
    _assert_bitexact(out.tie_margin.cpu().numpy(), ref["tie_margin"], "tie_margin")
    plain = ops.rq_forward(x.cuda(), cbs.cuda(), mode, 0.25, want_embs=False, want_residuals=False)
    assert torch.equal(plain.ids.cpu(), ids) and torch.equal(plain.loss.cpu(), out.loss.cpu())


def _filter_flavour(flavour, mode, D=32):
    """(x [B,D], cb [L,K,D]) torch CPU tensors built to sit on the filtered scan's decision boundaries."""
    g = torch.Generator().manual_seed(100 + mode + D)
    B, K, L = 100_000, 256, 3
    if flavour == "random":
        B = 300_000
        x = torch.randn(B, D, generator=g) * 0.5
        cb = torch.randn(L, K, D, generator=g) * 0.3
    elif flavour == "clustered":          # codebook = perturbed data points: many rows close to two codes
        B = 300_000
        x = torch.randn(B, D, generator=g) * 0.5
        cb = torch.stack([x[torch.randperm(B, generator=g)[:K]] / (l + 1) + 0.02 * torch.randn(K, D, generator=g)
                          for l in range(L)])
    elif flavour == "near_duplicate_codes":   # every code has a twin a few ulps away: every row is a near tie
        x = torch.randn(B, D, generator=g) * 0.5
        half = torch.randn(L, K // 2, D, generator=g) * 0.3
        twin = half * (1.0 + 3e-7 * torch.randn(L, K // 2, D, generator=g))
        cb = torch.cat([half, twin], dim=1)
    elif flavour == "tiny_scale":           # |x|^2 max|c|^2 underflows: the bound is meaningless, every row goes exact
        x = torch.randn(B, D, generator=g) * 1e-18
        cb = torch.randn(L, K, D, generator=g) * 1e-18
    elif flavour in ("rows_dwarf_codes", "codes_dwarf_rows"):
        # |x| >> every |c| (or the reverse): all distances of a row agree in their leading digits, neighbouring codes
        # differ by a few ulps of the distance itself -- the fp32 rounding of d decides, not the dot-product error
        big, small = (1.0e5, 1.0) if flavour == "rows_dwarf_codes" else (1.0, 1.0e5)
        x = torch.randn(B, D, generator=g) * big
        cb = torch.randn(L, K, D, generator=g) * small
    elif flavour == "split_worst_case":
        # the operands of tests/test_filter_bound.py: every mantissa on the bf16 split's worst case, |x_d| proportional to
        # |c_d|, error signs aligned for one code and opposed for its mirror -- tiled to full size, every level
        import test_filter_bound as fb
        xw, cw = fb._worst_case_operands(D, n_rows=64, seed=mode)
        rep = torch.from_numpy(xw).repeat(B // 64 + 1, 1)[:B]
        x = rep * (1.0 + 2.0 ** -20 * torch.randint(-8, 9, (B, 1), generator=g).float())      # rows a few ulps apart
        cb = torch.from_numpy(cw[:K]).unsqueeze(0).repeat(L, 1, 1) * torch.tensor([1.0, 0.5, 0.25]).view(L, 1, 1)
    else:                                   # mixed_scale: rows and codes spread over twelve orders of magnitude
        x = torch.randn(B, D, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (B, 1), generator=g).float())
        cb = torch.randn(L, K, D, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (L, K, 1), generator=g).float())
    return x.contiguous(), cb.contiguous()


def _recheck_rate(x, cb, mode):
    """Fraction of (row, level) decisions whose EXACT top-2 distance gap is below the filter's threshold T: an estimate
    of how many rows the filtered kernel re-decides exactly (it tests the approximate gap, which differs by < T)."""
    from rqhip import ops
    c1, c2 = ops.filter_bound(cb.shape[-1])
    out = ops.rq_forward(x, cb, mode, 0.25, want_margin=True, want_embs=False)
    res = out.residuals                                     # [L,B,D]
    xsq = (res.double() ** 2).sum(-1)                       # [L,B]
    csq = (cb.double() ** 2).sum(-1)                        # [L,K]
    cwin = torch.gather(csq, 1, out.ids)                    # [L,B]
    gap = out.tie_margin.double() * (xsq + cwin)
    T = c1 * torch.sqrt(xsq * csq.max(dim=1, keepdim=True).values) + c2 * (xsq + csq.max(dim=1, keepdim=True).values)
    return float((gap <= T).double().mean())


@pytest.mark.parametrize("flavour", ["random", "clustered", "near_duplicate_codes", "tiny_scale", "mixed_scale",
                                     "rows_dwarf_codes", "codes_dwarf_rows", "split_worst_case"])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_filtered_scan_equals_the_oracle(flavour, mode):
    """D = 32 launches without margins rank the codes by bf16-split scores on the matrix cores and re-decide, with the
    oracle's arithmetic, every row whose two best scores are closer than the proven error bound (csrc/rq_forward.hip
    FILT; tests/test_filter_bound.py).  The product launch must reproduce THE ORACLE bit for bit -- ids, losses, sums --
    on >= 100 000 rows of every flavour of data built to sit on the decision boundaries."""
    from rqhip import ops
    x, cb = _filter_flavour(flavour, mode)
    xg, cg = x.cuda(), cb.cuda()
    got = ops.rq_forward(xg, cg, mode, 0.25, want_embs=False, want_residuals=False)
    ref = o.rq_forward(x.numpy(), cb.numpy(), mode, 0.25)
    _assert_bitexact(got.ids.cpu().numpy(), ref["ids"], f"ids ({flavour}, mode {mode})")
    _assert_bitexact(got.loss.cpu().numpy(), ref["loss"], f"loss ({flavour}, mode {mode})")
    _assert_bitexact(got.emb_sum.cpu().numpy(), ref["emb_sum"], f"emb_sum ({flavour}, mode {mode})")
    _assert_bitexact(got.embs_norm.cpu().numpy(), ref["embs_norm"], f"embs_norm ({flavour}, mode {mode})")
    # the all-fp32 scan and the product scan are interchangeable
    f32 = ops.rq_forward(xg, cg, mode, 0.25, want_embs=False, want_residuals=False, scan="fp32")
    assert torch.equal(f32.ids, got.ids) and torch.equal(f32.loss.view(torch.int32), got.loss.view(torch.int32))
    print(f"{flavour} mode {mode}: {x.shape[0]} rows, estimated exact re-decisions {100 * _recheck_rate(xg, cg, mode):.3f} % "
          "of the (row, level) decisions")


@pytest.mark.parametrize("flavour", ["random", "clustered", "split_worst_case"])
def test_filtered_scan_equals_the_oracle_other_shapes(flavour):
    """The K = 1024, four-level shape of configuration 4 (one level at a time in LDS, half-width group maxima), a K that
    is not a multiple of the tile and D = 64 (configuration 3's width) -- against the oracle, bit for bit."""
    from rqhip import ops
    g = torch.Generator().manual_seed(7)
    x, cb = _filter_flavour(flavour, 1)
    x4 = x[:125_000]
    cb4 = torch.cat([cb[:, :, :], cb[:, :, :] * 0.7 + 0.01 * torch.randn(cb.shape, generator=g),
                     torch.randn(3, 512, 32, generator=g) * cb.abs().mean()], dim=1)            # [3,1024,32]
    cb4 = torch.cat([cb4, cb4[:1] * 0.3], dim=0).contiguous()                                   # [4,1024,32]
    cases = [(x4, cb4, 1), (x[:60_000], cb[:, :200].contiguous(), 0)]
    x64, cb64 = _filter_flavour(flavour, 2, D=64)
    cases.append((x64[:100_000], cb64, 2))
    cases.append((x64[:40_000], cb64, 1))
    for xx, cc, mode in cases:
        got = ops.rq_forward(xx.cuda(), cc.cuda(), mode, 0.25, want_embs=False, want_residuals=False)
        ref = o.rq_forward(xx.numpy(), cc.numpy(), mode, 0.25)
        what = f"({flavour}, {tuple(xx.shape)} x {tuple(cc.shape)}, mode {mode})"
        _assert_bitexact(got.ids.cpu().numpy(), ref["ids"], "ids " + what)
        _assert_bitexact(got.loss.cpu().numpy(), ref["loss"], "loss " + what)
        _assert_bitexact(got.emb_sum.cpu().numpy(), ref["emb_sum"], "emb_sum " + what)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("B,K,L", [(1, 256, 3), (1000, 256, 3), (100_000, 256, 3), (5000, 1024, 4), (777, 37, 2)])
def test_valu_scan_equals_the_oracle(mode, B, K, L):
    """The LDS / vector-ALU form of the forward (csrc/rq_forward_valu.hip, RQHIP_FWD_SCAN_VALU) -- the kernel the MFMA
    form is measured against -- returns the oracle's bits too."""
    from rqhip import ops
    rng = np.random.default_rng(B + K + L + mode)
    x = (rng.standard_normal((B, 32)) * 0.7).astype(np.float32)
    cbs = (rng.standard_normal((L, K, 32)) * np.array([0.6 / (l + 1) for l in range(L)])[:, None, None]).astype(np.float32)
    if B >= 1000:
        x[5] = np.nan
        x[6, 3] = np.inf
        x[7] = 3e19
    ref = o.rq_forward(x, cbs, mode, 0.25)
    out = ops.rq_forward(_gpu(x), _gpu(cbs), mode, 0.25, scan="valu")
    for k in ("ids", "embs", "residuals", "emb_sum", "loss", "embs_norm"):
        _assert_bitexact(getattr(out, k).cpu().numpy(), ref[k], f"{k} (VALU scan, mode {mode}, B={B}, K={K})")
    with pytest.raises(ops.RqHipError):
        ops.rq_forward(_gpu(x), _gpu(cbs), 2, 0.25, scan="valu")          # rotation trick: MFMA kernels only


# ---------------------------------------------------------------- backward ---------------------------

def _bwd_order(B, D, L, K, mode):
    """(n_wg, units per workgroup, rows per unit) of the fixed-order backward kernels, or None when the shape takes the
    atomic scatter path."""
    import ctypes as C
    from rqhip import _lib
    n_wg, nw, unit = C.c_int(0), C.c_int(0), C.c_int(0)
    fixed = _lib.lib().rqhip_rq_backward_plan(B, D, L, K, mode, C.byref(n_wg), C.byref(nw), C.byref(unit))
    return (n_wg.value, nw.value, unit.value) if fixed else None


def _assert_cb_grad(g_cb, x, cbs, mode, ids, g, r_cb, what=""):
    """Codebook gradient: bit-exact against the oracle's restatement of the kernel's summation order where the
    fixed-order kernels run (L <= 4 and D <= 32, or EVAL / STE with D % 4 == 0, D <= 64); fp32-rounding tolerance on the
    atomic scatter path."""
    B, D = x.shape
    L, K, _ = cbs.shape
    order = _bwd_order(B, D, L, K, mode)
    if order is not None:
        _, o_cb = o.rq_backward(x, cbs, mode, 0.25, ids, order=order, **g)
        _assert_bitexact(g_cb, o_cb, "g_codebooks (fixed order) " + what)
    scale = max(1e-6, float(np.abs(r_cb).max()))
    np.testing.assert_allclose(g_cb, r_cb, rtol=1e-4, atol=3e-6 * scale)


def _run_backward(x, cbs, mode, beta, ids, **g):
    from rqhip import ops
    gg = {k: (None if v is None else _gpu(v)) for k, v in g.items()}
    g_res0, g_cb = ops.rq_backward(_gpu(x), _gpu(cbs), mode, beta, _gpu(ids), **gg)
    torch.cuda.synchronize()
    return g_res0.cpu().numpy(), g_cb.cpu().numpy()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("B,D,K,L", [(1, 32, 256, 3), (333, 32, 256, 3), (64, 64, 256, 3), (100, 16, 32, 1),
                                     (77, 24, 100, 4), (40, 128, 64, 2), (2000, 32, 1024, 4), (500, 32, 512, 3),
                                     # the bench shapes at full size: config 2, and a config-4 slice
                                     (100_000, 32, 256, 3), (60_000, 32, 1024, 4)])
@pytest.mark.parametrize("which", ["all", "train_like"])
def test_backward_vs_oracle(mode, B, D, K, L, which):
    rng = np.random.default_rng(B + D + K + L + mode)
    x = (rng.standard_normal((B, D)) * 0.7).astype(np.float32)
    cbs = (rng.standard_normal((L, K, D)) * np.array([0.6 / (l + 1) for l in range(L)])[:, None, None]).astype(np.float32)
    ref = o.rq_forward(x, cbs, mode, 0.25)
    if which == "all":
        g = dict(g_embs=rng.standard_normal((L, B, D)).astype(np.float32),
                 g_embsum=rng.standard_normal((B, D)).astype(np.float32),
                 g_resid=rng.standard_normal((L, B, D)).astype(np.float32),
                 g_loss=rng.random(B).astype(np.float32))
    else:  # what RqVae.forward's loss produces: decoder gradient through emb_sum, 1/B through the loss
        g = dict(g_embs=None, g_embsum=(rng.standard_normal((B, D)) / B).astype(np.float32), g_resid=None,
                 g_loss=np.full((B,), 1.0 / B, np.float32))
    r_res0, r_cb = o.rq_backward(x, cbs, mode, 0.25, ref["ids"], **g)
    g_res0, g_cb = _run_backward(x, cbs, mode, 0.25, ref["ids"], **g)
    _assert_bitexact(g_res0, r_res0, "g_res0")          # per-row arithmetic: exact
    _assert_cb_grad(g_cb, x, cbs, mode, ref["ids"], g, r_cb)
    g_res0_b, g_cb_b = _run_backward(x, cbs, mode, 0.25, ref["ids"], **g)
    # fixed-order kernels, and the small-batch kernel of the general path (one workgroup per level, rows in order: csrc/rq_backward.hip
    # rq_cbgrad_small_kernel -- e.g. batch 64, D = 64, rotation trick: the reference's rqvae_ml32m.gin)
    if _bwd_order(B, D, L, K, mode) is not None or (B <= 2048 and D <= 64 and K * D <= 16384):
        _assert_bitexact(g_cb_b, g_cb, "g_codebooks run-to-run")


@pytest.mark.parametrize("pattern", ["argmin", "one_code", "two_codes", "runs", "ragged"])
@pytest.mark.parametrize("B,K,L", [(100_000, 256, 3), (3000, 256, 3), (60_000, 1024, 4), (7, 256, 3), (5000, 1024, 3), (4000, 96, 3)])
def test_backward_matrix_form_of_the_codebook_gradient(B, K, L, pattern):
    """RQHIP_BWD_CBGRAD_MATRIX (round 5): the codebook gradient as a one-hot matrix product on the bf16 matrix cores -- what the training
    path runs at D = 32 / STE.  g_res0 keeps the bits of the ordered kernel (== the oracle's); the codebook gradient is the exact sum of the
    same fp32 terms in another order: no further from the fp64 sum than the ordered kernel is (both are fp32 accumulations), run-to-run
    identical, collapsed codebooks and ragged tails included.  Shapes the matrix form does not cover fall back to the ordered kernel."""
    from rqhip import _lib, ops
    D, mode = 32, 1
    rng = np.random.default_rng(B + K + L)
    if pattern == "ragged" and B > 5000:
        B = B - 37
    x = (rng.standard_normal((B, D)) * 0.7).astype(np.float32)
    cbs = (rng.standard_normal((L, K, D)) * np.array([0.6 / (l + 1) for l in range(L)])[:, None, None]).astype(np.float32)
    if pattern in ("argmin", "ragged"):
        ids = o.rq_forward(x, cbs, mode, 0.25)["ids"]
    elif pattern == "one_code":
        ids = np.full((L, B), 5, np.int64)
    elif pattern == "two_codes":
        ids = rng.choice([3, K - 1], size=(L, B)).astype(np.int64)
    else:
        ids = np.repeat(rng.integers(0, K, size=(L, B // 3 + 1)), 3, axis=1)[:, :B].astype(np.int64)
    g_sum = (rng.standard_normal((B, D)) / B).astype(np.float32)
    g_loss = np.full((B,), 1.0 / B, np.float32) if pattern != "runs" else (rng.random(B) / B).astype(np.float32)
    kw = dict(g_embsum=_gpu(g_sum), g_loss=_gpu(g_loss))
    r0_o, cb_o = ops.rq_backward(_gpu(x), _gpu(cbs), mode, 0.25, _gpu(ids), **kw)
    r0_m, cb_m = ops.rq_backward(_gpu(x), _gpu(cbs), mode, 0.25, _gpu(ids), cbgrad="matrix", **kw)
    r0_m2, cb_m2 = ops.rq_backward(_gpu(x), _gpu(cbs), mode, 0.25, _gpu(ids), cbgrad="matrix", **kw)
    assert torch.equal(r0_m, r0_o) and torch.equal(cb_m, cb_m2) and torch.equal(r0_m, r0_m2)
    covered = bool(_lib.lib().rqhip_rq_backward_matrix_form(D, K, L, mode))
    assert covered == ((K <= 256 and K % 32 == 0 and L == 3) or K == 1024)
    if not covered:
        assert torch.equal(cb_m, cb_o)              # the flag is ignored: same kernel
        return
    # the fp64 sum of the terms the kernels add: 2 g_loss (e_l - r_l) per row into code id_l (STE: r_{l+1} = r_l - (r_l + (e_l - r_l)))
    xt, ct, it = _gpu(x), _gpu(cbs), _gpu(ids)
    gl = _gpu(g_loss)
    ref = torch.zeros((L, K, D), dtype=torch.float64, device="cuda")
    r = xt.clone()
    for l in range(L):
        e = ct[l][it[l]]
        term = (2.0 * (e - r)) * gl[:, None]                       # fp32, as the kernels form it
        ref[l].index_add_(0, it[l], term.double())
        r = r - (r + (e - r))
    scale = ref.abs().max().item()
    err_m = (cb_m.double() - ref).abs().max().item() / scale
    err_o = (cb_o.double() - ref).abs().max().item() / scale
    print(f"B={B} K={K} L={L} {pattern}: codebook-gradient error vs fp64: matrix {err_m:.3e}, ordered {err_o:.3e}")
    assert err_m <= max(2.0 * err_o, 2e-7), (err_m, err_o)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("pattern", ["one_code", "two_codes", "same_owner", "runs"])
def test_backward_skewed_ids(mode, pattern):
    """Codebook gradient when many rows of a round share a code or an owner (collapsed codebooks, duplicated items): the
    owner's list of the flat kernel overflows into further passes and equal codes meet inside one step (register
    forwarding).  The backward takes the ids as given, so they need not be the argmin."""
    B, D, K, L = 3000, 32, 256, 3
    rng = np.random.default_rng(17 + mode)
    x = (rng.standard_normal((B, D)) * 0.7).astype(np.float32)
    cbs = (rng.standard_normal((L, K, D)) * 0.4).astype(np.float32)
    if pattern == "one_code":
        ids = np.full((L, B), 5, np.int64)
    elif pattern == "two_codes":
        ids = rng.choice([3, 200], size=(L, B)).astype(np.int64)
    elif pattern == "same_owner":          # every code is congruent mod 32: one owner takes all rows of a level
        ids = (rng.integers(0, K // 32, size=(L, B)) * 32 + 7).astype(np.int64)
    else:                                   # runs of equal codes of random length
        ids = np.repeat(rng.integers(0, K, size=(L, B // 3 + 1)), 3, axis=1)[:, :B].astype(np.int64)
    g = dict(g_embs=None, g_embsum=(rng.standard_normal((B, D)) / B).astype(np.float32), g_resid=None,
             g_loss=rng.random(B).astype(np.float32))
    r_res0, r_cb = o.rq_backward(x, cbs, mode, 0.25, ids, **g)
    g_res0, g_cb = _run_backward(x, cbs, mode, 0.25, ids, **g)
    _assert_bitexact(g_res0, r_res0, "g_res0")
    _assert_cb_grad(g_cb, x, cbs, mode, ids, g, r_cb, pattern)


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "quantize_*.npz"))))
def test_backward_vs_reference_autograd_golden(name):
    g = load_golden(name)
    mode = MODES[name.split("_")[1]]
    g_x, g_cb = _run_backward(g["x"], g["codebook"][None], mode, float(g["beta"]), g["ids"][None],
                              g_embs=g["g_emb"][None], g_loss=g["g_loss"])
    np.testing.assert_allclose(g_x, g["grad_x"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(g_cb[0], g["grad_codebook"], rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------- k-means -----------------------------

@pytest.mark.parametrize("B,D,K", [(300, 16, 8), (20000, 32, 256), (5000, 64, 256), (3000, 32, 1024), (257, 7, 5),
                                   (64, 128, 33)])
def test_kmeans_assign_and_update_bitexact(B, D, K):
    from rqhip import ops
    rng = np.random.default_rng(B + D + K)
    centers = rng.standard_normal((K, D)).astype(np.float32) * 2
    x = (centers[rng.integers(0, K, B)] + 0.3 * rng.standard_normal((B, D))).astype(np.float32)
    cent = x[rng.choice(B, K, replace=False)].copy()
    if K > 4:
        cent[3] = cent[1]           # duplicate centroid -> index 3 can never win -> empty cluster
    a_ref = o.kmeans_assign(x, cent)
    a_gpu = ops.kmeans_assign(_gpu(x), _gpu(cent))
    _assert_bitexact(a_gpu.cpu().numpy(), a_ref, "assignment")
    c_ref = cent.copy()
    counts_ref = o.kmeans_update(x, a_ref, c_ref)
    c_gpu = _gpu(cent)
    counts, shift = ops.kmeans_update(_gpu(x), a_gpu, c_gpu)
    _assert_bitexact(counts.cpu().numpy(), counts_ref, "counts")
    _assert_bitexact(c_gpu.cpu().numpy(), c_ref, "centroids")
    if K > 4:
        assert counts_ref[3] == 0
    ref_shift = o.kmeans_shift(c_ref, cent)
    assert np.float32(np.sqrt(np.float32(shift.item()))) == np.float32(ref_shift)


@pytest.mark.parametrize("name", ["kmeans_a.npz", "kmeans_b.npz", "kmeans_dup.npz"])
def test_kmeans_golden_one_iteration_matches_reference_start(name):
    """First Lloyd iteration from the reference's own seed rows: assignments exact (init/kmeans.py:40-43)."""
    from rqhip import ops
    g = load_golden(name)
    x = g["x"]
    cent = x[g["init_idx"]].copy()
    a_ref = o.kmeans_assign(x, cent)
    a_gpu = ops.kmeans_assign(_gpu(x), _gpu(cent)).cpu().numpy()
    assert np.array_equal(a_gpu, a_ref)


# ---------------------------------------------------------------- gumbel ------------------------------

def _gumbel_inputs(B, D, K, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, D)) * 0.5).astype(np.float32)
    cb = (rng.standard_normal((K, D)) * 0.5).astype(np.float32)
    U = rng.random((B, K)).astype(np.float32)
    return x, cb, U


@pytest.mark.parametrize("B,D,K,T", [(40, 32, 256, 0.2), (17, 16, 32, 0.5), (300, 64, 256, 0.2), (5, 8, 5, 1.0),
                                     (1000, 32, 100, 0.2), (64, 128, 96, 0.3)])
def test_gumbel_forward_backward_vs_oracle(B, D, K, T):
    from rqhip import ops
    x, cb, U = _gumbel_inputs(B, D, K, B + D + K)
    ref = o.gumbel_forward(x, cb, U, T, 0.25)
    ids, emb, loss = ops.gumbel_forward(_gpu(x), _gpu(cb), _gpu(U), T, 0.25)
    assert np.array_equal(ids.cpu().numpy(), ref["ids"])                      # noise-free argmin: exact
    np.testing.assert_allclose(emb.cpu().numpy(), ref["emb"], rtol=2e-4, atol=1e-5)  # exp((.)/T) amplifies 1-ulp distance differences
    np.testing.assert_allclose(loss.cpu().numpy(), ref["loss"], rtol=1e-5, atol=1e-5)   # north_star: losses within 1e-5
    rng = np.random.default_rng(1)
    g_emb = rng.standard_normal((B, D)).astype(np.float32)
    g_loss = rng.random(B).astype(np.float32)
    r_x, r_cb = o.gumbel_backward(x, cb, U, T, 0.25, g_emb=g_emb, g_loss=g_loss)
    g_x, g_cb = ops.gumbel_backward(_gpu(x), _gpu(cb), _gpu(U), T, 0.25, g_emb=_gpu(g_emb), g_loss=_gpu(g_loss))
    sx, sc = max(1.0, float(np.abs(r_x).max())), max(1.0, float(np.abs(r_cb).max()))
    np.testing.assert_allclose(g_x.cpu().numpy(), r_x, rtol=2e-3, atol=5e-5 * sx)
    np.testing.assert_allclose(g_cb.cpu().numpy(), r_cb, rtol=2e-3, atol=5e-5 * sc)


@pytest.mark.parametrize("name", ["gumbel_a.npz", "gumbel_b.npz"])
def test_gumbel_vs_reference_golden(name):
    """Same U as the reference drew (recovered in oracle/gen_golden.py): ids exact, emb/loss/grads close."""
    from rqhip import ops
    g = load_golden(name)
    T, beta = float(g["temperature"]), float(g["beta"])
    ids, emb, loss = ops.gumbel_forward(_gpu(g["x"]), _gpu(g["codebook"]), _gpu(g["U"]), T, beta)
    assert np.array_equal(ids.cpu().numpy(), g["ids"])
    np.testing.assert_allclose(emb.cpu().numpy(), g["embeddings"], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=1e-5, atol=1e-5)   # north_star: losses within 1e-5
    g_x, g_cb = ops.gumbel_backward(_gpu(g["x"]), _gpu(g["codebook"]), _gpu(g["U"]), T, beta, g_emb=_gpu(g["g_emb"]),
                                    g_loss=_gpu(g["g_loss"]))
    sx, sc = max(1.0, float(np.abs(g["grad_x"]).max())), max(1.0, float(np.abs(g["grad_codebook"]).max()))
    np.testing.assert_allclose(g_x.cpu().numpy(), g["grad_x"], rtol=2e-3, atol=5e-5 * sx)
    np.testing.assert_allclose(g_cb.cpu().numpy(), g["grad_codebook"], rtol=2e-3, atol=5e-5 * sc)


@pytest.mark.parametrize("B,K,T", [(4096, 256, 0.2), (4133, 64, 0.5), (5000, 128, 0.2), (4100, 32, 1.0)])
def test_gumbel_matrix_path_vs_oracle(B, K, T):
    """From 4096 rows of width 32 the Gumbel level runs on the matrix instructions (csrc/gumbel_mfma.hip), 32 rows
    per wave; same bars as the one-row-per-wave kernels: ids exact, everything else to rounding."""
    test_gumbel_forward_backward_vs_oracle(B, 32, K, T)


def test_gumbel_matrix_path_small_batches():
    """The same kernels forced on for a ragged small batch (rows past the end of the last tile must not contribute)."""
    from rqhip import ops
    before = ops.gumbel_matrix_path_min_rows(1)
    try:
        assert ops.gumbel_matrix_path_min_rows() == 1
        test_gumbel_forward_backward_vs_oracle(77, 32, 256, 0.2)
        test_gumbel_forward_backward_vs_oracle(1, 32, 64, 0.3)
    finally:
        ops.gumbel_matrix_path_min_rows(before)
    assert ops.gumbel_matrix_path_min_rows() == before == 4096


def test_gumbel_unsupported_shape_fails_loudly():
    from rqhip import RqHipError, ops
    x, cb, U = _gumbel_inputs(8, 32, 4096, 0)
    with pytest.raises(RqHipError, match="Gumbel"):
        ops.gumbel_forward(_gpu(x), _gpu(cb), _gpu(U), 0.2, 0.25)


# ---------------------------------------------------------------- reconstruction loss -----------------

@pytest.mark.parametrize("B,N", [(1, 768), (100, 768), (333, 48), (50, 7), (64, 130), (4097, 64)])
def test_recon_loss_forward_bitexact_backward_exact(B, N):
    from rqhip import ops
    rng = np.random.default_rng(B + N)
    a = rng.standard_normal((B, N)).astype(np.float32)
    b = rng.standard_normal((B, N)).astype(np.float32)
    g = rng.random(B).astype(np.float32)
    out = ops.recon_loss_forward(_gpu(a), _gpu(b)).cpu().numpy()
    _assert_bitexact(out, o.recon_loss(a, b), "recon loss")
    gh, gx = ops.recon_loss_backward(_gpu(a), _gpu(b), _gpu(g), True, True)
    want = ((2.0 * (a - b)) * g[:, None]).astype(np.float32)
    _assert_bitexact(gh.cpu().numpy(), want, "g_x_hat")
    _assert_bitexact(gx.cpu().numpy(), -want, "g_x")


def test_recon_loss_strided_views_and_autograd():
    """CategoricalReconstuctionLoss hands column slices of wider matrices (loss.py:22-24): row stride > N."""
    from modules.loss import CategoricalReconstuctionLoss, ReconstructionLoss
    torch.manual_seed(0)
    xh = torch.randn(70, 40, device="cuda", requires_grad=True)
    x = torch.randn(70, 40, device="cuda")
    x[:, -6:] = (x[:, -6:] > 0).float()
    loss = CategoricalReconstuctionLoss(6)(xh, x)
    ref = ((xh[:, :-6] - x[:, :-6]) ** 2).sum(-1) + torch.nn.functional.binary_cross_entropy_with_logits(
        xh[:, -6:], x[:, -6:], reduction="none").sum(-1)
    assert torch.allclose(loss, ref, rtol=1e-5, atol=1e-5)
    loss.sum().backward()
    g1 = xh.grad.clone()
    xh.grad = None
    ref.sum().backward()
    assert torch.allclose(g1, xh.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(ReconstructionLoss()(xh.detach(), x), ((xh.detach() - x) ** 2).sum(-1), rtol=1e-5)


def test_recon_loss_speculative_gradient_equals_plain_pair():
    """ReconLossFunction's training form (csrc/recon_loss.hip *_spec): the forward pre-writes 2 (x_hat - x) / B and the
    backward only redoes rows whose upstream gradient is not 1/B.  Bitwise the same as the plain forward + backward,
    whether the expectation holds (mean().backward()), fails for every row (sum().backward()) or for a few rows."""
    from rqhip import ops
    from rqhip.autograd import ReconLossFunction
    torch.manual_seed(5)
    B, N = 3000, 768
    x = torch.randn(B, N, device="cuda")
    base = torch.randn(B, N, device="cuda")
    w = torch.rand(B, device="cuda")
    w[::7] = 1.0 / B
    for reduce in (lambda t: t.mean(), lambda t: t.sum(), lambda t: (t * w).sum()):
        xh = base.clone().requires_grad_(True)
        out = ReconLossFunction.apply(xh, x)
        assert torch.equal(out, ops.recon_loss_forward(base, x))
        g_rows = torch.autograd.grad(reduce(out), out, retain_graph=True)[0]
        reduce(out).backward()
        want, _ = ops.recon_loss_backward(base, x, g_rows.contiguous(), True, False)
        assert torch.equal(xh.grad, want)
    # gradient accumulation: the caller announces the factor it applies to the mean loss (rqhip.autograd.loss_scale);
    # right or wrong, the hint never changes the result
    from rqhip.autograd import loss_scale
    for hint, applied in ((0.25, 0.25), (1.0 / 3.0, 1.0 / 3.0), (0.5, 0.125)):
        xh = base.clone().requires_grad_(True)
        with loss_scale(hint):
            out = ReconLossFunction.apply(xh, x)
        g_rows = torch.autograd.grad(out.mean() * applied, out, retain_graph=True)[0]
        (out.mean() * applied).backward()
        want, _ = ops.recon_loss_backward(base, x, g_rows.contiguous(), True, False)
        assert torch.equal(xh.grad, want)
        if hint == applied:   # the expectation held: every row's upstream gradient is the announced fp32 value
            want_row = float(torch.tensor(hint, dtype=torch.float32) * (torch.tensor(1.0, dtype=torch.float32) / B))
            assert torch.equal(g_rows, torch.full_like(g_rows, want_row))
    # inference (no gradient requested) takes the plain kernel
    with torch.no_grad():
        assert torch.equal(ReconLossFunction.apply(base, x), ops.recon_loss_forward(base, x))


def test_loss_means_and_their_backward_equal_autograd():
    """LossMeansFunction (csrc/recon_loss.hip loss_means / loss_means_bwd): the three means to fp32 rounding of a
    different summation order, and the row gradients equal to what autograd derives for mean(r + q), mean(r), mean(q)
    under any mix of used / unused outputs -- bit for bit when one mean feeds an input (the training step), to one
    rounding when two do ((a + b) * (1/n) here, a * (1/n) + b * (1/n) there)."""
    from rqhip.autograd import LossMeansFunction
    torch.manual_seed(11)
    for n in (1, 7, 100_000):
        r0 = torch.rand(n, device="cuda") * 3
        q0 = torch.rand(n, device="cuda")
        for weights in ((1.0, None, None), (0.5, 2.0, None), (None, None, 3.0), (1.0, 0.25, 0.125), (None, 1.0, None)):
            r, q = r0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
            outs = LossMeansFunction.apply(r, q)
            rr, qr = r0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
            refs = ((rr + qr).mean(), rr.mean(), qr.mean())
            for a, b in zip(outs, refs):
                assert torch.allclose(a, b, rtol=2e-6, atol=0)
            sum(w * o for w, o in zip(weights, outs) if w is not None).backward()
            sum(w * o for w, o in zip(weights, refs) if w is not None).backward()
            for got, want, terms in ((r.grad, rr.grad, (weights[0], weights[1])), (q.grad, qr.grad, (weights[0], weights[2]))):
                if want is None:
                    assert got is None
                elif sum(t is not None for t in terms) == 1:
                    assert torch.equal(got, want), (n, weights)
                else:
                    assert torch.allclose(got, want, rtol=3e-7, atol=0), (n, weights)


# ---------------------------------------------------------------- property tests ----------------------

def test_forward_property_random_shapes_and_special_values():
    """Randomised (hypothesis) shapes and adversarial values -- duplicated codes, exact hits, zeros, huge and tiny
    magnitudes, NaN/Inf sprinkled into rows and codes -- all outputs bit-exact vs the oracle in every mode."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(B=st.integers(1, 200), D=st.sampled_from([1, 3, 8, 16, 24, 32, 33, 64, 100, 128]),
           K=st.sampled_from([1, 2, 5, 31, 32, 33, 64, 100, 256, 300]), L=st.integers(1, 5), mode=st.integers(0, 2),
           seed=st.integers(0, 2 ** 16), flavour=st.sampled_from(["normal", "dups", "tiny", "huge", "special", "zeros"]))
    def run(B, D, K, L, mode, seed, flavour):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((B, D)).astype(np.float32)
        cbs = rng.standard_normal((L, K, D)).astype(np.float32) * 0.5
        if flavour == "dups" and K > 1:
            cbs[:, K - 1] = cbs[:, 0]
            x[0] = cbs[0, 0]
            if B > 1:
                x[1] = 0.5 * (cbs[0, 0] + cbs[0, min(1, K - 1)])      # equidistant in exact arithmetic
        elif flavour == "tiny":
            x *= 1e-20
            cbs *= 1e-19
        elif flavour == "huge":
            x *= 3e18
            cbs[0] *= 1e18
        elif flavour == "zeros":
            x[:] = 0
            cbs[:, ::2] = 0
        elif flavour == "special":
            x[rng.integers(0, B), rng.integers(0, D)] = np.nan
            x[rng.integers(0, B), rng.integers(0, D)] = np.inf
            cbs[rng.integers(0, L), rng.integers(0, K), rng.integers(0, D)] = -np.inf if seed % 2 else np.nan
        _check_forward(x, cbs, mode)

    run()


def test_backward_property_random_shapes():
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(B=st.integers(1, 300), D=st.sampled_from([3, 8, 16, 32, 33, 64]), K=st.sampled_from([2, 31, 64, 256, 1100]),
           L=st.integers(1, 6), mode=st.integers(0, 2), seed=st.integers(0, 2 ** 16))
    def run(B, D, K, L, mode, seed):
        rng = np.random.default_rng(seed)
        x = (rng.standard_normal((B, D)) * 0.7).astype(np.float32)
        cbs = (rng.standard_normal((L, K, D)) * 0.4).astype(np.float32)
        ref = o.rq_forward(x, cbs, mode, 0.25)
        g = dict(g_embs=rng.standard_normal((L, B, D)).astype(np.float32) if seed % 2 else None,
                 g_embsum=rng.standard_normal((B, D)).astype(np.float32),
                 g_resid=rng.standard_normal((L, B, D)).astype(np.float32) if seed % 3 == 0 else None,
                 g_loss=rng.random(B).astype(np.float32))
        r_res0, r_cb = o.rq_backward(x, cbs, mode, 0.25, ref["ids"], **g)
        g_res0, g_cb = _run_backward(x, cbs, mode, 0.25, ref["ids"], **g)
        _assert_bitexact(g_res0, r_res0, f"g_res0 B={B} D={D} K={K} L={L} mode={mode}")
        _assert_cb_grad(g_cb, x, cbs, mode, ref["ids"], g, r_cb, f"B={B} D={D} K={K} L={L} mode={mode}")

    run()
