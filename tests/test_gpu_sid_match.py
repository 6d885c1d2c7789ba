"""GPU parity for the decoder-side consumers of semantic ids:
valid-prefix lookup (modules/model.py:169-182) and first-match rank / retrieval metrics
(evaluate/metrics.py:7-28).  Integer work: every result must be bit-exact against the oracle and against the
fixtures the reference itself produced."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import rq_oracle as o

pytestmark = pytest.mark.gpu


def _names(pat):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, pat)))


def _gpu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", _names("prefix_*.npz"))
def test_prefix_lookup_matches_reference_fixture(name):
    from modules.sid_prefix import SemIdPrefixIndex
    g = load_golden(name)
    index = SemIdPrefixIndex(_gpu(g["corpus"]))
    for h in range(1, g["corpus"].shape[1] + 1):
        got = index.check_valid_prefix(_gpu(g[f"prefix_h{h}"]))
        assert got.dtype == torch.bool and got.shape == (g[f"prefix_h{h}"].shape[0],)
        assert np.array_equal(got.cpu().numpy(), g[f"valid_h{h}"])


@pytest.mark.parametrize("N,H,K,P,seed", [(1, 1, 2, 5, 0), (37, 2, 3, 100, 1), (5000, 3, 16, 4000, 2),
                                          (20000, 4, 256, 3000, 3), (3000, 6, 4, 2000, 4)])
def test_prefix_lookup_matches_oracle(N, H, K, P, seed):
    from rqhip import ops
    rng = np.random.default_rng(seed)
    corpus = rng.integers(0, K, size=(N, H)).astype(np.int64)
    dc = _gpu(corpus)
    index = ops.prefix_index_build(dc)
    for h in range(0, H + 1):
        prefix = np.concatenate([corpus[rng.integers(0, N, size=P // 2), :h],
                                 rng.integers(-1, K + 2, size=(P - P // 2, h)).astype(np.int64)], axis=0)
        got = ops.prefix_lookup(index, dc, _gpu(prefix)).cpu().numpy()
        assert np.array_equal(got, o.prefix_valid(corpus, prefix)), f"h={h}"


def test_prefix_lookup_on_strided_views_and_wide_values():
    """The model's codebooks are a column slice of the tokenizer's [N, L+1] table (train_decoder.py:131) and
    beam prefixes are built by torch.cat; neither needs to be dense.  Values are compared as full int64."""
    from rqhip import ops
    rng = np.random.default_rng(9)
    table = rng.integers(0, 7, size=(900, 5)).astype(np.int64)
    table[11, :3] = [2**40 + 5, -3, 2**33]
    dt = _gpu(table)
    corpus = dt[:, :3]                      # row stride 5
    index = ops.prefix_index_build(corpus)
    q = np.stack([table[11, :3], [5, -3, 2**33], table[12, :3], [6, 6, 99]]).astype(np.int64)
    wide = torch.zeros((4, 9), dtype=torch.int64, device="cuda")
    wide[:, 2:5] = _gpu(q)
    got = ops.prefix_lookup(index, corpus, wide[:, 2:5]).cpu().numpy()
    assert np.array_equal(got, o.prefix_valid(table[:, :3], q))
    assert got.tolist()[:2] == [True, False]


def test_prefix_index_full_size_properties():
    """10^6 items x 4 levels (a C4-scale shard): every corpus row's prefixes are valid; a random sample agrees
    with a packed-key membership test done on the host; extending an invalid prefix never makes it valid;
    tuples containing an id outside the codebook are never valid."""
    from rqhip import ops
    rng = np.random.default_rng(5)
    N, H, K = 1_000_000, 4, 1024
    corpus = rng.integers(0, K, size=(N, H)).astype(np.int64)
    corpus[:, 0] = rng.integers(0, 64, size=N)       # skewed first level: 64 prefixes shared by 10^6 rows
    dc = _gpu(corpus)
    index = ops.prefix_index_build(dc)
    rows = _gpu(rng.integers(0, N, size=200_000))
    for h in range(1, H + 1):
        assert bool(ops.prefix_lookup(index, dc, dc[rows, :h]).all())
        q = rng.integers(0, K, size=(300_000, h)).astype(np.int64)
        q[:, 0] = rng.integers(0, 80, size=len(q))
        got = ops.prefix_lookup(index, dc, _gpu(q)).cpu().numpy()
        shifts = (10 * np.arange(h))[None, :]        # ids < 2^10, h <= 4: 40-bit packed keys
        keys = np.unique((corpus[:, :h] << shifts).sum(axis=1))
        assert np.array_equal(got, np.isin((q << shifts).sum(axis=1), keys)), f"h={h}"
        assert got.any() or h == H
        bad = q.copy()
        bad[:, rng.integers(0, h)] = K + 3
        assert not bool(ops.prefix_lookup(index, dc, _gpu(bad)).any())
        if h < H:
            longer = np.concatenate([q, rng.integers(0, K, size=(len(q), 1)).astype(np.int64)], axis=1)
            got_longer = ops.prefix_lookup(index, dc, _gpu(longer)).cpu().numpy()
            assert not (got_longer & ~got).any()


def test_prefix_index_error_behaviour():
    from modules.sid_prefix import SemIdPrefixIndex
    from rqhip._lib import RqHipError
    index = SemIdPrefixIndex(torch.arange(12, device="cuda").reshape(4, 3))
    with pytest.raises(RuntimeError):            # the reference ends in torch.cat([]) on an empty batch
        index.check_valid_prefix(torch.zeros((0, 2), dtype=torch.int64, device="cuda"))
    with pytest.raises(RuntimeError):            # more columns than the corpus has
        index.check_valid_prefix(torch.zeros((2, 4), dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        SemIdPrefixIndex(torch.zeros((4, 3), dtype=torch.int32))
    with pytest.raises(RqHipError):              # no CPU path
        from rqhip import ops
        ops.prefix_index_build(torch.zeros((4, 3), dtype=torch.int64))
    assert index.check_valid_prefix(torch.tensor([[3, 4], [3, 5]], device="cuda")).tolist() == [True, False]
    assert index.check_valid_prefix(torch.zeros((2, 0), dtype=torch.int64, device="cuda")).tolist() == [True, True]


@pytest.mark.parametrize("name", _names("topk_*.npz"))
def test_topk_accumulator_matches_reference_fixture(name):
    from evaluate.metrics import TopKAccumulator
    from rqhip import ops
    g = load_golden(name)
    acc = TopKAccumulator(ks=[1, 5, 10])
    for part in range(2):
        a, t = _gpu(g[f"actual_{part}"]), _gpu(g[f"top_k_{part}"])
        assert np.array_equal(ops.topk_first_match(a, t).cpu().numpy(), g[f"rank_{part}"])
        acc.accumulate(actual=a, top_k=t)
    red = acc.reduce()
    assert sorted(red) == g["metric_names"].tolist()
    for k, v in zip(g["metric_names"].tolist(), g["metric_values"].tolist()):
        assert abs(red[k] - v) <= 1e-6, (k, red[k], v)   # fp32 gains, summation order
    acc.reset()
    assert acc.total == 0 and len(acc.metrics) == 0


@pytest.mark.parametrize("B,K,D,vocab,seed", [(1, 1, 1, 2, 0), (257, 10, 3, 4, 1), (1000, 64, 4, 3, 2),
                                              (4096, 10, 0, 5, 3), (50_000, 10, 3, 8, 4)])
def test_topk_first_match_matches_oracle(B, K, D, vocab, seed):
    from rqhip import ops
    rng = np.random.default_rng(seed)
    actual = rng.integers(0, vocab, size=(B, D)).astype(np.int64)
    top_k = rng.integers(0, vocab, size=(B, K, D)).astype(np.int64)
    got = ops.topk_first_match(_gpu(actual), _gpu(top_k)).cpu().numpy()
    assert np.array_equal(got, o.topk_first_match(actual, top_k))
