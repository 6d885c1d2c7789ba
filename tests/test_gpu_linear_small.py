"""csrc/mlp_small.hip (rqhip_linear_small): the encoder / decoder Linear layers below 4096 rows, forward and data gradient -- reference
modules/encoder.py:25-38 (`relu(x W^T)`; autograd's `g W` + the ReLU backward of the layer below).  Bit-exact against the oracle's
restatement of the kernel's summation order (oracle/rq_oracle.c:rqo_linear_small), for every launch plan, both weight orientations, the
three epilogues, ragged batches; and no further from fp64 than the library's fp32 GEMM."""
import numpy as np
import pytest
import torch

from oracle import rq_oracle as oracle
from rqhip import _lib, ops

pytestmark = pytest.mark.gpu

LAYERS = [(512, 768), (256, 512), (128, 256), (32, 128), (128, 32), (256, 128), (512, 256), (768, 512), (64, 128), (128, 64)]   # (n_out, n_in)


def _data(M, n_out, n_in, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, n_in, generator=g) * torch.rand(M, 1, generator=g)
    w = (torch.rand(n_out, n_in, generator=g) - 0.5) * (2.0 / n_in ** 0.5)
    return a, w


@pytest.mark.parametrize("M", [640, 581, 64, 1, 33])
def test_forward_and_data_gradient_of_every_shipped_layer_match_the_oracle_bit_for_bit(M):
    for li, (n_out, n_in) in enumerate(LAYERS):
        a, w = _data(M, n_out, n_in, 100 + li)
        cb, ks = ops.linear_small_plan(M, n_out, n_in)
        y = ops.linear_small(a.cuda(), w.cuda(), epilogue=_lib.EPI_RELU)
        ref = oracle.linear_small(a.numpy(), w.numpy(), False, ks, 1)
        assert np.array_equal(y.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (M, n_out, n_in, cb, ks)
        # the data gradient of the same layer: g [M, n_out] . w [n_out, n_in], masked by the activation below
        g = torch.randn(M, n_out, generator=torch.Generator().manual_seed(7 + li))
        below = torch.randn(M, n_in, generator=torch.Generator().manual_seed(9 + li)).clamp_min(0.0)
        cb2, ks2 = ops.linear_small_plan(M, n_in, n_out)
        gx = ops.linear_small(g.cuda(), w.cuda(), w_kn=True, epilogue=_lib.EPI_MASK, aux=below.cuda())
        ref = oracle.linear_small(g.numpy(), w.numpy(), True, ks2, 3, below.numpy())
        assert np.array_equal(gx.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (M, n_in, n_out, cb2, ks2)
        assert (gx.cpu()[below <= 0] == 0).all()


@pytest.mark.parametrize("cb,ks", [(1, 4), (1, 8), (1, 16), (2, 4), (2, 8)])
@pytest.mark.parametrize("w_kn", [False, True])
def test_every_launch_plan(cb, ks, w_kn):
    """The plan only changes how many partial chains an output is summed from (`waves`), never which terms: each plan against the oracle
    at its own `waves`, on a reduction whose group count does not divide evenly (Kr = 96: 3 groups over 4, 8 or 16 waves, some empty)."""
    for M, N, Kr in ((100, 128, 96), (640, 256, 512), (37, 64, 32)):
        g = torch.Generator().manual_seed(M + N + Kr)
        a = torch.randn(M, Kr, generator=g)
        w = torch.randn(Kr, N, generator=g) if w_kn else torch.randn(N, Kr, generator=g)
        y = ops.linear_small(a.cuda(), w.cuda(), w_kn=w_kn, col_blocks=cb, waves=ks)
        ref = oracle.linear_small(a.numpy(), w.numpy(), w_kn, ks, 0)
        assert np.array_equal(y.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (M, N, Kr)


def test_special_values_and_epilogues():
    M, N, Kr = 70, 64, 64
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, Kr, generator=g)
    w = torch.randn(N, Kr, generator=g)
    a[0, 0], a[1, 5], a[2, :] = float("nan"), float("inf"), 0.0
    a[3, :] = 1e-41          # denormal inputs: the matrix instruction keeps them (as the fp32 FMA does)
    a[4, :] = -a[5, :]
    for epi in (_lib.EPI_STORE, _lib.EPI_RELU):
        y = ops.linear_small(a.cuda(), w.cuda(), epilogue=epi).cpu().numpy()
        ref = oracle.linear_small(a.numpy(), w.numpy(), False, ops.linear_small_plan(M, N, Kr)[1], epi)
        same = (y.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(y) & np.isnan(ref))
        assert same.all()
        assert np.isnan(y[0]).all()          # ReLU keeps a NaN, as torch.relu
    with pytest.raises(_lib.RqHipError):
        ops.linear_small(a.cuda(), w.cuda(), epilogue=_lib.EPI_MASK)       # mask without aux
    with pytest.raises(_lib.RqHipError):
        ops.linear_small(a[:, :40].contiguous().cuda(), w[:, :40].contiguous().cuda())   # Kr = 40
    with pytest.raises(_lib.RqHipError):
        ops.linear_small(a, w)               # CPU tensors: no fallback
    assert ops.linear_small(a[:0].cuda(), w.cuda()).shape == (0, N)


def test_error_against_fp64_is_no_larger_than_the_librarys():
    worst = 0.0
    for li, (n_out, n_in) in enumerate(LAYERS[:8]):
        a, w = _data(640, n_out, n_in, 500 + li)
        exact = a.double() @ w.double().t()
        ours = ops.linear_small(a.cuda(), w.cuda()).cpu().double()
        lib = (a.cuda() @ w.cuda().t()).cpu().double()
        scale = (a.double().abs() @ w.double().abs().t()).clamp_min(1e-30)
        e_ours, e_lib = ((ours - exact).abs() / scale).max().item(), ((lib - exact).abs() / scale).max().item()
        worst = max(worst, e_ours / max(e_lib, 1e-12))
        assert e_ours <= 2.0 ** -20, (n_out, n_in, e_ours)          # chains of fp32 FMAs: far inside n_in * 2^-24 of the term magnitudes
        assert e_ours <= 2.0 * e_lib + 2.0 ** -24, (n_out, n_in, e_ours, e_lib)
    print("max error ratio ours / library:", worst)
