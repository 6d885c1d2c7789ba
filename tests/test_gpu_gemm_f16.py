"""The two-piece fp16 form of the activation GEMMs (csrc/gemm_split.hip built with GS_F16=1: exact power-of-two row scales,
products hh + hm + mh).  Skipped while the library is built without it (the round-3 default); these are the
gates the path has to pass before it replaces the three-piece bf16 split: on every operand family no less exact against fp64
than the library's fp32 GEMM, and inside the |A||B| bound of tests/test_gpu_gemm_split.py over twelve decades of row scales."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib_f16():
    from rqhip import _lib
    h = _lib.lib()
    if not hasattr(h, "rqhip_gemm_split_f16"):
        pytest.skip("librqhip.so is built without GS_F16")
    h.rqhip_weight_planes_bytes.restype = C.c_size_t
    return h


def _gemm_f16(h, a, w, relu=False):
    vp = C.c_void_p
    M, R = a.shape
    Nc = w.shape[0]
    nb = h.rqhip_weight_planes_bytes(Nc, R)
    planes = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    assert h.rqhip_weight_planes(vp(w.data_ptr()), Nc, R, 0, vp(planes.data_ptr()), C.c_size_t(nb), None) == 0
    ex = torch.empty((M,), dtype=torch.int32, device="cuda")
    c = torch.empty((M, Nc), device="cuda")
    assert h.rqhip_row_exponents(vp(a.data_ptr()), C.c_int64(M), R, vp(ex.data_ptr()), None) == 0
    assert h.rqhip_gemm_split_f16(vp(a.data_ptr()), vp(ex.data_ptr()), C.c_int64(M), R, vp(planes.data_ptr()), Nc, int(relu),
                                  vp(c.data_ptr()), None) == 0
    return c


def _families():
    g = torch.Generator().manual_seed(0)
    M, K, N = 8192, 768, 512
    x = torch.nn.functional.normalize(torch.randn(M, K, generator=g), dim=-1)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    hdn = torch.relu(x @ w.t())
    yield "unit-norm rows", x, w
    yield "post-ReLU activations", hdn, torch.randn(256, N, generator=g) / N ** 0.5
    yield "1e-5-scale masked gradient", torch.randn(M, N, generator=g) * 1e-5 * (torch.rand(M, N, generator=g) > 0.5), w.t().contiguous()
    yield ("twelve decades of row scales", torch.randn(M, K, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float()),
           torch.randn(N, K, generator=g) * torch.pow(10.0, torch.randint(-3, 4, (N, 1), generator=g).float()))
    yield "five decades inside every row", torch.randn(M, K, generator=g) * torch.pow(10.0, torch.randint(-4, 1, (M, K), generator=g).float()), w


def test_gemm_f16_no_less_exact_than_the_library_and_inside_the_bound():
    h = _lib_f16()
    for name, a, w in _families():
        a, w = a.cuda().contiguous(), w.cuda().contiguous()
        ref = a.double() @ w.double().t()
        scale = ref.abs().max().item()
        bound = (a.double().abs() @ w.double().abs().t()) * ((a.shape[1] ** 0.5 + 8) * 2.0 ** -24) + 1e-300
        c = _gemm_f16(h, a, w)
        err = (c.double() - ref).abs()
        lerr = ((a @ w.t()).double() - ref).abs().max().item() / scale
        assert (err <= bound).all(), (name, float((err / bound).max()))
        assert err.max().item() / scale <= max(lerr, 2e-7), (name, err.max().item() / scale, lerr)
        assert torch.equal(c, _gemm_f16(h, a, w)), name          # bit-reproducible


def test_gemm_f16_relu_and_ragged_rows():
    h = _lib_f16()
    g = torch.Generator().manual_seed(3)
    for M in (1, 127, 4099):
        a = torch.randn(M, 512, generator=g).cuda()
        w = (torch.randn(256, 512, generator=g) / 512 ** 0.5).cuda()
        c = _gemm_f16(h, a, w, relu=True)
        ref = torch.relu(a.double() @ w.double().t())
        assert (c >= 0).all()
        assert (c.double() - ref).abs().max().item() <= 2e-6 * max(ref.abs().max().item(), 1e-30)
