"""csrc/gemm_img.hip (round 5): the MLP GEMMs and weight gradients on operand IMAGES -- an activation / gradient is split into its
two fp16 pieces ONCE, by the kernel that produces it, in the layouts its consumers stream (R planes: the A operand of a GEMM; T planes:
the weight gradient's operands; E: one exact power-of-two exponent per (256-column segment, row)).  Reference: modules/encoder.py:25-38
and its autograd, modules/rqvae.py:146,152 + modules/loss.py:5-10.

Gates, the same as tests/test_gpu_gemm_split.py / test_gpu_wgrad.py hold the round-4 kernels to: max error against fp64 <= the library
fp32 GEMM's on the same inputs (full size, every operand family), ragged row counts against an elementwise bound, special rows,
bit-reproducibility -- plus what is new here: the image formats themselves (both layouts stand for the same values, the representation
error bound tests/test_split_bound.py proves, exponents exact), segment boundaries in the reduction (rescaled accumulators, dropped
segments), epilogue-written images == images packed from the fp32 result, the run-flag protocol of the speculative reconstruction gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu

E_ZERO = -200


def _exp_of_max(mx: torch.Tensor) -> torch.Tensor:
    """the exponent the kernels give a segment whose largest |value| is mx (fp32): exponent(mx) - 14; -200 for 0; 0 for inf / nan"""
    bits = mx.float().view(torch.int32) & 0x7fffffff
    e = (bits >> 23)
    out = torch.where(e == 0, torch.full_like(e, -126), e - 127) - 14
    out = torch.where(bits == 0, torch.full_like(out, E_ZERO), out)
    return torch.where(e == 255, torch.zeros_like(out), out)


def _seg_exponents(a: torch.Tensor, seg: int) -> torch.Tensor:
    M, N = a.shape
    mx = a.abs().view(M, N // seg, seg)
    mx = torch.where(torch.isnan(mx), torch.full_like(mx, float("inf")), mx).amax(dim=2)
    return _exp_of_max(mx).t().contiguous()          # [N / seg, M]


def _repr_bound(a: torch.Tensor, E: torch.Tensor, seg: int) -> torch.Tensor:
    """|v - (h + m) 2^E| <= max(2^-23 |v|, 2^-25 2^E) (tests/test_split_bound.py)"""
    M, N = a.shape
    e = E.t().double().repeat_interleave(seg, dim=1)           # [M, N]
    # (+ half a quantum of fp32's subnormals: img_unpack returns fp32, which cannot hold the represented value of a subnormal row finer)
    return torch.maximum(a.double().abs() * 2.0 ** -23, torch.pow(2.0, e - 25)) + 2.0 ** -150


@pytest.mark.parametrize("M,N", [(100_000, 768), (5003, 512), (77, 256), (1, 128), (33, 384), (4099, 128)])
def test_img_pack_both_layouts_exponents_and_representation(M, N):
    from rqhip import ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, N, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float())
    a = a * torch.pow(10.0, torch.randint(-2, 1, (M, N), generator=g).float())
    if M > 40:
        a[3] = 0.0                                   # a zero row: every segment gets the zero exponent
        a[5, : min(N, 256)] = 0.0                    # one zero segment
        a[7] *= 1e-41 / max(a[7].abs().max().item(), 1e-30)   # subnormals
    a = a.cuda()
    img, _ = ops.img_pack(a)
    seg = img.seg
    assert seg == (256 if N % 256 == 0 else 128) and tuple(img.E.shape) == (N // seg, M)
    assert torch.equal(img.E, _seg_exponents(a, seg).to(torch.int32))
    r, t = ops.img_unpack(img), ops.img_unpack(img, from_t=True)
    assert torch.equal(r, t)                                                   # both layouts stand for the same values
    assert ((r.double() - a.double()).abs() <= _repr_bound(a, img.E, seg)).all()
    if M > 40:
        assert (r[3] == 0).all() and (img.E[:, 3] == E_ZERO).all() and img.E[0, 5] == E_ZERO
    # the ReLU backward inside the pass
    y = torch.randn(M, N, generator=g).cuda()
    want = torch.where(y > 0, a, torch.zeros_like(a))
    im2, masked = ops.img_pack(a, y, write_masked=True)
    assert torch.equal(masked, want) and torch.equal(im2.E, _seg_exponents(want, seg).to(torch.int32))
    assert torch.equal(ops.img_unpack(im2), ops.img_unpack(ops.img_pack(want)[0]))
    # rows past M of the T planes are zeros (the weight gradient reduces over them)
    t_bytes = img.T.view(torch.int16)
    n_pad = (M + 31) // 32 * 32
    assert t_bytes.numel() >= n_pad * N * 2
    img_pad = ops.img_pack(torch.cat([a, torch.zeros(n_pad - M, N, device="cuda")]))[0] if n_pad > M else None
    if img_pad is not None:
        assert torch.equal(img_pad.T[: n_pad * N * 4], img.T[: n_pad * N * 4])


def test_img_pack_special_values():
    from rqhip import ops
    g = torch.Generator().manual_seed(2)
    a = torch.randn(64, 256, generator=g)
    a[1, 7] = float("inf")
    a[2, 9] = float("nan")
    a[4] *= 1e30
    a = a.cuda()
    img, _ = ops.img_pack(a)
    assert img.E[0, 1] == 0 and img.E[0, 2] == 0                     # inf / nan segments are not scaled
    r = ops.img_unpack(img)
    assert not torch.isfinite(r[1, 7]) and torch.isnan(r[2, 9])
    ok = torch.ones(64, dtype=torch.bool)
    ok[[1, 2]] = False
    assert ((r.double() - a.double()).abs()[ok] <= _repr_bound(a, img.E, 256)[ok]).all()


def _gemm(a, w, transpose, epi=None, **kw):
    from rqhip import _lib, ops
    image = ops.weight_planes(w, transpose=transpose, arith=ops.F16X2)
    n_cols = w.shape[1] if transpose else w.shape[0]
    img = ops.img_pack(a, want_t=False)[0]
    return ops.gemm_img(img, image, n_cols, epilogue=_lib.EPI_STORE if epi is None else epi, **kw)


SHAPES = [(512, 768), (768, 512), (256, 512), (512, 256), (128, 256), (256, 128), (384, 128), (128, 384)]


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("Nc,R", SHAPES)
def test_gemm_img_full_size_vs_fp64(Nc, R, transpose, relu):
    from rqhip import _lib, ops
    g = torch.Generator().manual_seed(Nc + R)
    M = 100_000
    a = torch.randn(M, R, generator=g).cuda()
    w = (torch.randn(R, Nc, generator=g) if transpose else torch.randn(Nc, R, generator=g)).cuda() / R ** 0.5
    c, out, _ = _gemm(a, w, transpose, _lib.EPI_RELU if relu else _lib.EPI_STORE, want_c=True, want_r=True, want_t=True)
    b = w if transpose else w.t()
    ref, lib = a.double() @ b.double(), a @ b
    if relu:
        ref, lib = torch.relu(ref), torch.relu(lib)
    scale = ref.abs().max().item()
    err = (c.double() - ref).abs().max().item() / scale
    lerr = (lib.double() - ref).abs().max().item() / scale
    print(f"img: C [{M},{Nc}] = A [{M},{R}] B^T (transpose={transpose}, relu={relu}): max err / max|C| = {err:.3e} (library {lerr:.3e})")
    assert err < 2e-6 and err <= max(lerr, 2e-7), (err, lerr)
    # the image the epilogue wrote == the image packed from the fp32 result it stored (same exponents, same pieces, both layouts)
    packed = ops.img_pack(c)[0]
    assert torch.equal(out.E, packed.E)
    assert torch.equal(ops.img_unpack(out), ops.img_unpack(packed)) and torch.equal(ops.img_unpack(out, from_t=True), ops.img_unpack(packed))
    n_r, n_t = M * Nc * 4, (M + 31) // 32 * 32 * Nc * 4
    assert torch.equal(out.T[:n_t], packed.T[:n_t])
    # bit-reproducible, and the per-XCD dispensers give the same bits
    c2 = _gemm(a, w, transpose, _lib.EPI_RELU if relu else _lib.EPI_STORE, want_c=True)[0]
    c3 = _gemm(a, w, transpose, _lib.EPI_RELU if relu else _lib.EPI_STORE, want_c=True, xcd_queues=True)[0]
    assert torch.equal(c, c2) and torch.equal(c, c3)
    del n_r


def _families(Nc, R):
    from test_gpu_gemm_split import _families as fam
    return fam(Nc, R)


@pytest.mark.parametrize("Nc,R", [(512, 768), (768, 512), (256, 512), (512, 256)])
def test_gemm_img_operand_families_vs_fp64(Nc, R):
    for name, A, B in _families(Nc, R):
        A, B = A.cuda().contiguous(), B.cuda().contiguous()
        c = _gemm(A, B, False, want_c=True)[0]
        ref = A.double() @ B.double().t()
        scale = ref.abs().max().item()
        err = (c.double() - ref).abs().max().item() / scale
        lerr = ((A @ B.t()).double() - ref).abs().max().item() / scale
        print(f"img {R}->{Nc} {name}: {err:.3e} (library {lerr:.3e})")
        assert err <= max(lerr, 2e-7), (name, err, lerr)


@pytest.mark.parametrize("M", [1, 63, 64, 65, 127, 129, 4099])
def test_gemm_img_ragged_rows_and_scales(M):
    from rqhip import _lib, ops
    g = torch.Generator().manual_seed(M)
    a = (torch.randn(M, 768, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float())).cuda()
    w = (torch.randn(512, 768, generator=g) * torch.pow(10.0, torch.randint(-3, 4, (512, 1), generator=g).float())).cuda()
    c, out, _ = _gemm(a, w, False, want_c=True, want_r=True, want_t=True)
    ref = a.double() @ w.double().t()
    bound = a.double().abs() @ w.double().abs().t() * (768 ** 0.5 + 8) * 2.0 ** -24 + 1e-30
    assert ((c.double() - ref).abs() <= bound).all(), float(((c.double() - ref).abs() / bound).max())
    packed = ops.img_pack(c)[0]
    assert torch.equal(out.E, packed.E) and torch.equal(ops.img_unpack(out), ops.img_unpack(packed))
    assert torch.equal(ops.img_unpack(out, from_t=True), ops.img_unpack(packed))
    n_t = (M + 31) // 32 * 32 * 512 * 4
    assert torch.equal(out.T[:n_t], packed.T[:n_t])                     # including the zero rows past M
    # the 128-column tile
    w2 = (torch.randn(128, 768, generator=g) * torch.pow(10.0, torch.randint(-3, 4, (128, 1), generator=g).float())).cuda()
    c2, out2, _ = _gemm(a, w2, False, _lib.EPI_RELU, want_c=True, want_r=True, want_t=True)
    ref2 = torch.relu(a.double() @ w2.double().t())
    bound2 = a.double().abs() @ w2.double().abs().t() * (768 ** 0.5 + 8) * 2.0 ** -24 + 1e-30
    assert ((c2.double() - ref2).abs() <= bound2).all()
    p2 = ops.img_pack(c2)[0]
    assert torch.equal(out2.E, p2.E) and torch.equal(ops.img_unpack(out2), ops.img_unpack(p2)) and torch.equal(ops.img_unpack(out2, from_t=True), ops.img_unpack(p2))


def test_gemm_img_segments_of_very_different_size_zero_segments_special_rows():
    """The reduction crosses segment boundaries with a rescale of the accumulators: segments whose exponents differ by decades, zero
    segments anywhere in the row, segments more than 2^64 below the row's largest (dropped: below fp32's resolution of the sum),
    and rows with inf / nan."""
    from rqhip import ops
    g = torch.Generator().manual_seed(13)
    M, R, Nc = 700, 768, 256
    a = torch.randn(M, R, generator=g)
    seg_scale = torch.pow(10.0, torch.randint(-9, 10, (M, 3), generator=g).float())
    a = (a.view(M, 3, 256) * seg_scale[:, :, None]).view(M, R)
    a[10, :256] = 0.0
    a[11, 256:512] = 0.0
    a[12, 512:] = 0.0
    a[13] = 0.0
    a[14, :256] *= 1e-30
    a[14, 256:] *= 1e+5          # first segment ~2^-116 below the others: dropped
    a[15, 3] = float("inf")
    a[16, 300] = float("nan")
    w = torch.randn(Nc, R, generator=g) / R ** 0.5
    a, w = a.cuda(), w.cuda()
    c = _gemm(a, w, False, want_c=True)[0]
    ref = a.double() @ w.double().t()
    assert (c[13] == 0).all() and not torch.isfinite(c[15]).all() and torch.isnan(c[16]).all()
    ok = torch.ones(M, dtype=torch.bool)
    ok[[15, 16]] = False
    bound = a.double().abs() @ w.double().abs().t() * (R ** 0.5 + 8) * 2.0 ** -24 + 1e-30
    # a dropped segment's terms are missing: add their whole magnitude for row 14 (2^-64 of the row's scale and far less here)
    bound[14] += a[14, :256].double().abs() @ w[:, :256].double().abs().t()
    assert ((c.double() - ref).abs()[ok] <= bound[ok]).all(), float(((c.double() - ref).abs()[ok] / bound[ok]).max())


def test_gemm_img_mask_epilogue_and_chain():
    """RQHIP_EPI_MASK: the result zeroed where y's image holds no positive value; a GEMM fed by an epilogue-written image equals one fed
    by the image packed from the fp32 matrix (bits)."""
    from rqhip import _lib, ops
    g = torch.Generator().manual_seed(17)
    M = 5003
    a = torch.randn(M, 512, generator=g).cuda()
    w1 = (torch.randn(768, 512, generator=g) / 512 ** 0.5).cuda()
    w2 = (torch.randn(256, 768, generator=g) / 768 ** 0.5).cuda()
    i1, i2 = ops.weight_images([(w1, False), (w2, False)])
    ia = ops.img_pack(a, want_t=False)[0]
    y = torch.relu(torch.randn(M, 768, generator=g)).cuda()
    iy = ops.img_pack(y, want_t=False)[0]
    plain = ops.gemm_img(ia, i1, 768, want_c=True)[0]
    masked, mo, _ = ops.gemm_img(ia, i1, 768, epilogue=_lib.EPI_MASK, y=iy, want_c=True, want_r=True, want_t=True)
    want = torch.where(ops.img_unpack(iy) > 0, plain, torch.zeros_like(plain))
    assert torch.equal(masked, want)
    masked_t = ops.gemm_img(ia, i1, 768, epilogue=_lib.EPI_MASK, y=ops.img_pack(y, want_r=False)[0], want_c=True)[0]   # mask from the T planes
    assert torch.equal(masked_t, want)
    assert torch.equal(want, torch.where(y > 0, plain, torch.zeros_like(plain)))      # (no y here is small enough to flush)
    pm = ops.img_pack(masked)[0]
    assert torch.equal(mo.E, pm.E) and torch.equal(ops.img_unpack(mo), ops.img_unpack(pm))
    h, ho, _ = ops.gemm_img(ia, i1, 768, epilogue=_lib.EPI_RELU, want_c=True, want_r=True)
    chained = ops.gemm_img(ho, i2, 256, want_c=True)[0]
    passed = ops.gemm_img(ops.img_pack(h, want_t=False)[0], i2, 256, want_c=True)[0]
    assert torch.equal(chained, passed)
    ref = torch.relu(a.double() @ w1.double().t()) @ w2.double().t()
    assert (chained.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("M", [100_000, 5003, 77])
def test_gemm_img_recon_epilogue_and_run_flag(M):
    """Epilogue 2 (last decoder layer + ReconstructionLoss, reference modules/rqvae.py:146,152 + loss.py:5-10): the gradient has the bits
    of the plain GEMM followed by recon_loss_forward_spec; the loss rows to 1e-6; with row_scales the gradient is scaled per row; under a
    clear run flag nothing is written, under a set one everything is."""
    from rqhip import _lib, ops
    g = torch.Generator().manual_seed(M)
    a = torch.relu(torch.randn(M, 512, generator=g)).cuda()
    w = (torch.randn(768, 512, generator=g) / 512 ** 0.5).cuda()
    x = torch.nn.functional.normalize(torch.randn(M, 768, generator=g), dim=-1).cuda()
    image = ops.weight_planes(w, arith=ops.F16X2)
    ia = ops.img_pack(a, want_t=False)[0]
    rs = 1.0 / M
    grad, gimg, loss = ops.gemm_img(ia, image, 768, epilogue=_lib.EPI_RECON, aux=x, row_scale=rs, want_c=True, want_r=True, want_t=True)
    xhat = ops.gemm_img(ia, image, 768, want_c=True)[0]
    want_loss, want_grad = ops.recon_loss_forward_spec(xhat, x, rs)
    assert torch.equal(grad, want_grad)
    assert (loss - want_loss).abs().max().item() <= 1e-6 * want_loss.abs().max().item()
    pg = ops.img_pack(grad)[0]
    assert torch.equal(gimg.E, pg.E) and torch.equal(ops.img_unpack(gimg), ops.img_unpack(pg)) and torch.equal(ops.img_unpack(gimg, from_t=True), ops.img_unpack(pg))
    # per-row upstream gradients instead of the announced scalar
    up = (torch.rand(M, generator=g) + 0.5).cuda() / M
    grad2 = ops.gemm_img(ia, image, 768, epilogue=_lib.EPI_RECON, aux=x, row_scales=up, want_c=True)[0]
    assert torch.equal(grad2, (2.0 * (xhat - x)) * up[:, None])
    # the run flag: clear -> untouched; set -> rewritten
    flag = ops.rows_differ(torch.full((M,), rs, device="cuda"), rs)
    assert int(flag) == 0
    keep_c, keep_e, keep_loss = grad.clone(), gimg.E.clone(), loss.clone()
    ops.gemm_img(ia, image, 768, epilogue=_lib.EPI_RECON, aux=x, row_scales=up, run_flag=flag, out=gimg, c_out=grad, loss_out=loss)
    assert torch.equal(grad, keep_c) and torch.equal(gimg.E, keep_e) and torch.equal(loss, keep_loss)
    flag = ops.rows_differ(up, rs)
    assert int(flag) == 1
    ops.gemm_img(ia, image, 768, epilogue=_lib.EPI_RECON, aux=x, row_scales=up, run_flag=flag, out=gimg, c_out=grad, loss_out=loss)
    assert torch.equal(grad, grad2)
    p2 = ops.img_pack(grad2)[0]
    assert torch.equal(gimg.E, p2.E) and torch.equal(ops.img_unpack(gimg), ops.img_unpack(p2))


# ---- the weight gradient ---------------------------------------------------------------------------------------------------------
LAYERS = [(512, 768), (256, 512), (128, 256), (256, 128), (512, 256), (768, 512)]


def _wgrad(gy, x, **kw):
    from rqhip import ops
    return ops.linear_wgrad_img(ops.img_pack(gy, want_r=False)[0], ops.img_pack(x, want_r=False)[0], **kw)


@pytest.mark.parametrize("N,K", LAYERS)
def test_wgrad_img_full_size_vs_fp64(N, K):
    M = 100_000
    g = torch.Generator().manual_seed(N * 7 + K)
    gy = (torch.randn(M, N, generator=g) * 0.3 * (torch.rand(M, N, generator=g) > 0.5)).cuda()     # a masked gradient
    x = torch.randn(M, K, generator=g).cuda()
    dw = _wgrad(gy, x)
    ref = gy.double().t().mm(x.double())
    scale = ref.abs().max().item()
    err = (dw.double() - ref).abs().max().item() / scale
    lib = (gy.t().mm(x).double() - ref).abs().max().item() / scale
    print(f"img dW [{N},{K}]: max err / max|dW| = {err:.3e} (library fp32 GEMM: {lib:.3e})")
    assert err < 2e-6 and err <= max(lib, 2e-7), (err, lib)
    assert torch.equal(dw, _wgrad(gy, x))                                                         # fixed reduction order
    sink = torch.empty(N, K, device="cuda")
    assert _wgrad(gy, x, out=sink) is sink and torch.equal(sink, dw)


@pytest.mark.parametrize("M,N,K", [(1, 256, 256), (17, 128, 256), (1000, 256, 128), (4099, 512, 768), (33, 768, 512), (31, 256, 256), (32, 256, 256)])
def test_wgrad_img_ragged_rows_vs_fp64(M, N, K):
    """Row counts that do not fill a 32-row block, and rows spread over twelve decades (the per-row exponents carry them; rows far below
    the range's largest lose low-order bits of their low piece, which the bound -- a few ulps of the sum of the terms' magnitudes --
    has room for)."""
    g = torch.Generator().manual_seed(M + N)
    gy = (torch.randn(M, N, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float())).cuda()
    x = (torch.randn(M, K, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float())).cuda()
    dw = _wgrad(gy, x)
    ref = gy.double().t().mm(x.double())
    bound = gy.double().abs().t().mm(x.double().abs()) * (M ** 0.5 + 8) * 2.0 ** -24 + 1e-30
    assert ((dw.double() - ref).abs() <= bound).all(), float(((dw.double() - ref).abs() / bound).max())


@pytest.mark.parametrize("N,K", [(512, 768), (256, 512), (768, 512)])
def test_wgrad_img_operand_families_vs_fp64(N, K):
    from test_gpu_wgrad import _col_worst_mantissa
    g = torch.Generator().manual_seed(N + K)
    M = 16384
    fams = {
        "1/B-scale gradient x unit-norm rows": (torch.randn(M, N, generator=g) / 1e5, torch.nn.functional.normalize(torch.randn(M, K, generator=g), dim=-1)),
        "twelve decades of row scales": (torch.randn(M, N, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float()),
                                         torch.randn(M, K, generator=g) * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float())),
        "three decades of column scales": (torch.randn(M, N, generator=g) * torch.pow(10.0, torch.randint(-3, 1, (1, N), generator=g).float()),
                                           torch.randn(M, K, generator=g) * torch.pow(10.0, torch.randint(-3, 1, (1, K), generator=g).float())),
        "five decades inside every column": (torch.randn(M, N, generator=g) * torch.pow(10.0, torch.randint(-4, 1, (M, N), generator=g).float()),
                                             torch.relu(torch.randn(M, K, generator=g))),
        "worst-case mantissas": (_col_worst_mantissa((M, N), g), _col_worst_mantissa((M, K), g)),
        "worst-case mantissas, all positive": (_col_worst_mantissa((M, N), g).abs(), _col_worst_mantissa((M, K), g).abs()),
    }
    for name, (gy, x) in fams.items():
        gy, x = gy.cuda(), x.cuda()
        dw = _wgrad(gy, x)
        ref = gy.double().t().mm(x.double())
        scale = ref.abs().max().item()
        err = (dw.double() - ref).abs().max().item() / scale
        lib = (gy.t().mm(x).double() - ref).abs().max().item() / scale
        print(f"img dW [{N},{K}], {name}: {err:.3e} (library {lib:.3e})")
        assert err <= max(lib, 2e-7), (name, err, lib)


def test_wgrad_img_zero_rows_zero_operands_and_run_flag():
    from rqhip import ops
    g = torch.Generator().manual_seed(4)
    M, N, K = 3000, 256, 512
    gy = torch.randn(M, N, generator=g) * 1e-6
    x = torch.randn(M, K, generator=g)
    gy[100:400] = 0.0                       # zero rows must not set the range's reference exponent
    x[1000:1100] = 0.0
    gy, x = gy.cuda(), x.cuda()
    dw = _wgrad(gy, x)
    ref = gy.double().t().mm(x.double())
    assert (dw.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    assert (_wgrad(torch.zeros_like(gy), x) == 0).all()
    sink = torch.full((N, K), 7.0, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    _wgrad(gy, x, out=sink, run_flag=flag)
    assert (sink == 7.0).all()
    flag.fill_(1)
    _wgrad(gy, x, out=sink, run_flag=flag)
    assert torch.equal(sink, dw)
