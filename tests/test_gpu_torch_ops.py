"""rqhip/torch_ops.py: the hot-path kernels as registered torch.library operators (`torch.ops.rqhip.*`).

opcheck (schema, fake tensors, autograd registration), equality with the default autograd-Function path, and a
`torch.compile(..., backend="aot_eager")` training step that traces through them -- no inductor, no Triton: the ops
ARE the HIP kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3,
              n_cat_features=0, codebook_kmeans_init=False, codebook_mode=QuantizeForwardMode.STE).cuda()
    with torch.no_grad():
        for l, layer in enumerate(m.layers):
            layer.embedding.weight.copy_(torch.randn(256, 32, device="cuda") * (0.05 / (l + 1)))
    return m.train()


def test_opcheck_registered_ops():
    from rqhip import torch_ops  # noqa: F401  (registers the ops)
    g = torch.Generator(device="cuda").manual_seed(1)
    res0 = (torch.randn(300, 32, device="cuda", generator=g) * 0.5).requires_grad_(True)
    cbs = (torch.randn(3, 256, 32, device="cuda", generator=g) * 0.3).requires_grad_(True)
    checks = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.rqhip.rq_stack.default, (res0, cbs, 1, 0.25, True), test_utils=checks)
    x = torch.randn(500, 256, device="cuda", generator=g, requires_grad=True)
    w = torch.randn(128, 256, device="cuda", generator=g, requires_grad=True)
    torch.library.opcheck(torch.ops.rqhip.linear_relu.default, (x, w), test_utils=checks)
    torch.library.opcheck(torch.ops.rqhip.linear_plain.default, (x, w), test_utils=checks)
    a = torch.randn(200, 768, device="cuda", generator=g, requires_grad=True)
    b = torch.randn(200, 768, device="cuda", generator=g)
    torch.library.opcheck(torch.ops.rqhip.recon_loss.default, (a, b), test_utils=checks)
    r = torch.rand(1000, device="cuda", generator=g, requires_grad=True)
    q = torch.rand(1000, device="cuda", generator=g, requires_grad=True)
    torch.library.opcheck(torch.ops.rqhip.loss_means.default, (r, q), test_utils=checks)


def _step(m, x):
    from data.schemas import SeqBatch
    for p in m.parameters():
        p.grad = None
    out = m(SeqBatch(None, None, None, x, None, None), 0.2)
    out.loss.backward()
    return out, [p.grad.clone() for p in m.parameters()]


def test_registered_ops_equal_the_function_path_and_compile_traces_them():
    from rqhip import torch_ops
    m = _model()
    x = torch.nn.functional.normalize(torch.randn(5000, 768, device="cuda"), dim=-1)
    from rqhip import linear
    batched_out, batched_g = _step(m, x)        # the function path as shipped: its weight gradients in batched launches (csrc/wgrad_split.hip)
    before = linear.use_wgrad_batch(False)      # ... and with one launch per layer, the operators' own form: the same arithmetic, bit for bit
    try:
        ref_out, ref_g = _step(m, x)
    finally:
        linear.use_wgrad_batch(before)
    assert torch.equal(batched_out.loss, ref_out.loss)
    for a, b in zip(batched_g, ref_g):          # (another number of row ranges: another balanced tree over the partial blocks)
        assert (a - b).abs().max().item() <= 4e-6 * b.abs().max().item() + 1e-12
    torch_ops.enable(True)
    try:
        out, g = _step(m, x)
        # (the two paths form the reconstruction loss rows in different kernels -- fused GEMM epilogue vs decoder + loss kernel: the rows
        # agree to rounding, their batch mean to an ulp or two; everything that is the same arithmetic is compared bit for bit)
        assert abs(float(out.loss) - float(ref_out.loss)) <= 3e-7 * abs(float(ref_out.loss))
        assert torch.equal(out.rqvae_loss, ref_out.rqvae_loss) and torch.equal(out.p_unique_ids, ref_out.p_unique_ids)
        for a, b in zip(g, ref_g):
            assert torch.equal(a, b)
        # a traced training step: dynamo + AOT autograd see the kernels as opaque operators
        from torch import _dynamo
        _dynamo.reset()
        compiled = torch.compile(m, backend="aot_eager")
        from data.schemas import SeqBatch
        for p in m.parameters():
            p.grad = None
        c_out = compiled(SeqBatch(None, None, None, x, None, None), 0.2)
        c_out.loss.backward()
        assert abs(float(c_out.loss) - float(ref_out.loss)) < 1e-6
        for p, b in zip(m.parameters(), ref_g):
            assert torch.allclose(p.grad, b, rtol=1e-5, atol=1e-7 * float(b.abs().max()) + 1e-12)
    finally:
        torch_ops.enable(False)
