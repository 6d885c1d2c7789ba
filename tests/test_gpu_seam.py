"""The RQ <-> MLP seam kernel (csrc/rq_forward.hip: rq_seam_kernel; include/rqhip.h: rqhip_rq_seam) -- SURVEY.md section 8 row f2,
first clause: the encoder's last Linear (128 -> 32), all quantisation levels and the decoder's first Linear (32 -> 128) + ReLU in one
row-local launch (reference modules/rqvae.py:118-139,146; modules/encoder.py:25-38).

Parity, bit for bit:
  * the GEMMs against the oracle's one-FMA-chain-per-output linear layer (oracle/rq_oracle.c:rqo_linear_chain), every switch of the
    kernel (transposed weights, the ReLU backward on load / in the epilogue, row and column maxima);
  * the fused launch against the COMPOSED path -- stand-alone input GEMM (the same kernel with everything else switched off), then
    rqhip_rq_forward on its result, then the stand-alone output GEMM -- and against the oracle's quantisation of the oracle's res0:
    ids, losses, sums, norms, decoder activations identical, at the batch sizes that take the cooperative tiles (small batches, the
    partly filled last round) and the plain rounds.
"""
import numpy as np
import pytest
import torch

from oracle import rq_oracle as o
from rqhip import _lib, ops

pytestmark = pytest.mark.gpu

D, H = 32, 128


def _bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    w_in = (torch.randn(D, H, generator=g) / H ** 0.5).cuda()
    w_out = (torch.randn(H, D, generator=g) / D ** 0.5).cuda()
    return w_in, w_out


def _hidden(B, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.relu(torch.randn(B, H, generator=g)).cuda()        # what the encoder's ReLU hands over


def _codebooks(L, K, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randn(K, D, generator=g) * (0.35 / (l + 1)) for l in range(L)]).cuda()


@pytest.mark.parametrize("B", [1, 31, 640, 4097, 20000])
def test_input_gemm_is_the_oracles_fma_chain(B):
    w_in, _ = _weights(1)
    h = _hidden(B, 2)
    got = ops.rq_seam(h=h, w_in=w_in).res0
    want = o.linear_chain(h.cpu().numpy(), w_in.cpu().numpy())
    assert np.array_equal(_bits(got), want.view(np.uint32))
    # the weight given transposed ([128, 32], as the backward holds it): same numbers
    got_t = ops.rq_seam(h=h, w_in=w_in.t().contiguous(), w_in_transposed=True).res0
    assert torch.equal(got, got_t)
    # the ReLU backward applied on load
    g = torch.Generator().manual_seed(3)
    mask = torch.randn(B, H, generator=g).cuda()
    got_m = ops.rq_seam(h=h, w_in=w_in, h_mask=mask).res0
    want_m = o.linear_chain(h.cpu().numpy(), w_in.cpu().numpy(), xmask=mask.cpu().numpy())
    assert np.array_equal(_bits(got_m), want_m.view(np.uint32))
    assert torch.equal(got_m, ops.rq_seam(h=torch.ops.aten.threshold_backward(h, mask, 0.0), w_in=w_in).res0)


@pytest.mark.parametrize("B", [1, 33, 640, 5000])
@pytest.mark.parametrize("epi", [_lib.EPI_STORE, _lib.EPI_RELU, _lib.EPI_MASK])
def test_output_gemm_epilogues_and_maxima(B, epi):
    _, w_out = _weights(4)
    g = torch.Generator().manual_seed(5)
    rows = (torch.randn(B, D, generator=g) * 0.4).cuda()
    mask = torch.randn(B, H, generator=g).cuda() if epi == _lib.EPI_MASK else None
    cm = torch.zeros(H, dtype=torch.int32, device="cuda")
    r = ops.rq_seam(res0=rows, w_out=w_out, epilogue=epi, out_mask=mask, want_row_max=True, col_max_out=cm)
    want = o.linear_chain(rows.cpu().numpy(), w_out.cpu().numpy(), epilogue=epi, omask=None if mask is None else mask.cpu().numpy())
    assert np.array_equal(_bits(r.out), want.view(np.uint32))
    r_t = ops.rq_seam(res0=rows, w_out=w_out.t().contiguous(), w_out_transposed=True, epilogue=epi, out_mask=mask)
    assert torch.equal(r.out, r_t.out)
    # the maxima the next kernels scale by: exactly rqhip_maxima's over the stored matrix
    rm, cmx, _ = ops.maxima(r.out)
    assert torch.equal(r.out_row_max.max(dim=0).values, rm[0]) and torch.equal(cm, cmx)
    blocks = r.out.abs().view(B, H // 32, 32).amax(dim=2).t().contiguous()
    assert np.array_equal(r.out_row_max.cpu().numpy().view(np.uint32), _bits(blocks))


@pytest.mark.parametrize("mode", [ops.MODE_EVAL, ops.MODE_STE, ops.MODE_ROTATION])
@pytest.mark.parametrize("B,L,K", [(640, 3, 256), (64, 3, 256), (4097, 3, 256), (100000 // 4 + 13, 3, 256), (2000, 2, 100), (3000, 4, 128)])
def test_fused_seam_equals_composed_and_oracle(mode, B, L, K):
    assert ops.rq_seam_supported(D, H, L, K)
    w_in, w_out = _weights(6)
    h = _hidden(B, 7)
    cbs = _codebooks(L, K, 8)
    cm = torch.zeros(H, dtype=torch.int32, device="cuda")
    f = ops.rq_seam(h=h, w_in=w_in, codebooks=cbs, mode=mode, beta=0.25, w_out=w_out, epilogue=_lib.EPI_RELU, want_row_max=True,
                    col_max_out=cm)
    # composed on the device: the three pieces as separate launches
    res0 = ops.rq_seam(h=h, w_in=w_in).res0
    k = ops.rq_forward(res0, cbs, mode, 0.25, want_embs=False, want_residuals=False)
    d1 = ops.rq_seam(res0=k.emb_sum, w_out=w_out, epilogue=_lib.EPI_RELU).out
    assert torch.equal(f.res0, res0) and torch.equal(f.ids, k.ids)
    assert np.array_equal(_bits(f.loss), _bits(k.loss)) and np.array_equal(_bits(f.emb_sum), _bits(k.emb_sum))
    assert np.array_equal(_bits(f.embs_norm), _bits(k.embs_norm)) and np.array_equal(_bits(f.out), _bits(d1))
    rm, cmx, _ = ops.maxima(f.out)
    assert torch.equal(f.out_row_max.max(dim=0).values, rm[0]) and torch.equal(cm, cmx)
    # ... and the oracle end to end (smaller batches: it is a scalar C loop)
    if B <= 5000:
        r0 = o.linear_chain(h.cpu().numpy(), w_in.cpu().numpy())
        ref = o.rq_forward(r0, cbs.cpu().numpy(), mode, 0.25)
        assert np.array_equal(f.ids.cpu().numpy(), ref["ids"])
        assert np.array_equal(_bits(f.loss), ref["loss"].view(np.uint32))
        assert np.array_equal(_bits(f.emb_sum), ref["emb_sum"].view(np.uint32))
        want_d1 = o.linear_chain(ref["emb_sum"], w_out.cpu().numpy(), epilogue=1)
        assert np.array_equal(_bits(f.out), want_d1.view(np.uint32))


def test_unsupported_shapes_are_refused():
    assert not ops.rq_seam_supported(64, 128, 3, 256) and not ops.rq_seam_supported(32, 256, 3, 256)
    assert not ops.rq_seam_supported(32, 128, 4, 1024)            # BASELINE configuration 4's codebooks do not fit the LDS beside the weights
    assert ops.rq_seam_supported(32, 128, 0, 0) and ops.rq_seam_supported(32, 128, 3, 256)
    with pytest.raises(_lib.RqHipError):
        ops.rq_seam(h=torch.zeros(4, H, device="cuda"), w_in=torch.zeros(D, H, device="cuda"),
                    codebooks=torch.zeros(4, 1024, D, device="cuda"))
    with pytest.raises(_lib.RqHipError):
        ops.rq_seam(h=torch.zeros(4, H), w_in=torch.zeros(D, H))      # host tensors: no fallback


# ---- module level: RqVae.forward through the fused seam node == the same layers as separate launches ---------------------------------
def _model(mode, L=3, K=256):
    from modules.quantize import QuantizeForwardMode
    from modules.rqvae import RqVae
    torch.manual_seed(0)
    m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=K, n_layers=L, n_cat_features=0,
              codebook_kmeans_init=False, codebook_mode=mode, commitment_weight=0.25).cuda()
    with torch.no_grad():
        for l, layer in enumerate(m.layers):
            layer.embedding.weight.copy_(torch.randn(K, 32, generator=torch.Generator().manual_seed(50 + l)).cuda() * (0.3 / (l + 1)))
    return m


@pytest.mark.parametrize("B", [4096, 4500, 20001])
@pytest.mark.parametrize("mode_name", ["STE", "ROTATION_TRICK"])
def test_rqvae_forward_backward_fused_seam_equals_separate_launches(B, mode_name):
    import modules.rqvae as rqvae_mod
    from data.schemas import SeqBatch
    from modules.quantize import QuantizeForwardMode
    mode = getattr(QuantizeForwardMode, mode_name)
    g = torch.Generator().manual_seed(11)
    x = torch.nn.functional.normalize(torch.randn(B, 768, generator=g), dim=-1).cuda()
    results = []
    for fuse in (True, False):
        prev, rqvae_mod.FUSE_SEAM = rqvae_mod.FUSE_SEAM, fuse
        try:
            m = _model(mode)
            m.train()
            assert (m._seam_weights(x) is not None) == fuse
            out = m(SeqBatch(None, None, None, x, None, None), 0.2)
            out.loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
            m.eval()
            with torch.no_grad():
                ev = m(SeqBatch(None, None, None, x, None, None), 0.2)
            results.append((out, grads, ev))
        finally:
            rqvae_mod.FUSE_SEAM = prev
    (a, ga, ea), (b, gb, eb) = results
    for name in ("loss", "reconstruction_loss", "rqvae_loss", "embs_norm", "p_unique_ids"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert torch.equal(getattr(ea, name), getattr(eb, name)), "eval " + name
    assert set(ga) == set(gb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_tokenisation_and_training_forward_agree_on_the_ids():
    """get_semantic_ids (encode -> stack kernel, per-level outputs) and the fused training forward see the same res0 bits: the encoder's
    last Linear is the seam kernel's GEMM on every path."""
    m = _model(__import__("modules.quantize", fromlist=["QuantizeForwardMode"]).QuantizeForwardMode.STE)
    g = torch.Generator().manual_seed(12)
    x = torch.nn.functional.normalize(torch.randn(5000, 768, generator=g), dim=-1).cuda()
    m.eval()
    with torch.no_grad():
        sem = m.get_semantic_ids(x)
        hidden = m.encoder.run_before_tail(x)
        w_in, w_out = m._seam_weights(x)
        r = ops.rq_seam(h=hidden, w_in=w_in, codebooks=torch.stack([l.codebook() for l in m.layers]), mode=ops.MODE_EVAL, beta=0.25,
                        w_out=w_out, epilogue=_lib.EPI_RELU)
        assert torch.equal(m.encode(x), r.res0)
        assert torch.equal(sem.sem_ids, r.ids.t()) and torch.equal(sem.quantize_loss, r.loss)
        # the decoder's first Linear + ReLU through the module stack (the stand-alone launch of the same GEMM) on the stack kernel's sum
        k = ops.rq_forward(m.encode(x), torch.stack([l.codebook() for l in m.layers]), ops.MODE_EVAL, 0.25, want_embs=False, want_residuals=False)
        assert torch.equal(m.decoder._run(k.emb_sum, list(m.decoder.mlp)[:2]), r.out)
