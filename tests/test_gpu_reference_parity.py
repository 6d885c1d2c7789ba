"""HIP path against reference-generated ids at the BASELINE.json shapes, on every row (`-m gpu`).

Fixtures: tests/golden/parity_c2.npz (100 000 rows, 3 x 256 codes) and parity_c4.npz (300 000 rows, 4 x 1024
codes), written by oracle/gen_parity_fixtures.py from a run of the reference itself.

Tie policy (SURVEY.md section 7 "hard parts", section 8b): the kernel reports, per row and level, the relative margin
between the two smallest distances.  Where it is below tau another correct fp32 evaluation of
quantize.py:113-117 -- the reference's BLAS -- may pick the other code.  The contract checked here:
  (a) the exact-match rate against the reference is reported,
  (b) EVERY mismatching row is a flagged near-tie at its first differing level, and is adjudicated in fp64,
  (c) on identical input bits the HIP ids equal the oracle's on all rows (so (a)/(b) carry over from the CPU
      tests in tests/test_reference_parity_cpu.py).
"""
import numpy as np
import pytest
import torch

from oracle import rq_oracle as o
from rqhip import ops
import parity_gate as parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["c2", "c4"])
def fx(request):
    return request.param, parity.load_fixture(request.param)


def _hip_forward(x_np, cbs_np, mode, beta):
    out = ops.rq_forward(torch.from_numpy(x_np).cuda(), torch.from_numpy(cbs_np).cuda(), mode, beta, want_margin=True)
    torch.cuda.synchronize()
    return {k: (None if v is None else v.cpu().numpy()) for k, v in out._asdict().items()}


def _report(tag, what, cmp, gaps=None):
    print(f"[parity {tag}] {what}: exact {cmp['rows_total'] - cmp['mismatches']}/{cmp['rows_total']} "
          f"(rate {cmp['ids_exact_rate']:.6f}), flagged rows {cmp['rows_flagged']} at tau {cmp['tau']:g}, "
          f"mismatch margins {[float(f'{m:.2e}') for m in cmp['mismatch_margin']]}"
          + (f", fp64 gaps {[float(f'{g:.2e}') for g in gaps]}" if gaps is not None else ""))


def test_kernel_vs_reference_on_regenerable_latents_all_rows(fx):
    tag, f = fx
    z = parity.regenerable_latents(int(f["z_ids_eval"].shape[0]), float(f["z_scale"]), int(f["z_seed"]))
    assert parity.sha(z) == str(f["z_sha256"])
    got = _hip_forward(z, f["codebooks"], ops.MODE_EVAL, float(f["beta"]))
    ref = f["z_ids_eval"].astype(np.int64)
    cmp = parity.compare_ids(got["ids"].T, ref, got["tie_margin"], parity.TAU_KERNEL)
    gaps = parity.adjudicate_fp64(got["residuals"], f["codebooks"], cmp["mismatch_rows"], cmp["mismatch_level"],
                                  got["ids"].T, ref)
    _report(tag, "kernel, regenerable latents", cmp, gaps)
    assert cmp["all_mismatches_flagged"], cmp
    assert np.abs(gaps).max(initial=0.0) < 1e-5
    # and bit-identical to the oracle on all rows, margins included
    orc = o.rq_forward(z, f["codebooks"], o.MODE_EVAL, float(f["beta"]), want_margin=True)
    assert np.array_equal(got["ids"], orc["ids"])
    assert np.array_equal(got["tie_margin"].view(np.uint32), orc["tie_margin"].view(np.uint32))
    assert np.array_equal(got["loss"].view(np.uint32), orc["loss"].view(np.uint32))
    n = len(f["z_loss_eval_head"])
    ok = np.ones(n, bool)
    ok[cmp["mismatch_rows"][cmp["mismatch_rows"] < n]] = False
    np.testing.assert_allclose(got["loss"][:n][ok], f["z_loss_eval_head"][ok], rtol=1e-5, atol=1e-5)


def test_kernel_vs_reference_on_hard_rows(fx):
    """The reference's own encoder output bits for the 2048 rows with the smallest top-2 margins."""
    tag, f = fx
    got = _hip_forward(f["hard_res0"], f["codebooks"], ops.MODE_EVAL, float(f["beta"]))
    ref = parity.reference_ids(f, training=False)[f["hard_rows"]]
    cmp = parity.compare_ids(got["ids"].T, ref, got["tie_margin"], parity.TAU_KERNEL)
    gaps = parity.adjudicate_fp64(got["residuals"], f["codebooks"], cmp["mismatch_rows"], cmp["mismatch_level"],
                                  got["ids"].T, ref)
    _report(tag, "kernel, hard rows", cmp, gaps)
    assert cmp["all_mismatches_flagged"], cmp
    # the rows that differ are exactly the ones the fixture run recorded for the oracle (c2: none; c4: one)
    assert set(f["hard_rows"][cmp["mismatch_rows"]]) == set(f["oracle_mismatch_rows_eval"])
    np.testing.assert_allclose(gaps, f["oracle_mismatch_fp64_gap_eval"], rtol=1e-6)


def test_end_to_end_from_item_features(fx):
    """768-d items -> encoder GEMMs on the GPU (hipBLASLt, another summation order than the reference's MKL, so res0
    differs in the last bits) -> HIP quantisation: ids vs the reference on all rows, eval and STE training mode,
    plus the scalar losses of the full-batch training step."""
    tag, f = fx
    model = parity.build_fixture_model(f, "cuda")
    x = parity.synthetic_items(int(f["n_rows"]), int(f["x_seed"]))
    assert parity.sha(x.numpy()) == str(f["x_sha256"])
    xg = x.cuda()
    cbs = torch.from_numpy(f["codebooks"]).cuda()
    for training, mode in ((False, ops.MODE_EVAL), (True, ops.MODE_STE)):
        model.train(training)
        with torch.no_grad():
            res0 = model.encode(xg)
            sem = model.get_semantic_ids(xg, 0.2)
            out = ops.rq_forward(res0, cbs, mode, float(f["beta"]), want_margin=True, want_embs=False)
        ids = sem.sem_ids.cpu().numpy()
        assert np.array_equal(ids, out.ids.t().cpu().numpy())      # the module path IS the kernel path
        ref = parity.reference_ids(f, training)
        cmp = parity.compare_ids(ids, ref, out.tie_margin.cpu().numpy(), parity.TAU_E2E)
        gaps = parity.adjudicate_fp64(out.residuals.cpu().numpy(), f["codebooks"], cmp["mismatch_rows"],
                                      cmp["mismatch_level"], ids, ref)
        _report(tag, f"end to end, {'STE train' if training else 'eval'}", cmp, gaps)
        assert cmp["all_mismatches_flagged"], cmp
        assert cmp["mismatches"] <= parity.E2E_MISMATCH_CEILING[tag], (tag, cmp["mismatches"])
        assert np.abs(gaps).max(initial=0.0) < 1e-4
        p = "train" if training else "eval"
        n = len(f[f"loss_{p}_head"])
        ok = np.ones(n, bool)
        ok[cmp["mismatch_rows"][cmp["mismatch_rows"] < n]] = False
        np.testing.assert_allclose(sem.quantize_loss.cpu().numpy()[:n][ok], f[f"loss_{p}_head"][ok], rtol=1e-4,
                                   atol=1e-5)
    # full-batch training step: loss = mean(recon + quantize) (rqvae.py:152-154); 1e-5 absolute
    from data.schemas import SeqBatch
    model.train(True)
    with torch.no_grad():
        losses = model(SeqBatch(None, None, None, xg, None, None), 0.2)
    assert abs(float(losses.loss) - float(f["train_loss"])) < 1e-5
    assert abs(float(losses.reconstruction_loss) - float(f["train_reconstruction_loss"])) < 1e-5
    assert abs(float(losses.rqvae_loss) - float(f["train_rqvae_loss"])) < 1e-5


def test_kernel_on_reference_encoder_bits_when_host_reproduces_them(fx):
    """If this host's torch-CPU encoder reproduces the fixture run's res0 bits, the HIP kernel runs on exactly the
    reference's level-0 input: all rows, both modes; mismatches == the recorded near-ties."""
    tag, f = fx
    cpu_model = parity.build_fixture_model(f)
    x = parity.synthetic_items(int(f["n_rows"]), int(f["x_seed"]))
    with torch.no_grad():
        res0 = cpu_model.encoder.mlp(x).numpy()
    if parity.sha(res0) != str(f["res0_sha256"]):
        pytest.skip("this host's CPU GEMM does not reproduce the encoder output bits of the fixture run")
    for training, mode in ((False, ops.MODE_EVAL), (True, ops.MODE_STE)):
        p = "train" if training else "eval"
        got = _hip_forward(res0, f["codebooks"], mode, float(f["beta"]))
        ref = parity.reference_ids(f, training)
        cmp = parity.compare_ids(got["ids"].T, ref, got["tie_margin"], parity.TAU_KERNEL)
        _report(tag, f"kernel on reference res0, {p}", cmp)
        assert cmp["all_mismatches_flagged"], cmp
        assert np.array_equal(cmp["mismatch_rows"], f[f"oracle_mismatch_rows_{p}"])
