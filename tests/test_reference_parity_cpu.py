"""The oracle against reference-generated ids at the BASELINE.json shapes (CPU; `-m "not gpu"`).

tests/golden/parity_c2.npz / parity_c4.npz were produced by oracle/gen_parity_fixtures.py, which ran the
reference itself on 100 000 (3 x 256 codes) and 300 000 (4 x 1024 codes) rows.  Here the C oracle has to
reproduce those ids on EVERY row, except rows its own tie_margin output flags as near-ties (two correct fp32
evaluations of quantize.py:113-117 may disagree there); the fixtures record which rows those were when they
were generated, and an fp64 adjudication of each.
"""
import numpy as np
import pytest
import torch

from oracle import rq_oracle as o
import parity_gate as parity


@pytest.fixture(scope="module", params=["c2", "c4"])
def fx(request):
    return request.param, parity.load_fixture(request.param)


def test_fixture_inputs_regenerate_bit_for_bit(fx):
    tag, f = fx
    z = parity.regenerable_latents(int(f["z_ids_eval"].shape[0]), float(f["z_scale"]), int(f["z_seed"]))
    assert parity.sha(z) == str(f["z_sha256"])
    if tag == "c2":   # the 768-d items and the seed-0 weights (same generator for c4, three times the rows)
        x = parity.synthetic_items(int(f["n_rows"]), int(f["x_seed"]))
        assert parity.sha(x.numpy()) == str(f["x_sha256"])
    parity.build_fixture_model(f)   # raises unless our RqVae's seed-0 construction equals the reference's


def test_oracle_matches_reference_on_regenerable_latents(fx):
    """All 100 000 rows of latents any host can regenerate: the reference's level loop ran on exactly these bits."""
    tag, f = fx
    z = parity.regenerable_latents(int(f["z_ids_eval"].shape[0]), float(f["z_scale"]), int(f["z_seed"]))
    out = o.rq_forward(z, f["codebooks"], o.MODE_EVAL, float(f["beta"]), want_margin=True)
    cmp = parity.compare_ids(out["ids"].T, f["z_ids_eval"].astype(np.int64), out["tie_margin"], parity.TAU_KERNEL)
    assert cmp["all_mismatches_flagged"], cmp
    assert np.array_equal(cmp["mismatch_rows"], f["z_oracle_mismatch_rows"])
    n = len(f["z_loss_eval_head"])
    ok = np.ones(n, bool)
    ok[cmp["mismatch_rows"][cmp["mismatch_rows"] < n]] = False
    np.testing.assert_allclose(out["loss"][:n][ok], f["z_loss_eval_head"][ok], rtol=1e-5, atol=1e-5)


def test_oracle_matches_reference_on_hard_rows(fx):
    """The 2048 rows of the reference's own encoder output with the smallest top-2 margins."""
    tag, f = fx
    out = o.rq_forward(f["hard_res0"], f["codebooks"], o.MODE_EVAL, float(f["beta"]), want_margin=True)
    ref = parity.reference_ids(f, training=False)[f["hard_rows"]]
    cmp = parity.compare_ids(out["ids"].T, ref, out["tie_margin"], parity.TAU_KERNEL)
    assert cmp["all_mismatches_flagged"], cmp
    assert set(f["hard_rows"][cmp["mismatch_rows"]]) == set(f["oracle_mismatch_rows_eval"])
    assert cmp["rows_flagged"] >= int(f["oracle_flagged_rows_eval"])   # every flagged row is among the hard rows


def test_oracle_matches_reference_end_to_end_when_host_reproduces_encoder(fx):
    """From the 768-d items: needs this host's torch-CPU GEMMs to reproduce the reference run's encoder bits
    (sha256 of res0 recorded in the fixture); skipped otherwise -- the two tests above do not depend on it."""
    tag, f = fx
    model = parity.build_fixture_model(f)
    x = parity.synthetic_items(int(f["n_rows"]), int(f["x_seed"]))
    with torch.no_grad():
        res0 = model.encoder.mlp(x).numpy()
    if parity.sha(res0) != str(f["res0_sha256"]):
        pytest.skip("this host's CPU GEMM does not reproduce the encoder output bits of the fixture run")
    for training, mode in ((False, o.MODE_EVAL), (True, o.MODE_STE)):
        p = "train" if training else "eval"
        out = o.rq_forward(res0, f["codebooks"], mode, float(f["beta"]), want_margin=True)
        ref = parity.reference_ids(f, training)
        cmp = parity.compare_ids(out["ids"].T, ref, out["tie_margin"], parity.TAU_KERNEL)
        assert cmp["all_mismatches_flagged"], cmp
        assert np.array_equal(cmp["mismatch_rows"], f[f"oracle_mismatch_rows_{p}"])
        gaps = parity.adjudicate_fp64(out["residuals"], f["codebooks"], cmp["mismatch_rows"], cmp["mismatch_level"],
                                      out["ids"].T, ref)
        np.testing.assert_allclose(gaps, f[f"oracle_mismatch_fp64_gap_{p}"], rtol=1e-6)
        assert np.abs(gaps).max(initial=0.0) < 1e-5          # genuinely sub-ulp ties in exact arithmetic too
        n = len(f[f"loss_{p}_head"])
        np.testing.assert_allclose(out["loss"][:n], f[f"loss_{p}_head"], rtol=1e-5, atol=1e-5)
