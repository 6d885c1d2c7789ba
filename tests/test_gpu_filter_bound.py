"""The filtered scan's error bound against the real matrix instruction (GPU half of tests/test_filter_bound.py).

rqhip_filter_scores writes out the approximate scores exactly as `rq_forward_kernel<.., FILT>` forms them (same staging,
split and v_mfma_f32_32x32x16_bf16 chain).  Checked here, on operands built to hurt:
  * assumption (H) of the proof: the chain's accumulation error stays inside the share the bound assigns to it
    ((3 D + 2) steps of relative error 2^-23 on the running magnitude) -- measured, with the worst ratio reported;
  * end to end: for every row, the two largest |d_oracle - (xsq - 2 score~)| sum to less than the threshold T the
    kernel tests the top-2 gap against, i.e. a row the filter keeps cannot have a different exact argmin.
"""
import numpy as np
import pytest
import torch

import test_filter_bound as fb

pytestmark = pytest.mark.gpu


def _scores(x, cb):
    from rqhip import ops
    out = ops.filter_scores(torch.from_numpy(x).cuda(), torch.from_numpy(cb).cuda())
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _operand_sets(D):
    rng = np.random.default_rng(11 + D)
    sets = {}
    sets["worst_case_mantissas"] = fb._worst_case_operands(D, n_rows=96, seed=3)
    x = (rng.standard_normal((256, D)) * 0.5).astype(np.float32)
    sets["random"] = (x, (x[rng.choice(256, 200, replace=False)] + 0.05 * rng.standard_normal((200, D))).astype(np.float32))
    # heavy cancellation: the dot product is ~1e-4 of sum |x_d c_d| (partial sums dwarf the result)
    a = rng.standard_normal((128, D)).astype(np.float32)
    c = a.copy()
    c[:, 1::2] *= -1.0
    c *= (1.0 + 1e-4 * rng.standard_normal(c.shape)).astype(np.float32)
    sets["cancellation"] = (np.abs(a), c)
    # forty decades of scale between rows, codes and features
    xs = (rng.standard_normal((128, D)) * 10.0 ** rng.integers(-15, 15, (128, 1)) * 10.0 ** rng.uniform(-4, 0, (128, D))).astype(np.float32)
    cs = (rng.standard_normal((96, D)) * 10.0 ** rng.integers(-15, 15, (96, 1)) * 10.0 ** rng.uniform(-4, 0, (96, D))).astype(np.float32)
    sets["scales"] = (xs, cs)
    return sets


@pytest.mark.parametrize("D", [32, 64])
def test_matrix_chain_accumulation_error_within_assumption(D):
    worst = {}
    for name, (x, cb) in _operand_sets(D).items():
        _, xsq, csq = fb.oracle_dist(x, cb)
        ok_rows = (xsq.astype(np.float64) * float(csq.max()) > 1e-30) & (xsq.astype(np.float64) + float(csq.max()) < 1e38)
        got = _scores(x, cb).astype(np.float64)
        xh, xl = fb.split2(x)
        ch, cl = fb.split2(cb)
        q = (-0.5 * csq).astype(np.float32).astype(np.float64)
        f = lambda a: a.astype(np.float64)  # noqa: E731
        exact = f(xh) @ f(ch).T + f(xh) @ f(cl).T + f(xl) @ f(ch).T + q[None, :]        # the split score, real arithmetic
        mag = np.abs(f(xh)) @ np.abs(f(ch)).T + np.abs(f(xh)) @ np.abs(f(cl)).T + np.abs(f(xl)) @ np.abs(f(ch)).T
        allowed = (3 * D + 2) * fb.H_U * mag + 3 * fb.H_U * np.abs(q)[None, :]
        with np.errstate(invalid="ignore", divide="ignore"):
            ratio = np.abs(got - exact) / allowed
        ratio = ratio[ok_rows]
        ratio = ratio[np.isfinite(ratio)]
        worst[name] = float(ratio.max()) if ratio.size else 0.0
    print(f"D={D}: worst accumulation error / allowance per operand set: " + ", ".join(f"{k} {v:.3f}" for k, v in worst.items()))
    assert max(worst.values()) <= 1.0, worst


@pytest.mark.parametrize("D", [32, 64])
def test_hardware_scores_keep_two_code_errors_under_the_threshold(D):
    from rqhip import ops
    c1, c2 = ops.filter_bound(D)
    for name, (x, cb) in _operand_sets(D).items():
        d, xsq, csq = fb.oracle_dist(x, cb)
        sc = _scores(x, cb).astype(np.float64)
        e = np.abs(d.astype(np.float64) - (xsq[:, None].astype(np.float64) - 2.0 * sc))
        e2 = np.sort(e, axis=1)[:, -2:].sum(axis=1)
        csqmax = np.float32(csq.max())
        with np.errstate(over="ignore"):
            T = c1 * np.sqrt(xsq * csqmax, dtype=np.float32) + c2 * (xsq + csqmax)
            ok = (xsq * csqmax > 1e-30) & (xsq + csqmax < 1e38)
        if ok.any():
            w = float(np.max(e2[ok] / T[ok]))
            print(f"D={D} {name}: worst (e_a + e_b) / T = {w:.3f}")
            assert w <= 1.0, (name, w)
