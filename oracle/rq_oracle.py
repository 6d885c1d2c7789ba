"""ctypes/numpy binding of the CPU oracle (oracle/rq_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of rq_oracle.c.  Importable from tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg; the product package
(``rq-vae-recommender_amd/``) never imports it.

Every function takes/returns numpy arrays (fp32 / int64, C-contiguous) and mirrors one entry point
of the C file; the reference lines each restates are cited there.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librq_oracle.so")

MODE_EVAL, MODE_STE, MODE_ROTATION, MODE_GUMBEL = 0, 1, 2, 3


def build(force: bool = False) -> str:
    """Compile librq_oracle.so with the committed Makefile (gcc only, no third-party deps)."""
    src = os.path.join(_HERE, "rq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.rqo_kmeans_shift.restype = C.c_float
        _lib.rqo_sumsq2.restype = C.c_float
        _lib.rqo_count_rows_without_later_duplicate.restype = C.c_int64
    return _lib


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def rq_forward(res0, codebooks, mode: int, beta: float = 0.25, want_margin: bool = False):
    """L chained quantisation levels.  Returns dict(ids [L,B], embs [L,B,D], residuals [L,B,D],
    emb_sum [B,D], loss [B], embs_norm [B,L]) and, with want_margin, tie_margin [L,B] (see rq_oracle.c)."""
    res0, codebooks = _f(res0), _f(codebooks)
    B, D = res0.shape
    L, K, D2 = codebooks.shape
    assert D == D2
    out = dict(ids=np.empty((L, B), np.int64), embs=np.empty((L, B, D), np.float32),
               residuals=np.empty((L, B, D), np.float32), emb_sum=np.empty((B, D), np.float32),
               loss=np.empty((B,), np.float32), embs_norm=np.empty((B, L), np.float32))
    if want_margin:
        out["tie_margin"] = np.empty((L, B), np.float32)
    rc = lib().rqo_rq_forward_ex(_p(res0), C.c_int64(B), C.c_int(D), _p(codebooks), C.c_int(L), C.c_int(K),
                                 C.c_int(mode), C.c_float(beta), _p(out["ids"]), _p(out["embs"]),
                                 _p(out["residuals"]), _p(out["emb_sum"]), _p(out["loss"]),
                                 _p(out["embs_norm"]), _p(out.get("tie_margin")))
    _chk(rc, "rq_forward")
    return out


def rq_backward(res0, codebooks, mode: int, beta: float, ids, g_embs=None, g_embsum=None,
                g_resid=None, g_loss=None, order=None):
    """Closed-form backward of rq_forward.  Returns (g_res0 [B,D], g_codebooks [L,K,D]).
    order=None: codebook gradients summed over rows in ascending order; order=(n_wg, units_per_wg, unit_rows): in the
    fixed order of the HIP kernel with that geometry (rqhip_rq_backward_plan), for bit-exact comparison."""
    res0, codebooks = _f(res0), _f(codebooks)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, D = res0.shape
    L, K, _ = codebooks.shape
    ge = None if g_embs is None else _f(g_embs)
    gs = None if g_embsum is None else _f(g_embsum)
    gr = None if g_resid is None else _f(g_resid)
    gl = None if g_loss is None else _f(g_loss)
    g_res0 = np.empty((B, D), np.float32)
    g_cb = np.empty((L, K, D), np.float32)
    args = (_p(res0), C.c_int64(B), C.c_int(D), _p(codebooks), C.c_int(L), C.c_int(K), C.c_int(mode), C.c_float(beta),
            _p(ids), _p(ge), _p(gs), _p(gr), _p(gl), _p(g_res0), _p(g_cb))
    if order is None:
        rc = lib().rqo_rq_backward(*args)
    else:
        rc = lib().rqo_rq_backward_ordered(*args, C.c_int(int(order[0])), C.c_int(int(order[1])), C.c_int(int(order[2])))
    _chk(rc, "rq_backward")
    return g_res0, g_cb


def gumbel_forward(x, cb, U, temperature: float, beta: float = 0.25):
    x, cb, U = _f(x), _f(cb), _f(U)
    B, D = x.shape
    K = cb.shape[0]
    out = dict(ids=np.empty((B,), np.int64), emb=np.empty((B, D), np.float32),
               loss=np.empty((B,), np.float32), weights=np.empty((B, K), np.float32))
    rc = lib().rqo_gumbel_forward(_p(x), C.c_int64(B), C.c_int(D), _p(cb), C.c_int(K), _p(U),
                                  C.c_float(temperature), C.c_float(beta), _p(out["ids"]), _p(out["emb"]),
                                  _p(out["loss"]), _p(out["weights"]))
    _chk(rc, "gumbel_forward")
    return out


def gumbel_backward(x, cb, U, temperature: float, beta: float, g_emb=None, g_loss=None):
    x, cb, U = _f(x), _f(cb), _f(U)
    B, D = x.shape
    K = cb.shape[0]
    ge = None if g_emb is None else _f(g_emb)
    gl = None if g_loss is None else _f(g_loss)
    g_x = np.empty((B, D), np.float32)
    g_cb = np.empty((K, D), np.float32)
    rc = lib().rqo_gumbel_backward(_p(x), C.c_int64(B), C.c_int(D), _p(cb), C.c_int(K), _p(U),
                                   C.c_float(temperature), C.c_float(beta), _p(ge), _p(gl), _p(g_x), _p(g_cb))
    _chk(rc, "gumbel_backward")
    return g_x, g_cb


def kmeans_assign(x, cent) -> np.ndarray:
    x, cent = _f(x), _f(cent)
    B, D = x.shape
    K = cent.shape[0]
    a = np.empty((B,), np.int64)
    _chk(lib().rqo_kmeans_assign(_p(x), C.c_int64(B), C.c_int(D), _p(cent), C.c_int(K), _p(a)), "kmeans_assign")
    return a


def kmeans_update(x, assign, cent):
    """In-place centroid update; returns counts [K] (0 => cluster empty, centroid untouched)."""
    x = _f(x)
    assert cent.dtype == np.float32 and cent.flags.c_contiguous
    assign = np.ascontiguousarray(assign, dtype=np.int64)
    B, D = x.shape
    K = cent.shape[0]
    counts = np.empty((K,), np.int64)
    _chk(lib().rqo_kmeans_update(_p(x), C.c_int64(B), C.c_int(D), _p(assign), C.c_int(K), _p(cent), _p(counts)),
         "kmeans_update")
    return counts


def kmeans_shift(cent, old) -> float:
    cent, old = _f(cent), _f(old)
    K, D = cent.shape
    return float(lib().rqo_kmeans_shift(_p(cent), _p(old), C.c_int(K), C.c_int(D)))


def kmeans_run(x, init_idx, reseed_draws=None, max_iters=None, stop_threshold: float = 1e-10):
    """Lloyd loop of init/kmeans.py:61-72 given the rows np.random.choice picked (init_idx) and a
    callable ``reseed_draws() -> int`` standing in for torch.randint(0, B, (1,)) (kmeans.py:53).
    Returns (centroids [K,D], assignment [B], n_update_calls)."""
    x = _f(x)
    cent = x[np.asarray(init_idx)].copy()
    K = cent.shape[0]
    assign = None
    i = 0
    calls = 0
    while max_iters is None or i < max_iters:
        old = cent.copy()
        assign = kmeans_assign(x, cent)
        counts = kmeans_update(x, assign, cent)
        calls += 1
        for k in range(K):
            if counts[k] == 0:
                if x.shape[0] == 0:
                    raise ValueError("Can not choose random element from x, x is empty")
                cent[k] = x[int(reseed_draws())]
        if kmeans_shift(cent, old) < stop_threshold:
            break
        i += 1
    return cent, assign, calls


def dedup_rank(ids) -> np.ndarray:
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    L, B = ids.shape
    r = np.empty((B,), np.int64)
    _chk(lib().rqo_dedup_rank(_p(ids), C.c_int64(B), C.c_int(L), _p(r)), "dedup_rank")
    return r


def count_rows_without_later_duplicate(ids) -> int:
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    L, B = ids.shape
    return int(lib().rqo_count_rows_without_later_duplicate(_p(ids), C.c_int64(B), C.c_int(L)))


def prefix_valid(corpus, prefix) -> np.ndarray:
    """modules/model.py:169-182: which prefix rows [P,h] occur as the first h ids of a corpus row [N,H]."""
    corpus = np.ascontiguousarray(corpus, dtype=np.int64)
    prefix = np.ascontiguousarray(prefix, dtype=np.int64)
    N, H = corpus.shape
    P, h = prefix.shape
    valid = np.empty((P,), np.uint8)
    _chk(lib().rqo_prefix_valid(_p(corpus), C.c_int64(N), C.c_int(H), _p(prefix), C.c_int64(P), C.c_int(h),
                                _p(valid)), "prefix_valid")
    return valid.astype(bool)


def topk_first_match(actual, top_k) -> np.ndarray:
    """evaluate/metrics.py:16-19: first position k with top_k[b,k,:] == actual[b,:], or -1."""
    actual = np.ascontiguousarray(actual, dtype=np.int64)
    top_k = np.ascontiguousarray(top_k, dtype=np.int64)
    B, D = actual.shape
    K = top_k.shape[1]
    rank = np.empty((B,), np.int64)
    _chk(lib().rqo_topk_first_match(_p(actual), _p(top_k), C.c_int64(B), C.c_int(K), C.c_int(D), _p(rank)),
         "topk_first_match")
    return rank


def topk_metrics(rank, ks=(1, 5, 10)) -> dict:
    """evaluate/metrics.py:19-25 + reduce(): ndcg and hit rates from first-match positions (-1 = no match)."""
    rank = np.asarray(rank, np.int64)
    hit = rank >= 0
    gain = (np.float32(1.0) / np.log2(rank[hit].astype(np.float32) + np.float32(2.0))).astype(np.float32)
    out = {"ndcg": float(gain.sum(dtype=np.float64)) / len(rank)}
    for k in ks:
        out[f"h@{k}"] = float(np.count_nonzero(rank[hit] < k)) / len(rank)
    return out


def recon_loss(x_hat, x) -> np.ndarray:
    """Row-wise sum of squared differences in the kernel's fixed order (modules/loss.py:5-10)."""
    x_hat, x = _f(x_hat), _f(x)
    B, N = x.shape
    out = np.empty((B,), np.float32)
    _chk(lib().rqo_recon_loss(_p(x_hat), _p(x), C.c_int64(B), C.c_int(N), _p(out)), "recon_loss")
    return out


def linear_wgrad(g, y, x, msplit: int):
    """dW = (g * (y > 0))^T x in csrc/wgrad.hip's summation order (see rq_oracle.c).  Returns (dW [N,K], g_pre [M,N])."""
    g, x = _f(g), _f(x)
    y = None if y is None else _f(y)
    M, N = g.shape
    K = x.shape[1]
    dw = np.empty((N, K), np.float32)
    gm = np.empty((M, N), np.float32)
    rc = lib().rqo_linear_wgrad(_p(g), _p(y), _p(x), C.c_int64(M), C.c_int(N), C.c_int(K), C.c_int(msplit), _p(gm),
                                _p(dw))
    _chk(rc, "linear_wgrad")
    return dw, gm


def linear_chain(x, w, transposed: bool = False, epilogue: int = 0, xmask=None, omask=None) -> np.ndarray:
    """out = epilogue(x' . W^T) with every output ONE fp32 FMA chain over the input features (rqo_linear_chain: the seam's GEMMs).
    w: [n_out, n_in], or [n_in, n_out] with `transposed`; epilogue 0 store / 1 relu / 3 mask by omask > 0; xmask: x kept where > 0."""
    x, w = _f(x), _f(w)
    B, n_in = x.shape
    n_out = w.shape[1] if transposed else w.shape[0]
    assert (w.shape[0] if transposed else w.shape[1]) == n_in
    xm = None if xmask is None else _f(xmask)
    om = None if omask is None else _f(omask)
    out = np.empty((B, n_out), dtype=np.float32)
    _chk(lib().rqo_linear_chain(_p(x), _p(xm), C.c_int64(B), C.c_int(n_in), _p(w), C.c_int(n_out), C.c_int(1 if transposed else 0),
                                C.c_int(int(epilogue)), _p(om), _p(out)), "linear_chain")
    return out


def linear_small(a, w, w_kn: bool = False, waves: int = 4, epilogue: int = 0, aux=None) -> np.ndarray:
    """out = epilogue(a . B) as csrc/mlp_small.hip sums it (rqo_linear_small): `waves` partial fp32 FMA chains over contiguous ranges of
    32-term groups, group order 0 8 16 24 1 9 17 25 ..., partials added in wave order.  w: [n_out, n_red], or [n_red, n_out] with `w_kn`;
    epilogue 0 store / 1 relu / 3 mask by aux > 0."""
    a, w = _f(a), _f(w)
    M, n_red = a.shape
    n_out = w.shape[1] if w_kn else w.shape[0]
    assert (w.shape[0] if w_kn else w.shape[1]) == n_red
    ax = None if aux is None else _f(aux)
    out = np.empty((M, n_out), dtype=np.float32)
    _chk(lib().rqo_linear_small(_p(a), C.c_int64(M), C.c_int(n_red), _p(w), C.c_int(n_out), C.c_int(1 if w_kn else 0), C.c_int(int(waves)),
                                C.c_int(int(epilogue)), _p(ax), _p(out)), "linear_small")
    return out
