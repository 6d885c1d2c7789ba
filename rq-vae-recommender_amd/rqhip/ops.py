"""Tensor-level wrappers over the C ABI (include/rqhip.h).

Every function takes CUDA(ROCm) fp32 tensors, allocates outputs/scratch with torch (the library never
owns memory), enqueues on torch's current stream and returns without synchronising.  CPU tensors are
rejected loudly: the HIP kernels are the product path, there is no host fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import MODE_EVAL, MODE_GUMBEL, MODE_ROTATION, MODE_STE, RqHipError, check  # noqa: F401


def _need_gpu(*tensors: Optional[Tensor]) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RqHipError(
                "rqhip ops need ROCm device tensors (got a %s tensor); the HIP kernels are the only "
                "implementation of this path -- there is no CPU fallback." % t.device.type)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RqHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _f32c(t: Optional[Tensor], name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RqHipError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current stream's handle.  `torch.cuda.current_stream().cuda_stream` builds a Stream object per call (11 us of host time, five
    times per eager small-batch step: tools/eager_host_profile.py); the raw getter behind it returns the same handle in well under 1 us."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


class RqForwardOut(NamedTuple):
    ids: Tensor                  # [L,B] int64
    embs: Optional[Tensor]       # [L,B,D]
    residuals: Optional[Tensor]  # [L,B,D]
    emb_sum: Optional[Tensor]    # [B,D]
    loss: Optional[Tensor]       # [B]
    embs_norm: Optional[Tensor]  # [B,L]
    tie_margin: Optional[Tensor] = None  # [L,B] relative top-2 distance margin of each level's argmin


TIE_TAU = 1e-6  # rows with tie_margin below this are near-ties: another correct fp32 evaluation may pick the other code


_SCAN_FLAGS = {"auto": 0, "fp32": _lib.FWD_SCAN_FP32, "valu": _lib.FWD_SCAN_VALU}


def filter_bound(D: int = 32):
    """(c1, c2) of the filtered scan's too-close-to-call threshold at embedding width D (rqhip_filter_bound_d; host-side, no
    GPU needed): c1 = 2^-12 at D = 32, 2^-11 at D = 64."""
    import ctypes as C
    c1, c2 = C.c_float(0), C.c_float(0)
    _lib.lib().rqhip_filter_bound_d(int(D), C.byref(c1), C.byref(c2))
    return c1.value, c2.value


def filter_scores(x: Tensor, codebook: Tensor) -> Tensor:
    """The approximate scores x.c_k - |c_k|^2/2 the filtered scan ranks by (rqhip_filter_scores; test hook of the error
    bound).  x [B,D], codebook [K,D], D = 32 or 64 -> [B,K] fp32."""
    _need_gpu(x, codebook)
    x, codebook = _f32c(x, "x"), _f32c(codebook, "codebook")
    B, D = x.shape
    K = codebook.shape[0]
    with torch.cuda.device(x.device):
        l = _lib.lib()
        out = torch.empty((B, K), dtype=torch.float32, device=x.device)
        wsb = l.rqhip_rq_forward_workspace_bytes(1, K)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
        check(l.rqhip_filter_scores(_ptr(x), B, D, _ptr(codebook), K, _ptr(out), _ptr(ws), wsb, _stream()),
              "rqhip_filter_scores")
    return out


class RqSeamOut(NamedTuple):
    res0: Optional[Tensor]       # [B,D]   the input GEMM's result (h given)
    ids: Optional[Tensor]        # [L,B] int64
    emb_sum: Optional[Tensor]    # [B,D]
    loss: Optional[Tensor]       # [B]
    embs_norm: Optional[Tensor]  # [B,L]
    out: Optional[Tensor]        # [B,H]   the output GEMM's result
    out_row_max: Optional[Tensor]   # [H/32, B] int32
    out_col_max: Optional[Tensor]   # [H] int32 (the caller's zeroed buffer)


SEAM_H = 128


def linear_small_supported(M: int, N: int, Kr: int) -> bool:
    """Does rqhip_linear_small take out [M, N] = a [M, Kr] . B (N, Kr multiples of 32)?  Host-side query."""
    return bool(_lib.lib().rqhip_linear_small_supported(int(M), int(N), int(Kr)))


def linear_small_plan(M: int, N: int, Kr: int):
    """(col_blocks, waves) rqhip_linear_small chooses for a shape: `waves` is the number of partial chains an output is summed from."""
    cb, ks = C.c_int(0), C.c_int(0)
    check(_lib.lib().rqhip_linear_small_plan(int(M), int(N), int(Kr), C.byref(cb), C.byref(ks)), "linear_small_plan")
    return cb.value, ks.value


def linear_small(a: Tensor, w: Tensor, *, w_kn: bool = False, epilogue: int = _lib.EPI_STORE, aux: Optional[Tensor] = None,
                 col_blocks: int = 0, waves: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """epilogue(a . w^T) for w [N, Kr] (the forward of a bias-free nn.Linear) or, with w_kn, epilogue(a . w) for w [Kr, N] (its data
    gradient) -- csrc/mlp_small.hip, exact fp32, for batches below 4096 rows (any M is computed correctly).  epilogue: EPI_STORE /
    EPI_RELU / EPI_MASK (out where aux > 0 else 0: the ReLU backward of the layer below).  reference modules/encoder.py:25-38."""
    _need_gpu(a, w, aux, out)
    a, w = _f32c(a, "a"), _f32c(w, "w")
    M, Kr = a.shape
    N = w.shape[1] if w_kn else w.shape[0]
    if (w.shape[0] if w_kn else w.shape[1]) != Kr:
        raise RqHipError(f"linear_small: a {tuple(a.shape)} and w {tuple(w.shape)} (w_kn={w_kn}) do not share the reduction dimension")
    if aux is not None:
        aux = _f32c(aux, "aux")
        if tuple(aux.shape) != (M, N):
            raise RqHipError(f"linear_small: aux must be [{M}, {N}], got {tuple(aux.shape)}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)

    def launch():
        check(_lib.lib().rqhip_linear_small(_ptr(a), _ptr(w), 1 if w_kn else 0, _ptr(out), M, N, Kr, int(epilogue), _ptr(aux),
                                            int(col_blocks), int(waves), _stream()), "rqhip_linear_small")
    if _RAW_DEVICE is not None and a.device.index == _RAW_DEVICE():     # (the device context manager costs more host time than the launch)
        launch()
    else:
        with torch.cuda.device(a.device):
            launch()
    return out


def rq_seam_supported(D: int, H: int, L: int, K: int) -> bool:
    """Does rqhip_rq_seam take these shapes (D = 32, H = 128, all levels resident in LDS beside the two weights)?  Host-side query."""
    return bool(_lib.lib().rqhip_rq_seam_supported(int(D), int(H), int(L), int(K)))


def rq_seam(*, h: Optional[Tensor] = None, w_in: Optional[Tensor] = None, w_in_transposed: bool = False, h_mask: Optional[Tensor] = None,
            res0: Optional[Tensor] = None, want_res0: bool = True,
            codebooks: Optional[Tensor] = None, mode: int = MODE_EVAL, beta: float = 0.25, want_emb_sum: bool = True,
            want_loss: bool = True, want_norm: bool = True,
            w_out: Optional[Tensor] = None, w_out_transposed: bool = False, epilogue: int = _lib.EPI_STORE,
            out_mask: Optional[Tensor] = None, want_row_max: bool = False, col_max_out: Optional[Tensor] = None) -> RqSeamOut:
    """The RQ <-> MLP seam in one launch (rqhip_rq_seam): [res0 = h' . w_in^T] -> [L quantisation levels] -> [out = epilogue(s . w_out^T)].
    h [B,128] (h' = h where h_mask > 0), w_in [32,128] (or [128,32] with w_in_transposed); without h the rows are `res0` [B,32];
    codebooks [L,K,32] or None (no quantisation: s = the rows); w_out [128,32] (or [32,128] transposed) or None.
    epilogue: EPI_STORE / EPI_RELU / EPI_MASK (out_mask [B,128]).  want_row_max: int32 [4,B] row maxima of `out`; col_max_out: a ZEROED
    int32 [128] that receives its column maxima."""
    _need_gpu(h, w_in, h_mask, res0, codebooks, w_out, out_mask, col_max_out)
    h, w_in, h_mask, res0 = _f32c(h, "h"), _f32c(w_in, "w_in"), _f32c(h_mask, "h_mask"), _f32c(res0, "res0")
    codebooks, w_out, out_mask = _f32c(codebooks, "codebooks"), _f32c(w_out, "w_out"), _f32c(out_mask, "out_mask")
    rows = h if h is not None else res0
    if rows is None or rows.dim() != 2:
        raise RqHipError("rq_seam: h [B,128] or res0 [B,32] must be given")
    B, dev = rows.shape[0], rows.device
    D, H = 32, SEAM_H
    L, K = (codebooks.shape[0], codebooks.shape[1]) if codebooks is not None else (0, 0)
    if h is not None and (tuple(h.shape) != (B, H) or w_in is None or tuple(w_in.shape) != ((H, D) if w_in_transposed else (D, H))
                          or (h_mask is not None and h_mask.shape != h.shape)):
        raise RqHipError(f"rq_seam: h {tuple(h.shape)} / w_in {None if w_in is None else tuple(w_in.shape)} / h_mask do not fit")
    if h is None and tuple(res0.shape) != (B, D):
        raise RqHipError(f"rq_seam: res0 must be [B,{D}]")
    if codebooks is not None and (codebooks.dim() != 3 or codebooks.shape[2] != D):
        raise RqHipError(f"rq_seam: codebooks must be [L,K,{D}]")
    if w_out is not None and tuple(w_out.shape) != ((D, H) if w_out_transposed else (H, D)):
        raise RqHipError(f"rq_seam: w_out {tuple(w_out.shape)} does not fit")
    if epilogue == _lib.EPI_MASK and (out_mask is None or tuple(out_mask.shape) != (B, H)):
        raise RqHipError("rq_seam: EPI_MASK needs out_mask [B,128]")
    if col_max_out is not None and (col_max_out.dtype != torch.int32 or col_max_out.numel() != H or not col_max_out.is_contiguous()):
        raise RqHipError("rq_seam: col_max_out must be a contiguous int32 [128] tensor (zeroed by the caller)")
    with torch.cuda.device(dev):
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        a = _lib.SeamArgs()
        a.B, a.D, a.H = B, D, H
        a.h, a.h_mask, a.w_in, a.w_in_transposed = _ptr(h), _ptr(h_mask), _ptr(w_in), int(bool(w_in_transposed))
        r0 = f(B, D) if (h is not None and want_res0) else None
        a.res0, a.res0_out = _ptr(res0) if h is None else None, _ptr(r0)
        ids = torch.empty((L, B), dtype=torch.int64, device=dev) if L > 0 else None
        es = f(B, D) if (L > 0 and want_emb_sum) else None
        loss = f(B) if (L > 0 and want_loss) else None
        norms = f(B, L) if (L > 0 and want_norm) else None
        a.codebooks, a.L, a.K, a.mode, a.beta = _ptr(codebooks), L, K, int(mode), float(beta)
        a.ids, a.emb_sum, a.loss, a.embs_norm = _ptr(ids), _ptr(es), _ptr(loss), _ptr(norms)
        out = f(B, H) if w_out is not None else None
        rmx = torch.empty((H // 32, B), dtype=torch.int32, device=dev) if (w_out is not None and want_row_max) else None
        a.w_out, a.w_out_transposed, a.out_epilogue = _ptr(w_out), int(bool(w_out_transposed)), int(epilogue)
        a.out_mask, a.out, a.out_row_max, a.out_col_max = _ptr(out_mask), _ptr(out), _ptr(rmx), _ptr(col_max_out if w_out is not None else None)
        check(_lib.lib().rqhip_rq_seam(C.byref(a), _stream()), "rqhip_rq_seam")
    return RqSeamOut(r0, ids, es, loss, norms, out, rmx, col_max_out if w_out is not None else None)


def rq_forward(res0: Tensor, codebooks: Tensor, mode: int, beta: float, *, want_embs: bool = True,
               want_residuals: bool = True, want_emb_sum: bool = True, want_loss: bool = True,
               want_norm: bool = True, want_margin: bool = False, scan: str = "auto",
               coop_tail: bool = True) -> RqForwardOut:
    """L fused quantisation levels (rqhip_rq_forward_ex).  res0 [B,D], codebooks [L,K,D].
    scan: "auto" (the filtered bf16-split scan where it applies, else fp32 MFMA), "fp32" (always the fp32 MFMA scan),
    "valu" (LDS/VALU scan, D = 32 only) -- same bits from all three, for A/B timing and tests."""
    _need_gpu(res0, codebooks)
    res0, codebooks = _f32c(res0, "res0"), _f32c(codebooks, "codebooks")
    if res0.dim() != 2 or codebooks.dim() != 3 or codebooks.shape[2] != res0.shape[1]:
        raise RqHipError(f"shape mismatch: res0 {tuple(res0.shape)}, codebooks {tuple(codebooks.shape)}")
    B, D = res0.shape
    L, K, _ = codebooks.shape
    dev = res0.device
    with torch.cuda.device(dev):
        l = _lib.lib()
        ids = torch.empty((L, B), dtype=torch.int64, device=dev)
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        embs = f(L, B, D) if want_embs else None
        residuals = f(L, B, D) if want_residuals else None
        emb_sum = f(B, D) if want_emb_sum else None
        loss = f(B) if want_loss else None
        norm = f(B, L) if want_norm else None
        margin = f(L, B) if want_margin else None
        wsb = l.rqhip_rq_forward_workspace_bytes(L, K)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        flags = _SCAN_FLAGS[scan] | (0 if coop_tail else _lib.FWD_NO_COOP_TAIL)
        rc = l.rqhip_rq_forward_ex(_ptr(res0), B, D, _ptr(codebooks), L, K, mode, beta, _ptr(ids), _ptr(embs),
                                   _ptr(residuals), _ptr(emb_sum), _ptr(loss), _ptr(norm), _ptr(margin), _ptr(ws), wsb,
                                   flags, _stream())
        check(rc, "rqhip_rq_forward_ex")
    return RqForwardOut(ids, embs, residuals, emb_sum, loss, norm, margin)


_CBGRAD = "ordered"


def use_cbgrad(form: str) -> str:
    """How the TRAINING path (rqhip/autograd.py, rqhip/torch_ops.py) accumulates the codebook gradient where both forms exist (D = 32, STE,
    3 x <= 256 or 3-4 x 1024 codes): "ordered" (default: row order, bit-exact against oracle/rq_oracle.c:rqo_rq_backward_ordered) or
    "matrix" (round 5's experiment, kept selectable: a one-hot matrix product on the bf16 matrix cores, three exact pieces per value, the
    sum's order is the matrix pipe's -- correct to the ordered kernel's own error level and MEASURED SLOWER: 38 vs 32 us at 100 000 rows,
    242 vs 184 us at 1 M, profiles/r05_cbgrad_matrix_ab.txt).  g_res0 has the same bits either way.  Returns the previous setting."""
    global _CBGRAD
    if form not in ("matrix", "ordered"):
        raise ValueError(form)
    before, _CBGRAD = _CBGRAD, form
    return before


def cbgrad_default() -> str:
    return _CBGRAD


def rq_backward(res0: Tensor, codebooks: Tensor, mode: int, beta: float, ids: Tensor, *,
                g_embs: Optional[Tensor] = None, g_embsum: Optional[Tensor] = None,
                g_resid: Optional[Tensor] = None, g_loss: Optional[Tensor] = None,
                need_res0: bool = True, need_codebooks: bool = True, out_g_codebooks: Optional[Tensor] = None,
                cbgrad: str = "ordered"):
    """Closed-form backward of rq_forward (rqhip_rq_backward_ex) -> (g_res0 [B,D] | None, g_codebooks [L,K,D] | None).
    cbgrad: "ordered" (this function's default: the form the oracle restates bit for bit) or "matrix" (see `use_cbgrad`)."""
    _need_gpu(res0, codebooks, ids, g_embs, g_embsum, g_resid, g_loss)
    res0, codebooks = _f32c(res0, "res0"), _f32c(codebooks, "codebooks")
    g_embs, g_embsum = _f32c(g_embs, "g_embs"), _f32c(g_embsum, "g_embsum")
    g_resid, g_loss = _f32c(g_resid, "g_resid"), _f32c(g_loss, "g_loss")
    if ids.dtype != torch.int64:
        raise RqHipError("ids must be int64")
    ids = ids.contiguous()
    B, D = res0.shape
    L, K, _ = codebooks.shape
    if tuple(ids.shape) != (L, B):
        raise RqHipError(f"ids must be [L,B]=({L},{B}), got {tuple(ids.shape)}")
    dev = res0.device
    with torch.cuda.device(dev):
        l = _lib.lib()
        g_res0 = torch.empty((B, D), dtype=torch.float32, device=dev) if need_res0 else None
        g_cb = None
        if need_codebooks:
            g_cb = out_g_codebooks if out_g_codebooks is not None else torch.empty((L, K, D), dtype=torch.float32, device=dev)
            if tuple(g_cb.shape) != (L, K, D) or g_cb.dtype != torch.float32 or not g_cb.is_contiguous():
                raise RqHipError("rq_backward: out_g_codebooks must be a contiguous float32 [L,K,D] tensor")
        wsb = l.rqhip_rq_backward_workspace_bytes(B, D, L, K)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        rc = l.rqhip_rq_backward_ex(_ptr(res0), B, D, _ptr(codebooks), L, K, mode, beta, _ptr(ids), _ptr(g_embs),
                                    _ptr(g_embsum), _ptr(g_resid), _ptr(g_loss), _ptr(g_res0), _ptr(g_cb), _ptr(ws),
                                    wsb, _lib.BWD_CBGRAD_MATRIX if cbgrad == "matrix" else 0, _stream())
        check(rc, "rqhip_rq_backward_ex")
    return g_res0, g_cb


def kmeans_assign(x: Tensor, centroids: Tensor) -> Tensor:
    """assign[i] = argmin_k |x_i - c_k|^2 (rqhip_kmeans_assign; reference init/kmeans.py:40-43)."""
    _need_gpu(x, centroids)
    x, centroids = _f32c(x, "x"), _f32c(centroids, "centroids")
    B, D = x.shape
    K = centroids.shape[0]
    with torch.cuda.device(x.device):
        assign = torch.empty((B,), dtype=torch.int64, device=x.device)
        rc = _lib.lib().rqhip_kmeans_assign(_ptr(x), B, D, _ptr(centroids), K, _ptr(assign), _stream())
        check(rc, "rqhip_kmeans_assign")
    return assign


def kmeans_update(x: Tensor, assign: Tensor, centroids: Tensor):
    """In-place centroid update (rqhip_kmeans_update; init/kmeans.py:44-59).  `centroids` must be a contiguous
    fp32 device tensor.  Returns (counts [K] int64, shift_sq_max [] fp32), both on the device."""
    _need_gpu(x, assign, centroids)
    x = _f32c(x, "x")
    if centroids.dtype != torch.float32 or not centroids.is_contiguous():
        raise RqHipError("centroids must be contiguous float32 (updated in place)")
    if assign.dtype != torch.int64:
        raise RqHipError("assign must be int64")
    assign = assign.contiguous()
    B, D = x.shape
    K = centroids.shape[0]
    with torch.cuda.device(x.device):
        counts = torch.empty((K,), dtype=torch.int64, device=x.device)
        shift = torch.empty((), dtype=torch.float32, device=x.device)
        rc = _lib.lib().rqhip_kmeans_update(_ptr(x), B, D, _ptr(assign), K, _ptr(centroids), _ptr(counts),
                                            _ptr(shift), _stream())
        check(rc, "rqhip_kmeans_update")
    return counts, shift


def kmeans_lloyd(x: Tensor, centroids: Tensor, assign: Tensor, counts: Tensor, state: Tensor, n_iters: int,
                 stop_threshold: float) -> None:
    """Enqueue a batch of up to n_iters Lloyd iterations (rqhip_kmeans_lloyd); nothing is synchronised.  centroids
    [K,D] fp32 (updated in place), assign [B] int64, counts [K] int64, state [4] int32 -- all device tensors the
    caller keeps across batches (see include/rqhip.h for the state words)."""
    _need_gpu(x, centroids, assign, counts, state)
    if x.dtype != torch.float32 or not x.is_contiguous() or centroids.dtype != torch.float32 or not centroids.is_contiguous():
        raise RqHipError("kmeans_lloyd: x and centroids must be contiguous float32")
    if assign.dtype != torch.int64 or counts.dtype != torch.int64 or state.dtype != torch.int32 or state.numel() < 4:
        raise RqHipError("kmeans_lloyd: assign/counts must be int64, state int32[4]")
    B, D = x.shape
    K = centroids.shape[0]
    with torch.cuda.device(x.device):
        rc = _lib.lib().rqhip_kmeans_lloyd(_ptr(x), B, D, _ptr(centroids), K, _ptr(assign), _ptr(counts), _ptr(state),
                                           int(n_iters), float(stop_threshold), _stream())
        check(rc, "rqhip_kmeans_lloyd")


def kmeans_partial_sums(x: Tensor, centroids: Tensor, assign: Tensor, sums: Tensor, state: Tensor) -> None:
    """Row-sharded Lloyd step, first half (rqhip_kmeans_partial_sums): assign this rank's rows and write their
    per-cluster sums and counts to sums [K, D+1]; the caller all-reduces `sums`."""
    _need_gpu(x, centroids, assign, sums, state)
    B, D = x.shape
    K = centroids.shape[0]
    with torch.cuda.device(centroids.device):
        rc = _lib.lib().rqhip_kmeans_partial_sums(_ptr(x), B, D, _ptr(centroids), K, _ptr(assign), _ptr(sums),
                                                  _ptr(state), _stream())
        check(rc, "rqhip_kmeans_partial_sums")


def kmeans_apply_sums(sums: Tensor, centroids: Tensor, counts: Tensor, state: Tensor, stop_threshold: float) -> None:
    """Row-sharded Lloyd step, second half (rqhip_kmeans_apply_sums): means / shift / flags from the reduced sums."""
    _need_gpu(sums, centroids, counts, state)
    K, D = centroids.shape
    with torch.cuda.device(centroids.device):
        rc = _lib.lib().rqhip_kmeans_apply_sums(_ptr(sums), K, D, _ptr(centroids), _ptr(counts), _ptr(state),
                                                float(stop_threshold), _stream())
        check(rc, "rqhip_kmeans_apply_sums")


def dedup_rank(ids: Tensor, codebook_size: int = 0, *, want_rank: bool = True):
    """Duplicate statistics of semantic-id tuples (rqhip_dedup_rank).  ids [L,B] int64 ->
    (rank [B] int64 | None, n_distinct [] int64): rank[i] = number of earlier rows with row i's tuple
    (reference modules/tokenizer/semids.py:92-108); n_distinct / B = p_unique_ids (modules/rqvae.py:159-167)."""
    _need_gpu(ids)
    if ids.dtype != torch.int64 or ids.dim() != 2:
        raise RqHipError("ids must be an int64 [L,B] tensor")
    ids = ids.contiguous()
    L, B = ids.shape
    dev = ids.device
    with torch.cuda.device(dev):
        l = _lib.lib()
        rank = torch.empty((B,), dtype=torch.int64, device=dev) if want_rank else None
        n = torch.empty((), dtype=torch.int64, device=dev)
        wsb = l.rqhip_dedup_workspace_bytes(B)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        rc = l.rqhip_dedup_rank(_ptr(ids), B, L, int(codebook_size), _ptr(rank), _ptr(n), _ptr(ws), wsb, _stream())
        check(rc, "rqhip_dedup_rank")
    return rank, n


_DEDUP_WS = {}      # (device index, B) -> workspace: one allocation per batch size instead of one per step


def unique_fraction(ids: Tensor) -> Tensor:
    """p_unique_ids of reference modules/rqvae.py:159-167 for ids [L,B] int64 (B >= 1): the fraction of rows without a later duplicate,
    a device fp32 scalar with torch's `int64 tensor / int` rounding -- one fill + one kernel (rqhip_unique_fraction)."""
    _need_gpu(ids)
    if ids.dtype != torch.int64 or ids.dim() != 2 or ids.shape[1] < 1:
        raise RqHipError("ids must be an int64 [L,B] tensor with B >= 1")
    ids = ids.contiguous()
    L, B = ids.shape
    dev = ids.device
    with torch.cuda.device(dev):
        l = _lib.lib()
        out = torch.empty((), dtype=torch.float32, device=dev)
        key = (dev.index, B)
        ws = _DEDUP_WS.get(key)
        if ws is None or torch.cuda.is_current_stream_capturing():
            ws = torch.empty((l.rqhip_dedup_workspace_bytes(B),), dtype=torch.uint8, device=dev)
            if not torch.cuda.is_current_stream_capturing() and len(_DEDUP_WS) < 16:
                _DEDUP_WS[key] = ws        # (stream-ordered reuse: every use starts with the fill that clears it)
        check(l.rqhip_unique_fraction(_ptr(ids), B, L, _ptr(out), _ptr(ws), ws.numel(), _stream()), "rqhip_unique_fraction")
    return out


def _i64_rows(t: Tensor, name: str) -> Tensor:
    """int64 [rows, cols] with unit column stride (a column slice of a wider table is taken as is)."""
    if t.dtype != torch.int64 or t.dim() != 2:
        raise RqHipError(f"{name} must be an int64 2-D tensor, got {t.dtype} {tuple(t.shape)}")
    cols = t.shape[1]
    if cols > 0 and (t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < cols)):
        t = t.contiguous()
    return t


def _row_stride(t: Tensor) -> int:
    return int(t.stride(0)) if t.shape[0] > 1 else max(int(t.shape[1]), 1)


def prefix_index_build(corpus: Tensor) -> Tensor:
    """Set of all prefixes of the corpus' semantic-id rows (rqhip_prefix_index_build).  corpus [N,H] int64
    (row-strided views are fine) -> opaque uint8 index tensor; keep `corpus` alive and unchanged next to it."""
    _need_gpu(corpus)
    corpus = _i64_rows(corpus, "corpus")
    N, H = corpus.shape
    dev = corpus.device
    with torch.cuda.device(dev):
        l = _lib.lib()
        nbytes = l.rqhip_prefix_index_bytes(N, H)
        if nbytes == 0:
            raise RqHipError(f"prefix_index_build: unsupported corpus shape {tuple(corpus.shape)}")
        index = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        rc = l.rqhip_prefix_index_build(_ptr(corpus), N, H, _row_stride(corpus), _ptr(index), nbytes, _stream())
        check(rc, "rqhip_prefix_index_build")
    return index


def prefix_lookup(index: Tensor, corpus: Tensor, prefix: Tensor) -> Tensor:
    """valid[q] = prefix[q] occurs as the first h ids of some corpus row (rqhip_prefix_lookup); reference
    modules/model.py:169-182.  prefix [P,h] int64 -> bool [P]."""
    _need_gpu(index, corpus, prefix)
    corpus = _i64_rows(corpus, "corpus")
    prefix = _i64_rows(prefix, "prefix")
    N, H = corpus.shape
    P, h = prefix.shape
    dev = corpus.device
    with torch.cuda.device(dev):
        valid = torch.empty((P,), dtype=torch.bool, device=dev)
        rc = _lib.lib().rqhip_prefix_lookup(_ptr(index), index.numel(), _ptr(corpus), N, H, _row_stride(corpus),
                                            _ptr(prefix), P, h, _row_stride(prefix), _ptr(valid), _stream())
        check(rc, "rqhip_prefix_lookup")
    return valid


def topk_first_match(actual: Tensor, top_k: Tensor) -> Tensor:
    """rank[b] = first k with top_k[b,k,:] == actual[b,:], else -1 (rqhip_topk_first_match); the array step of
    the reference's TopKAccumulator.accumulate (evaluate/metrics.py:16-25).  actual [B,D], top_k [B,K,D] int64."""
    _need_gpu(actual, top_k)
    if actual.dtype != torch.int64 or top_k.dtype != torch.int64 or actual.dim() != 2 or top_k.dim() != 3:
        raise RqHipError("topk_first_match: actual [B,D] and top_k [B,K,D] must be int64")
    B, D = actual.shape
    if top_k.shape[0] != B or top_k.shape[2] != D:
        raise RqHipError(f"topk_first_match: top_k {tuple(top_k.shape)} does not match actual {tuple(actual.shape)}")
    K = top_k.shape[1]
    actual, top_k = actual.contiguous(), top_k.contiguous()
    dev = actual.device
    with torch.cuda.device(dev):
        rank = torch.empty((B,), dtype=torch.int64, device=dev)
        rc = _lib.lib().rqhip_topk_first_match(_ptr(actual), _ptr(top_k), B, K, D, _ptr(rank), _stream())
        check(rc, "rqhip_topk_first_match")
    return rank


def gumbel_matrix_path_min_rows(set_to: int = 0) -> int:
    """Query (set_to <= 0) or set the batch size from which the Gumbel level runs on the matrix instructions
    (rqhip_gumbel_matrix_path_min_rows); returns the previous value."""
    return int(_lib.lib().rqhip_gumbel_matrix_path_min_rows(int(set_to)))


def gumbel_forward(x: Tensor, codebook: Tensor, U: Tensor, temperature: float, beta: float):
    """One GUMBEL_SOFTMAX level, training (rqhip_gumbel_forward) -> (ids [B], emb [B,D], loss [B])."""
    _need_gpu(x, codebook, U)
    x, codebook, U = _f32c(x, "x"), _f32c(codebook, "codebook"), _f32c(U, "U")
    B, D = x.shape
    K = codebook.shape[0]
    if tuple(U.shape) != (B, K):
        raise RqHipError(f"U must be [B,K]=({B},{K}), got {tuple(U.shape)}")
    dev = x.device
    with torch.cuda.device(dev):
        ids = torch.empty((B,), dtype=torch.int64, device=dev)
        emb = torch.empty((B, D), dtype=torch.float32, device=dev)
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        rc = _lib.lib().rqhip_gumbel_forward(_ptr(x), B, D, _ptr(codebook), K, _ptr(U), temperature, beta, _ptr(ids),
                                             _ptr(emb), _ptr(loss), _stream())
        check(rc, "rqhip_gumbel_forward")
    return ids, emb, loss


def gumbel_backward(x: Tensor, codebook: Tensor, U: Tensor, temperature: float, beta: float, *,
                    g_emb: Optional[Tensor] = None, g_loss: Optional[Tensor] = None):
    """Backward of gumbel_forward (rqhip_gumbel_backward) -> (g_x [B,D], g_codebook [K,D])."""
    _need_gpu(x, codebook, U, g_emb, g_loss)
    x, codebook, U = _f32c(x, "x"), _f32c(codebook, "codebook"), _f32c(U, "U")
    g_emb, g_loss = _f32c(g_emb, "g_emb"), _f32c(g_loss, "g_loss")
    B, D = x.shape
    K = codebook.shape[0]
    dev = x.device
    with torch.cuda.device(dev):
        l = _lib.lib()
        g_x = torch.empty((B, D), dtype=torch.float32, device=dev)
        g_cb = torch.empty((K, D), dtype=torch.float32, device=dev)
        wsb = l.rqhip_gumbel_backward_workspace_bytes(B, D, K)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        rc = l.rqhip_gumbel_backward(_ptr(x), B, D, _ptr(codebook), K, _ptr(U), temperature, beta, _ptr(g_emb),
                                     _ptr(g_loss), _ptr(g_x), _ptr(g_cb), _ptr(ws), wsb, _stream())
        check(rc, "rqhip_gumbel_backward")
    return g_x, g_cb


def profile_enable(max_records: int) -> None:
    """Bench-only: time the main rq_forward kernel of every following call with HIP events on its stream."""
    check(_lib.lib().rqhip_profile_enable(int(max_records)), "rqhip_profile_enable")


def profile_read(cap: int = 4096):
    """Durations (ms) of the rq_forward kernels recorded since the last read (synchronises them)."""
    import ctypes as C
    buf = (C.c_float * cap)()
    n = C.c_int(0)
    check(_lib.lib().rqhip_profile_read(buf, cap, C.byref(n)), "rqhip_profile_read")
    return [buf[i] for i in range(n.value)]


def profile_select(*kinds: str) -> None:
    """Record only launches of these kinds (names of _lib.PROF_TAGS); no argument = every kind."""
    mask = 0
    for k in kinds:
        if k == "none":          # record nothing until the next select
            check(_lib.lib().rqhip_profile_select(0), "rqhip_profile_select")
            return
        mask |= 1 << next(t for t, n in _lib.PROF_TAGS.items() if n == k)
    check(_lib.lib().rqhip_profile_select(mask if kinds else 0xFFFFFFFF), "rqhip_profile_select")


def profile_read_tagged(cap: int = 65536):
    """Every record since the last read as (kind, ms, algorithmic flops, algorithmic bytes); kinds: _lib.PROF_TAGS."""
    import ctypes as C
    buf = (_lib.ProfileRecord * cap)()
    n = C.c_int(0)
    check(_lib.lib().rqhip_profile_read_tagged(buf, cap, C.byref(n)), "rqhip_profile_read_tagged")
    return [(_lib.PROF_TAGS.get(buf[i].tag, str(buf[i].tag)), buf[i].ms, buf[i].flops, buf[i].bytes) for i in range(n.value)]


def _rows(t: Tensor, name: str) -> Tensor:
    """[B,N] fp32 with unit column stride (row stride may exceed N: slices of a wider matrix are fine)."""
    if t.dtype != torch.float32 or t.dim() != 2:
        raise RqHipError(f"{name} must be a 2-D float32 tensor")
    return t if t.stride(1) == 1 and t.stride(0) >= t.shape[1] else t.contiguous()


def recon_loss_forward(x_hat: Tensor, x: Tensor) -> Tensor:
    """out[b] = sum_d (x_hat - x)^2 (rqhip_recon_loss_forward; reference modules/loss.py:5-10)."""
    _need_gpu(x_hat, x)
    x_hat, x = _rows(x_hat, "x_hat"), _rows(x, "x")
    if x_hat.shape != x.shape:
        raise RqHipError(f"shape mismatch {tuple(x_hat.shape)} vs {tuple(x.shape)}")
    B, N = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty((B,), dtype=torch.float32, device=x.device)
        rc = _lib.lib().rqhip_recon_loss_forward(_ptr(x_hat), x_hat.stride(0), _ptr(x), x.stride(0), B, N, _ptr(out),
                                                 _stream())
        check(rc, "rqhip_recon_loss_forward")
    return out


def recon_loss_backward(x_hat: Tensor, x: Tensor, g_out: Tensor, need_hat: bool = True, need_x: bool = False):
    _need_gpu(x_hat, x, g_out)
    x_hat, x, g_out = _rows(x_hat, "x_hat"), _rows(x, "x"), _f32c(g_out, "g_out")
    B, N = x.shape
    with torch.cuda.device(x.device):
        g_hat = torch.empty((B, N), dtype=torch.float32, device=x.device) if need_hat else None
        g_x = torch.empty((B, N), dtype=torch.float32, device=x.device) if need_x else None
        rc = _lib.lib().rqhip_recon_loss_backward(_ptr(x_hat), x_hat.stride(0), _ptr(x), x.stride(0), _ptr(g_out), B, N,
                                                  _ptr(g_hat), _ptr(g_x), _stream())
        check(rc, "rqhip_recon_loss_backward")
    return g_hat, g_x


def recon_spec_ok(x_hat: Tensor, x: Tensor) -> bool:
    """Can the speculative reconstruction-loss pair take these tensors (float4 layout)?"""
    return (x_hat.dim() == 2 and x_hat.shape == x.shape and x_hat.shape[1] % 4 == 0 and x_hat.stride(1) == 1
            and x.stride(1) == 1 and x_hat.stride(0) % 4 == 0 and x.stride(0) % 4 == 0
            and x_hat.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and x_hat.dtype == torch.float32
            and x.dtype == torch.float32)


def recon_loss_forward_spec(x_hat: Tensor, x: Tensor, row_scale: float):
    """out[b] = sum_d (x_hat - x)^2 and the expected gradient g_spec = 2 (x_hat - x) * row_scale in one pass
    (rqhip_recon_loss_forward_spec).  Returns (out [B], g_spec [B,N])."""
    _need_gpu(x_hat, x)
    B, N = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty((B,), dtype=torch.float32, device=x.device)
        g_spec = torch.empty((B, N), dtype=torch.float32, device=x.device)
        rc = _lib.lib().rqhip_recon_loss_forward_spec(_ptr(x_hat), x_hat.stride(0), _ptr(x), x.stride(0), B, N,
                                                      row_scale, _ptr(out), _ptr(g_spec), _stream())
        check(rc, "rqhip_recon_loss_forward_spec")
    return out, g_spec


def recon_loss_backward_spec(x_hat: Tensor, x: Tensor, g_out: Tensor, row_scale: float, g_spec: Tensor) -> Tensor:
    """Fix up g_spec in place for rows whose upstream gradient differs from row_scale; returns g_spec."""
    _need_gpu(x_hat, x, g_out, g_spec)
    g_out = _f32c(g_out, "g_out")
    B, N = x.shape
    with torch.cuda.device(x.device):
        rc = _lib.lib().rqhip_recon_loss_backward_spec(_ptr(x_hat), x_hat.stride(0), _ptr(x), x.stride(0), _ptr(g_out),
                                                       B, N, row_scale, _ptr(g_spec), _stream())
        check(rc, "rqhip_recon_loss_backward_spec")
    return g_spec


def loss_means_backward(g_loss: Optional[Tensor], g_recon: Optional[Tensor], g_quant: Optional[Tensor], n: int,
                        want_recon: bool = True, want_quant: bool = True):
    """Row gradients of the three means (rqhip_loss_means_backward): two dense [n] vectors (None where not wanted)."""
    ref = next((g for g in (g_loss, g_recon, g_quant) if g is not None), None)
    if ref is None or not (want_recon or want_quant):
        return None, None
    _need_gpu(ref)
    gs = [None if g is None else _f32c(g.reshape(()), "g") for g in (g_loss, g_recon, g_quant)]
    with torch.cuda.device(ref.device):
        rows_r = torch.empty((n,), dtype=torch.float32, device=ref.device) if want_recon else None
        rows_q = torch.empty((n,), dtype=torch.float32, device=ref.device) if want_quant else None
        check(_lib.lib().rqhip_loss_means_backward(_ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), n, _ptr(rows_r), _ptr(rows_q),
                                                   _stream()), "rqhip_loss_means_backward")
    return rows_r, rows_q


def loss_means(recon: Tensor, quant: Tensor) -> Tensor:
    """[3] = mean(recon + quant), mean(recon), mean(quant) in one launch (rqhip_loss_means)."""
    _need_gpu(recon, quant)
    recon, quant = _f32c(recon, "recon"), _f32c(quant, "quant")
    if recon.dim() != 1 or recon.shape != quant.shape or recon.numel() == 0:
        raise RqHipError(f"loss_means: need two non-empty [B] tensors, got {tuple(recon.shape)}, {tuple(quant.shape)}")
    with torch.cuda.device(recon.device):
        out = torch.empty((3,), dtype=torch.float32, device=recon.device)
        l = _lib.lib()
        from . import linear as _lin_mod
        key = (recon.device.index, _stream())
        # (the meeting place must have been zeroed by an EXECUTED fill: a first use under hipGraph capture takes the one-workgroup kernel)
        fresh_in_capture = key not in _LOSS_MEANS_WS and torch.cuda.is_current_stream_capturing()
        if recon.numel() <= 4096 or not _lin_mod.trims_on() or fresh_in_capture:        # one workgroup's worth: the single-workgroup kernel
            check(l.rqhip_loss_means(_ptr(recon), _ptr(quant), recon.numel(), _ptr(out), _stream()), "rqhip_loss_means")
        else:                            # many workgroups; their meeting place is zeroed once per (device, stream) and re-armed by the kernel
            ws = _LOSS_MEANS_WS.get(key)
            if ws is None:
                ws = _LOSS_MEANS_WS[key] = torch.zeros((l.rqhip_loss_means_workspace_bytes(),), dtype=torch.uint8, device=recon.device)
            check(l.rqhip_loss_means_ws(_ptr(recon), _ptr(quant), recon.numel(), _ptr(out), _ptr(ws), ws.numel(), _stream()),
                  "rqhip_loss_means_ws")
    return out


_LOSS_MEANS_WS = {}


def linear_wgrad_supported(n_out: int, n_in: int) -> bool:
    return bool(_lib.lib().rqhip_linear_wgrad_supported(int(n_out), int(n_in)))


def linear_wgrad(g: Tensor, y: Optional[Tensor], x: Tensor, *, want_masked: bool = True, out: Optional[Tensor] = None,
                 exact_fp32: bool = False, g_col_max: Optional[Tensor] = None, x_col_max: Optional[Tensor] = None):
    """dW [N,K] = (g * (y > 0))^T x with the ReLU backward fused (rqhip_linear_wgrad*); y None = layer without ReLU.
    Returns (dW, g_pre): g_pre = the masked gradient in a fresh tensor when `want_masked` and y is given (the input
    of the data-gradient GEMM that follows), g itself when there is no mask, None when not wanted.  `out`: a
    contiguous fp32 [N,K] tensor to receive dW (e.g. the parameter's slice of a flat gradient buffer).
    g_col_max / x_col_max (int32 [N] / [K], `maxima`): the two-piece fp16 arithmetic under exact column scales
    (rqhip_linear_wgrad_f16, the product path); without them the three-piece bf16 kernel of round 3.
    exact_fp32: force the fp32-MFMA kernel with the oracle-restated summation order (RQHIP_WGRAD_FP32)."""
    _need_gpu(g, y, x, g_col_max, x_col_max)
    g, x = _f32c(g, "g"), _f32c(x, "x")
    y = _f32c(y, "y")
    if g.dim() != 2 or x.dim() != 2 or g.shape[0] != x.shape[0] or (y is not None and y.shape != g.shape):
        raise RqHipError(f"linear_wgrad: shapes g {tuple(g.shape)}, x {tuple(x.shape)}")
    M, N = g.shape
    K = x.shape[1]
    f16 = g_col_max is not None or x_col_max is not None
    if f16 and (exact_fp32 or g_col_max is None or x_col_max is None or g_col_max.numel() != N or x_col_max.numel() != K
                or g_col_max.dtype != torch.int32 or x_col_max.dtype != torch.int32):
        raise RqHipError("linear_wgrad: the fp16 path needs BOTH column-maxima vectors (int32 [N] and [K]) and no exact_fp32")
    dev = g.device
    with torch.cuda.device(dev):
        l = _lib.lib()
        if out is not None and (tuple(out.shape) != (N, K) or out.dtype != torch.float32 or not out.is_contiguous()):
            raise RqHipError("linear_wgrad: `out` must be a contiguous float32 [N,K] tensor")
        dw = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
        wsb = l.rqhip_linear_wgrad_workspace_bytes(M, N, K)
        ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
        gm = torch.empty_like(g) if (want_masked and y is not None) else None
        if f16:
            rc = l.rqhip_linear_wgrad_f16(_ptr(g), _ptr(y), _ptr(x), M, N, K, _ptr(g_col_max), _ptr(x_col_max), _ptr(gm),
                                          _ptr(dw), _ptr(ws), wsb, _stream())
            check(rc, "rqhip_linear_wgrad_f16")
        else:
            rc = l.rqhip_linear_wgrad_ex(_ptr(g), _ptr(y), _ptr(x), M, N, K, _ptr(gm), _ptr(dw), _ptr(ws), wsb,
                                         _lib.WGRAD_FP32 if exact_fp32 else 0, _stream())
            check(rc, "rqhip_linear_wgrad_ex")
    return dw, (g if y is None else gm)


def linear_wgrad_f16_batch_ranges(M: int, shapes) -> int:
    """The common number of row ranges rqhip_linear_wgrad_f16_batch would cut `shapes` = [(N, K), ...] into over M rows; 0 = not batchable
    (2..4 layers, every dW tiled 256 x 256, more than 128 rows)."""
    n = len(shapes)
    if n < 2 or n > 4:
        return 0
    ci = C.c_int * n
    return int(_lib.lib().rqhip_linear_wgrad_f16_batch_plan(int(M), ci(*[int(a) for a, _ in shapes]), ci(*[int(b) for _, b in shapes]), n))


def linear_wgrad_f16_batch(jobs, outs=None):
    """The weight gradients of 2..4 layers in ONE launch (rqhip_linear_wgrad_f16_batch): `jobs` = [(g [M, N_j] already masked by the layer's
    ReLU, x [M, K_j], g_col_max int32 [N_j], x_col_max int32 [K_j]), ...], every dW tiled 256 x 256; `outs[j]`: a contiguous fp32
    [N_j, K_j] tensor to receive dW_j (or None).  Returns [dW_j]."""
    n = len(jobs)
    gs = [_f32c(j[0], "g") for j in jobs]
    xs = [_f32c(j[1], "x") for j in jobs]
    _need_gpu(*gs, *xs, *[j[2] for j in jobs], *[j[3] for j in jobs])
    M, dev = gs[0].shape[0], gs[0].device
    shapes = [(g.shape[1], x.shape[1]) for g, x in zip(gs, xs)]
    for i, (g, x, gm, xm) in enumerate(zip(gs, xs, [j[2] for j in jobs], [j[3] for j in jobs])):
        if (g.dim() != 2 or x.dim() != 2 or g.shape[0] != M or x.shape[0] != M or gm.dtype != torch.int32 or xm.dtype != torch.int32
                or gm.numel() != g.shape[1] or xm.numel() != x.shape[1] or not gm.is_contiguous() or not xm.is_contiguous()):
            raise RqHipError(f"linear_wgrad_f16_batch: job {i}: g {tuple(g.shape)}, x {tuple(x.shape)}, maxima int32 [N] / [K]; one M")
    if linear_wgrad_f16_batch_ranges(M, shapes) < 1:
        raise RqHipError(f"linear_wgrad_f16_batch: {shapes} over {M} rows is not batchable (linear_wgrad_f16_batch_ranges)")
    with torch.cuda.device(dev):
        l = _lib.lib()
        dws = []
        for i, (N, K) in enumerate(shapes):
            out = outs[i] if outs is not None else None
            if out is not None and (tuple(out.shape) != (N, K) or out.dtype != torch.float32 or not out.is_contiguous()):
                raise RqHipError("linear_wgrad_f16_batch: `outs[j]` must be a contiguous float32 [N, K] tensor")
            dws.append(out if out is not None else torch.empty((N, K), dtype=torch.float32, device=dev))
        ci = C.c_int * n
        Ns, Ks = ci(*[a for a, _ in shapes]), ci(*[b for _, b in shapes])
        wsb = l.rqhip_linear_wgrad_f16_batch_workspace_bytes(M, Ns, Ks, n)
        ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
        arr = (_lib.WgradJob * n)()
        for i in range(n):
            arr[i].g, arr[i].x, arr[i].N, arr[i].K = gs[i].data_ptr(), xs[i].data_ptr(), shapes[i][0], shapes[i][1]
            arr[i].g_col_max, arr[i].x_col_max, arr[i].dW = jobs[i][2].data_ptr(), jobs[i][3].data_ptr(), dws[i].data_ptr()
        check(l.rqhip_linear_wgrad_f16_batch(arr, n, M, _ptr(ws), wsb, _stream()), "rqhip_linear_wgrad_f16_batch")
    return dws


def linear_wgrad_jobs_supported(n_out: int, n_in: int) -> bool:
    return bool(_lib.lib().rqhip_linear_wgrad_jobs_supported(int(n_out), int(n_in)))


WGRAD_JOBS_MAX = 8


def linear_wgrad_jobs(jobs, outs=None):
    """The weight gradients of several layers in ONE launch (rqhip_linear_wgrad_jobs, csrc/wgrad_jobs.hip): `jobs` =
    [(g [M, N_i] ALREADY masked by the layer's ReLU, x [M, K_i]), ...], at most 8, one M; `outs[i]`: a contiguous fp32
    [N_i, K_i] tensor to receive dW_i (or None).  Returns [dW_i].  The small-batch path of the MLP stacks."""
    if not jobs:
        return []
    if len(jobs) > WGRAD_JOBS_MAX:
        raise RqHipError(f"linear_wgrad_jobs: at most {WGRAD_JOBS_MAX} jobs per launch (got {len(jobs)})")
    gs = [_f32c(g, "g") for g, _ in jobs]
    xs = [_f32c(x, "x") for _, x in jobs]
    _need_gpu(*gs, *xs)
    M, dev = gs[0].shape[0], gs[0].device
    dws = []
    for i, (g, x) in enumerate(zip(gs, xs)):
        if g.dim() != 2 or x.dim() != 2 or g.shape[0] != M or x.shape[0] != M or g.device != dev or x.device != dev:
            raise RqHipError(f"linear_wgrad_jobs: job {i}: g {tuple(g.shape)}, x {tuple(x.shape)}; every job has M = {M} rows")
        N, K = g.shape[1], x.shape[1]
        out = outs[i] if outs is not None else None
        if out is not None and (tuple(out.shape) != (N, K) or out.dtype != torch.float32 or not out.is_contiguous()):
            raise RqHipError("linear_wgrad_jobs: `outs[i]` must be a contiguous float32 [N,K] tensor")
        dws.append(out if out is not None else torch.empty((N, K), dtype=torch.float32, device=dev))
    if M == 0:          # no rows: every gradient is zero (and an empty tensor has no address to hand over)
        for d in dws:
            d.zero_()
        return dws
    n = len(jobs)
    vp, ci = C.c_void_p * n, C.c_int * n
    with torch.cuda.device(dev):
        rc = _lib.lib().rqhip_linear_wgrad_jobs(vp(*[_ptr(g) for g in gs]), vp(*[_ptr(x) for x in xs]), vp(*[_ptr(d) for d in dws]),
                                                ci(*[g.shape[1] for g in gs]), ci(*[x.shape[1] for x in xs]), n, M, _stream())
        check(rc, "rqhip_linear_wgrad_jobs")
    return dws


def gemm_split_supported(n_cols: int, n_red: int) -> bool:
    return bool(_lib.lib().rqhip_gemm_split_supported(int(n_cols), int(n_red)))


F16X2, BF16X3 = _lib.SPLIT_F16X2, _lib.SPLIT_BF16X3


def weight_images(jobs, arith: int = F16X2):
    """The split-GEMM images of several weight matrices in ONE launch (rqhip_weight_images): `jobs` = [(w, transpose), ...];
    of w [rows, cols] itself (forward: C = A w^T) or of w^T (`transpose`: data gradient, C = A w).  Returns one opaque
    uint8 tensor per job (views of a single allocation)."""
    if not jobs:
        return []
    ws = [_f32c(w.detach(), "w") for w, _ in jobs]
    dev = _need_gpu(*ws)
    l = _lib.lib()
    sizes = []
    for w, (_, tr) in zip(ws, jobs):
        rows, cols = w.shape
        Nc, R = (cols, rows) if tr else (rows, cols)
        nb = l.rqhip_weight_image_bytes(Nc, R, int(arith))
        if nb == 0:
            raise RqHipError(f"weight_images: unsupported shape {tuple(w.shape)} (transpose={bool(tr)})")
        sizes.append(nb)
    with torch.cuda.device(dev):
        slots = [(nb + 255) // 256 * 256 for nb in sizes]          # every image 256-byte aligned inside the arena
        arena = torch.empty((sum(slots),), dtype=torch.uint8, device=dev)
        arr = (_lib.ImageJob * len(jobs))()
        images, off = [], 0
        for i, (w, (_, tr), nb, slot) in enumerate(zip(ws, jobs, sizes, slots)):
            img = arena[off:off + nb]                               # exactly the image: no uninitialised padding in the view
            off += slot
            arr[i].w, arr[i].rows, arr[i].cols = w.data_ptr(), w.shape[0], w.shape[1]
            arr[i].transpose, arr[i].arith = int(bool(tr)), int(arith)
            arr[i].image, arr[i].image_bytes = img.data_ptr(), nb
            images.append(img)
        check(l.rqhip_weight_images(arr, len(jobs), _stream()), "rqhip_weight_images")
    return images


def weight_planes(w: Tensor, transpose: bool = False, arith: int = BF16X3) -> Tensor:
    """One image (`weight_images` with a single job; round 3's name and default arithmetic)."""
    return weight_images([(w, transpose)], arith)[0]


def maxima(a: Tensor, y: Optional[Tensor] = None, *, rows: bool = True, cols: bool = True, write_masked: bool = False):
    """(row_max [1, M] or None, col_max [R] or None, masked or None): int32 tensors holding the bit patterns of the largest
    |value| of every row / column of a [M, R] (rqhip_maxima) -- the exact power-of-two scales of the fp16 split kernels are
    derived from them.  With y: of `a` masked by y > 0 (the ReLU backward), returned as `masked` when `write_masked`."""
    _need_gpu(a, y)
    a, y = _f32c(a, "a"), _f32c(y, "y")
    M, R = a.shape
    if y is not None and y.shape != a.shape:
        raise RqHipError(f"maxima: y is {tuple(y.shape)}, expected {tuple(a.shape)}")
    with torch.cuda.device(a.device):
        rm = torch.empty((1, M), dtype=torch.int32, device=a.device) if rows else None
        cm = torch.zeros((R,), dtype=torch.int32, device=a.device) if cols else None
        out = torch.empty_like(a) if (write_masked and y is not None) else None
        check(_lib.lib().rqhip_maxima(_ptr(a), _ptr(y), _ptr(out), M, R, _ptr(rm), _ptr(cm), _stream()), "rqhip_maxima")
    return rm, cm, out


def gemm_split_ex(a: Tensor, image: Tensor, n_cols: int, *, arith: int = F16X2, epilogue: int = _lib.EPI_STORE,
                  aux: Optional[Tensor] = None, row_scale: float = 0.0, a_row_max: Optional[Tensor] = None,
                  want_row_max: bool = False, col_max_out: Optional[Tensor] = None, tile_rows: int = 0):
    """C [M, n_cols] = epilogue(a [M, R] . image^T) (rqhip_gemm_split_ex).  Returns (C, loss_rows or None, c_row_max or None):
    loss_rows for EPI_RECON; c_row_max [column tiles, M] int32 when `want_row_max`; `col_max_out` (int32 [n_cols], ZEROED by
    the caller, or a slice of a zeroed arena) receives the column maxima of C.  arith F16X2 needs `a_row_max` [parts, M]."""
    _need_gpu(a, image, aux, a_row_max, col_max_out)
    a, aux = _f32c(a, "a"), _f32c(aux, "aux")
    M, R = a.shape
    if aux is not None and tuple(aux.shape) != (M, n_cols):
        raise RqHipError(f"gemm_split: aux is {tuple(aux.shape)}, expected {(M, n_cols)}")
    if a_row_max is not None and (a_row_max.dtype != torch.int32 or a_row_max.dim() != 2 or a_row_max.shape[1] != M
                                  or not a_row_max.is_contiguous()):
        raise RqHipError("gemm_split: a_row_max must be a contiguous int32 [parts, M] tensor")
    if col_max_out is not None and (col_max_out.dtype != torch.int32 or col_max_out.numel() != n_cols
                                    or not col_max_out.is_contiguous()):
        raise RqHipError("gemm_split: col_max_out must be a contiguous int32 [n_cols] tensor")
    with torch.cuda.device(a.device):
        l = _lib.lib()
        c = torch.empty((M, n_cols), dtype=torch.float32, device=a.device)
        args = _lib.GemmArgs()
        args.A, args.M, args.R, args.image, args.Nc = a.data_ptr(), M, R, image.data_ptr(), int(n_cols)
        args.arith, args.epilogue, args.tile_rows, args.C = int(arith), int(epilogue), int(tile_rows), c.data_ptr()
        args.aux, args.row_scale = _ptr(aux), float(row_scale)
        rows = ws = None
        if epilogue == _lib.EPI_RECON:
            rows = torch.empty((M,), dtype=torch.float32, device=a.device)
            nbytes = l.rqhip_gemm_split_recon_workspace_bytes(M, int(n_cols))
            ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=a.device)
            args.loss_rows, args.workspace, args.workspace_bytes = rows.data_ptr(), ws.data_ptr(), nbytes
        if a_row_max is not None:
            args.a_row_max, args.a_row_parts = a_row_max.data_ptr(), a_row_max.shape[0]
        crm = torch.empty((n_cols // (256 if n_cols % 256 == 0 else 128), M), dtype=torch.int32, device=a.device) if want_row_max else None
        args.c_row_max, args.c_col_max = _ptr(crm), _ptr(col_max_out)
        import ctypes as C
        check(l.rqhip_gemm_split_ex(C.byref(args), _stream()), "rqhip_gemm_split_ex")
    return c, rows, crm


def gemm_split(a: Tensor, planes: Tensor, n_cols: int, relu: bool = False, tile_rows: int = 0, arith: int = BF16X3) -> Tensor:
    """C [M, n_cols] = a [M, R] . image^T with the optional ReLU epilogue.  arith BF16X3 (default here: round 3's call) or
    F16X2, for which the row maxima of `a` are computed by a `maxima` pass first (callers that chain layers use
    `gemm_split_ex` and hand the maxima over instead)."""
    rm = maxima(a, cols=False)[0] if arith == F16X2 else None
    return gemm_split_ex(a, planes, n_cols, arith=arith, epilogue=_lib.EPI_RELU if relu else _lib.EPI_STORE, a_row_max=rm,
                         tile_rows=tile_rows)[0]


def gemm_split_recon(a: Tensor, planes: Tensor, n_cols: int, x: Tensor, row_scale: float, arith: int = BF16X3):
    """The last decoder layer fused with the reconstruction loss (RQHIP_EPI_RECON): with x_hat = a . image^T (never stored)
    returns (g, loss_rows): g [M, n_cols] = (2 (x_hat - x)) * row_scale and loss_rows [M] = sum (x_hat - x)^2."""
    if tuple(x.shape) != (a.shape[0], n_cols):
        raise RqHipError(f"gemm_split_recon: x is {tuple(x.shape)}, expected {(a.shape[0], n_cols)}")
    rm = maxima(a, cols=False)[0] if arith == F16X2 else None
    g, rows, _ = gemm_split_ex(a, planes, n_cols, arith=arith, epilogue=_lib.EPI_RECON, aux=x, row_scale=row_scale, a_row_max=rm)
    return g, rows


def recon_rescale_rows(g_spec: Tensor, g_out: Tensor, row_scale: float, row_max: Optional[Tensor] = None,
                       col_max: Optional[Tensor] = None) -> Tensor:
    """In place: rows of g_spec whose upstream gradient g_out[row] is not row_scale (bit compare) are multiplied by
    g_out[row] / row_scale (rqhip_recon_rescale_rows_ex); returns g_spec.  row_max [parts, B] / col_max [N] (int32, the
    maxima the fused epilogue emitted for g_spec) are brought up to date for the rows that change."""
    _need_gpu(g_spec, g_out, row_max, col_max)
    g_out = _f32c(g_out, "g_out")
    B, N = g_spec.shape
    with torch.cuda.device(g_spec.device):
        check(_lib.lib().rqhip_recon_rescale_rows_ex(_ptr(g_out), B, N, float(row_scale), _ptr(g_spec), _ptr(row_max),
                                                     0 if row_max is None else row_max.shape[0], _ptr(col_max), _stream()),
              "rqhip_recon_rescale_rows_ex")
    return g_spec
