"""Shapes and mode combinations the HIP kernels do not keep on chip, as PyTorch-ROCm operators on the GPU.

The kernels behind `Quantize` / `RqVae` / `Kmeans` hold a latent row in registers and a codebook in LDS: latent width
D <= 128 (csrc/rq_forward.hip, rq_backward.hip, kmeans.hip), and for the Gumbel-softmax level K <= 1024 codes with the
LDS carve of csrc/gumbel.hip below 160 KiB.  The reference has no such limits (modules/quantize.py:54-81 takes any
`embed_dim`, `n_embed`), and it also combines COSINE distance with GUMBEL_SOFTMAX (quantize.py:118-136), for which there is
no kernel.  Instead of raising, those calls run on the ROCm tensors they were given as torch operators in ROW TILES (a level's work is
row-local: no [B, K] matrix is ever formed; the reference's formulas per row, quantize.py:110-161, init/kmeans.py:39-70), differentiated by autograd.  This is not a CPU path and not the
oracle: the tensors stay on the device and nothing here imports `oracle/`.  Ties in the argmin are broken as torch breaks
them on the GPU; losses follow the reference's summation expression, in the device's reduction order.

Every entry point warns once per process, so that a run that was meant to use the kernels does not end up here silently.
"""
import warnings

import torch
from torch import Tensor

_warned = set()


def _note(what: str) -> None:
    if what not in _warned:
        _warned.add(what)
        warnings.warn(f"rqhip: {what} is outside what the HIP kernels cover; running the reference's expression as "
                      "PyTorch-ROCm operators on the GPU (rqhip/wide.py)", RuntimeWarning, stacklevel=3)


def _need_gpu(x: Tensor, who: str) -> None:
    if not x.is_cuda:
        from ._lib import RqHipError
        raise RqHipError(f"{who}: tensors must live on a ROCm device (there is no CPU path in this package)")


# ---- what the kernels cover (mirrors the checks in csrc/*.hip) --------------------------------------------------------------
def stack_covers(D: int, K: int, L: int = 1) -> bool:
    """csrc/rq_forward.hip / rq_backward.hip: 1 <= D <= 128, 1 <= K <= 65536, 1 <= L <= 16."""
    return 1 <= D <= 128 and 1 <= K <= 65536 and 1 <= L <= 16


def gumbel_covers(D: int, K: int) -> bool:
    """csrc/gumbel.hip:check_shape for the backward (the larger LDS carve): D <= 128, K <= 1024, carve <= 160 KiB."""
    kpad = (K + 63) & ~63
    floats = D * (K + 1) + kpad + 4 * (2 * kpad + 3 * 128) + K * (D + 1)
    return 1 <= D <= 128 and kpad <= 1024 and floats * 4 <= 160 * 1024


def kmeans_covers(D: int) -> bool:
    return 1 <= D <= 128


# ---- one quantisation level on shapes the kernels do not take, in ROW TILES -----------------------------------------------------
# What the reference computes per level (modules/quantize.py:110-161) is row-local: a row's distances to the K codes, its argmin, its
# output and loss depend on that row and the codebook only.  The reference forms them for all B rows at once -- a [B, K] distance matrix
# (and, for Gumbel-softmax, [B, K] noise, weights and their autograd copies: ~5 B K floats).  Here a tile of at most _TILE_ROWS rows goes
# through `_level_tile` at a time, so the temporaries are [tile, K] whatever B is (10 M rows x 2048 codes would be 82 GB per matrix), and
# autograd sees one small graph per tile.  Per row the arithmetic is the reference's (same operators in the same order: its goldens,
# tests/golden/wide_*.npz, are reproduced to the ids), and `torch.rand` is drawn per tile in row order.
_TILE_ROWS = 16384


def _scores_l2(rows: Tensor, codes: Tensor) -> Tensor:
    """[tile, K] squared distances, (|x|^2 + |c|^2) - 2 x.c in the reference's operator order (quantize.py:113-117)."""
    return (rows ** 2).sum(dim=1, keepdim=True) + (codes.T ** 2).sum(dim=0, keepdim=True) - 2 * rows @ codes.T


def _scores_cosine(rows: Tensor, codes: Tensor) -> Tensor:
    """[tile, K] negative cosine similarities (quantize.py:118-124)."""
    return -(rows / rows.norm(dim=1, keepdim=True) @ codes.T / codes.T.norm(dim=0, keepdim=True))


def _level_tile(layer, rows: Tensor, codes: Tensor, temperature: float, modes):
    """(emb_out, ids, emb) of one row tile; `emb` is what the commitment loss compares the rows with."""
    QuantizeDistance, QuantizeForwardMode, rotate = modes
    if layer.distance_mode == QuantizeDistance.L2:
        scores = _scores_l2(rows, codes)
    elif layer.distance_mode == QuantizeDistance.COSINE:
        scores = _scores_cosine(rows, codes)
    else:
        raise Exception("Unsupported Quantize distance mode.")
    ids = scores.detach().min(dim=1).indices
    if not layer.training:
        picked = layer.get_item_embeddings(ids)
        return picked, ids, picked
    mode = layer.forward_mode
    if mode == QuantizeForwardMode.GUMBEL_SOFTMAX:
        noise = torch.rand(scores.shape, device=layer.device)                 # distributions/gumbel.py:10-11
        soft = torch.softmax((-scores + -torch.log(-torch.log(noise + 1e-20) + 1e-20)) / temperature, dim=-1)
        mix = soft @ codes
        return mix, ids, mix
    picked = layer.get_item_embeddings(ids)
    if mode == QuantizeForwardMode.STE:
        return rows + (picked - rows).detach(), ids, picked
    if mode == QuantizeForwardMode.ROTATION_TRICK:
        unit = lambda t: t / (t.norm(dim=-1, keepdim=True) + 1e-8)            # noqa: E731
        turned = rotate(unit(rows), unit(picked), rows)
        return turned * (picked.norm(dim=1, keepdim=True) / (rows.norm(dim=1, keepdim=True) + 1e-6)).detach(), ids, picked
    raise Exception("Unsupported Quantize forward mode.")


def quantize_forward(layer, x: Tensor, temperature: float):
    """`Quantize.forward` after the lazy k-means init for a level the kernels do not cover.  Returns (emb_out [B, D], ids [B], loss [B])."""
    from modules.quantize import QuantizeDistance, QuantizeForwardMode, efficient_rotation_trick_transform
    _need_gpu(x, "Quantize.forward")
    _note(f"Quantize(embed_dim={layer.embed_dim}, n_embed={layer.n_embed}, {layer.forward_mode.name}, {layer.distance_mode.name})")
    codes = layer.codebook()
    modes = (QuantizeDistance, QuantizeForwardMode, efficient_rotation_trick_transform)
    outs, ids, losses = [], [], []
    for r0 in range(0, max(x.shape[0], 1), _TILE_ROWS):
        rows = x[r0:r0 + _TILE_ROWS]
        out_t, ids_t, emb_t = _level_tile(layer, rows, codes, temperature, modes)
        outs.append(out_t)
        ids.append(ids_t)
        losses.append(layer.quantize_loss(query=rows, value=emb_t))
    if len(outs) == 1:
        return outs[0], ids[0], losses[0]
    return torch.cat(outs, dim=0), torch.cat(ids, dim=0), torch.cat(losses, dim=0)


# ---- reference init/kmeans.py:39-59 ----------------------------------------------------------------------------------------
_ROWS_PER_CHUNK = 1 << 22   # elements of the [rows, K, D] difference tensor formed at a time


def kmeans_update(x: Tensor, centroids: Tensor):
    """One Lloyd step of init/kmeans.py:39-59 on the device: (assignment [B] int64, counts [K] int64); `centroids` is
    updated in place for the non-empty clusters (empty ones are left for the caller's host RNG, in cluster order).
    The direct-difference distance of kmeans.py:40-43 is formed in row chunks (the reference's [B, K, D] temporary would
    be B K D floats)."""
    _need_gpu(x, "Kmeans")
    B, D = x.shape
    K = centroids.shape[0]
    step = max(1, _ROWS_PER_CHUNK // max(1, K * D))
    assign = torch.empty((B,), dtype=torch.int64, device=x.device)
    for r0 in range(0, B, step):
        d = ((x[r0:r0 + step, None, :] - centroids[None, :, :]) ** 2).sum(dim=2)
        assign[r0:r0 + step] = d.min(dim=1).indices
    counts = torch.bincount(assign, minlength=K)
    # per-cluster sums in a FIXED order (rows sorted by cluster, stable; segment sums): `index_add_` on the GPU adds with float
    # atomics in whatever order they land, the centroids then differ in their last bits from one iteration to the next and the
    # loop's stop test (max shift < 1e-10, kmeans.py:68) never fires
    order = torch.argsort(assign, stable=True)
    sums = torch.segment_reduce(x[order], "sum", lengths=counts, axis=0)
    filled = counts > 0
    centroids[filled] = sums[filled] / counts[filled].unsqueeze(1).to(x.dtype)
    return assign, counts
