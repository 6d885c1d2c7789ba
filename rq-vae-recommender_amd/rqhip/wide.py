"""Shapes and mode combinations the HIP kernels do not keep on chip, as PyTorch-ROCm operators on the GPU.

The kernels behind `Quantize` / `RqVae` / `Kmeans` hold a latent row in registers and a codebook in LDS: latent width
D <= 128 (csrc/rq_forward.hip, rq_backward.hip, kmeans.hip), and for the Gumbel-softmax level K <= 1024 codes with the
LDS carve of csrc/gumbel.hip below 160 KiB.  The reference has no such limits (modules/quantize.py:54-81 takes any
`embed_dim`, `n_embed`), and it also combines COSINE distance with GUMBEL_SOFTMAX (quantize.py:118-136), for which there is
no kernel.  Instead of raising, those calls run the reference's EXPRESSIONS (quantize.py:110-161, init/kmeans.py:39-70) with
torch operators on the ROCm tensors they were given, differentiated by autograd.  This is not a CPU path and not the
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


# ---- reference modules/quantize.py:110-161 ---------------------------------------------------------------------------------
def quantize_forward(layer, x: Tensor, temperature: float):
    """`Quantize.forward` after the lazy k-means init, as torch operators.  Returns (emb_out, ids, loss)."""
    from modules.quantize import QuantizeDistance, QuantizeForwardMode, efficient_rotation_trick_transform
    _need_gpu(x, "Quantize.forward")
    _note(f"Quantize(embed_dim={layer.embed_dim}, n_embed={layer.n_embed}, {layer.forward_mode.name}, {layer.distance_mode.name})")
    codebook = layer.codebook()
    if layer.distance_mode == QuantizeDistance.L2:
        dist = (x ** 2).sum(dim=1, keepdim=True) + (codebook.T ** 2).sum(dim=0, keepdim=True) - 2 * x @ codebook.T
    elif layer.distance_mode == QuantizeDistance.COSINE:
        dist = -(x / x.norm(dim=1, keepdim=True) @ codebook.T / codebook.T.norm(dim=0, keepdim=True))
    else:
        raise Exception("Unsupported Quantize distance mode.")
    ids = dist.detach().min(dim=1).indices
    if not layer.training:
        emb_out = layer.get_item_embeddings(ids)
        return emb_out, ids, layer.quantize_loss(query=x, value=emb_out)
    if layer.forward_mode == QuantizeForwardMode.GUMBEL_SOFTMAX:
        u = torch.rand(dist.shape, device=layer.device)                      # distributions/gumbel.py:10-11
        gumbel = -torch.log(-torch.log(u + 1e-20) + 1e-20)
        weights = torch.softmax((-dist + gumbel) / temperature, dim=-1)
        emb = weights @ codebook
        emb_out = emb
    elif layer.forward_mode == QuantizeForwardMode.STE:
        emb = layer.get_item_embeddings(ids)
        emb_out = x + (emb - x).detach()
    elif layer.forward_mode == QuantizeForwardMode.ROTATION_TRICK:
        emb = layer.get_item_embeddings(ids)
        rot = efficient_rotation_trick_transform(x / (x.norm(dim=-1, keepdim=True) + 1e-8),
                                                 emb / (emb.norm(dim=-1, keepdim=True) + 1e-8), x)
        emb_out = rot * (emb.norm(dim=1, keepdim=True) / (x.norm(dim=1, keepdim=True) + 1e-6)).detach()
    else:
        raise Exception("Unsupported Quantize forward mode.")
    return emb_out, ids, layer.quantize_loss(query=x, value=emb)


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
