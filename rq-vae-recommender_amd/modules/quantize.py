"""One residual-quantisation level on the MI355X, behind the reference's `Quantize` API.

Mirror of reference modules/quantize.py (names, constructor arguments, attributes, return type and error
behaviour), but `forward` does not build a distance matrix with torch ops: distance, argmin, codeword
gather, STE / rotation-trick / Gumbel-softmax output and the quantize loss are ONE HIP kernel launch
(csrc/rq_forward.hip, csrc/gumbel.hip) and its autograd backward another (csrc/rq_backward.hip), reached
through the C ABI of include/rqhip.h.  `RqVae` does not even call this per level: it hands all its levels to
the same kernel at once (modules/rqvae.py); `Quantize.forward` is that kernel with L = 1.

There is no CPU implementation: tensors must live on a ROCm device (rqhip.RqHipError otherwise).  Shapes the kernels do not
keep on chip (embed_dim > 128; Gumbel-softmax with more than 1024 codes) and COSINE x GUMBEL_SOFTMAX run the reference's
expression as PyTorch-ROCm operators on the same device tensors (rqhip/wide.py), with a one-time warning.
"""
from enum import Enum
from typing import NamedTuple

import torch
from torch import Tensor, nn

from init.kmeans import kmeans_init_
from modules.loss import QuantizeLoss
from modules.normalize import L2NormalizationLayer
from rqhip import MODE_EVAL, MODE_ROTATION, MODE_STE, wide
from rqhip.autograd import GumbelLevelFunction, RqStackFunction

try:
    import gin
except ImportError:  # pragma: no cover - environment dependent
    from rqhip import ginlite as gin


@gin.constants_from_enum
class QuantizeForwardMode(Enum):
    GUMBEL_SOFTMAX = 1
    STE = 2
    ROTATION_TRICK = 3


class QuantizeDistance(Enum):
    L2 = 1
    COSINE = 2


class QuantizeOutput(NamedTuple):
    embeddings: Tensor
    ids: Tensor
    loss: Tensor


_TRAIN_MODE = {QuantizeForwardMode.STE: MODE_STE, QuantizeForwardMode.ROTATION_TRICK: MODE_ROTATION}


def efficient_rotation_trick_transform(u: Tensor, q: Tensor, e: Tensor) -> Tensor:
    """Section 4.2 of arXiv:2410.06424 applied row-wise: e - 2 (e.w) w + 2 (e.u) q with w = normalize(u+q);
    u, q and w are constants for autograd (reference modules/quantize.py:34-50).  Kept as a public helper;
    the ROTATION_TRICK forward mode computes the same thing inside the fused kernel."""
    u, q = u.detach(), q.detach()
    w = torch.nn.functional.normalize(u + q, p=2, dim=1, eps=1e-6)
    ew = (e * w).sum(dim=1, keepdim=True)
    eu = (e * u).sum(dim=1, keepdim=True)
    return e - 2 * ew * w + 2 * eu * q


class Quantize(nn.Module):
    def __init__(
        self,
        embed_dim: int,
        n_embed: int,
        do_kmeans_init: bool = True,
        codebook_normalize: bool = False,
        sim_vq: bool = False,  # https://arxiv.org/pdf/2411.02038
        commitment_weight: float = 0.25,
        forward_mode: QuantizeForwardMode = QuantizeForwardMode.GUMBEL_SOFTMAX,
        distance_mode: QuantizeDistance = QuantizeDistance.L2,
    ) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.n_embed = n_embed
        self.forward_mode = forward_mode
        self.distance_mode = distance_mode
        self.do_kmeans_init = do_kmeans_init
        self.kmeans_initted = False
        self.kmeans_rows_sharded = False   # multi-GPU warm-up: the lazy init sees this rank's block of the rows only
        self.embedding = nn.Embedding(n_embed, embed_dim)
        # identity unless sim_vq / codebook_normalize: then ordinary torch ops in front of the kernel
        self.out_proj = nn.Sequential(
            nn.Linear(embed_dim, embed_dim, bias=False) if sim_vq else nn.Identity(),
            L2NormalizationLayer(dim=-1) if codebook_normalize else nn.Identity(),
        )
        self.quantize_loss = QuantizeLoss(commitment_weight)
        nn.init.uniform_(self.embedding.weight)  # U(0,1), reference quantize.py:91-94

    @property
    def weight(self) -> Tensor:
        return self.embedding.weight

    @property
    def device(self) -> torch.device:
        return self.embedding.weight.device

    @property
    def commitment_weight(self) -> float:
        return self.quantize_loss.commitment_weight

    @property
    def plain_codebook(self) -> bool:
        """True when out_proj is the identity (all shipped configs)."""
        return all(isinstance(m, nn.Identity) for m in self.out_proj)

    def kernel_covers(self) -> bool:
        """Do the HIP kernels take this level's shape in its current mode?  (else: rqhip/wide.py)"""
        if self.training and self.forward_mode == QuantizeForwardMode.GUMBEL_SOFTMAX:
            return self.distance_mode == QuantizeDistance.L2 and wide.gumbel_covers(self.embed_dim, self.n_embed)
        return wide.stack_covers(self.embed_dim, self.n_embed)

    def codebook(self) -> Tensor:
        """out_proj(embedding.weight): the [K,D] matrix distances are taken against (quantize.py:110)."""
        return self.embedding.weight if self.plain_codebook else self.out_proj(self.embedding.weight)

    def hip_mode(self) -> int:
        """RQHIP_MODE_* for the current training flag; raises for modes the stack kernel does not cover."""
        if not self.training:
            return MODE_EVAL
        try:
            return _TRAIN_MODE[self.forward_mode]
        except KeyError:
            raise Exception("Unsupported Quantize forward mode.") from None

    @torch.no_grad()
    def _kmeans_init(self, x: Tensor) -> None:
        kmeans_init_(self.embedding.weight, x=x, rows_sharded=self.kmeans_rows_sharded)
        self.kmeans_initted = True

    def get_item_embeddings(self, item_ids: Tensor) -> Tensor:
        return self.out_proj(self.embedding(item_ids))

    def _forward_cosine(self, x: Tensor, temperature: float) -> QuantizeOutput:
        """QuantizeDistance.COSINE (reference quantize.py:118-124; RqVae never selects it).  The argmin of
        -(x/|x|).c_k/|c_k| runs on the HIP kernel as the nearest unit codeword of the unit query (for unit vectors
        |a-b|^2 = 2 - 2 a.b, the same ordering); everything after the ids -- gather, STE / rotation output, loss --
        is the reference's expression in PyTorch-ROCm ops, differentiated by autograd.
        Near-tie caveat: the kernel ranks by (|xn|^2 + |cn_k|^2) - 2 xn.cn_k, where |cn_k|^2 is 1 up to an ulp PER CODE,
        while the reference ranks by -(xn.cn_k) alone; two codes whose cosines differ by less than ~1e-7 can therefore
        be ordered differently (the same class of sub-ulp ties as tests/parity_gate.py describes for the L2 path).
        GUMBEL_SOFTMAX with COSINE (softmax over the cosines, gradient through both normalisations) has no kernel: the
        reference's expression in PyTorch-ROCm operators (rqhip/wide.py)."""
        if (self.training and self.forward_mode == QuantizeForwardMode.GUMBEL_SOFTMAX) or not wide.stack_covers(self.embed_dim, self.n_embed):
            return QuantizeOutput(*wide.quantize_forward(self, x, float(temperature)))
        codebook = self.codebook()
        with torch.no_grad():
            xn = x / x.norm(dim=1, keepdim=True)
            cn = codebook / codebook.norm(dim=1, keepdim=True)
            ids = RqStackFunction.apply(xn, cn.unsqueeze(0), MODE_EVAL, 0.0, False, None)[2][0]
        emb = self.get_item_embeddings(ids)
        if not self.training:
            return QuantizeOutput(embeddings=emb, ids=ids, loss=self.quantize_loss(query=x, value=emb))
        if self.forward_mode == QuantizeForwardMode.STE:
            emb_out = x + (emb - x).detach()
        elif self.forward_mode == QuantizeForwardMode.ROTATION_TRICK:
            rot = efficient_rotation_trick_transform(x / (x.norm(dim=-1, keepdim=True) + 1e-8),
                                                     emb / (emb.norm(dim=-1, keepdim=True) + 1e-8), x)
            emb_out = rot * (emb.norm(dim=1, keepdim=True) / (x.norm(dim=1, keepdim=True) + 1e-6)).detach()
        else:
            raise Exception("Unsupported Quantize forward mode.")
        return QuantizeOutput(embeddings=emb_out, ids=ids, loss=self.quantize_loss(query=x, value=emb))

    def forward(self, x: Tensor, temperature: float) -> QuantizeOutput:
        assert x.shape[-1] == self.embed_dim
        if self.do_kmeans_init and not self.kmeans_initted:
            self._kmeans_init(x=x)
        if self.distance_mode == QuantizeDistance.COSINE:
            return self._forward_cosine(x, temperature)
        if self.distance_mode != QuantizeDistance.L2:
            raise Exception("Unsupported Quantize distance mode.")
        if not self.kernel_covers():
            return QuantizeOutput(*wide.quantize_forward(self, x, float(temperature)))

        codebook = self.codebook()
        beta = float(self.commitment_weight)
        if self.training and self.forward_mode == QuantizeForwardMode.GUMBEL_SOFTMAX:
            # the reference draws torch.rand(B, K) on self.device here (distributions/gumbel.py:10)
            noise = torch.rand((x.shape[0], self.n_embed), device=self.device)
            emb, ids, loss = GumbelLevelFunction.apply(x, codebook, noise, float(temperature), beta)
            return QuantizeOutput(embeddings=emb, ids=ids, loss=loss)
        if self.training and self.forward_mode not in _TRAIN_MODE:
            raise Exception("Unsupported Quantize forward mode.")

        embs, _res, ids, loss, _sum, _norm = RqStackFunction.apply(x, codebook.unsqueeze(0), self.hip_mode(), beta,
                                                                   True, None)
        return QuantizeOutput(embeddings=embs[0], ids=ids[0], loss=loss)
