"""Loss modules with the reference's names and semantics (reference modules/loss.py:5-41).

`QuantizeLoss` exists for API compatibility: inside Quantize/RqVae the same quantity is produced by the
fused HIP kernel (csrc/rq_forward.hip) and its gradient by csrc/rq_backward.hip.  `ReconstructionLoss` is a fused
HIP kernel too (csrc/recon_loss.hip); the categorical variant adds PyTorch-ROCm BCE on the trailing columns."""
import torch
from torch import Tensor, nn
from torch.nn import functional as F

from rqhip.autograd import ReconLossFunction


class ReconstructionLoss(nn.Module):
    """Row-wise squared error, summed over features: one fused HIP pass (csrc/recon_loss.hip) instead of the
    sub / pow / sum kernels the expression would launch; GPU tensors only, like the rest of the path."""

    def forward(self, x_hat: Tensor, x: Tensor) -> Tensor:
        lead = x.shape[:-1]
        from rqhip import torch_ops
        if torch_ops.enabled():
            out = torch.ops.rqhip.recon_loss(x_hat.reshape(-1, x_hat.shape[-1]), x.reshape(-1, x.shape[-1]))
            return out.reshape(lead)
        out = ReconLossFunction.apply(x_hat.reshape(-1, x_hat.shape[-1]), x.reshape(-1, x.shape[-1]))
        return out.reshape(lead)


class CategoricalReconstuctionLoss(nn.Module):  # (sic) the reference's spelling is part of its API
    """Squared error on the dense columns + BCE-with-logits on the trailing `n_cat_feats` columns."""

    def __init__(self, n_cat_feats: int) -> None:
        super().__init__()
        self.reconstruction_loss = ReconstructionLoss()
        self.n_cat_feats = n_cat_feats

    def forward(self, x_hat: Tensor, x: Tensor) -> Tensor:
        n = self.n_cat_feats
        total = self.reconstruction_loss(x_hat[:, :-n], x[:, :-n])
        if n > 0:
            bce = F.binary_cross_entropy_with_logits(x_hat[:, -n:], x[:, -n:], reduction="none")
            total = total + bce.sum(dim=-1)
        return total


class QuantizeLoss(nn.Module):
    """||sg(query) - value||^2 + commitment_weight * ||query - sg(value)||^2 per row."""

    def __init__(self, commitment_weight: float = 1.0) -> None:
        super().__init__()
        self.commitment_weight = commitment_weight

    def forward(self, query: Tensor, value: Tensor) -> Tensor:
        codebook_term = (query.detach() - value).pow(2).sum(dim=-1)
        commit_term = (query - value.detach()).pow(2).sum(dim=-1)
        return codebook_term + self.commitment_weight * commit_term
