"""L2 normalisation helpers (API of reference modules/normalize.py:6-17).

Used only by the optional `codebook_normalize` paths; they run as ordinary PyTorch-ROCm ops in front of
the HIP kernels (off in every shipped config)."""
from torch import Tensor, nn
from torch.nn import functional as F


def l2norm(x: Tensor, dim: int = -1, eps: float = 1e-12) -> Tensor:
    """x / max(||x||_2, eps) along `dim`."""
    return F.normalize(x, p=2, dim=dim, eps=eps)


class L2NormalizationLayer(nn.Module):
    def __init__(self, dim: int = -1, eps: float = 1e-12) -> None:
        super().__init__()
        self.dim, self.eps = dim, eps

    def forward(self, x: Tensor) -> Tensor:
        return l2norm(x, dim=self.dim, eps=self.eps)

    def extra_repr(self) -> str:
        return f"dim={self.dim}, eps={self.eps}"
