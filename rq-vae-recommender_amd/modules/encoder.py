"""Bias-free Linear+ReLU stack (API and state_dict keys of reference modules/encoder.py:7-38).

The encoder/decoder GEMMs stay on PyTorch-ROCm (rocBLAS/hipBLASLt fp32); SURVEY.md section 8f lists fusing
them around the RQ kernel as the next step.  Parameter names are `mlp.{0,2,4,...}.weight`, as in the
reference, so checkpoints load in both directions."""
from typing import List

from torch import Tensor, nn

from modules.normalize import L2NormalizationLayer


class MLP(nn.Module):
    def __init__(self, input_dim: int, hidden_dims: List[int], out_dim: int, dropout: float = 0.0,
                 normalize: bool = False) -> None:
        super().__init__()
        self.input_dim, self.hidden_dims, self.out_dim, self.dropout = input_dim, hidden_dims, out_dim, dropout
        widths = [input_dim, *hidden_dims, out_dim]
        stack = nn.Sequential()
        last = len(widths) - 2
        for i in range(len(widths) - 1):
            stack.append(nn.Linear(widths[i], widths[i + 1], bias=False))
            if i == last:
                break
            stack.append(nn.ReLU())
            if dropout != 0:
                stack.append(nn.Dropout(dropout))
        stack.append(L2NormalizationLayer() if normalize else nn.Identity())
        self.mlp = stack

    def forward(self, x: Tensor) -> Tensor:
        assert x.shape[-1] == self.input_dim, f"Invalid input dim: Expected {self.input_dim}, found {x.shape[-1]}"
        return self.mlp(x)
