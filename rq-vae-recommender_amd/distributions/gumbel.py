"""Gumbel noise helpers with the reference's names (reference distributions/gumbel.py:8-20).

The accelerated Quantize draws the SAME uniform tensor the reference does -- torch.rand(shape, device) --
and hands it to the HIP kernel, which applies -log(-log(U+eps)+eps), the temperature and the softmax
on chip (csrc/gumbel.hip).  These functions remain for code that calls them directly."""
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


def sample_gumbel(shape: Tuple, device: torch.device, eps: float = 1e-20) -> Tensor:
    u = torch.rand(shape, device=device)
    return -torch.log(eps - torch.log(u + eps))


def gumbel_softmax_sample(logits: Tensor, temperature: float, device: torch.device) -> Tensor:
    noisy = logits + sample_gumbel(logits.shape, device)
    return F.softmax(noisy / temperature, dim=-1)
