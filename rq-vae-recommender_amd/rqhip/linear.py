"""Kernel selection for the bias-free Linear(+ReLU) layers of the MLPs, shared by the autograd Functions of
modules/encoder.py and the registered operators of rqhip/torch_ops.py (both must run the same kernels: the tests compare
them bit for bit)."""
import torch
from torch import Tensor

from . import ops

# ---- the large activation GEMMs on the bf16 matrix cores (csrc/gemm_split.hip) ------------------------------------------
# Measured at 100 000 rows against the tuned library fp32 GEMM of the same layer (tools/bench_gemm_split.py, and in the
# step: profiles/r03_bench_kernel_stats_summary.txt): every supported forward and data gradient is faster -- the first
# encoder layer's forward (768 -> 512 with the ReLU in the hipBLASLt epilogue) was a tie in isolation (456 vs 445-489 us)
# but takes 626 us inside the step, where the chip runs at the clocks the other matrix kernels leave it.  Small batches
# are launch-bound and stay with the library.
_SPLIT_MIN_ROWS = 4096
_SPLIT_GEMMS = True


def use_split_gemms(on: bool = True) -> bool:
    """Route the large activation GEMMs through csrc/gemm_split.hip (default) or the library (A/B: tools/ab_step.py).
    Returns the previous setting."""
    global _SPLIT_GEMMS
    before, _SPLIT_GEMMS = _SPLIT_GEMMS, bool(on)
    return before


def split_ok(x: Tensor, n_cols: int, n_red: int, forward_relu: bool = False) -> bool:
    """Does this GEMM (x [M, n_red] against a weight image of n_cols columns) take the bf16-split kernel?"""
    del forward_relu   # (round 3 excluded the first encoder layer's forward here; see the note above)
    return bool(_SPLIT_GEMMS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                and x.shape[0] >= _SPLIT_MIN_ROWS and x.is_contiguous() and _wide(n_cols)
                and ops.gemm_split_supported(n_cols, n_red))


def _wide(n_cols: int) -> bool:
    """Layers of 256 (mod 256) output columns only.  The kernel also has a 256 x 128 tile for 128 (mod 256) columns
    (csrc/gemm_split.hip, tested), but the layers that would take it are HBM-bound at these widths and the library is as
    fast: 100 000 x 256 -> 128 forward 58 us vs 67, its data gradient 59 vs 56, 100 000 x 32 -> 128 24 vs 25 -- and every use
    adds the 6 us image rebuild (round 3, tools/bench_gemm_split.py)."""
    return n_cols % 256 == 0


def split_shape_ok(rows: int, n_cols: int, n_red: int) -> bool:
    """`split_ok` for a contiguous fp32 ROCm operand of `rows` rows that does not exist yet."""
    return bool(_SPLIT_GEMMS and rows >= _SPLIT_MIN_ROWS and _wide(n_cols) and ops.gemm_split_supported(n_cols, n_red))


def planes(w: Tensor, transpose: bool) -> Tensor:
    """The bf16-piece image of `w` (or of its transpose) for csrc/gemm_split.hip, rebuilt at EVERY use (one 6 us kernel).
    A first version cached it per `w._version` -- and trained on stale weights: the fused AdamW update (and any
    `w.data` write) does not bump the version counter.  Nothing cheaper than the rebuild is safe, and it is 0.1 % of the
    GEMM it feeds; inside a hipGraph capture the rebuild is part of the graph, as it has to be."""
    return ops.weight_planes(w.detach(), transpose=transpose)


def input_grad(g: Tensor, w: Tensor) -> Tensor:
    """g [M, N] . w [N, K]: the bf16-split kernel with the image of w^T where it applies, else the library GEMM."""
    g = g if g.is_contiguous() else g.contiguous()
    if split_ok(g, w.shape[1], w.shape[0], False):
        return ops.gemm_split(g, planes(w, True), w.shape[1])
    return g.mm(w)


def forward(x: Tensor, w: Tensor, relu: bool, zero_bias: Tensor = None) -> Tensor:
    """relu(x w^T) or x w^T for 2-D fp32 ROCm tensors: the bf16-split kernel where it applies, else the library GEMM (with
    the ReLU in the hipBLASLt epilogue)."""
    if split_ok(x, w.shape[0], w.shape[1], relu):
        return ops.gemm_split(x, planes(w, False), w.shape[0], relu=relu)
    if relu:
        zb = zero_bias if zero_bias is not None else x.new_zeros((w.shape[0],))
        return torch._addmm_activation(zb, x, w.t())
    return x.mm(w.t())
