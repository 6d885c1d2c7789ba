"""Kernel selection for the bias-free Linear(+ReLU) layers of the MLPs, shared by the autograd Functions of
modules/encoder.py and the registered operators of rqhip/torch_ops.py (both must run the same kernels: the tests compare
them bit for bit)."""
import torch
from torch import Tensor

from . import ops

# ---- the large activation GEMMs on the bf16 matrix cores (csrc/gemm_split.hip) ------------------------------------------
# Measured at 100 000 rows against the tuned library fp32 GEMM of the same layer (tools/bench_gemm_split.py):
# every supported data gradient and every supported forward but the first encoder layer's (768 -> 512 with the ReLU
# epilogue: 470 vs 445-489 us, a tie) is faster; small batches are launch-bound and stay with the library.
_SPLIT_MIN_ROWS = 4096
_SPLIT_GEMMS = True


def use_split_gemms(on: bool = True) -> bool:
    """Route the large activation GEMMs through csrc/gemm_split.hip (default) or the library (A/B: tools/ab_step.py).
    Returns the previous setting."""
    global _SPLIT_GEMMS
    before, _SPLIT_GEMMS = _SPLIT_GEMMS, bool(on)
    return before


_PLANES = {}   # (id(weight), transpose) -> (weight version, weakref, image): the image is rebuilt when the weight changes


def split_ok(x: Tensor, n_cols: int, n_red: int, forward_relu: bool) -> bool:
    if not (_SPLIT_GEMMS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= _SPLIT_MIN_ROWS
            and x.is_contiguous() and ops.gemm_split_supported(n_cols, n_red)):
        return False
    return not (forward_relu and (n_cols, n_red) == (512, 768))


def planes(w: Tensor, transpose: bool) -> Tensor:
    """The bf16-piece image of `w` (or of its transpose), cached per weight VERSION: the optimizer's in-place update bumps
    it, eval / tokenisation loops reuse the image.  While a hipGraph is being captured the image is always rebuilt -- the
    replayed step updates the weights without running this Python code, so the rebuild has to be part of the graph."""
    import weakref
    wd = w.detach()
    if torch.cuda.is_current_stream_capturing():
        return ops.weight_planes(wd, transpose=transpose)
    key = (id(w), transpose)
    hit = _PLANES.get(key)
    if hit is not None and hit[0] == w._version and hit[1]() is w and hit[2].device == w.device:
        return hit[2]
    img = ops.weight_planes(wd, transpose=transpose)
    if len(_PLANES) > 64:
        _PLANES.clear()
    _PLANES[key] = (w._version, weakref.ref(w), img)
    return img


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
