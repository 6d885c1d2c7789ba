"""Bias-free Linear+ReLU stack (API and state_dict keys of reference modules/encoder.py:7-38).

The large encoder/decoder GEMMs (output or reduction width >= 256) run as six-term bf16-split products on the bf16 matrix
cores (csrc/gemm_split.hip: fp32 operands as three exact bf16 pieces, fp32 accumulation; no less exact than the library's
fp32 GEMM, tests/test_gpu_gemm_split.py); the rest stay library GEMMs on PyTorch-ROCm (fp32, `rqhip/tuning.py` picks the
kernels), where every Linear that is followed by a ReLU runs as ONE hipBLASLt call with the ReLU in the GEMM epilogue
(`torch._addmm_activation` with a zero bias) instead of a GEMM plus an elementwise pass over the activations:
-0.29 ms of a 6.1 ms step at 100 000 rows (`tools/relu_epilogue_probe.py`).  The op has no autograd formula, so
`_LinearReLU` supplies the backward itself: the weight gradient WITH the ReLU mask fused is one hand-written kernel
(`csrc/wgrad.hip`), the input gradient a library GEMM on the masked gradient it hands over.  Parameter names are `mlp.{0,2,4,...}.weight`, as in the reference, so checkpoints load in both
directions."""
from typing import List

import torch
from torch import Tensor, nn

from modules.normalize import L2NormalizationLayer
from rqhip import linear as _lin
from rqhip import ops, torch_ops
from rqhip.linear import use_split_gemms  # noqa: F401  (A/B switch, tools/ab_step.py)


def _grad_sink(w: Tensor):
    """The parameter's slice of a flat gradient buffer (rqhip.dist.FlatGradReducer.attach), for the FIRST producer of
    this parameter's gradient in a step only: later ones (the same MLP applied twice under one loss) get None and
    return ordinary tensors, which autograd accumulates."""
    from rqhip.dist import claim_grad_sink
    return claim_grad_sink(w)


def _adopt(gw: Tensor, sink) -> Tensor:
    """A fresh alias of the sink: autograd takes a gradient tensor over as `.grad` only when nobody else holds it."""
    return gw.view_as(gw) if sink is not None else gw


def _hip_wgrad_ok(g: Tensor, w: Tensor) -> bool:
    return (g.is_cuda and g.dtype == torch.float32 and g.dim() == 2 and g.shape[0] > 0
            and ops.linear_wgrad_supported(w.shape[0], w.shape[1]))


class _LinearReLU(torch.autograd.Function):
    """relu(x @ w.T) for 2-D fp32 ROCm tensors, ReLU fused into the GEMM epilogue.  Backward: the weight gradient and
    the ReLU mask are ONE hand-written kernel (csrc/wgrad.hip), which also hands the masked gradient to the library
    GEMM that forms the input gradient; shapes the kernel does not tile keep the three library kernels autograd
    would run (mask, two GEMMs)."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, zero_bias: Tensor) -> Tensor:
        y = _lin.forward(x, w, True, zero_bias)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, w, y = ctx.saved_tensors
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        sink = _grad_sink(w) if need_w else None
        if need_w and _hip_wgrad_ok(gy, w):
            gw, g = ops.linear_wgrad(gy, y, x, want_masked=need_x, out=sink)
        else:
            g = torch.ops.aten.threshold_backward(gy, y, 0.0)      # gy where y > 0 (what autograd does for relu)
            gw = (torch.mm(g.t(), x, out=sink) if sink is not None else g.t().mm(x)) if need_w else None
        gx = _lin.input_grad(g, w) if need_x else None
        return gx, (_adopt(gw, sink) if need_w else None), None


class _LinearPlain(torch.autograd.Function):
    """x @ w.T (the last layer of each MLP: no ReLU), weight gradient by csrc/wgrad.hip."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor) -> Tensor:
        ctx.save_for_backward(x, w)
        return _lin.forward(x, w, False)

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, w = ctx.saved_tensors
        need_x, need_w = ctx.needs_input_grad
        gw = None
        if need_w:
            # (round 2 kept the 768 x 512 layer with the library: 626 vs 600 us for the fp32-MFMA kernel without a mask to
            # fuse; the bf16-split kernel of round 3 takes it: csrc/wgrad_split.hip)
            sink = _grad_sink(w)
            if _hip_wgrad_ok(gy, w):
                gw = ops.linear_wgrad(gy, None, x, out=sink)[0]
            else:
                gw = torch.mm(gy.t(), x, out=sink) if sink is not None else gy.t().mm(x)
            gw = _adopt(gw, sink)
        gx = _lin.input_grad(gy, w) if need_x else None
        return gx, gw


class _LinearRecon(torch.autograd.Function):
    """reconstruction_loss(h @ w.T, x) per row (reference modules/rqvae.py:146,152 + modules/loss.py:5-10) with the LAST
    decoder layer and the loss in ONE kernel (csrc/gemm_split.hip, epilogue 2): x_hat never reaches memory; the epilogue
    reads x, sums the squared error of its row and writes the gradient the step is going to ask for, (2 (x_hat - x)) * s / B
    (s: rqhip.autograd.loss_scale).  Backward compares the upstream rows with s / B on the device and rescales the rows
    that differ (csrc/recon_loss.hip: recon_rescale_rows_kernel), then forms the two GEMM gradients from that matrix as
    _LinearPlain does.  A second backward through a retained graph recomputes x_hat with the library."""

    @staticmethod
    def forward(ctx, h: Tensor, w: Tensor, x: Tensor) -> Tensor:
        from rqhip import autograd as _ag
        one = torch.tensor(1.0, dtype=torch.float32)   # fp32 (loss scale) * fp32 (1 / B), as ReconLossFunction
        ctx.row_scale = float(torch.tensor(_ag._LOSS_SCALE, dtype=torch.float32) * (one / x.shape[0]))
        g, rows = ops.gemm_split_recon(h, _lin.planes(w, False), w.shape[0], x, ctx.row_scale)
        ctx.save_for_backward(h, w, x, g)
        return rows

    @staticmethod
    def backward(ctx, g_out: Tensor):
        h, w, x, g = ctx.saved_tensors
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_out = g_out.contiguous()
        if not getattr(ctx, "consumed", False):
            ctx.consumed = True
            g = ops.recon_rescale_rows(g, g_out, ctx.row_scale)        # in place; a no-op in a training step
        else:
            g = ops.recon_loss_backward(h.mm(w.t()), x, g_out, True, False)[0]
        gw = None
        if need_w:
            sink = _grad_sink(w)
            if _hip_wgrad_ok(g, w):
                gw = ops.linear_wgrad(g, None, h, out=sink)[0]
            else:
                gw = torch.mm(g.t(), h, out=sink) if sink is not None else g.t().mm(h)
            gw = _adopt(gw, sink)
        gh = _lin.input_grad(g, w) if need_h else None
        return gh, gw, None


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
        self._zeros = {}  # zero "bias" vectors for the fused epilogue call, per width (not parameters, not saved)

    def _zero_bias(self, n: int, like: Tensor) -> Tensor:
        z = self._zeros.get(n)
        if z is None or z.device != like.device:
            z = torch.zeros(n, dtype=like.dtype, device=like.device)
            self._zeros[n] = z
        return z

    def forward(self, x: Tensor) -> Tensor:
        assert x.shape[-1] == self.input_dim, f"Invalid input dim: Expected {self.input_dim}, found {x.shape[-1]}"
        if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32):
            return self.mlp(x)
        return self._run(x, list(self.mlp))

    def _run(self, x: Tensor, layers) -> Tensor:
        as_ops = torch_ops.enabled()   # registered torch.library operators instead of the autograd Functions
        i = 0
        while i < len(layers):
            layer = layers[i]
            if (isinstance(layer, nn.Linear) and layer.bias is None and i + 1 < len(layers)
                    and isinstance(layers[i + 1], nn.ReLU)):
                x = (torch.ops.rqhip.linear_relu(x, layer.weight) if as_ops
                     else _LinearReLU.apply(x, layer.weight, self._zero_bias(layer.out_features, x)))
                i += 2
            elif isinstance(layer, nn.Linear) and layer.bias is None:
                x = torch.ops.rqhip.linear_plain(x, layer.weight) if as_ops else _LinearPlain.apply(x, layer.weight)
                i += 1
            else:
                x = layer(x)
                i += 1
        return x

    def reconstruction_rows(self, z: Tensor, target: Tensor):
        """ReconstructionLoss(self(z), target) per row with the last layer and the loss fused (`_LinearRecon`), or None
        when that kernel does not apply here (the caller then composes the two, as the reference does)."""
        layers = list(self.mlp)
        last = layers[-2] if len(layers) >= 2 else None
        if not (isinstance(last, nn.Linear) and last.bias is None and isinstance(layers[-1], nn.Identity)
                and not torch_ops.enabled() and z.is_cuda and z.dim() == 2 and z.dtype == torch.float32
                and target.is_cuda and target.dtype == torch.float32 and not target.requires_grad
                and tuple(target.shape) == (z.shape[0], last.out_features) and target.is_contiguous()
                and target.data_ptr() % 16 == 0
                and last.out_features % 256 == 0   # (the fused epilogue sums four 64-column waves: 256-column tiles only)
                and _lin.split_shape_ok(z.shape[0], last.out_features, last.in_features)):
            return None
        assert z.shape[-1] == self.input_dim, f"Invalid input dim: Expected {self.input_dim}, found {z.shape[-1]}"
        hdn = self._run(z, layers[:-2])
        return _LinearRecon.apply(hdn if hdn.is_contiguous() else hdn.contiguous(), last.weight, target)
