"""torch.autograd bridges: forward and backward both run in HIP through the C ABI.

These Functions are the seam between the reference-shaped Python modules (modules/quantize.py,
modules/rqvae.py) and librqhip.so.  They hold no arithmetic of their own.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops


def _dense(g: Optional[Tensor]) -> Optional[Tensor]:
    """Upstream gradients may arrive as None, expanded views or non-contiguous tensors."""
    if g is None:
        return None
    return g.contiguous()


class RqStackFunction(torch.autograd.Function):
    """L chained quantisation levels (EVAL / STE / ROTATION) as one differentiable op.

    forward(res0 [B,D], codebooks [L,K,D], mode, beta, want_levels) ->
        embs [L,B,D], residuals [L,B,D], ids [L,B], loss [B], emb_sum [B,D], embs_norm [B,L]
    (embs / residuals are empty tensors when want_levels is False: the fused training step only consumes
    emb_sum, loss and embs_norm, see modules/rqvae.py.)
    """

    @staticmethod
    def forward(ctx, res0: Tensor, codebooks: Tensor, mode: int, beta: float, want_levels: bool, grad_sink=None):
        """grad_sink (optional): object with `.view` ([L,K,D] slice of a flat gradient buffer) and `.params` (the L
        codebook parameters); the codebook gradient is written there when it is the first gradient of the step."""
        out = ops.rq_forward(res0, codebooks, mode, beta, want_embs=want_levels, want_residuals=want_levels)
        ctx.set_materialize_grads(False)   # outputs nobody used arrive as None (= NULL at the C ABI), not as zero tensors
        ctx.save_for_backward(res0, codebooks, out.ids)
        ctx.mode, ctx.beta, ctx.want_levels, ctx.grad_sink = mode, beta, want_levels, grad_sink
        embs = out.embs if want_levels else res0.new_empty((0,))
        residuals = out.residuals if want_levels else res0.new_empty((0,))
        if want_levels:
            ctx.mark_non_differentiable(out.ids, out.embs_norm)
        else:  # one call: a second mark_non_differentiable would replace the first
            ctx.mark_non_differentiable(out.ids, out.embs_norm, embs, residuals)
        return embs, residuals, out.ids, out.loss, out.emb_sum, out.embs_norm

    @staticmethod
    def backward(ctx, g_embs, g_resid, _g_ids, g_loss, g_embsum, _g_norm):
        res0, codebooks, ids = ctx.saved_tensors
        need_res0, need_cb = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_res0 or need_cb):
            return None, None, None, None, None, None
        if not ctx.want_levels:
            g_embs = g_resid = None
        sink = ctx.grad_sink
        out_cb = None
        if need_cb and sink is not None and tuple(sink.view.shape) == tuple(codebooks.shape) and all(
                p.grad is None and getattr(p, "_rq_sink_epoch", -1) != sink.owner.epoch for p in sink.params):
            out_cb = sink.view      # first gradient of the step: straight into the flat buffer (autograd's stack backward
                                    # hands each level's slice to its parameter as a view); claimed for this epoch, so a
                                    # second pass of the same model under one loss accumulates through autograd instead
            for p in sink.params:
                p._rq_sink_epoch = sink.owner.epoch
        g_res0, g_cb = ops.rq_backward(res0, codebooks, ctx.mode, ctx.beta, ids, g_embs=_dense(g_embs),
                                       g_embsum=_dense(g_embsum), g_resid=_dense(g_resid), g_loss=_dense(g_loss),
                                       need_res0=need_res0, need_codebooks=need_cb, out_g_codebooks=out_cb,
                                       cbgrad=ops.cbgrad_default())
        return g_res0, (g_cb.view_as(g_cb) if out_cb is not None else g_cb), None, None, None, None


class GumbelLevelFunction(torch.autograd.Function):
    """One GUMBEL_SOFTMAX level (training): forward(x [B,D], codebook [K,D], U [B,K], T, beta) -> emb, ids, loss."""

    @staticmethod
    def forward(ctx, x: Tensor, codebook: Tensor, U: Tensor, temperature: float, beta: float):
        ids, emb, loss = ops.gumbel_forward(x, codebook, U, temperature, beta)
        ctx.save_for_backward(x, codebook, U)
        ctx.temperature, ctx.beta = temperature, beta
        ctx.mark_non_differentiable(ids)
        return emb, ids, loss

    @staticmethod
    def backward(ctx, g_emb, _g_ids, g_loss):
        x, codebook, U = ctx.saved_tensors
        g_x, g_cb = ops.gumbel_backward(x, codebook, U, ctx.temperature, ctx.beta, g_emb=_dense(g_emb),
                                        g_loss=_dense(g_loss))
        return g_x, g_cb, None, None, None


# The factor the caller will apply to the batch-mean loss before calling backward() (1 / gradient_accumulate_every, or a
# micro-batch's share of a larger batch).  Only a HINT for the speculative reconstruction-loss gradient below: a wrong
# value costs the saved pass, never correctness.
_LOSS_SCALE = 1.0


class loss_scale:
    """`with loss_scale(s): out = model(batch, t)` -- tell the forward that `out.loss * s` is what will be backpropagated."""

    def __init__(self, s: float) -> None:
        self.s, self.prev = float(s), 1.0

    def __enter__(self):
        global _LOSS_SCALE
        self.prev, _LOSS_SCALE = _LOSS_SCALE, self.s
        return self

    def __exit__(self, *exc):
        global _LOSS_SCALE
        _LOSS_SCALE = self.prev
        return False


class ReconLossFunction(torch.autograd.Function):
    """Row-wise squared error (reference modules/loss.py:5-10) as one HIP pass forward and one backward.

    When only x_hat needs a gradient (the training step), the forward pass also writes the gradient it expects to be
    asked for -- 2 (x_hat - x) / B, what `.mean().backward()` of rqvae.py:152-154 sends -- and the backward pass merely
    verifies the upstream rows on the device, redoing the ones that differ: same results, one pass over the [B, 768]
    tensors instead of two."""

    @staticmethod
    def forward(ctx, x_hat: Tensor, x: Tensor):
        B = x.shape[0]
        ctx.spec = B > 0 and ctx.needs_input_grad[0] and not ctx.needs_input_grad[1] and ops.recon_spec_ok(x_hat, x)
        if ctx.spec:
            # fp32 (loss scale) * fp32 (1 / B): what autograd's multiply + mean backward hand to every row on the device
            # (a scalar divisor becomes a multiplication by its fp32 reciprocal there, and in rqhip_loss_means_backward)
            one = torch.tensor(1.0, dtype=torch.float32)
            ctx.row_scale = float(torch.tensor(_LOSS_SCALE, dtype=torch.float32) * (one / B))
            out, g_spec = ops.recon_loss_forward_spec(x_hat, x, ctx.row_scale)
            ctx.save_for_backward(x_hat, x, g_spec)
            return out
        ctx.save_for_backward(x_hat, x)
        return ops.recon_loss_forward(x_hat, x)

    @staticmethod
    def backward(ctx, g_out):
        if ctx.spec and not getattr(ctx, "spec_consumed", False):
            # the speculative buffer is handed to autograd (and fixed up in place) ONCE; a second backward through a
            # retained graph recomputes from x_hat and x below instead of returning the first call's tensor again
            ctx.spec_consumed = True
            x_hat, x, g_spec = ctx.saved_tensors
            return ops.recon_loss_backward_spec(x_hat, x, _dense(g_out), ctx.row_scale, g_spec), None
        x_hat, x = ctx.saved_tensors[:2]
        need_hat, need_x = ctx.needs_input_grad
        if not (need_hat or need_x):
            return None, None
        return ops.recon_loss_backward(x_hat, x, _dense(g_out), need_hat, need_x)


class LossMeansFunction(torch.autograd.Function):
    """(loss, reconstruction_loss, rqvae_loss) of RqVae.forward -- mean(recon + quant), mean(recon), mean(quant) -- as one
    launch.  Backward is what autograd derives for the three means, also one launch: every row of `recon` receives
    (g_loss + g_recon_mean) * (1/B), every row of `quant` (g_loss + g_quant_mean) * (1/B)."""

    @staticmethod
    def forward(ctx, recon: Tensor, quant: Tensor):
        ctx.n = recon.numel()
        ctx.set_materialize_grads(False)   # unused means arrive as None, not as zero tensors to add
        out = ops.loss_means(recon, quant)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_loss, g_rmean, g_qmean):
        need_r = ctx.needs_input_grad[0] and (g_loss is not None or g_rmean is not None)
        need_q = ctx.needs_input_grad[1] and (g_loss is not None or g_qmean is not None)
        if g_rmean is None and g_qmean is None and g_loss is not None and need_r and need_q:
            rows, _ = ops.loss_means_backward(g_loss, None, None, ctx.n, True, False)
            return rows, rows       # the usual case (only `loss` is backpropagated): one vector serves both inputs
        return ops.loss_means_backward(g_loss, g_rmean, g_qmean, ctx.n, need_r, need_q)
