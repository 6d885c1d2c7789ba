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


class RqSeamFunction(torch.autograd.Function):
    """The RQ <-> MLP seam as ONE differentiable op and ONE forward launch (rqhip_rq_seam; reference modules/rqvae.py:118-139,146 with the
    last encoder Linear of modules/encoder.py:25-38 in front and the first decoder Linear + ReLU behind):

        forward(h [B,128], w_in [32,128], codebooks [L,K,32], w_out [128,32], mode, beta, grad_sink, want_scales)
            -> ids [L,B], loss [B], embs_norm [B,L], d [B,128] = relu((sum of the levels' outputs) w_out^T),
               row maxima [4,B] and column maxima [128] of d (int32 bit patterns; empty without want_scales)

    res0 = h w_in^T and the sum of the levels' outputs stay inside (saved for the backward).  Backward, composed of the same kernels:
    the decoder-side data gradient with the ReLU backward applied on load, the quantiser's closed-form backward (csrc/rq_backward.hip),
    the encoder-side data gradient with the ReLU backward of the layer below and its maxima in the epilogue (handed to the encoder
    stack's node: rqhip/linear.py:handoff_grad), and the two 32-wide weight gradients.  Same bits as running the three pieces as separate
    launches (tests/test_gpu_seam.py), which is what every other path through these layers does (rqhip/linear.py:chain_*)."""

    @staticmethod
    def forward(ctx, h: Tensor, w_in: Tensor, codebooks: Tensor, w_out: Tensor, mode: int, beta: float, grad_sink, want_scales: bool):
        from . import _lib
        from . import linear as _lin
        # (one zeroed buffer for the column maxima of this launch's output and of the backward's: one fill launch, not two)
        both = _lin.zeros_i32(2 * ops.SEAM_H, h.device) if want_scales else None
        cols = both[:ops.SEAM_H] if want_scales else None
        ctx.bwd_cols = both[ops.SEAM_H:] if (want_scales and torch.is_grad_enabled()) else None
        r = ops.rq_seam(h=h, w_in=w_in.detach(), codebooks=codebooks.detach(), mode=mode, beta=beta, w_out=w_out.detach(),
                        epilogue=_lib.EPI_RELU, want_row_max=want_scales, col_max_out=cols)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(h, w_in, codebooks, w_out, r.res0, r.ids, r.emb_sum, r.out)
        ctx.mode, ctx.beta, ctx.grad_sink, ctx.want_scales = mode, beta, grad_sink, want_scales
        # the maxima of d the decoder's split kernels scale by travel as two more (non-differentiable) outputs; empty when not wanted
        rmax = r.out_row_max if want_scales else h.new_empty((0,), dtype=torch.int32)
        cmax = cols if want_scales else h.new_empty((0,), dtype=torch.int32)
        ctx.mark_non_differentiable(r.ids, r.embs_norm, rmax, cmax)
        return r.ids, r.loss, r.embs_norm, r.out, rmax, cmax

    @staticmethod
    def backward(ctx, _g_ids, g_loss, _g_norm, g_d, _g_rmax=None, _g_cmax=None):
        from . import linear as _lin
        from .dist import claim_grad_sink
        h, w_in, codebooks, w_out, res0, ids, emb_sum, d = ctx.saved_tensors
        need_h, need_win, need_cb, need_wout = ctx.needs_input_grad[:4]
        g_d, g_loss = _dense(g_d), _dense(g_loss)
        B = h.shape[0]
        jobs = _lin.wgrad_jobs_ok(B, [tuple(w_out.shape), tuple(w_in.shape)])     # small batches: both weight gradients in ONE launch, at the end
        pending = []
        gw_out = gw_in = None
        # 1. decoder side: d = relu(s w_out^T)
        g_es = None
        if g_d is not None:
            g_dm = None
            if need_wout:
                if jobs:
                    g_dm = torch.ops.aten.threshold_backward(g_d, d, 0.0)     # (the job-table kernel takes the masked gradient as a tensor)
                    pending.append(("out", g_dm, emb_sum, claim_grad_sink(w_out), w_out))
                else:
                    sink = claim_grad_sink(w_out)
                    gw, gp, _ = _lin.weight_grad(g_d, d, emb_sum, w_out, out=sink, want_masked=False)
                    gw_out = gw.view_as(gw) if sink is not None else gw
                    g_dm = gp if (gp is not None and gp is not g_d) else None
            if need_h or need_win or need_cb:
                g_es = (_lin.chain_input_grad(g_dm, w_out)[0] if g_dm is not None
                        else _lin.chain_input_grad(g_d, w_out, g_mask=d)[0])       # the ReLU backward on load
        # 2. the quantiser (closed form)
        g_res0 = g_cb = None
        if need_h or need_win or need_cb:
            sink = ctx.grad_sink
            out_cb = None
            if need_cb and sink is not None and tuple(sink.view.shape) == tuple(codebooks.shape) and all(
                    p.grad is None and getattr(p, "_rq_sink_epoch", -1) != sink.owner.epoch for p in sink.params):
                out_cb = sink.view
                for p in sink.params:
                    p._rq_sink_epoch = sink.owner.epoch
            g_res0, g_cb = ops.rq_backward(res0, codebooks, ctx.mode, ctx.beta, ids, g_embsum=g_es, g_loss=g_loss,
                                           need_res0=need_h or need_win, need_codebooks=need_cb, out_g_codebooks=out_cb,
                                           cbgrad=ops.cbgrad_default())
            if out_cb is not None and g_cb is not None:
                g_cb = g_cb.view_as(g_cb)
        # 3. encoder side: res0 = h w_in^T, h = relu(...) of the encoder stack's last layer
        g_h = None
        if g_res0 is not None and need_win:
            if jobs:
                pending.append(("in", g_res0, h, claim_grad_sink(w_in), w_in))
            else:
                sink = claim_grad_sink(w_in)
                gw, _, _ = _lin.weight_grad(g_res0, None, h, w_in, out=sink)
                gw_in = gw.view_as(gw) if sink is not None else gw
        if pending:
            outs = [sk if sk is not None else torch.empty_like(w) for (_, _, _, sk, w) in pending]
            for (which, _, _, sk, _), gw in zip(pending, ops.linear_wgrad_jobs([(gm, a) for _, gm, a, _, _ in pending], outs=outs)):
                gw = gw.view_as(gw) if sk is not None else gw
                if which == "out":
                    gw_out = gw
                else:
                    gw_in = gw
        if g_res0 is not None and need_h:
            cols = None
            if ctx.want_scales:      # the forward's spare half, once; a second backward through a retained graph zeroes its own
                cols, ctx.bwd_cols = ctx.bwd_cols, None
                if cols is None:
                    cols = _lin.zeros_i32(ops.SEAM_H, h.device)
            g_h, sc = _lin.chain_input_grad(g_res0, w_in, out_mask=h, want_rows=ctx.want_scales, col_out=cols)
            _lin.handoff_grad(g_h, sc if ctx.want_scales else _lin.Scales())
        return g_h, gw_in, g_cb, gw_out, None, None, None, None


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
