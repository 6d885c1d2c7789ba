"""Bias-free Linear+ReLU stack (API and state_dict keys of reference modules/encoder.py:7-38).

On ROCm tensors the whole run of Linear (+ ReLU) layers is ONE autograd node (`_MLPStack`).  Its large GEMMs (output width a
multiple of 256, batches of 4096 rows and more) run on the 16-bit matrix cores with fp32's accuracy (csrc/gemm_split.hip,
csrc/wgrad_split.hip: two fp16 pieces per operand under exact power-of-two row / column scales, three piece products, fp32
accumulation; no less exact than the library's fp32 GEMM, tests/test_gpu_gemm_split.py, tests/test_gpu_wgrad.py) with the
ReLU, the ReLU backward and the reconstruction loss in their epilogues; the 128 <-> 32 layers either side of the quantiser are the seam
kernel's GEMMs (csrc/rq_forward.hip).  Batches below 4096 rows -- the reference's 640 / 64 -- run every layer's forward and data gradient
on csrc/mlp_small.hip (exact fp32 on the fp32 matrix instruction, ReLU / ReLU backward of the layer below in the epilogue) and all weight
gradients of a stack as one job-table launch (csrc/wgrad_jobs.hip).  Library GEMMs on PyTorch-ROCm (fp32, `rqhip/tuning.py` picks the
kernels; a Linear followed by a ReLU is ONE hipBLASLt call, `torch._addmm_activation` with a zero bias) are left for shapes no kernel
tiles (a width that is not a multiple of 32) and for the strict-fp32 arm.  Large-batch weight gradients: csrc/wgrad_split.hip / csrc/wgrad.hip
(SURVEY section 8 row f2).  Parameter names are `mlp.{0,2,4,...}.weight`, as in the reference, so checkpoints load in both
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


class _LinearReLU(torch.autograd.Function):
    """relu(x @ w.T) for 2-D fp32 tensors, one layer (the registered operators, stacks with dropout, CPU): ReLU fused into the
    GEMM epilogue; backward by rqhip/linear.py:backward -- weight gradient with the ReLU mask, data gradient on the masked
    gradient it hands over; shapes no kernel tiles keep the three library kernels autograd would run (mask, two GEMMs)."""

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
        gx, gw = _lin.backward(gy, y, x, w, need_x, need_w, sink)
        return gx, (_adopt(gw, sink) if need_w else None), None


class _LinearPlain(torch.autograd.Function):
    """x @ w.T (the last layer of each MLP: no ReLU), one layer."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor) -> Tensor:
        ctx.save_for_backward(x, w)
        return _lin.forward(x, w, False)

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, w = ctx.saved_tensors
        need_x, need_w = ctx.needs_input_grad
        sink = _grad_sink(w) if need_w else None
        gx, gw = _lin.backward(gy, None, x, w, need_x, need_w, sink)
        return gx, (_adopt(gw, sink) if need_w else None)


class _MLPStack(torch.autograd.Function):
    """A whole bias-free Linear(+ReLU) stack as ONE autograd node, optionally ending in the reconstruction loss
    (reference modules/encoder.py:25-38; with `target`: modules/rqvae.py:146,152 + modules/loss.py:5-10, the last layer and
    the loss in one kernel -- x_hat never reaches memory).  What the single node buys over one node per layer:
      * the weight images of every forward and data-gradient GEMM of the stack are built by one launch when the forward
        starts;
      * the power-of-two scales of the fp16 split kernels travel with the data: every GEMM epilogue emits the row maxima
        (for the GEMM that reads its output next) and the column maxima (for the weight-gradient kernel that does) of what
        it stores, so the only maxima PASS of a training step is the one over the input batch;
      * a data gradient applies the ReLU backward of the layer below in its epilogue (RQHIP_EPI_MASK), so the masked
        gradient is written once, with its maxima, and the weight-gradient kernels read it unmasked.
    Same kernels, maxima and therefore result bits as the per-layer Functions above (tests/test_gpu_modules.py).
    forward(x, target, relus, zero_bias, *weights): relus[i] = layer i is followed by a ReLU; zero_bias(n, like) -> zeros
    [n] for the library GEMM's fused-ReLU call.  Returns the last layer's output, or the loss rows [M] when target is given
    (then the last layer has no ReLU and takes the split kernel: the caller checks)."""

    @staticmethod
    def forward(ctx, x: Tensor, target, relus, zero_bias, *weights):
        from rqhip import _lib
        from rqhip import autograd as _ag
        n, M = len(weights), x.shape[0]
        given = _lin.take_scales()          # maxima that came with the input (rqhip/linear.py:attach_scales), or None
        need_w = [bool(f) for f in ctx.needs_input_grad[4:]]
        need_in = [bool(ctx.needs_input_grad[0]) or any(need_w[:i]) for i in range(n)]   # gradient wrt layer i's input wanted
        # the 128 <-> 32 layers either side of the quantiser: the seam kernel's GEMMs (rqhip/linear.py:chain_*), forward and data gradient
        aligned = x.data_ptr() % 16 == 0 and M > 0
        chain = [(_lin.chain_shape(w.shape[0], w.shape[1], M) if aligned else 0) for w in weights]
        chain = [0 if (k == 1 and relus[i]) else k for i, k in enumerate(chain)]
        # batches below 4096 rows: forward and data gradient on csrc/mlp_small.hip (exact fp32; ReLU / ReLU backward in the epilogue)
        small = [aligned and w.is_contiguous() and w.data_ptr() % 16 == 0 and _lin.small_shape_ok(M, w.shape[0], w.shape[1])
                 and _lin.small_shape_ok(M, w.shape[1], w.shape[0]) for w in weights]
        fwd_split = [not chain[i] and _lin.split_shape_ok(M, w.shape[0], w.shape[1]) for i, w in enumerate(weights)]
        dg_split = [not chain[i] and need_in[i] and _lin.split_shape_ok(M, w.shape[1], w.shape[0]) for i, w in enumerate(weights)]
        wg_f16 = [need_w[i] and _lin.wgrad_f16_ok(w.shape[0], w.shape[1], M) for i, w in enumerate(weights)]
        jobs = [(w, False) for i, w in enumerate(weights) if fwd_split[i]] + [(w, True) for i, w in enumerate(weights) if dg_split[i]]
        imgs = iter(_lin.images(jobs))
        img_f = [next(imgs) if fwd_split[i] else None for i in range(n)]
        img_t = [next(imgs) if dg_split[i] else None for i in range(n)]
        f16 = _lin.f16()
        # column maxima the epilogues emit in this forward: of layer i's output when layer i + 1's weight gradient wants them
        # (and of the reconstruction gradient for the last layer's own weight gradient) -- one zeroed arena
        emit = [f16 and (fwd_split[i] or chain[i] == 2) and ((i + 1 < n and wg_f16[i + 1]) or (i + 1 == n and target is not None and wg_f16[i]))
                for i in range(n)]
        arena = _lin.zeros_i32(sum(w.shape[0] for i, w in enumerate(weights) if emit[i]), x.device) if any(emit) else None
        off = 0
        acts, scs = [x], [given if given is not None else _lin.Scales()]
        if f16 and (fwd_split[0] or wg_f16[0]):   # the input batch: the one maxima pass of the step -- unless its maxima came with it
            _lin.ensure_scales(x, scs[0], fwd_split[0], wg_f16[0])
        out = g_recon = g_scales = None
        for i, w in enumerate(weights):
            a, sc, last = acts[-1], scs[-1], i + 1 == n
            col_out = None
            if emit[i]:
                col_out = arena[off:off + w.shape[0]]
                off += w.shape[0]
            if last and target is not None:
                one = torch.tensor(1.0, dtype=torch.float32)   # fp32 (loss scale) * fp32 (1 / B), as ReconLossFunction
                ctx.row_scale = float(torch.tensor(_ag._LOSS_SCALE, dtype=torch.float32) * (one / M))
                g_recon, out, g_scales = _lin.gemm(a, img_f[i], w.shape[0], epilogue=_lib.EPI_RECON, aux=target,
                                                   row_scale=ctx.row_scale, a_scales=sc, want_rows=dg_split[i], col_out=col_out)
                break
            if chain[i]:
                y, ysc = _lin.chain_forward(a, w, relus[i], want_rows=not last and fwd_split[i + 1], col_out=col_out)
            elif fwd_split[i]:
                y, _, ysc = _lin.gemm(a, img_f[i], w.shape[0], epilogue=_lib.EPI_RELU if relus[i] else _lib.EPI_STORE,
                                      a_scales=sc, want_rows=not last and fwd_split[i + 1], col_out=col_out)
            elif small[i]:
                y, ysc = _lin.small_forward(a, w, relus[i]), _lin.Scales()
            else:
                y, ysc = _lin.library_forward(a, w, relus[i], zero_bias(w.shape[0], a) if relus[i] else None), _lin.Scales()
            acts.append(y)
            scs.append(ysc)
            out = y
        # (the OUTPUT goes through save_for_backward: as a plain attribute it would close a reference cycle output -> node ->
        # ctx -> output and keep the activations alive until the garbage collector runs; intermediates carry no grad_fn)
        ctx.has_target = target is not None
        ctx.save_for_backward(x, *weights, target if ctx.has_target else out)
        ctx.acts_mid, ctx.scs, ctx.relus = (acts[1:] if ctx.has_target else acts[1:-1]), scs, tuple(relus)
        ctx.img_t, ctx.dg_split, ctx.wg_f16, ctx.need_in, ctx.need_w, ctx.chain = img_t, dg_split, wg_f16, need_in, need_w, chain
        ctx.small = small
        ctx.defer = _lin.take_defer_flag()
        ctx.g_recon, ctx.g_scales, ctx.consumed = g_recon, g_scales, False
        return out

    @staticmethod
    def backward(ctx, g_out: Tensor):
        from rqhip import _lib
        x, weights, tail = ctx.saved_tensors[0], ctx.saved_tensors[1:-1], ctx.saved_tensors[-1]
        n = len(weights)
        target = tail if ctx.has_target else None
        acts = [x] + list(ctx.acts_mid) + ([] if ctx.has_target else [tail])
        scs, relus, need_in, need_w = ctx.scs, ctx.relus, ctx.need_in, ctx.need_w
        dg_split, wg_f16, f16, chain, small = ctx.dg_split, ctx.wg_f16, _lin.f16(), ctx.chain, ctx.small
        g_out = g_out.contiguous()
        handed = _lin.take_grad_handoff(g_out) if target is None else None
        if handed is not None:       # the node above (modules/rqvae.py's seam) masked this gradient by our last ReLU and took its maxima
            g, gsc = g_out, handed
        elif target is None:
            g, gsc = g_out, _lin.Scales()
        elif not ctx.consumed:
            ctx.consumed = True
            gsc = ctx.g_scales       # rows whose upstream gradient is not the announced one are rescaled in place, maxima too
            g = ops.recon_rescale_rows(ctx.g_recon, g_out, ctx.row_scale, gsc.rows, gsc.cols)
        else:                        # a second backward through a retained graph: x_hat is recomputed with the library
            g = ops.recon_loss_backward(acts[-1].mm(weights[-1].t()), target, g_out, True, False)[0]
            gsc = _lin.Scales()
        # column maxima the data-gradient epilogues emit: of the gradient wrt layer i - 1's output (masked) when that
        # layer's weight gradient wants them
        emit = [f16 and (dg_split[i] or chain[i] == 1) and i > 0 and wg_f16[i - 1] for i in range(n)]
        arena = _lin.zeros_i32(sum(w.shape[1] for i, w in enumerate(weights) if emit[i]), g.device) if any(emit) else None
        off = 0
        premasked = (not relus[n - 1]) or handed is not None      # is g already masked by this layer's ReLU (or is there none)?
        gws = [None] * n
        # batches below the split kernels' row count: every weight gradient of the stack in ONE launch after the data gradients
        # (csrc/wgrad_jobs.hip; same bits as the per-layer path, which runs the same kernel with one job)
        deferred = g.is_cuda and _lin.wgrad_jobs_ok(g.shape[0], [tuple(w.shape) for i, w in enumerate(weights) if need_w[i]])
        pending = []
        # split-kernel batches: the weight gradients of the layers tiled 256 x 256 whose gradient arrives masked wait for ONE launch at the
        # end of the stack (rqhip_linear_wgrad_f16_batch: one workgroup's partial block per CU for all of them instead of per layer)
        batched = []
        for i in range(n - 1, -1, -1):
            w, a = weights[i], acts[i]
            y = acts[i + 1] if (relus[i] and not premasked) else None
            # a 32 -> 128 seam layer applies its own ReLU backward on LOAD in its data gradient: the masked gradient is then only
            # written out when the job-table weight gradient needs it as a tensor
            mask_on_load = chain[i] == 2 and need_in[i] and y is not None and g.data_ptr() % 16 == 0
            g_unmasked = g
            if need_w[i] and deferred:
                if y is not None:
                    g, gsc = torch.ops.aten.threshold_backward(g, y, 0.0), _lin.Scales()
                pending.append((i, g, a, _grad_sink(w)))
            elif need_w[i] and y is None and g.is_cuda and wg_f16[i] and _lin.wgrad_batch_shape_ok(w.shape[0], w.shape[1], g.shape[0]):
                gsc = _lin.ensure_scales(g, gsc, False, True)
                batched.append((i, g, a, gsc.cols, _lin.ensure_scales(a, scs[i], False, True).cols, _grad_sink(w)))
            elif need_w[i]:
                sink = _grad_sink(w)
                gw, g, gsc = _lin.weight_grad(g, y, a, w, out=sink, want_masked=need_in[i] and not mask_on_load, g_scales=gsc,
                                              x_scales=scs[i], premasked=premasked)
                gws[i] = _adopt(gw, sink)
            elif y is not None and need_in[i] and not mask_on_load:
                g, gsc = torch.ops.aten.threshold_backward(g, y, 0.0), _lin.Scales()
            if not need_in[i]:
                g = None
                break
            lower_relu = i > 0 and relus[i - 1]
            if chain[i] and (g_unmasked if mask_on_load else g).data_ptr() % 16 == 0:
                col_out = None
                if emit[i]:
                    col_out = arena[off:off + w.shape[1]]
                    off += w.shape[1]
                if chain[i] == 2:     # 32 -> 128 layer: gx [M, 32]; its own ReLU backward on load when the gradient is still unmasked
                    src, msk = (g_unmasked, y) if (mask_on_load and (g is None or g is g_unmasked)) else (g, None)
                    g, gsc = _lin.chain_input_grad(src, w, g_mask=msk)
                    premasked = not lower_relu
                else:                 # 128 -> 32 layer: gx [M, 128], the ReLU backward of the layer below in the epilogue, with its maxima
                    g, gsc = _lin.chain_input_grad(g, w, out_mask=a if lower_relu else None,
                                                   want_rows=i > 0 and dg_split[i - 1], col_out=col_out)
                    premasked = True
            elif dg_split[i]:
                col_out = None
                if emit[i]:
                    col_out = arena[off:off + w.shape[1]]
                    off += w.shape[1]
                fuse = f16 and lower_relu              # the ReLU backward of the layer below in this GEMM's epilogue
                g, _, gsc = _lin.gemm(g, ctx.img_t[i], w.shape[1], epilogue=_lib.EPI_MASK if fuse else _lib.EPI_STORE,
                                      aux=a if fuse else None, a_scales=gsc,
                                      want_rows=i > 0 and dg_split[i - 1] and (fuse or not lower_relu),
                                      col_out=col_out if (fuse or not lower_relu) else None)
                premasked = fuse or not lower_relu
            elif small[i] and g.data_ptr() % 16 == 0:      # the ReLU backward of the layer below in the epilogue: no mask launch
                g, gsc = _lin.small_input_grad(g, w, a if lower_relu else None), _lin.Scales()
                premasked = True
            else:
                g, gsc = g.mm(w), _lin.Scales()
                premasked = not lower_relu
        waiting = _lin.xstack_take() if g_out.is_cuda else []      # an earlier stack's weight gradients that waited for this launch
        if batched or waiting:
            M = g_out.shape[0]
            sunk = all(sk is not None for *_, sk in batched)
            if batched and not waiting and ctx.defer and sunk and _lin.xstack_ok():
                # (this stack's turn to wait: a later node, or the engine's end-of-backward callback, launches them into the flat buffer)
                _lin.xstack_push([(weights[i], gm, a, gc, xc, sk) for i, gm, a, gc, xc, sk in batched])
                for i, *_, sk in batched:
                    gws[i] = _adopt(sk, sk)
            elif sunk:
                _lin._launch_wgrads(waiting + [(weights[i], gm, a, gc, xc, sk) for i, gm, a, gc, xc, sk in batched])
                for i, *_, sk in batched:
                    gws[i] = _adopt(sk, sk)
            else:       # (some gradient has no slice of a flat buffer to land in: fresh tensors, launched now)
                _lin._launch_wgrads(waiting)
                full = [b for b in batched if weights[b[0]].shape[0] % 256 == 0 and weights[b[0]].shape[1] % 256 == 0]
                for group in (full, [b for b in batched if b not in full]):
                    if len(group) >= 2 and ops.linear_wgrad_f16_batch_ranges(M, [tuple(weights[i].shape) for i, *_ in group]) >= 1:
                        dws = ops.linear_wgrad_f16_batch([(gm, a, gc, xc) for _, gm, a, gc, xc, _ in group], outs=[sk for *_, sk in group])
                        for (i, *_, sk), gw in zip(group, dws):
                            gws[i] = _adopt(gw, sk)
                    else:
                        for i, gm, a, gc, xc, sk in group:
                            gw, _, _ = _lin.weight_grad(gm, None, a, weights[i], out=sk, want_masked=False, g_scales=_lin.Scales(None, gc),
                                                        x_scales=_lin.Scales(None, xc), premasked=True)
                            gws[i] = _adopt(gw, sk)
        waiting_small = _lin.xsmall_take() if g_out.is_cuda else []      # an earlier stack's job-table weight gradients
        if pending or waiting_small:
            sunk = all(sk is not None for *_, sk in pending)
            if pending and not waiting_small and ctx.defer and sunk and _lin.xsmall_ok():
                _lin.xsmall_push([(gm, a, sk) for _, gm, a, sk in pending])      # (a later stack's launch, or the end-of-backward callback)
                for i, *_, sk in pending:
                    gws[i] = _adopt(sk, sk)
            else:
                outs = [sk if sk is not None else torch.empty_like(weights[i]) for (i, _, _, sk) in pending]
                _lin._launch_small(waiting_small + [(gm, a, o) for (_, gm, a, _), o in zip(pending, outs)])
                for (i, _, _, sk), o in zip(pending, outs):
                    gws[i] = _adopt(o, sk)
        return (g if ctx.needs_input_grad[0] else None), None, None, None, *gws


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
        return self._run(x if x.is_contiguous() else x.contiguous(), list(self.mlp))

    def _run(self, x: Tensor, layers, target: Tensor = None) -> Tensor:
        if torch_ops.enabled():
            return self._run_layerwise(x, layers)
        # the leading run of bias-free Linear (+ ReLU) layers is one autograd node; what follows (Identity, L2 norm) is applied
        # after it; a Dropout inside the run ends it (the rest goes layer by layer)
        weights, relus, i = [], [], 0
        while i < len(layers) and isinstance(layers[i], nn.Linear) and layers[i].bias is None:
            relu = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
            weights.append(layers[i].weight)
            relus.append(relu)
            i += 2 if relu else 1
        if weights:
            _lin.handoff_scales(_lin.attached_scales(x))
            _lin.mark_next_stack_defers(getattr(self, "_defer_wgrads", False))
            try:
                x = _MLPStack.apply(x, target, tuple(relus), self._zero_bias, *weights)
            finally:
                _lin.mark_next_stack_defers(False)
        return self._run_layerwise(x, layers[i:]) if i < len(layers) else x

    def _run_layerwise(self, x: Tensor, layers) -> Tensor:
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

    # ---- the RQ <-> MLP seam (modules/rqvae.py): the stack without its last Linear / without its first Linear + ReLU -----------------
    def seam_tail_weight(self):
        """The weight of the LAST Linear when the stack ends `..., ReLU, Linear(128 -> 32, no bias), Identity` (no dropout): the layer
        the seam kernel runs in front of the quantiser; else None."""
        layers = list(self.mlp)
        if (self.dropout == 0 and len(layers) >= 4 and isinstance(layers[-1], nn.Identity) and isinstance(layers[-2], nn.Linear)
                and layers[-2].bias is None and isinstance(layers[-3], nn.ReLU)
                and tuple(layers[-2].weight.shape) == (_lin.CHAIN_D, _lin.CHAIN_H)
                and all((isinstance(l, nn.Linear) and l.bias is None) or isinstance(l, nn.ReLU) for l in layers[:-1])):
            return layers[-2].weight
        return None

    def seam_head_weight(self):
        """The weight of the FIRST Linear when the stack starts `Linear(32 -> 128, no bias), ReLU, Linear, ...` (no dropout); else None."""
        layers = list(self.mlp)
        if (self.dropout == 0 and len(layers) >= 4 and isinstance(layers[0], nn.Linear) and layers[0].bias is None
                and isinstance(layers[1], nn.ReLU) and isinstance(layers[2], nn.Linear)
                and tuple(layers[0].weight.shape) == (_lin.CHAIN_H, _lin.CHAIN_D)
                and all((isinstance(l, nn.Linear) and l.bias is None) or isinstance(l, nn.ReLU) for l in layers[:-1])):
            return layers[0].weight
        return None

    def run_before_tail(self, x: Tensor) -> Tensor:
        """The hidden activation in front of the last Linear (through its ReLU)."""
        return self._run(x if x.is_contiguous() else x.contiguous(), list(self.mlp)[:-2])

    def run_after_head(self, d: Tensor) -> Tensor:
        """The rest of the stack behind the first Linear + ReLU."""
        return self._run(d if d.is_contiguous() else d.contiguous(), list(self.mlp)[2:])

    def reconstruction_rows(self, z: Tensor, target: Tensor, first: int = 0):
        """ReconstructionLoss(self(z), target) per row with the last layer and the loss fused (`_MLPStack` with a target), or
        None when that kernel does not apply here (the caller then composes the two, as the reference does).  The fused
        epilogue writes the gradient matrix the backward is going to ask for, so it only runs when a backward can follow.
        first: index of the first module of the stack to run (2: z is already behind the first Linear + ReLU -- the seam)."""
        layers = list(self.mlp)[first:]
        lin = [l for l in layers[:-1]]
        last = layers[-2] if len(layers) >= 2 else None
        chain_ok = all((isinstance(l, nn.Linear) and l.bias is None) or isinstance(l, nn.ReLU) for l in lin) \
            and not any(isinstance(a, nn.ReLU) and isinstance(b, nn.ReLU) for a, b in zip(lin, lin[1:]))
        if not (isinstance(last, nn.Linear) and last.bias is None and isinstance(layers[-1], nn.Identity) and chain_ok
                and isinstance(layers[0], nn.Linear)
                and not torch_ops.enabled() and z.is_cuda and z.dim() == 2 and z.dtype == torch.float32
                and torch.is_grad_enabled() and (z.requires_grad or any(l.weight.requires_grad for l in layers if isinstance(l, nn.Linear)))
                and target.is_cuda and target.dtype == torch.float32 and not target.requires_grad
                and tuple(target.shape) == (z.shape[0], last.out_features) and target.is_contiguous()
                and target.data_ptr() % 16 == 0
                and last.out_features % 256 == 0   # (the fused epilogue sums four 64-column waves: 256-column tiles only)
                and _lin.split_shape_ok(z.shape[0], last.out_features, last.in_features)):
            return None
        assert z.shape[-1] == layers[0].in_features, f"Invalid input dim: Expected {layers[0].in_features}, found {z.shape[-1]}"
        return self._run(z if z.is_contiguous() else z.contiguous(), layers[:-1], target)
