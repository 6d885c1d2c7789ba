"""AdamW over all parameters in ONE kernel launch (csrc/adamw.hip) -- the update `train_rqvae.py:136-138` of the reference asks for
(`AdamW(params=model.parameters(), lr, weight_decay)`: decoupled weight decay on every parameter, codebooks included).

A `torch.optim.Optimizer` with AdamW's constructor, `param_groups` and per-parameter state (`step`, `exp_avg`, `exp_avg_sq`), so that
`optimizer.state_dict()` / `load_state_dict()` -- the `"optimizer"` entry of the reference's checkpoints (train_rqvae.py:260-265) -- are
interchangeable with `torch.optim.AdamW`'s in both directions.  What differs is the launch: torch's fused kernel works on 64 K-element chunks
(18 workgroups for this model's 1.15 M parameters: 40-46 us per step, plus a launch that bumps the step counters); here 1 100 workgroups
update everything in a few microseconds behind a one-thread kernel that bumps the one device-side step counter -- which also makes the step replayable from a
captured hipGraph without torch's `capturable` machinery.  The hyper-parameters are kernel arguments: a captured graph replays the values
it was captured with (train_rqvae.py re-captures after every eager excursion; it has no scheduler)."""
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import RqHipError, check
from .ops import _RAW_DEVICE, _stream as ops_stream


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, amsgrad: bool = False, *,
                 maximize: bool = False, foreach: Optional[bool] = None, capturable: bool = False, differentiable: bool = False,
                 fused: Optional[bool] = None) -> None:
        if amsgrad or maximize or differentiable:
            raise ValueError("FlatAdamW implements plain AdamW (no amsgrad / maximize / differentiable)")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameters")
        del foreach, capturable, fused          # accepted for signature compatibility: one kernel, always graph-safe
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                        capturable=True, differentiable=False, fused=True)
        super().__init__(params, defaults)
        self._steps = {}        # group index -> the device float scalar every parameter of the group shares as state["step"]
        self._scratch = None
        self._cache = {}

    def _shared_step(self, gi: int, group) -> torch.Tensor:
        """One step counter per group, on the device; parameters loaded from a torch AdamW checkpoint bring their own (all equal).
        CONTRACT (differs from torch.optim.AdamW, which counts per parameter): every parameter of a group takes part in every step() --
        a parameter whose .grad is None on some steps, or that receives its first gradient later than the others, is bias-corrected
        with the GROUP's count.  The RQ-VAE training loops of this package (every parameter gets a gradient every step) satisfy it."""
        st = self._steps.get(gi)
        live = [self.state[p]["step"] for p in group["params"] if p in self.state and "step" in self.state[p]]
        if st is None or any(s is not st for s in live):
            dev = group["params"][0].device
            start = max([float(s) for s in live if s is not st] + ([float(st)] if st is not None else [0.0])) if live else 0.0   # (host sync)
            st = torch.full((), start, dtype=torch.float32, device=dev)
            self._steps[gi] = st
            for p in group["params"]:
                if p in self.state:
                    self.state[p]["step"] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        l = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise RqHipError("FlatAdamW updates ROCm device parameters (csrc/adamw.hip); use torch.optim.AdamW for host tensors")
            # the same parameters and the same (aligned) gradient buffers as last step -- every step of a training loop whose gradients
            # live in rqhip.dist.FlatGradReducer's flat buffer: the argument arrays, the validation and the state lookups of the first
            # such step stand (load_state_dict drops them); 11 parameters x 5 Python loops were a third of an eager step's optimizer time
            fast = (tuple([p.data_ptr() for p in ps]), tuple([p.grad.data_ptr() for p in ps]))
            arrs = self._cache.get(gi)
            step = self._steps.get(gi)
            if arrs is None or arrs[0] != fast or step is None:
                for p in ps:
                    if p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                        raise RqHipError("FlatAdamW: parameters and gradients must be contiguous float32 tensors")
                    st = self.state[p]
                    if "exp_avg" not in st:
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["step"] = self._steps.get(gi, torch.zeros((), dtype=torch.float32, device=dev))
                        self._steps.setdefault(gi, st["step"])
                step = self._shared_step(gi, group)
                # (the kernel reads 16 bytes at a time: a gradient that is a misaligned view of someone's packed buffer is copied once
                # per step; rqhip.dist.FlatGradReducer pads its slices, so its views never take this branch)
                direct = all(p.grad.is_contiguous() and p.grad.data_ptr() % 16 == 0 for p in ps)
                grads = [p.grad if (p.grad.is_contiguous() and p.grad.data_ptr() % 16 == 0) else p.grad.clone(memory_format=torch.contiguous_format)
                         for p in ps]
                n = len(ps)
                vp = C.c_void_p * n
                arrs = (fast if direct else None, vp(*[p.data_ptr() for p in ps]), vp(*[g.data_ptr() for g in grads]),
                        vp(*[self.state[p]["exp_avg"].data_ptr() for p in ps]), vp(*[self.state[p]["exp_avg_sq"].data_ptr() for p in ps]),
                        (C.c_int64 * n)(*[p.numel() for p in ps]), n, grads)      # (grads: keeps copies alive until the launch)
                self._cache[gi] = arrs
            if self._scratch is None or self._scratch.device != dev:
                self._scratch = torch.zeros((2,), dtype=torch.float32, device=dev)
            b1, b2 = group["betas"]

            def launch():
                check(l.rqhip_adamw_step(arrs[1], arrs[2], arrs[3], arrs[4], arrs[5], arrs[6], step.data_ptr(), self._scratch.data_ptr(),
                                         float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                         ops_stream()), "rqhip_adamw_step")
            if _RAW_DEVICE is not None and dev.index == _RAW_DEVICE():     # (the device context manager costs more than the launch)
                launch()
            else:
                with torch.cuda.device(dev):
                    launch()
        return loss

    def state_dict(self):
        """torch.optim.AdamW's format.  Every parameter gets its OWN copy of the step counter: the live one is shared by the group, and a
        torch AdamW that loaded aliased counters would advance the shared tensor once per parameter."""
        sd = super().state_dict()
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone()
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._steps, self._cache = {}, {}
        # the loaded per-parameter counters (all equal in an AdamW checkpoint) become the group's shared device scalar NOW -- reading
        # them back is a host sync, which the next step() may not do (it can run under hipGraph capture)
        for gi, group in enumerate(self.param_groups):
            if any(p in self.state and "step" in self.state[p] for p in group["params"]):
                self._shared_step(gi, group)
