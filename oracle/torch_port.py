"""A torch-CPU restatement of the reference's RQ-VAE training step, used ONLY as bench.py's cpu_baseline.

TEST / MEASUREMENT INFRASTRUCTURE (same rules as the rest of oracle/): never imported by the product.
The reference itself cannot travel to the GPU box, and the scalar C oracle is far slower than what the
reference actually executes on a CPU (multi-threaded MKL GEMMs), so the honest "reference CPU path" number is
a port that runs the same tensor program through the same library: encoder MLP -> L levels of
[dist = |x|^2 + |c|^2 - 2 x c^T ; argmin ; STE ; quantize loss ; residual] -> decoder -> sum-sq reconstruction
-> mean -> autograd backward -> AdamW step (reference modules/rqvae.py:141-175, modules/quantize.py:104-163,
modules/loss.py, modules/encoder.py; train_rqvae.py:185-214).  The O(B^2) p_unique_ids statistic of
rqvae.py:156-167 is included only up to `stat_rows` rows (it allocates B^2 L bytes).
"""
from __future__ import annotations

import time
from typing import List

import torch


class PortModel:
    def __init__(self, input_dim: int, hidden: List[int], embed_dim: int, n_levels: int, codebook_size: int,
                 beta: float = 0.25, seed: int = 0) -> None:
        g = torch.Generator().manual_seed(seed)
        dims = [input_dim, *hidden, embed_dim]

        def lin(i, o):
            bound = 1.0 / i ** 0.5
            return ((torch.rand(o, i, generator=g) * 2 - 1) * bound).requires_grad_(True)

        self.enc = [lin(a, b) for a, b in zip(dims[:-1], dims[1:])]
        rdims = dims[::-1]
        self.dec = [lin(a, b) for a, b in zip(rdims[:-1], rdims[1:])]
        self.codebooks = [(torch.randn(codebook_size, embed_dim, generator=g) * (0.3 / (l + 1))).requires_grad_(True)
                          for l in range(n_levels)]
        self.beta = beta

    def parameters(self):
        return [*self.codebooks, *self.enc, *self.dec]

    @staticmethod
    def _mlp(ws, h):
        for i, w in enumerate(ws):
            h = h @ w.t()
            if i != len(ws) - 1:
                h = torch.relu(h)
        return h

    def quantize_level(self, x, cb):
        dist = (x * x).sum(1, keepdim=True) + (cb * cb).sum(1).unsqueeze(0) - (2 * x) @ cb.t()
        ids = dist.detach().argmin(dim=1)
        emb = cb[ids]
        out = x + (emb - x).detach()
        loss = ((x.detach() - emb) ** 2).sum(-1) + self.beta * ((x - emb.detach()) ** 2).sum(-1)
        return out, ids, loss

    def step_loss(self, x, stat_rows: int = 4096):
        res = self._mlp(self.enc, x)
        total, qloss, ids = 0, 0, []
        for cb in self.codebooks:
            out, i, l = self.quantize_level(res, cb)
            res = res - out
            total = total + out
            qloss = qloss + l
            ids.append(i)
        x_hat = self._mlp(self.dec, total)
        recon = ((x_hat - x) ** 2).sum(-1)
        loss = (recon + qloss).mean()
        with torch.no_grad():
            s = torch.stack(ids, dim=1)[:stat_rows]
            dup = torch.triu((s.unsqueeze(1) == s.unsqueeze(0)).all(-1), diagonal=1)
            p_unique = (~dup).all(dim=1).sum() / s.shape[0]
        return loss, p_unique


def time_training_steps(x: torch.Tensor, steps: int, warmup: int = 1, **model_kw) -> dict:
    """items/s of fwd+bwd+AdamW on the host cores; x [B, input_dim] fp32 CPU tensor."""
    model = PortModel(input_dim=x.shape[1], **model_kw)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for i in range(warmup + steps):
        if i == warmup:
            t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = model.step_loss(x)
        loss.backward()
        opt.step()
    dt = time.perf_counter() - t0
    return {"items_per_s": x.shape[0] * steps / dt, "seconds": dt, "steps": steps, "rows": x.shape[0],
            "threads": torch.get_num_threads(), "final_loss": float(loss.detach())}
