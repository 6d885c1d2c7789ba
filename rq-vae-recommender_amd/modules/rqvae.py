"""RQ-VAE on the MI355X behind the reference's `RqVae` API (reference modules/rqvae.py).

Same constructor, attributes, state_dict keys (`layers.{l}.embedding.weight`, `encoder.mlp.*`,
`decoder.mlp.*`), `get_semantic_ids` / `forward` signatures and return types.  What differs is where the
arithmetic runs: the whole residual-quantisation stack -- every level's distance, argmin, codeword gather,
STE / rotation-trick output, quantize loss and residual update, plus the emb-sum / emb-norm the loss
consumes -- is ONE fused HIP kernel (csrc/rq_forward.hip) with a closed-form HIP backward
(csrc/rq_backward.hip); the O(B^2) duplicate statistic of rqvae.py:159-167 is a hash pass (csrc/ids.hip).  At batches of 4096
rows and more `forward` runs the encoder's last Linear, every level and the decoder's first Linear + ReLU as ONE launch
(rqhip_rq_seam: `_seam_weights`, rqhip/autograd.py:RqSeamFunction); the encoder / decoder MLPs around it are
modules/encoder.py's split-fp16 matrix kernels with the reconstruction loss in the last GEMM's epilogue.

Differences a caller can observe, all deliberate:
  * no `@torch.compile(mode="reduce-overhead")` on forward (rqvae.py:141): the hot path is already a
    handful of launches and a ctypes call cannot be traced by dynamo;
  * tensors must be on a ROCm device; a CPU call raises (no fallback path exists).
"""
from functools import cached_property
from typing import List, NamedTuple

import torch
from huggingface_hub import PyTorchModelHubMixin
from torch import Tensor, nn

from data.schemas import SeqBatch
from modules.encoder import MLP
from modules.loss import CategoricalReconstuctionLoss, ReconstructionLoss
from modules.normalize import l2norm
from modules.quantize import Quantize, QuantizeForwardMode
from rqhip import ops, torch_ops
from rqhip import linear as _lin
from rqhip.autograd import LossMeansFunction, RqSeamFunction, RqStackFunction

# The reference sets "high" here (rqvae.py:19): on its CPU path that is plain fp32 (bit-identical to "highest",
# SURVEY probe 3), but on ROCm "high" switches the MLP GEMMs to a reduced-precision tf32 class.  Parity is judged
# against the CPU results, so the GPU path pins true fp32; rqhip/tuning.py recovers the speed by kernel selection.
torch.set_float32_matmul_precision("highest")


# A/B and test switch: False runs the same layers as separate launches (the 128 <-> 32 GEMMs of rqhip/linear.py:chain_*, the stack kernel
# between them) -- bit-identical results (tests/test_gpu_seam.py), one launch more on either side of the quantiser
FUSE_SEAM = True


_SIDE_STREAMS = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


class RqVaeOutput(NamedTuple):
    embeddings: Tensor     # [B, D, L]
    residuals: Tensor      # [B, D, L]
    sem_ids: Tensor        # [B, L] int64
    quantize_loss: Tensor  # [B]


class RqVaeComputedLosses(NamedTuple):
    loss: Tensor
    reconstruction_loss: Tensor
    rqvae_loss: Tensor
    embs_norm: Tensor      # [B, L]
    p_unique_ids: Tensor


class _StackResult(NamedTuple):
    embs: Tensor       # [L,B,D] (empty when levels were not requested)
    residuals: Tensor  # [L,B,D] (")
    ids: Tensor        # [L,B]
    loss: Tensor       # [B]
    emb_sum: Tensor    # [B,D]
    embs_norm: Tensor  # [B,L]


class RqVae(nn.Module, PyTorchModelHubMixin):
    def __init__(
        self,
        input_dim: int,
        embed_dim: int,
        hidden_dims: List[int],
        codebook_size: int,
        codebook_kmeans_init: bool = True,
        codebook_normalize: bool = False,
        codebook_sim_vq: bool = False,
        codebook_mode: QuantizeForwardMode = QuantizeForwardMode.GUMBEL_SOFTMAX,
        n_layers: int = 3,
        commitment_weight: float = 0.25,
        n_cat_features: int = 18,
    ) -> None:
        self._config = locals()
        super().__init__()
        self.input_dim = input_dim
        self.embed_dim = embed_dim
        self.hidden_dims = hidden_dims
        self.n_layers = n_layers
        self.codebook_size = codebook_size
        self.commitment_weight = commitment_weight
        self.n_cat_feats = n_cat_features

        # construction order (levels, encoder, decoder) == the reference's, so a seeded init matches it
        self.layers = nn.ModuleList(
            Quantize(embed_dim=embed_dim, n_embed=codebook_size, forward_mode=codebook_mode,
                     do_kmeans_init=codebook_kmeans_init, codebook_normalize=(level == 0 and codebook_normalize),
                     sim_vq=codebook_sim_vq, commitment_weight=commitment_weight)
            for level in range(n_layers))
        self.encoder = MLP(input_dim=input_dim, hidden_dims=hidden_dims, out_dim=embed_dim,
                           normalize=codebook_normalize)
        self.decoder = MLP(input_dim=embed_dim, hidden_dims=hidden_dims[::-1], out_dim=input_dim, normalize=False)
        # (the decoder's backward runs before the encoder's: its batched weight gradients may wait for the encoder's launch, rqhip/linear.py)
        self.decoder._defer_wgrads = True
        self.reconstruction_loss = (CategoricalReconstuctionLoss(n_cat_features) if n_cat_features != 0
                                    else ReconstructionLoss())

    @cached_property
    def config(self) -> dict:
        return self._config

    def _first_param(self) -> Tensor:
        """The encoder's first parameter.  `next(self.encoder.parameters())` walks the module tree through three generators (7 us, twice per
        forward: tools/eager_host_profile.py); the Parameter object itself survives `.to()` / `load_state_dict`, so it is looked up once."""
        p = self.__dict__.get("_p0")
        if p is None:
            p = next(self.encoder.parameters())
            self.__dict__["_p0"] = p
        return p

    @property
    def device(self) -> torch.device:
        return self._first_param().device

    def load_pretrained(self, path: str) -> None:
        state = torch.load(path, map_location=self.device, weights_only=False)
        self.load_state_dict(state["model"])
        print(f"---Loaded RQVAE Iter {state['iter']}---")

    def encode(self, x: Tensor) -> Tensor:
        return self.encoder(x)

    def decode(self, x: Tensor) -> Tensor:
        return self.decoder(x)

    # ---- the hot path --------------------------------------------------------------------------------
    def _can_fuse(self) -> bool:
        """All levels in one launch: needs every level past its lazy k-means init and a mode the stack
        kernel implements (eval, STE, rotation trick)."""
        if len(self.layers) > 16:
            return False
        for layer in self.layers:
            if layer.do_kmeans_init and not layer.kmeans_initted:
                return False
            if not layer.kernel_covers():        # e.g. embed_dim > 128: level by level through rqhip/wide.py
                return False
            if layer.training and layer.forward_mode == QuantizeForwardMode.GUMBEL_SOFTMAX:
                return False
            if layer.training != self.layers[0].training or layer.forward_mode != self.layers[0].forward_mode:
                return False
        return True

    def _seam_weights(self, x: Tensor):
        """(w_in, w_out) when RqVae.forward can run the encoder's last Linear, every level and the decoder's first Linear + ReLU as ONE
        launch (rqhip_rq_seam): embed_dim 32 behind a 128-wide hidden layer on both sides (the reference's
        configs/rqvae_amazon.gin), plain codebooks that fit the LDS beside the two weights, a mode the stack kernel implements, and a
        batch of 4096 rows or more; else None.  (Smaller batches are launch-bound and keep round 5's path: measured at batch 640, the
        fused node 0.233 ms per graph step against 0.198 -- its two weight gradients leave the MLP stacks' job tables and every GEMM
        output is a long chain of dependent fp32 matrix instructions; profiles/r06_seam.txt.)"""
        if (not FUSE_SEAM or torch_ops.enabled() or not _lin._CHAIN or type(self.encoder) is not MLP or type(self.decoder) is not MLP
                or not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= _lin._SPLIT_MIN_ROWS)
                or not self._can_fuse() or not all(layer.plain_codebook for layer in self.layers)
                or self.embed_dim != _lin.CHAIN_D or not ops.rq_seam_supported(self.embed_dim, _lin.CHAIN_H, len(self.layers), self.codebook_size)):
            return None
        w_in, w_out = self.encoder.seam_tail_weight(), self.decoder.seam_head_weight()
        return None if (w_in is None or w_out is None) else (w_in, w_out)

    def _quantize_stack(self, res: Tensor, gumbel_t: float, want_levels: bool) -> _StackResult:
        if self._can_fuse():
            codebooks = torch.stack([layer.codebook() for layer in self.layers])  # [L,K,D], autograd splits it back
            if torch_ops.enabled():   # the same kernels as registered torch.library operators (traceable)
                return _StackResult(*torch.ops.rqhip.rq_stack(res, codebooks, self.layers[0].hip_mode(),
                                                              float(self.commitment_weight), want_levels))
            sink = getattr(self, "_rq_cb_grad_sink", None)
            if sink is not None and not all(layer.plain_codebook for layer in self.layers):
                sink = None
            out = RqStackFunction.apply(res, codebooks, self.layers[0].hip_mode(), float(self.commitment_weight),
                                        want_levels, sink)
            return _StackResult(*out)
        # level-by-level (first call with lazy k-means init, or Gumbel-softmax training): the reference's loop
        # (rqvae.py:125-132), each level being one L=1 launch of the same kernels
        embs, residuals, ids = [], [], []
        loss = 0
        for layer in self.layers:
            residuals.append(res)
            q = layer(res, temperature=gumbel_t)
            loss = loss + q.loss
            res = res - q.embeddings
            ids.append(q.ids)
            embs.append(q.embeddings)
        embs_t = torch.stack(embs)
        emb_sum = embs[0]
        for e in embs[1:]:
            emb_sum = emb_sum + e
        norms = torch.linalg.vector_norm(embs_t.detach(), dim=2).t()
        return _StackResult(embs_t, torch.stack(residuals), torch.stack(ids), loss, emb_sum, norms)

    def get_semantic_ids(self, x: Tensor, gumbel_t: float = 0.001) -> RqVaeOutput:
        x = x.to(self._first_param().dtype)
        res = self.encode(x)
        st = self._quantize_stack(res, gumbel_t, want_levels=True)
        return RqVaeOutput(
            embeddings=st.embs.permute(1, 2, 0),     # [B,D,L], same strides as the reference's rearrange
            residuals=st.residuals.permute(1, 2, 0),
            sem_ids=st.ids.t(),                      # [B,L] view of [L,B] (strides (1,B) like the reference)
            quantize_loss=st.loss,
        )

    def forward(self, batch: SeqBatch, gumbel_t: float) -> RqVaeComputedLosses:
        x = batch.x
        xin = x.to(self._first_param().dtype)
        reducer = getattr(self, "_rq_reducer", None)
        n = self.n_cat_feats
        seam = self._seam_weights(xin)
        _lin._XSTACK.clear()       # (weight gradients of a backward pass that ended in an exception do not outlive it)
        _lin._XSMALL.clear()
        reconstruction = x_hat = p_unique_ids = side = None
        if seam is not None:
            # the seam: the encoder up to its last hidden activation, then ONE launch for the last encoder Linear, every level and the
            # first decoder Linear + ReLU (res0 and the sum of the levels' outputs never travel), then the rest of the decoder
            hidden = self.encoder.run_before_tail(xin)
            codebooks = torch.stack([layer.codebook() for layer in self.layers])
            sink = getattr(self, "_rq_cb_grad_sink", None)
            want_scales = _lin.f16() and xin.shape[0] >= _lin._SPLIT_MIN_ROWS
            if reducer is not None and hidden.requires_grad:
                # multi-GPU: when the gradient of `hidden` exists, the decoder's, the codebooks' and every other gradient that is not an
                # encoder parameter's has been accumulated -- their all-reduce starts under the rest of the encoder's backward
                # (rqhip/dist.py:FlatGradReducer.boundary_hook; a no-op with one rank or an unarmed step).  The encoder's last weight,
                # whose gradient the seam node forms, is an encoder parameter: it travels with the late part.
                hidden.register_hook(reducer.boundary_hook)
            ids, qloss, norms, d, d_rows, d_cols = RqSeamFunction.apply(hidden, seam[0], codebooks, seam[1], self.layers[0].hip_mode(),
                                                                        float(self.commitment_weight), sink, want_scales)
            if want_scales:
                _lin.attach_scales(d, d_rows, d_cols)      # the maxima the decoder's split kernels scale by came with the launch
            st = _StackResult(None, None, ids, qloss, None, norms)
            if _lin.trims_on() and not torch.cuda.is_current_stream_capturing():
                # the duplicate statistic (rqvae.py:159-167: a debug output nothing in the step consumes) runs on a SIDE stream under the
                # decoder's GEMMs: a hash pass of CAS inserts + its fills and scalar kernels, 45 us of latency on the step's stream
                side = _side_stream(ids.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), torch.no_grad():
                    p_unique_ids = ops.unique_fraction(ids)
                ids.record_stream(side)
            if n == 0 and type(self.reconstruction_loss) is ReconstructionLoss and x.dim() == 2:
                reconstruction = self.decoder.reconstruction_rows(d, xin, first=2)
            if reconstruction is None:
                x_hat = self.decoder.run_after_head(d)
        else:
            res0 = self.encode(xin)
            if reducer is not None and res0.requires_grad:
                # multi-GPU: when this gradient exists, decoder and codebook gradients are final -- their all-reduce starts under
                # the encoder's backward (rqhip/dist.py:FlatGradReducer.boundary_hook; a no-op with one rank or an unarmed step)
                res0.register_hook(reducer.boundary_hook)
            st = self._quantize_stack(res0, gumbel_t, want_levels=False)
            if n == 0 and type(self.reconstruction_loss) is ReconstructionLoss and type(self.decoder) is MLP and x.dim() == 2:
                # large batches: the last decoder layer and the loss are one kernel, x_hat is never stored (modules/encoder.py)
                reconstruction = self.decoder.reconstruction_rows(st.emb_sum, xin)
        if reconstruction is None:
            if x_hat is None:
                x_hat = self.decode(st.emb_sum)                           # embs.sum(axis=-1), rqvae.py:146
            # rqvae.py:147-150: with n == 0 the `[..., :-0]` slice is EMPTY, so nothing is normalised
            x_hat = torch.cat([l2norm(x_hat[..., :-n]), x_hat[..., -n:]], dim=-1) if n != 0 else x_hat
            # (the kernels are fp32: a float64 / fp16 batch is compared in the model's dtype, as it was encoded)
            reconstruction = self.reconstruction_loss(x_hat, x if x.dtype == x_hat.dtype else x.to(x_hat.dtype))
        rqvae_loss = st.loss
        if (reconstruction.dim() == 1 and reconstruction.is_cuda and reconstruction.dtype == torch.float32
                and reconstruction.numel() > 0):   # (an empty batch keeps the reference's expression: three NaN means)
            # the three batch means of rqvae.py:154,171-172 in one launch
            if torch_ops.enabled():
                loss, recon_mean, rq_mean = torch.ops.rqhip.loss_means(reconstruction, rqvae_loss).unbind(0)
            else:
                loss, recon_mean, rq_mean = LossMeansFunction.apply(reconstruction, rqvae_loss)
        else:
            loss, recon_mean, rq_mean = (reconstruction + rqvae_loss).mean(), reconstruction.mean(), rqvae_loss.mean()
        if p_unique_ids is None:
            with torch.no_grad():
                if torch_ops.enabled():
                    n_distinct = torch.ops.rqhip.distinct_tuples(st.ids, self.codebook_size)
                    p_unique_ids = n_distinct / st.ids.shape[1]               # rqvae.py:159-167
                elif st.ids.is_cuda and st.ids.shape[1] > 0:
                    p_unique_ids = ops.unique_fraction(st.ids)                # the count and the division in one kernel
                else:
                    _, n_distinct = ops.dedup_rank(st.ids, self.codebook_size, want_rank=False)
                    p_unique_ids = n_distinct / st.ids.shape[1]
        elif side is not None:
            torch.cuda.current_stream().wait_stream(side)                   # join: the statistic is part of this call's result
            p_unique_ids.record_stream(torch.cuda.current_stream())
        return RqVaeComputedLosses(
            loss=loss,
            reconstruction_loss=recon_mean,
            rqvae_loss=rq_mean,
            embs_norm=st.embs_norm,
            p_unique_ids=p_unique_ids,
        )
