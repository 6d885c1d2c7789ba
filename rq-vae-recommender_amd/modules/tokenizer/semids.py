"""Frozen-RQ-VAE tokenizer: items -> semantic-id tuples (+ dedup column), API of the reference's
modules/tokenizer/semids.py.

`precompute_corpus_ids` is where the reference is quadratic: it walks the corpus in batches of 512 and
compares every batch with everything seen so far (semids.py:92-105, O(N^2 L)).  Here the corpus is
tokenised in large row blocks by the fused HIP kernel and the dedup column -- "how many earlier items have
the same tuple" -- comes from one hash + stable radix-sort pass on the device (csrc/ids.hip), O(N).
With torch.distributed initialised the rows are sharded across ranks and the id table is all-gathered.
"""
from typing import List, Optional

import torch
from torch import Tensor, nn

from data.schemas import SeqBatch, TokenizedSeqBatch
from modules.rqvae import RqVae
from modules.utils import eval_mode
from rqhip import dist as rqdist
from rqhip import ops

BATCH_SIZE = 16
CORPUS_BLOCK_ROWS = 1 << 18  # rows tokenised per launch (768-d fp32: 805 MB of features per block)


class SemanticIdTokenizer(nn.Module):
    """Tokenizes a batch of sequences of item features into a batch of sequences of semantic ids."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dims: List[int], codebook_size: int,
                 n_layers: int = 3, n_cat_feats: int = 18, commitment_weight: float = 0.25,
                 rqvae_weights_path: Optional[str] = None, rqvae_codebook_normalize: bool = False,
                 rqvae_sim_vq: bool = False) -> None:
        super().__init__()
        self.rq_vae = RqVae(input_dim=input_dim, embed_dim=output_dim, hidden_dims=hidden_dims,
                            codebook_size=codebook_size, codebook_kmeans_init=False,
                            codebook_normalize=rqvae_codebook_normalize, codebook_sim_vq=rqvae_sim_vq,
                            n_layers=n_layers, n_cat_features=n_cat_feats, commitment_weight=commitment_weight)
        if rqvae_weights_path is not None:
            self.rq_vae.load_pretrained(rqvae_weights_path)
        self.rq_vae.eval()
        self.codebook_size = codebook_size
        self.n_layers = n_layers
        self.reset()

    def reset(self) -> None:
        self.cached_ids = None

    @property
    def sem_ids_dim(self) -> int:
        return self.n_layers + 1

    def _get_hits(self, query: Tensor, key: Tensor) -> Tensor:
        """[Q, K] bool: does query row q equal key row k (kept for API parity; not used on the fast path)."""
        return (key.unsqueeze(0) == query.unsqueeze(1)).all(dim=-1)

    @torch.no_grad()
    @eval_mode
    def precompute_corpus_ids(self, movie_dataset) -> Tensor:
        """[N, n_layers + 1] int64: semantic ids of every item followed by the dedup counter."""
        device = self.rq_vae.device
        n = len(movie_dataset)
        lo, hi = rqdist.shard_bounds(n)
        blocks = []
        for start in range(lo, hi, CORPUS_BLOCK_ROWS):
            rows = torch.arange(start, min(hi, start + CORPUS_BLOCK_ROWS))
            x = movie_dataset[rows].x.to(device)
            blocks.append(self.rq_vae.get_semantic_ids(x).sem_ids)        # [b, L] view of [L, b]
        local = torch.cat(blocks, dim=0) if blocks else torch.empty((0, self.n_layers), dtype=torch.int64,
                                                                     device=device)
        ids = rqdist.allgather_rows(local.contiguous())                    # [N, L] on every rank
        rank, _ = ops.dedup_rank(ids.t().contiguous(), self.codebook_size)
        self.cached_ids = torch.cat([ids, rank.unsqueeze(1)], dim=1)
        return self.cached_ids

    def _tokenize_seq_batch_from_cached(self, ids: Tensor) -> Tensor:
        b, n = ids.shape
        return self.cached_ids[ids.flatten(), :].reshape(b, n * self.cached_ids.shape[1])

    @torch.no_grad()
    @eval_mode
    def forward(self, batch: SeqBatch) -> TokenizedSeqBatch:
        if self.cached_ids is None or batch.ids.max() >= self.cached_ids.shape[0]:
            B, N = batch.ids.shape
            sem_ids = self.rq_vae.get_semantic_ids(batch.x).sem_ids
            D = sem_ids.shape[-1]
            seq_mask, sem_ids_fut = None, None
        else:
            B, N = batch.ids.shape
            _, D = self.cached_ids.shape
            sem_ids = self._tokenize_seq_batch_from_cached(batch.ids)
            seq_mask = batch.seq_mask.repeat_interleave(D, dim=1)
            sem_ids[~seq_mask] = -1
            sem_ids_fut = self._tokenize_seq_batch_from_cached(batch.ids_fut)
        token_type_ids = torch.arange(D, device=sem_ids.device).repeat(B, N)
        token_type_ids_fut = torch.arange(D, device=sem_ids.device).repeat(B, 1)
        return TokenizedSeqBatch(user_ids=batch.user_ids, sem_ids=sem_ids, sem_ids_fut=sem_ids_fut,
                                 seq_mask=seq_mask, token_type_ids=token_type_ids,
                                 token_type_ids_fut=token_type_ids_fut)
