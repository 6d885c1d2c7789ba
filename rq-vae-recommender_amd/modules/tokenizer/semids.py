"""Frozen-RQ-VAE tokenizer: items -> semantic-id tuples (+ dedup column), API of the reference's
modules/tokenizer/semids.py.

`precompute_corpus_ids` is where the reference is quadratic: it walks the corpus in batches of 512 and
compares every batch with everything seen so far (semids.py:92-105, O(N^2 L)).  Here the corpus is
tokenised in large row blocks by the fused HIP kernel and the dedup column -- "how many earlier items have
the same tuple" -- comes from one hash + stable radix-sort pass on the device (csrc/ids.hip), O(N).
`precompute_corpus_ids(ds, sharded=True)` shards the rows across ranks and all-gathers the id table.
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
    def precompute_corpus_ids(self, movie_dataset, sharded: bool = False) -> Tensor:
        """[N, n_layers + 1] int64: semantic ids of every item followed by the dedup counter.

        Local by default, like the reference, whose callers run it on the main process only
        (train_rqvae.py:272-275): a rank-0-only call must not enter a collective.  `sharded=True` is the explicit
        multi-GPU form -- EVERY rank must call it: rows are split across ranks, tokenised, and the id table is
        all-gathered (ids are a function of each row alone, except that a different GEMM row count can flip a
        near-tie; tests/parity_gate.py describes the tie policy)."""
        device = self.rq_vae.device
        n = len(movie_dataset)
        lo, hi = rqdist.shard_bounds(n) if sharded else (0, n)
        blocks = []
        for start in range(lo, hi, CORPUS_BLOCK_ROWS):
            rows = torch.arange(start, min(hi, start + CORPUS_BLOCK_ROWS))
            x = movie_dataset[rows].x.to(device)
            blocks.append(self.rq_vae.get_semantic_ids(x).sem_ids)        # [b, L] view of [L, b]
        local = torch.cat(blocks, dim=0) if blocks else torch.empty((0, self.n_layers), dtype=torch.int64,
                                                                     device=device)
        ids = rqdist.allgather_rows(local.contiguous()) if sharded else local.contiguous()   # [N, L]
        rank, _ = ops.dedup_rank(ids.t().contiguous(), self.codebook_size)
        self.cached_ids = torch.cat([ids, rank.unsqueeze(1)], dim=1)
        return self.cached_ids

    # ---- sequences of items -> sequences of id tokens ------------------------------------------------------
    def _lookup(self, item_ids: Tensor) -> Tensor:
        """[b, n] item numbers -> [b, n * width] tokens: each item becomes the `width` cached ids of its row."""
        table = self.cached_ids
        return table.index_select(0, item_ids.reshape(-1)).view(item_ids.shape[0], -1)

    def _tokenize_seq_batch_from_cached(self, ids: Tensor) -> Tensor:   # the reference's name for `_lookup`
        return self._lookup(ids)

    @staticmethod
    def _positions(width: int, rows: int, items: int, device) -> Tensor:
        """token_type_ids: position of every token inside its item's tuple, 0..width-1 repeated `items` times."""
        return torch.arange(width, device=device).repeat(rows, items)

    @torch.no_grad()
    @eval_mode
    def forward(self, batch: SeqBatch) -> TokenizedSeqBatch:
        """Two regimes, as in the reference (semids.py:117-146).  Without a usable cache (none yet, or an item number
        beyond it) the batch's own features are quantised and the plain n_layers-wide tuples come back, no masks;
        with one, history and target item numbers are replaced by their cached (n_layers + 1)-wide rows and padded
        history positions are overwritten with -1."""
        n_rows, n_items = batch.ids.shape
        cache = self.cached_ids
        use_cache = cache is not None and not bool(batch.ids.max() >= cache.shape[0])
        if use_cache:
            width = cache.shape[1]
            valid = batch.seq_mask.repeat_interleave(width, dim=1)
            history = self._lookup(batch.ids).masked_fill(~valid, -1)
            target = self._lookup(batch.ids_fut)
        else:
            history = self.rq_vae.get_semantic_ids(batch.x).sem_ids
            width = history.shape[-1]
            valid = target = None
        dev = history.device
        return TokenizedSeqBatch(user_ids=batch.user_ids, sem_ids=history, sem_ids_fut=target, seq_mask=valid,
                                 token_type_ids=self._positions(width, n_rows, n_items, dev),
                                 token_type_ids_fut=self._positions(width, n_rows, 1, dev))
