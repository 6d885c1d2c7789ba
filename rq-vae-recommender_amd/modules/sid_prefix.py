"""Valid-prefix test for beam search over semantic ids.

The reference's retrieval model masks beam candidates whose id prefix does not occur in the corpus
(`EncoderDecoderRetrievalModel._check_valid_prefix`, modules/model.py:169-182, called at every hierarchy step
of `generate`, model.py:349,364).  It compares every candidate with every corpus row: an [N, P, h] boolean
tensor per call, chunked by `batch_size` to bound memory.  The corpus does not change during generation, so
here the set of all corpus prefixes is built once on the device (csrc/sid_match.hip: one exact hash set per
prefix length) and a call is one probe per candidate.

    index = SemIdPrefixIndex(codebooks)              # codebooks [N, n_layers] int64, as model.py:59,75
    is_valid = index.check_valid_prefix(prefix)      # prefix [P, h] int64 -> bool [P], same as the reference

Replicated per rank (the index is 8 bytes per item and level); generation shards by batch, no collective.
"""
import torch
from torch import Tensor

from rqhip import ops


class SemIdPrefixIndex:
    def __init__(self, codebooks: Tensor) -> None:
        if codebooks.dim() != 2 or codebooks.dtype != torch.int64:
            raise ValueError(f"codebooks must be an int64 [N, n_layers] tensor, got {codebooks.dtype} "
                             f"{tuple(codebooks.shape)}")
        self.codebooks = codebooks
        self._index = None
        if codebooks.is_cuda:
            self._build()

    def _build(self) -> None:
        # private dense copy: the index stores row numbers of exactly this tensor
        self._corpus = self.codebooks.detach().contiguous().clone()
        self._index = ops.prefix_index_build(self._corpus)

    def to(self, device) -> "SemIdPrefixIndex":
        device = torch.device(device)
        if self.codebooks.device != device:
            self.codebooks = self.codebooks.to(device)
            self._index = None
        if self._index is None and self.codebooks.is_cuda:
            self._build()
        return self

    @property
    def num_items(self) -> int:
        return self.codebooks.shape[0]

    @torch.no_grad()
    def check_valid_prefix(self, prefix: Tensor, batch_size: int = 100000) -> Tensor:
        """Boolean mask: which rows of `prefix` [P, h] occur as the first h ids of a corpus row.
        `batch_size` is accepted for signature parity (model.py:170); no chunking is needed here."""
        if prefix.device != self.codebooks.device:  # the reference moves its codebooks to the prefix (model.py:173-174)
            self.to(prefix.device)
        if self._index is None:
            self._build()  # on a CPU tensor this raises RqHipError: the lookup has no CPU implementation
        if prefix.shape[0] == 0:
            # the reference ends in torch.cat([]) here (model.py:182)
            raise RuntimeError("check_valid_prefix: expected a non-empty batch of prefixes")
        if prefix.shape[1] > self.codebooks.shape[1]:
            raise RuntimeError(f"check_valid_prefix: prefix length {prefix.shape[1]} exceeds the "
                               f"{self.codebooks.shape[1]} id levels of the corpus")
        return ops.prefix_lookup(self._index, self._corpus, prefix.to(torch.int64))


def check_valid_prefix(codebooks: Tensor, prefix: Tensor) -> Tensor:
    """One-shot form (builds a throw-away index); prefer a SemIdPrefixIndex kept next to the model."""
    return SemIdPrefixIndex(codebooks.to(prefix.device)).check_valid_prefix(prefix)
