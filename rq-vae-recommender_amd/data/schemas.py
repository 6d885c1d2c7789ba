"""Batch containers exchanged between the data layer, RqVae and the tokenizer.

Field names and order follow reference data/schemas.py:7-22 because callers construct and unpack these
positionally (`SeqBatch(*[...])` in data/utils.py, `batch.x` in modules/rqvae.py)."""
import collections

FUT_SUFFIX = "_fut"

#: one batch of items or user sequences; RqVae.forward reads only `.x` (reference modules/rqvae.py:143)
SeqBatch = collections.namedtuple("SeqBatch", ["user_ids", "ids", "ids" + FUT_SUFFIX, "x", "x" + FUT_SUFFIX, "seq_mask"])

#: output of SemanticIdTokenizer.forward (reference modules/tokenizer/semids.py:139-146)
TokenizedSeqBatch = collections.namedtuple(
    "TokenizedSeqBatch",
    ["user_ids", "sem_ids", "sem_ids" + FUT_SUFFIX, "seq_mask", "token_type_ids", "token_type_ids" + FUT_SUFFIX],
)
