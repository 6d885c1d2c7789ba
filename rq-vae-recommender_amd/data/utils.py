"""Batch plumbing helpers (API of reference data/utils.py:4-16)."""
import itertools

from data.schemas import SeqBatch


def cycle(dataloader):
    """Iterate over `dataloader` forever, restarting it when exhausted (a fresh epoch each time)."""
    for _epoch in itertools.count():
        yield from dataloader


def batch_to(batch: SeqBatch, device) -> SeqBatch:
    return SeqBatch._make(field.to(device) for field in batch)


def next_batch(dataloader, device) -> SeqBatch:
    return batch_to(next(dataloader), device)
