"""Item-feature dataset for RQ-VAE training: an HBM-resident stand-in for the reference's `ItemData`.

The reference's data layer (data/processed.py:39-86 on top of data/amazon.py, data/ml32m.py, ...) downloads
raw datasets and embeds item text with sentence-T5-XXL through torch_geometric / polars -- out of scope for
this repo and impossible offline.  What the hot path needs from it is small and is kept
verbatim: `len(ds)` and `ds[idx] -> SeqBatch` with `x = item_matrix[idx, :768]` for an int / list / tensor
index (processed.py:74-86), train/eval/all splits driven by an `is_train` mask, and the `RecDataset` enum
that gin configs name as `%data.processed.RecDataset.AMAZON`.

Source of the matrix, in order:
  1. `<root>/item_features.pt` -- a dict {"x": float32 [N, >=768], "is_train": bool [N] (optional)} that a
     user exports once from the reference's processed HeteroData (`data["item"].x`, `["is_train"]`);
  2. ONLY when the folder is NAMED "synthetic:<N>" (e.g. `train.dataset_folder="synthetic:87585"`: an explicit opt-in
     in the configuration itself, loud warning, `ds.synthetic` is True): a deterministic synthetic corpus of N unit-norm
     768-d rows (seed 1234) with the reference's 95/5 split (seed 42, data/amazon.py:154-156).  Otherwise a missing file
     raises FileNotFoundError -- the reference would download the data or fail here, never train on noise.
MI355X-first: the matrix is moved to the GPU once (`to_device`) and batches are gathered there -- 10 M x 768
fp32 = 30.7 GB fits one 288 GB HBM stack many times over -- so no PCIe copy sits in the training loop.
"""
import os
from enum import Enum
from typing import Optional

import torch
from torch import Tensor
from torch.utils.data import Dataset

from data.schemas import SeqBatch

try:
    import gin
except ImportError:  # pragma: no cover
    from rqhip import ginlite as gin

FEATURE_DIM = 768


@gin.constants_from_enum
class RecDataset(Enum):
    AMAZON = 1
    ML_1M = 2
    ML_32M = 3


def synthetic_item_matrix(n_items: int, dim: int = FEATURE_DIM, seed: int = 1234) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n_items, dim, generator=g), dim=-1)


def synthetic_train_mask(n_items: int, seed: int = 42, p_train: float = 0.95) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n_items, generator=g) > (1.0 - p_train)


class ItemData(Dataset):
    def __init__(self, root: str, *args, force_process: bool = False, dataset: RecDataset = RecDataset.ML_1M,
                 train_test_split: str = "all", item_matrix: Optional[Tensor] = None,
                 is_train: Optional[Tensor] = None, **kwargs) -> None:
        del args, kwargs, force_process  # accepted for signature compatibility (split=..., etc.)
        self.dataset = dataset
        self.synthetic = False
        if item_matrix is None:
            path = os.path.join(root, "item_features.pt")
            if str(root).startswith("synthetic:"):
                # explicit opt-in only, spelled out in the configuration: a run on noise must never look like a run on
                # the dataset
                n = int(str(root).split(":", 1)[1])
                print(f"[ItemData] WARNING: dataset folder {root!r} -- using {n} SYNTHETIC unit-norm items; "
                      "results say nothing about a real corpus", flush=True)
                item_matrix = synthetic_item_matrix(n)
                self.synthetic = True
            elif os.path.exists(path):
                blob = torch.load(path, map_location="cpu", weights_only=True)
                item_matrix, is_train = blob["x"].to(torch.float32), blob.get("is_train", is_train)
            else:
                raise FileNotFoundError(
                    f"{path} not found.  Export the item features first (INTEGRATION.md, 'Item features': "
                    "torch.save({'x': item_matrix_fp32[N, >=768], 'is_train': mask}, '<dataset_folder>/item_features.pt')) "
                    "or opt in to synthetic items explicitly by naming the folder 'synthetic:<n>'.")
        if is_train is None:
            is_train = synthetic_train_mask(item_matrix.shape[0])
        if train_test_split == "train":
            keep = is_train
        elif train_test_split == "eval":
            keep = ~is_train
        elif train_test_split == "all":
            keep = None            # every row: the matrix itself, no masked copy (10 M x 768 fp32 is 30.7 GB)
        else:
            raise ValueError(f"unknown train_test_split {train_test_split!r}")
        self.item_data = item_matrix if keep is None else item_matrix[keep.to(item_matrix.device)]
        self._row_max = self._col_max = None

    def to_device(self, device) -> "ItemData":
        """Make the feature matrix resident on `device` (HBM); later `ds[idx]` gathers happen there."""
        self.item_data = self.item_data.to(device)
        self._row_max = self._col_max = None
        return self

    # batches of this many rows and more run the MLPs' fp16-split kernels, which scale every row by its largest |value|
    # (rqhip/linear.py:_SPLIT_MIN_ROWS): that maximum is a property of the item, computed once per corpus and gathered with the batch
    @property
    def _SCALES_MIN_ROWS(self) -> int:
        from rqhip import linear as _lin
        return _lin._SPLIT_MIN_ROWS

    def _corpus_maxima(self):
        """(row maxima int32 [N], column maxima int32 [768]) of the resident matrix's feature columns, bit patterns (rqhip_maxima)."""
        if getattr(self, "_row_max", None) is None:
            from rqhip import ops
            x = self.item_data
            n, step = x.shape[0], 1 << 20
            rows, cols = [], None
            for lo in range(0, n, step):
                part = x[lo:lo + step, :FEATURE_DIM]
                r, c, _ = ops.maxima(part if part.is_contiguous() else part.contiguous())
                rows.append(r[0])
                cols = c if cols is None else torch.maximum(cols, c)       # (bit patterns of non-negative floats order like integers)
            self._row_max, self._col_max = torch.cat(rows), cols
        return self._row_max, self._col_max

    def __len__(self) -> int:
        return self.item_data.shape[0]

    def __getitem__(self, idx) -> SeqBatch:
        dev = self.item_data.device
        item_ids = idx.to(dev) if isinstance(idx, Tensor) else torch.tensor(idx, device=dev).unsqueeze(0)
        rows = idx.to(dev) if isinstance(idx, Tensor) else idx
        minus_one = -1 * torch.ones_like(item_ids.squeeze(0))
        x = self.item_data[rows, :FEATURE_DIM]
        if (x.is_cuda and isinstance(rows, Tensor) and rows.dim() == 1 and rows.numel() >= self._SCALES_MIN_ROWS
                and x.dtype == torch.float32 and x.shape[1] % 4 == 0):
            from rqhip import linear as _lin
            rm, cm = self._corpus_maxima()
            _lin.attach_scales(x, rm[rows].unsqueeze(0), cm)    # the rows' maxima travel with the rows; corpus-wide column bounds
        return SeqBatch(user_ids=minus_one, ids=item_ids, ids_fut=minus_one, x=x,
                        x_fut=minus_one, seq_mask=torch.ones_like(item_ids, dtype=torch.bool))
