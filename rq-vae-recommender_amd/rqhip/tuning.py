"""Library-GEMM selection for the encoder/decoder MLPs (PyTorch TunableOp).

The MLP GEMMs are plain library GEMMs (hipBLASLt / rocBLAS through PyTorch-ROCm): 98 % of the FLOPs of a
training step but not part of the hand-written hot path.  Two things matter on MI355X:

* precision: the reference asks for torch.set_float32_matmul_precision("high") (modules/rqvae.py:19), which on its
  CPU path is plain fp32 but on ROCm selects a reduced-precision "tf32" GEMM class.  Parity with the reference's
  CPU results needs true fp32, so modules/rqvae.py pins "highest".
* kernel choice: the default heuristic picks poor fp32 kernels for these tall-skinny shapes (32 TFLOP/s at
  100 000 x 768 x 512); TunableOp's per-shape selection reaches ~110 TFLOP/s.  The selections for the shipped
  shapes are committed in tuning/tunableop_gfx950.csv and loaded here with tuning DISABLED, so a run never pays
  tuning time; `enable_tuned_gemms(tune=True)` (tools/tune_gemms.py) tunes shapes that are not in the file (new batch sizes) and writes them back.
"""
from __future__ import annotations

import os

import torch

TUNING_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning",
                           "tunableop_gfx950.csv")


def enable_tuned_gemms(verbose: bool = False, tune: bool = False, out_file: str | None = None, tune_ms: int = 30) -> bool:
    """Turn on TunableOp with the committed selections; returns True when active.
    tune=True (tools/tune_gemms.py only): also tune shapes that are not in the file and write the table to `out_file`
    (default: the committed file) when the process exits."""
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tunable
    tune_now = bool(tune)
    if not tune_now and not os.path.exists(TUNING_FILE):
        return False
    tunable.enable(True)
    tunable.tuning_enable(tune_now)
    if tune_now:
        # results are (re)written to this file when the process exits
        tunable.set_filename(out_file or TUNING_FILE)
        tunable.set_max_tuning_duration(int(tune_ms))
        tunable.set_max_tuning_iterations(100)
    else:
        # read-only use: keep TunableOp's output name away from the committed file (N ranks of one job would
        # otherwise all rewrite it at exit)
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rq_tunableop_unused_{os.getpid()}.csv"))
    if os.path.exists(TUNING_FILE):
        ok = tunable.read_file(TUNING_FILE)
        if verbose:
            print(f"TunableOp: loaded {TUNING_FILE}: {ok}")
    return True
