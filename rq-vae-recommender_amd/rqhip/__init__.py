"""rqhip -- Python side of the C ABI in include/rqhip.h (hand-written HIP kernels for gfx950)."""
from ._lib import (MODE_EVAL, MODE_GUMBEL, MODE_ROTATION, MODE_STE, RqHipError, SIGNATURES, SO_PATH, build,  # noqa
                   lib)
