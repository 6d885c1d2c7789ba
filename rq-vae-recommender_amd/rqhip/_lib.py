"""ctypes loader of csrc/librqhip.so -- the only bridge between the Python mirror of the reference
API and the HIP kernels.  There is NO fallback: if the library is missing (or, at call time, the
tensors are not on a ROCm device) the call raises.

Load order matters: torch bundles its own libamdhip64 with the same SONAME as /opt/rocm's, so torch is
imported first and librqhip.so then binds to the HIP runtime already in the process.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch  # noqa: F401  (must precede CDLL: one HIP runtime per process)

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
SO_PATH = os.path.join(_CSRC, "librqhip.so")

MODE_EVAL, MODE_STE, MODE_ROTATION, MODE_GUMBEL = 0, 1, 2, 3
# rqhip_rq_forward_ex flags (include/rqhip.h)
FWD_SCAN_FP32, FWD_SCAN_VALU, FWD_NO_COOP_TAIL = 0x1, 0x2, 0x10
WGRAD_FP32 = 0x1   # rqhip_linear_wgrad_ex
BWD_CBGRAD_MATRIX = 0x1   # rqhip_rq_backward_ex
SPLIT_F16X2, SPLIT_BF16X3 = 0, 1                       # arithmetic of the split GEMM kernels (include/rqhip.h)
EPI_STORE, EPI_RELU, EPI_RECON, EPI_MASK = 0, 1, 2, 3  # rqhip_gemm_split_ex epilogues
PROF_TAGS = {1: "rq_forward", 2: "rq_backward", 3: "gemm_split", 4: "wgrad", 5: "maxima", 6: "weight_images", 7: "rq_seam", 8: "linear_small"}


class ImageJob(C.Structure):          # rqhip_image_job
    _fields_ = [("w", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("transpose", C.c_int), ("arith", C.c_int),
                ("image", C.c_void_p), ("image_bytes", C.c_size_t)]


class SeamArgs(C.Structure):          # rqhip_seam_args
    _fields_ = [("B", C.c_int64), ("D", C.c_int), ("H", C.c_int), ("h", C.c_void_p), ("h_mask", C.c_void_p), ("w_in", C.c_void_p),
                ("w_in_transposed", C.c_int), ("res0", C.c_void_p), ("res0_out", C.c_void_p), ("codebooks", C.c_void_p),
                ("L", C.c_int), ("K", C.c_int), ("mode", C.c_int), ("beta", C.c_float), ("ids", C.c_void_p), ("emb_sum", C.c_void_p),
                ("loss", C.c_void_p), ("embs_norm", C.c_void_p), ("w_out", C.c_void_p), ("w_out_transposed", C.c_int),
                ("out_epilogue", C.c_int), ("out_mask", C.c_void_p), ("out", C.c_void_p), ("out_row_max", C.c_void_p),
                ("out_col_max", C.c_void_p)]


class GemmArgs(C.Structure):          # rqhip_gemm_args
    _fields_ = [("A", C.c_void_p), ("M", C.c_int64), ("R", C.c_int), ("image", C.c_void_p), ("Nc", C.c_int),
                ("arith", C.c_int), ("epilogue", C.c_int), ("tile_rows", C.c_int), ("C", C.c_void_p), ("aux", C.c_void_p),
                ("row_scale", C.c_float), ("loss_rows", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("a_row_max", C.c_void_p), ("a_row_parts", C.c_int),
                ("c_row_max", C.c_void_p), ("c_col_max", C.c_void_p)]


class WgradJob(C.Structure):          # rqhip_wgrad_job
    _fields_ = [("g", C.c_void_p), ("x", C.c_void_p), ("N", C.c_int), ("K", C.c_int), ("g_col_max", C.c_void_p),
                ("x_col_max", C.c_void_p), ("dW", C.c_void_p)]


class ProfileRecord(C.Structure):     # rqhip_profile_record
    _fields_ = [("tag", C.c_int), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/rqhip.h declares: (restype, argtypes)
_i64, _int, _f32, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t
SIGNATURES = {
    "rqhip_version": (_int, []),
    "rqhip_last_error": (C.c_char_p, []),
    "rqhip_device_cu_count": (_int, [C.POINTER(_int)]),
    "rqhip_rq_forward_workspace_bytes": (_sz, [_int, _int]),
    "rqhip_rq_forward": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _sz, _vp]),
    "rqhip_rq_forward_ex": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _sz, C.c_uint, _vp]),
    "rqhip_filter_bound": (None, [C.POINTER(_f32), C.POINTER(_f32)]),
    "rqhip_filter_bound_d": (None, [_int, C.POINTER(_f32), C.POINTER(_f32)]),
    "rqhip_filter_scores": (_int, [_vp, _i64, _int, _vp, _int, _vp, _vp, _sz, _vp]),
    "rqhip_rq_backward_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "rqhip_rq_backward_plan": (_int, [_i64, _int, _int, _int, _int, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "rqhip_rq_backward": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _sz, _vp]),
    "rqhip_rq_backward_matrix_form": (_int, [_int, _int, _int, _int]),
    "rqhip_rq_backward_ex": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _sz, C.c_uint, _vp]),
    "rqhip_gumbel_forward": (_int, [_vp, _i64, _int, _vp, _int, _vp, _f32, _f32, _vp, _vp, _vp, _vp]),
    "rqhip_gumbel_backward_workspace_bytes": (_sz, [_i64, _int, _int]),
    "rqhip_gumbel_matrix_path_min_rows": (_i64, [_i64]),
    "rqhip_gumbel_backward": (_int, [_vp, _i64, _int, _vp, _int, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _sz,
                                     _vp]),
    "rqhip_kmeans_assign": (_int, [_vp, _i64, _int, _vp, _int, _vp, _vp]),
    "rqhip_kmeans_update": (_int, [_vp, _i64, _int, _vp, _int, _vp, _vp, _vp, _vp]),
    "rqhip_kmeans_lloyd": (_int, [_vp, _i64, _int, _vp, _int, _vp, _vp, _vp, _int, _f32, _vp]),
    "rqhip_kmeans_partial_sums": (_int, [_vp, _i64, _int, _vp, _int, _vp, _vp, _vp, _vp]),
    "rqhip_kmeans_apply_sums": (_int, [_vp, _int, _int, _vp, _vp, _vp, _f32, _vp]),
    "rqhip_dedup_workspace_bytes": (_sz, [_i64]),
    "rqhip_dedup_rank": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "rqhip_unique_fraction": (_int, [_vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "rqhip_prefix_index_bytes": (_sz, [_i64, _int]),
    "rqhip_prefix_index_build": (_int, [_vp, _i64, _int, _i64, _vp, _sz, _vp]),
    "rqhip_prefix_lookup": (_int, [_vp, _sz, _vp, _i64, _int, _i64, _vp, _i64, _int, _i64, _vp, _vp]),
    "rqhip_topk_first_match": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp]),
    "rqhip_recon_loss_forward": (_int, [_vp, _i64, _vp, _i64, _i64, _int, _vp, _vp]),
    "rqhip_recon_loss_backward": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _vp, _vp]),
    "rqhip_recon_loss_forward_spec": (_int, [_vp, _i64, _vp, _i64, _i64, _int, _f32, _vp, _vp, _vp]),
    "rqhip_recon_loss_backward_spec": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _int, _f32, _vp, _vp]),
    "rqhip_loss_means": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "rqhip_loss_means_workspace_bytes": (_sz, []),
    "rqhip_loss_means_ws": (_int, [_vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "rqhip_loss_means_backward": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "rqhip_linear_wgrad_supported": (_int, [_int, _int]),
    "rqhip_linear_wgrad_plan": (_int, [_i64, _int, _int, C.POINTER(_int)]),
    "rqhip_linear_wgrad_workspace_bytes": (_sz, [_i64, _int, _int]),
    "rqhip_linear_wgrad": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "rqhip_linear_wgrad_ex": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _sz, C.c_uint, _vp]),
    "rqhip_linear_wgrad_f16": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rqhip_linear_wgrad_f16_batch_plan": (_int, [_i64, C.POINTER(_int), C.POINTER(_int), _int]),
    "rqhip_linear_wgrad_f16_batch_workspace_bytes": (_sz, [_i64, C.POINTER(_int), C.POINTER(_int), _int]),
    "rqhip_linear_wgrad_f16_batch": (_int, [C.POINTER(WgradJob), _int, _i64, _vp, _sz, _vp]),
    "rqhip_linear_wgrad_jobs_supported": (_int, [_int, _int]),
    "rqhip_linear_wgrad_jobs": (_int, [_vp, _vp, _vp, C.POINTER(_int), C.POINTER(_int), _int, _i64, _vp]),
    "rqhip_gemm_split_supported": (_int, [_int, _int]),
    "rqhip_weight_image_bytes": (_sz, [_int, _int, _int]),
    "rqhip_weight_images": (_int, [C.POINTER(ImageJob), _int, _vp]),
    "rqhip_maxima": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp]),
    "rqhip_gemm_split_ex": (_int, [C.POINTER(GemmArgs), _vp]),
    "rqhip_weight_planes_bytes": (_sz, [_int, _int]),
    "rqhip_weight_planes": (_int, [_vp, _int, _int, _int, _vp, _sz, _vp]),
    "rqhip_gemm_split": (_int, [_vp, _i64, _int, _vp, _int, _int, _vp, _vp]),
    "rqhip_gemm_split_recon_workspace_bytes": (_sz, [_i64, _int]),
    "rqhip_gemm_split_recon": (_int, [_vp, _i64, _int, _vp, _int, _vp, _f32, _vp, _vp, _vp, _sz, _vp]),
    "rqhip_recon_rescale_rows": (_int, [_vp, _i64, _int, _f32, _vp, _vp]),
    "rqhip_recon_rescale_rows_ex": (_int, [_vp, _i64, _int, _f32, _vp, _vp, _int, _vp, _vp]),
    "rqhip_rq_seam_supported": (_int, [_int, _int, _int, _int]),
    "rqhip_rq_seam": (_int, [C.POINTER(SeamArgs), _vp]),
    "rqhip_linear_small_supported": (_int, [_i64, _int, _int]),
    "rqhip_linear_small_plan": (_int, [_i64, _int, _int, C.POINTER(_int), C.POINTER(_int)]),
    "rqhip_linear_small": (_int, [_vp, _vp, _int, _vp, _i64, _int, _int, _int, _vp, _int, _int, _vp]),
    "rqhip_adamw_step": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _vp]),
    "rqhip_profile_enable": (_int, [_int]),
    "rqhip_profile_select": (_int, [C.c_uint]),
    "rqhip_profile_read": (_int, [C.POINTER(_f32), _int, C.POINTER(_int)]),
    "rqhip_profile_read_tagged": (_int, [C.POINTER(ProfileRecord), _int, C.POINTER(_int)]),
}


class RqHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 with the committed Makefile (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, "-j8"] + (["-B"] if force else [])
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RqHipError("building librqhip.so failed:\n" + proc.stdout[-4000:])
    return SO_PATH


_lib = None


def load(path: str) -> C.CDLL:
    """Bind every symbol of include/rqhip.h in the library at `path` and make it the one `lib()` returns.  Called
    implicitly with the in-tree build; developer tools (tools/ab_*.py) call it with another build of the same ABI
    BEFORE the first op to compare two builds -- an explicit call, not an environment switch."""
    global _lib
    if not os.path.exists(path):
        raise RqHipError(
            f"{path} not found: build it with `python __graft_entry__.py` or `make -C {_CSRC}`. "
            "The HIP extension is the product path; there is no CPU fallback.")
    handle = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here == header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = handle
    return handle


def lib() -> C.CDLL:
    """The loaded library; raises (never falls back) when it has not been built."""
    return _lib if _lib is not None else load(SO_PATH)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().rqhip_last_error().decode("utf-8", "replace")
        kind = "HIP error" if rc > 0 else "argument error"
        raise RqHipError(f"{what}: {kind} {rc}: {msg}")
