"""The C-ABI library loads (no GPU needed) and exports exactly what include/rqhip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rqhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rqhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("rqhip_rq_forward", "rqhip_rq_backward", "rqhip_gumbel_forward", "rqhip_gumbel_backward",
                 "rqhip_kmeans_assign", "rqhip_kmeans_update", "rqhip_dedup_rank", "rqhip_prefix_index_build",
                 "rqhip_prefix_lookup", "rqhip_topk_first_match", "rqhip_version",
                 "rqhip_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol_and_binding_covers_them():
    from rqhip import _lib
    if not os.path.exists(_lib.SO_PATH):
        pytest.fail(f"{_lib.SO_PATH} missing: run `python __graft_entry__.py` (build) first")
    handle = ctypes.CDLL(_lib.SO_PATH)
    declared = _declared_functions()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in rqhip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "rqhip/_lib.py SIGNATURES out of sync with include/rqhip.h"


def test_version_and_error_string_without_gpu():
    from rqhip import _lib
    l = _lib.lib()
    assert l.rqhip_version() == 462
    # argument validation happens before any HIP call, so it is testable on a CPU-only box
    rc = l.rqhip_rq_forward(None, 4, 32, None, 3, 256, 1, 0.25, None, None, None, None, None, None, None, None, 0, None)
    assert rc == -1
    assert b"null pointer" in l.rqhip_last_error()
    rc = l.rqhip_kmeans_assign(None, -1, 32, None, 8, None, None)
    assert rc == -1


def test_sid_match_argument_checks_without_gpu():
    from rqhip import _lib
    l = _lib.lib()
    # 2 x N slots (power of two, at least 64) of 4 bytes per prefix length
    assert l.rqhip_prefix_index_bytes(1000, 3) == 2048 * 4 * 3
    assert l.rqhip_prefix_index_bytes(10, 1) == 64 * 4
    assert l.rqhip_prefix_index_bytes(-1, 3) == 0 and l.rqhip_prefix_index_bytes(5, 0) == 0
    assert l.rqhip_prefix_index_build(None, 5, 3, 3, None, 0, None) == -1       # null corpus
    assert l.rqhip_prefix_index_build(None, 0, 3, 2, None, 0, None) == -1       # row stride < H
    assert b"ld >= H" in l.rqhip_last_error()
    assert l.rqhip_prefix_lookup(None, 0, None, 0, 3, 3, None, 4, 5, 5, None, None) == -1   # h > H
    assert l.rqhip_topk_first_match(None, None, 4, 10, 3, None, None) == -1
    assert l.rqhip_topk_first_match(None, None, 0, 10, 3, None, None) == 0      # empty batch: nothing to do


def test_workspace_queries():
    from rqhip import _lib
    l = _lib.lib()
    assert l.rqhip_rq_forward_workspace_bytes(3, 256) == (3 * 256 + 3) * 4
    assert l.rqhip_rq_forward_workspace_bytes(2, 100) == (2 * 128 + 2) * 4
    # [L,B,D] row scratch + one [L,K,D] partial table per workgroup of the widest launch (16 x 64-row units here)
    assert l.rqhip_rq_backward_workspace_bytes(1000, 32, 3, 256) == 3 * 1000 * 32 * 4 + 16 * 3 * 256 * 32 * 4
    assert l.rqhip_dedup_workspace_bytes(1000) >= 2048 * 4 + 4 * 1000 * 4
