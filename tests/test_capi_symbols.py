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
                 "rqhip_kmeans_assign", "rqhip_kmeans_update", "rqhip_dedup_rank", "rqhip_version",
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
    assert l.rqhip_version() == 100
    # argument validation happens before any HIP call, so it is testable on a CPU-only box
    rc = l.rqhip_rq_forward(None, 4, 32, None, 3, 256, 1, 0.25, None, None, None, None, None, None, None, 0, None)
    assert rc == -1
    assert b"null pointer" in l.rqhip_last_error()
    rc = l.rqhip_kmeans_assign(None, -1, 32, None, 8, None, None)
    assert rc == -1


def test_workspace_queries():
    from rqhip import _lib
    l = _lib.lib()
    assert l.rqhip_rq_forward_workspace_bytes(3, 256) == (3 * 256 + 3) * 4
    assert l.rqhip_rq_forward_workspace_bytes(2, 100) == (2 * 128 + 2) * 4
    assert l.rqhip_rq_backward_workspace_bytes(1000, 32, 3, 256) == 3 * 1000 * 32 * 4 + 4 * 3 * 256 * 32 * 4
    assert l.rqhip_dedup_workspace_bytes(1000) >= 2048 * 4 + 4 * 1000 * 4
