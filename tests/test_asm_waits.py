"""tools/check_asm_waits.py: the kernels that issue global loads from inline asm and wait for them with hand-counted
`s_waitcnt vmcnt(N)` (csrc/wgrad_jobs.hip, csrc/wgrad_split.hip) are checked against the ISA this toolchain generates -- no instruction
may mention a register whose load can still be in flight on some path of the kernel.  CPU only: hipcc cross-compiles to assembly."""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_waits as caw  # noqa: E402

HAZARD = """
_Zhazard:
	global_load_dwordx4 v[4:7], v[0:1], off
	global_load_dwordx4 v[8:11], v[2:3], off
.LBB0_1:
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v12, v4, v5
	v_mov_b32_e32 v13, v8
	global_load_dwordx4 v[4:7], v[0:1], off
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v14, v8, v9
	global_load_dwordx4 v[8:11], v[2:3], off
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v12, v8, v4
	s_endpgm
.Lfunc_end0:
"""
CLEAN = HAZARD.replace("\tv_mov_b32_e32 v13, v8\n", "")


def test_the_checker_flags_a_register_read_before_its_wait_and_nothing_else():
    """Two register sets refilled in turn, each awaited with vmcnt(1) = "all but the youngest load": reading the younger set's register
    behind the older set's wait is the finding (first iteration and, through the back edge, every later one)."""
    _, bad = caw.check(HAZARD)
    assert len(bad) == 1 and "v_mov_b32_e32 v13, v8" in bad[0], bad
    _, ok = caw.check(CLEAN)
    assert ok == [], ok
    # the same loop with the second wait dropped: the read of v8 behind it is no longer covered
    _, bad = caw.check(CLEAN.replace("\ts_waitcnt vmcnt(1)\n\tv_add_f32_e32 v14", "\tv_add_f32_e32 v14"))
    assert len(bad) == 1 and "v14, v8, v9" in bad[0], bad


def test_both_sides_of_a_divergent_if_cannot_be_skipped():
    """if (role) { wait } else { wait }: the path that skips both sides needs an empty exec mask and is not a path."""
    asm = """
_Zroles:
	global_load_dword v4, v[0:1], off
	s_and_saveexec_b64 s[2:3], vcc
	s_xor_b64 s[2:3], exec, s[2:3]
	s_cbranch_execz .LBB0_2
	s_waitcnt vmcnt(0)
	v_mov_b32_e32 v5, v4
.LBB0_2:
	s_or_saveexec_b64 s[4:5], s[2:3]
	s_xor_b64 exec, exec, s[4:5]
	s_cbranch_execz .LBB0_4
	s_waitcnt vmcnt(0)
	v_mov_b32_e32 v5, v4
.LBB0_4:
	s_or_b64 exec, exec, s[4:5]
	v_mov_b32_e32 v6, v4
	s_endpgm
.Lfunc_end0:
"""
    _, bad = caw.check(asm)
    assert bad == [], bad
    _, bad = caw.check(asm.replace("\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v5, v4\n.LBB0_4:", "\tv_mov_b32_e32 v5, v4\n.LBB0_4:"))
    assert len(bad) >= 1      # the else side without its wait: caught


@pytest.mark.timeout(600)
def test_the_shipped_kernels_with_hand_placed_waits_are_clean():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    for src in caw.DEFAULT:
        ks, findings = caw.check(caw.compile_to_asm(os.path.join(caw.CSRC, src)))
        assert ks and not findings, (src, findings[:5])
