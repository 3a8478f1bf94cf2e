"""The ISA check of the inline-assembly .bed prefetch (tools/check_asm_prefetch.py) is part of the build; here its
own logic is pinned on a synthetic listing: a use of a destination register between request and wait must be caught,
also when it sits on the loop's back edge."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_asm_prefetch.py")

GOOD = """
_ZN4cusk14mxm_fp4_kernelILb1EEEvPKhPfmmmmim: ; @kernel
	global_load_dwordx4 v[10:13], v[2:3], off
	global_load_dwordx4 v[14:17], v[2:3], off offset:16
	global_load_dwordx4 v[20:23], v[2:3], off
	global_load_dwordx4 v[24:27], v[2:3], off offset:16
	s_branch .LBB0_2
.LBB0_1:
	s_waitcnt vmcnt(2)
	v_add_u32_e32 v30, v20, v24
	global_load_dwordx4 v[20:23], v[2:3], off
	global_load_dwordx4 v[24:27], v[2:3], off offset:16
.LBB0_2:
	s_waitcnt vmcnt(2)
	v_add_u32_e32 v31, v10, v14
	global_load_dwordx4 v[10:13], v[2:3], off
	global_load_dwordx4 v[14:17], v[2:3], off offset:16
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	v_add_u32_e32 v32, v10, v20
	s_endpgm
.Lfunc_end0:
"""


def _run(text, tmp_path, name):
    p = tmp_path / name
    p.write_text(text)
    return subprocess.run([sys.executable, TOOL, "--asm", str(p)], capture_output=True, text=True)


def test_clean_listing_passes(tmp_path):
    r = _run(GOOD, tmp_path, "good.s")
    assert r.returncode == 0, r.stdout + r.stderr


def test_copy_of_an_in_flight_register_on_the_back_edge_is_caught(tmp_path):
    bad = GOOD.replace("\ts_cbranch_scc1 .LBB0_1\n", "\tv_mov_b32_e32 v40, v11\n\ts_cbranch_scc1 .LBB0_1\n")
    r = _run(bad, tmp_path, "bad.s")
    assert r.returncode == 1 and "v_mov_b32_e32 v40, v11" in r.stderr


def test_wait_that_leaves_the_needed_set_in_flight_is_caught(tmp_path):
    bad = GOOD.replace(".LBB0_1:\n\ts_waitcnt vmcnt(2)", ".LBB0_1:\n\ts_waitcnt vmcnt(3)")
    r = _run(bad, tmp_path, "bad2.s")
    assert r.returncode == 1
