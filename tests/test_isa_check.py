"""The build-time check of the walk kernel's assembly (nbody_amd/csrc/check_pending_sgprs.py) on small
hand-written listings: registers of a scalar load in flight may not be touched before `s_waitcnt lgkmcnt(0)`."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location(
    "check_pending_sgprs", os.path.join(HERE, "..", "nbody_amd", "csrc", "check_pending_sgprs.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

GOOD = """
kern:
\ts_load_dwordx8 s[12:19], s[4:5], s0
\tv_add_f32_e32 v0, v1, v2
\ts_waitcnt lgkmcnt(0)
\tv_sub_f32_e32 v3, s14, v2
.LBB0_1:
\ts_lshl_b32 s0, s17, 5
\ts_load_dwordx8 s[20:27], s[4:5], s0
\tv_mov_b32_e32 v4, s15
\ts_cmp_lt_i32 s17, s30
\ts_cbranch_scc1 .LBB0_3
\ts_waitcnt lgkmcnt(0)
\tv_mov_b32_e32 v4, s21
\ts_endpgm
.LBB0_3:
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\tv_mov_b32_e32 v4, s22
\ts_branch .LBB0_1
\t.end_amdhsa_kernel
"""


def problems(asm):
    out = []
    for name, lines in chk.kernels(asm):
        out += chk.check(name, lines)
    return out


def test_clean_listing_passes():
    assert problems(GOOD) == []


def test_copy_before_the_wait_is_flagged():
    bad = GOOD.replace("\tv_add_f32_e32 v0, v1, v2\n", "\ts_mov_b64 s[40:41], s[14:15]\n")
    assert problems(bad) == ["s_mov_b64 s[40:41], s[14:15]"]


def test_read_on_the_branch_path_is_flagged():
    # the load of block .LBB0_1 is still in flight at .LBB0_3 if that block does not wait first
    bad = GOOD.replace("\ts_waitcnt vmcnt(0) lgkmcnt(0)\n", "\ts_waitcnt vmcnt(0)\n")
    assert "v_mov_b32_e32 v4, s22" in problems(bad)


def test_overwrite_of_pending_registers_is_flagged():
    bad = GOOD.replace("\tv_mov_b32_e32 v4, s15\n", "\ts_mov_b32 s23, 0\n")
    assert problems(bad) == ["s_mov_b32 s23, 0"]


def test_m0_may_only_serve_the_lane_write():
    ok = ["s_mov_b32 m0, s33", "s_nop 0", "v_writelane_b32 v3, s36, m0", "s_endpgm"]
    assert chk.check_m0(ok) == []
    assert chk.check_m0(ok[:1] + ["v_add_f32_e32 v0, v1, v2"] + ok[1:]) != []        # the sequence was split up
    assert chk.check_m0(ok + ["s_movrels_b32 s4, s8"]) == []                          # (no mention of m0: not this check's business)
    assert chk.check_m0(["s_mov_b32 m0, -1", "ds_read_b32 v1, v2", "s_endpgm"]) == ["s_mov_b32 m0, -1"]
    assert chk.check_m0(["v_readlane_b32 s4, v3, m0"]) == ["v_readlane_b32 s4, v3, m0"]
