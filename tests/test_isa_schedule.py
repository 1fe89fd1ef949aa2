"""the hand-scheduled LDS / global-load pipelines (wave_f64.hip's Row::issue / wait — the headline kernel — and
dfpwm_par.hip's AUKIT_DFF_ISSUE / WAIT, flac_fused.hip's masked window prefetch) are only correct if hipcc keeps its hands off the destination registers between the asm
statement that issues the loads and the one that waits for them.  tools/isa_check.py disassembles the translation units (hipcc
cross-compiles gfx950 without a GPU) and replays the hardware's counters over every kernel; this test runs it at every build."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_check  # noqa: E402

GOOD = """
_Z1kv: ; @_Z1kv
\tv_mov_b32_e32 v9, 0
\t;;#ASMSTART
\tds_read_b64 v[2:3], v9
\tds_read_b64 v[4:5], v9 offset:8
\t;;#ASMEND
\tv_add_f64 v[10:11], v[6:7], v[6:7]
\t;;#ASMSTART
\tds_read_b64 v[12:13], v9 offset:16
\t;;#ASMEND
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(1)
\t;;#ASMEND
\tv_add_f64 v[10:11], v[2:3], v[4:5]
\ts_waitcnt lgkmcnt(0)
\tv_add_f64 v[10:11], v[10:11], v[12:13]
\ts_endpgm
"""


def test_checker_accepts_a_correct_schedule_and_catches_an_early_use():
    v, n = isa_check.check(GOOD)
    assert n == 3 and v == []
    # the compiler copies a destination before its wait (what a spill or a coalesced move would look like)
    bad = GOOD.replace("\tv_add_f64 v[10:11], v[6:7], v[6:7]\n", "\tv_mov_b32_e32 v20, v3\n")
    v, _ = isa_check.check(bad)
    assert len(v) == 1 and v[0][3] == [3]
    # ... or uses the youngest load behind a wait that only covers the older ones
    bad = GOOD.replace("\tv_add_f64 v[10:11], v[2:3], v[4:5]\n", "\tv_add_f64 v[10:11], v[2:3], v[12:13]\n")
    v, _ = isa_check.check(bad)
    assert len(v) == 1 and v[0][3] == [12, 13]
    # global loads: a partial vmcnt retires nothing (loads and stores return out of order with respect to each other)
    vm = "_Z1gv: ; @_Z1gv\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[0:3], v[8:9], off\n\t;;#ASMEND\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v5, v0\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v5, v1\n\ts_endpgm\n"
    v, n = isa_check.check(vm)
    assert n == 1 and len(v) == 1 and v[0][3] == [0]


def test_cfg_checker_follows_branches():
    """check_vm_cfg: a hand-issued global load in one arm of a fork; the other arm may reuse the register, the join may not before vmcnt(0)."""
    base = ("_Z1hv: ; @_Z1hv\n\ts_cbranch_scc1 .LBB0_2\n; %bb.1:\n\t;;#ASMSTART\n\tglobal_load_dwordx2 v[2:3], v[8:9], off\n\t;;#ASMEND\n\ts_branch .LBB0_3\n"
            ".LBB0_2:\n\tv_mov_b32_e32 v2, 0\n.LBB0_3:\n\tv_add_u32_e32 v5, v6, v7\n{USE}\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v4, v3\n\ts_endpgm\n")
    v, n = isa_check.check_vm_cfg(base.replace("{USE}", ""))
    assert n == 1 and v == []          # (text order would flag the v_mov in the other arm)
    assert len(isa_check.check(base.replace("{USE}", ""))[0]) == 1
    v, _ = isa_check.check_vm_cfg(base.replace("{USE}", "\tv_mov_b32_e32 v4, v2\n"))
    assert len(v) == 1 and v[0][3] == [2]
    # a loop whose body requests at its end and waits at its top: the back edge carries the request to the instructions before the wait
    loop = ("_Z1iv: ; @_Z1iv\n.LBB1_1:\n\tv_mov_b32_e32 v10, v2\n\ts_waitcnt vmcnt(0)\n\t;;#ASMSTART\n\tglobal_load_dwordx2 v[2:3], v[8:9], off\n\t;;#ASMEND\n"
            "\ts_cbranch_scc1 .LBB1_1\n; %bb.2:\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n")
    v, _ = isa_check.check_vm_cfg(loop)
    assert len(v) == 1 and v[0][3] == [2]


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_rs_onepole_keeps_its_prefetch_registers():
    """flac_tail.hip: k_rs_onepole requests the next tile's window in inline asm and waits at the tile's end (every instantiation, on the control-flow graph)."""
    asm = isa_check.compile_asm(os.path.join(ROOT, "aukit_amd", "csrc", "flac_tail.hip"))
    v, n = isa_check.check_vm_cfg(asm)
    assert n >= 100, f"only {n} hand-issued loads found"
    assert not v, "\n".join(f"{k[:70]} line {ln}: `{s}` touches in-flight v{r}" for k, ln, s, r in v[:10])


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_rsp_keeps_its_prefetch_registers():
    """rs_periodic.hip: k_rsp requests the next tile's window the same way (two dwordx4 per lane), twelve instantiations."""
    asm = isa_check.compile_asm(os.path.join(ROOT, "aukit_amd", "csrc", "rs_periodic.hip"))
    v, n = isa_check.check_vm_cfg(asm)
    assert n >= 48, f"only {n} hand-issued loads found"
    assert not v, "\n".join(f"{k[:70]} line {ln}: `{s}` touches in-flight v{r}" for k, ln, s, r in v[:10])


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
@pytest.mark.parametrize("src,at_least", [("wave_f64.hip", 1000), ("dfpwm_par.hip", 16), ("wave_coef_f64.hip", 32), ("flac_fused.hip", 8)])
def test_hand_scheduled_kernels_keep_their_registers(src, at_least):
    asm = isa_check.compile_asm(os.path.join(ROOT, "aukit_amd", "csrc", src))
    v, n = isa_check.check(asm)
    assert n >= at_least, f"{src}: only {n} hand-issued loads found — the checker no longer sees the pipeline"
    assert not v, "\n".join(f"{k[:70]} line {ln}: `{s}` touches in-flight v{r}" for k, ln, s, r in v[:10])
