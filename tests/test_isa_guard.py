"""The build-time guard for the hand-written interpreters (tools/isa_check_asm_gaps.py): loads issued in one
asm statement of a word loop and waited for in a later one are invisible to the compiler; nothing it places inside that
loop may touch their registers. A synthetic listing shows the guard sees a planted spill and a planted copy; the real
kernels (stages.hip compiled to gfx950 assembly, no GPU needed) must come out clean."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_check_asm_gaps", os.path.join(ROOT, "tools", "isa_check_asm_gaps.py"))
guard = importlib.util.module_from_spec(spec)
spec.loader.exec_module(guard)

LISTING = """
_ZN3ovm20round0_cached_kernelILi1ELi1EEEvv:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:
\t;;#ASMSTART
\ts_bitcmp1_b32 s20, 9
\ts_load_dwordx16 s[36:51], s[2:3], 0x0
\tglobal_load_dword v40, v2, s[6:7]
\t;;#ASMEND
{between}
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\ts_bitcmp1_b32 s36, 9
\tv_add_u32 v3, v3, v40
\t;;#ASMEND
\ts_add_u32 s10, s10, 1
\ts_cmp_lt_u32 s10, s11
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
"""


@pytest.mark.parametrize("between,findings", [
    ("\ts_nop 0", 0),
    ("\ts_add_u32 s12, s12, 4", 0),
    ("\tv_writelane_b32 v60, s40, 3", 2),      # a spill, and it reads a register of the word group in flight
    ("\tv_mov_b32_e32 v41, v40", 1),            # a copy of a landing register whose load is in flight
    ("\ts_mov_b32 s44, 0", 1),                  # a re-materialisation into the word group
])
def test_guard_on_a_synthetic_word_loop(tmp_path, between, findings):
    p = tmp_path / "k.s"
    p.write_text(LISTING.format(between=between))
    stats, found = guard.check(str(p), ["round0_cached"])
    assert len(stats) == 1 and len(found) == findings, found


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_the_interpreters_word_loops_are_clean(tmp_path):
    out = str(tmp_path / "stages.s")
    guard.compile_to_asm(os.path.join(ROOT, "stark-backend_amd", "csrc", "stages.hip"), out)
    stats, found = guard.check(out, ["round0_cached", "gkr_leaves_cached"])
    assert len(stats) >= 10, stats           # every instantiation of the cached-column kernels was seen
    assert sum(s[2] for s in stats.values()) > 0  # ... and compiler-placed instructions inside word loops were examined
    assert found == [], found
