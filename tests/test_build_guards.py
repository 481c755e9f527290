"""Guards for hand-scheduled device code (CPU test: hipcc cross-compiles without a GPU).

k_partition_agg_lean (pass 2 of the partitioned GROUP BY, csrc/dfx_k_partition.hip) issues its row loads by inline
assembly into VGPRs v88..v119, which the kernel withholds from the register allocator (amdgpu_num_vgpr(88)).  That is
only sound while (a) the code object ALLOCATES those registers (its .vgpr_count covers v119: a wave that was given fewer
registers would have its loads land in another wave's registers) and (b) nothing the compiler generates touches them.
Round 2 found a silent-corruption bug of exactly this kind on the GPU; this test finds the next one at build time.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pass2_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from datafusion_archive_amd import build as b
    return b.pass2_guard_asm(hipcc, str(tmp_path_factory.mktemp("asm") / "partition.s"))


def test_pass2_reserved_registers_are_allocated_and_untouched(pass2_asm):
    """the check build() itself runs whenever dfx_k_partition.o is rebuilt (datafusion_archive_amd/build.py)"""
    from datafusion_archive_amd import build as b
    assert b.check_pass2_reserved_registers(pass2_asm) >= 8


def test_the_guard_refuses_a_build_that_breaks_the_register_window(pass2_asm):
    """(a) a code object whose .vgpr_count stops short of the window, (b) a compiler-generated instruction inside it"""
    from datafusion_archive_amd import build as b
    short = re.sub(r"(\.vgpr_count:\s+)\d+", r"\g<1>96", pass2_asm)
    with pytest.raises(RuntimeError, match="does not cover"):
        b.check_pass2_reserved_registers(short)
    m = re.search(r"^(_ZN3dfx\d+k_partition_agg_lean\w*):.*?\n", pass2_asm, re.M)
    poisoned = pass2_asm[:m.end()] + "\tv_add_u32_e32 v90, v1, v2\n" + pass2_asm[m.end():]
    with pytest.raises(RuntimeError, match="touches a reserved register"):
        b.check_pass2_reserved_registers(poisoned)
