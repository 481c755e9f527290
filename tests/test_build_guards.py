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
SRC = os.path.join(ROOT, "datafusion_archive_amd", "csrc", "dfx_k_partition.hip")
RESERVED = range(88, 120)


def _registers(operand_text):
    """VGPR numbers an operand string mentions: v93, v[92:94] ..."""
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", operand_text):
        regs.add(int(m.group(1)))
    return regs


@pytest.fixture(scope="module")
def pass2_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from datafusion_archive_amd import build as b
    out = str(tmp_path_factory.mktemp("asm") / "partition.s")
    flags = [f for f in b.CXXFLAGS if f not in ("-fPIC",)]
    subprocess.check_call([hipcc] + flags + ["--cuda-device-only", "-S", "-o", out, SRC], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm, name):
    """{mangled name: body} of every instantiation of `name`"""
    out = {}
    for m in re.finditer(r"^(_ZN3dfx\d+" + name + r"\w*):.*?\n(.*?)\n\s*s_endpgm", asm, re.S | re.M):
        out[m.group(1)] = m.group(2)
    return out


def test_pass2_reserved_registers_are_allocated_and_untouched(pass2_asm):
    kernels = _kernels(pass2_asm, "k_partition_agg_lean")
    assert len(kernels) >= 8, f"expected every k_partition_agg_lean instantiation, found {len(kernels)}"
    # (a) the code object's register allocation covers v119
    meta = {m.group(1): int(m.group(2)) for m in
            re.finditer(r"\.name:\s+(\S+)\n(?:(?!\.name:).*\n)*?\s+\.vgpr_count:\s+(\d+)", pass2_asm)}
    for k, body in kernels.items():
        assert k in meta, f"no metadata for {k}"
        used = max(_registers(body) & set(RESERVED))  # v118 with 12-byte rows (dwordx3 loads), v119 with 16-byte rows
        assert used >= 118, f"{k}: the in-flight row registers are gone?"
        assert meta[k] > used, f"{k}: .vgpr_count = {meta[k]} does not cover v{used} (the in-flight row registers v88..v{used})"
    # (b) only the hand-written instructions name v88..v119: loads INTO them, v_mov_b32 OUT of them
    for k, body in kernels.items():
        for line in body.split("\n"):
            ins = line.split(";")[0].strip()
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            op, _, rest = ins.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            touched = _registers(rest) & set(RESERVED)
            if not touched:
                continue
            if op in ("global_load_dwordx3", "global_load_dwordx4"):
                assert _registers(ops[0]) <= set(RESERVED) and not (_registers(",".join(ops[1:])) & set(RESERVED)), f"{k}: {ins}"
            elif op in ("v_mov_b32", "v_mov_b32_e32"):
                assert not (_registers(ops[0]) & set(RESERVED)) and _registers(ops[1]) <= set(RESERVED), f"{k}: {ins}"
            else:
                raise AssertionError(f"{k}: compiler-generated instruction touches a reserved register: {ins}")
