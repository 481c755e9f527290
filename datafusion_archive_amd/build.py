"""Builds the HIP extension in-tree: datafusion_archive_amd/lib/libdfx_hip.so (gfx950 only).

hipcc cross-compiles without a GPU.  The built library is git-ignored but travels to the GPU box
with the repo snapshot.  No JIT, no torch extension machinery: a plain C-ABI shared object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libdfx_hip.so")

# kernel translation units first (slowest): they compile in parallel
SOURCES = ["dfx_k_table1.hip", "dfx_k_table2.hip", "dfx_k_table3.hip", "dfx_k_table4.hip", "dfx_k_table8.hip", "dfx_k_core.hip",
           "dfx_k_reduce.hip", "dfx_k_partition.hip", "dfx_k_partition_v0.hip", "dfx_k_partition_v1.hip", "dfx_k_partition_v2.hip",
           "dfx_k_partition_v3.hip", "dfx_k_partition_v4.hip", "dfx_k_partition_v5.hip", "dfx_k_partition_v6.hip", "dfx_k_partition_v7.hip", "dfx_k_partition_v8.hip",
           "dfx_k_partition_v9.hip", "dfx_k_partition_v10.hip", "dfx_k_partition_v11.hip", "dfx_k_partition_v12.hip",
           "dfx_k_partition_v13.hip", "dfx_k_partition_v14.hip", "dfx_k_partition_v15.hip", "dfx_k_partition_v16.hip",
           "dfx_k_partition_v17.hip", "dfx_k_partition_v18.hip", "dfx_k_partition_v19.hip", "dfx_k_partition_v20.hip",
           "dfx_k_partition_v21.hip", "dfx_k_partition_v22.hip", "dfx_k_partition_v23.hip", "dfx_k_partition_v24.hip",
           "dfx_k_fewgroup.hip", "dfx_k_csv.hip", "dfx_k_sort.hip", "dfx_k_dict.hip", "dfx_host.cpp", "dfx_expr.cpp", "dfx_relation.cpp", "dfx_aggregate.cpp",
           "dfx_table.cpp", "dfx_csv.cpp", "dfx_sort.cpp", "dfx_exchange.cpp"]
HEADERS = ["dfx_device.hpp", "dfx_sigs.hpp", "dfx_numparse.hpp", "dfx_pow5_table.hpp", "dfx_csv_walk.hpp", "dfx_kernels.hpp", "dfx_kernels_inl.hpp", "dfx_k_table_inl.hpp", "dfx_k_partition_inl.hpp", "dfx_k_partition_ws_inl.hpp", "dfx_launch.hpp",
           "dfx_host.hpp", "dfx_relation.hpp", "../../include/dfx.h"]

# -ffp-contract=off : the reference never fuses a*b+c; projections must be bit-exact
# -munsafe-fp-atomics: hardware global_atomic_add_f64 / ds_add_f64 instead of CAS loops
CXXFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
            "-munsafe-fp-atomics", "-Wall", "-Wno-unused-result", "-fno-gpu-rdc"]
CXXFLAGS += os.environ.get("DFX_EXTRA_CXXFLAGS", "").split()  # debug builds (e.g. -DDFX_PA_TIMING)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm for the gfx950 build)")


def _stale(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _depfile_deps(depfile: str):
    """Prerequisites recorded by the compiler (-MD -MF) the last time this object was built; None if there is no record."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    if ":" not in text:
        return None
    deps = [d for d in text.split(":", 1)[1].split() if d.startswith(HERE) or d.startswith(os.path.dirname(HERE))]
    return deps or None


def _recorded_flags(obj: str) -> str:
    try:
        return open(obj + ".flags").read()
    except OSError:
        return ""


# ---- build-time guard: hand-scheduled registers of pass 2 ------------------------------------------------------------------
# k_partition_agg_lean (csrc/dfx_k_partition.hip) issues its row loads by inline assembly into VGPRs v88..v119, which the kernel
# withholds from the register allocator (amdgpu_num_vgpr(88)).  That is only sound while (a) the code object ALLOCATES those
# registers (its .vgpr_count covers them: a wave that was given fewer would have its loads land in another wave's registers) and
# (b) nothing the compiler generates touches them.  Round 2 found a silent-corruption bug of exactly this kind on the GPU;
# build() refuses to produce a library that has the next one (tests/test_build_guards.py runs the same check).
GUARD_SRC = "dfx_k_partition.hip"
RESERVED_VGPRS = range(88, 120)


def _asm_registers(operand_text: str):
    import re
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", operand_text):
        regs.add(int(m.group(1)))
    return regs


def check_pass2_reserved_registers(asm: str) -> int:
    """Raises RuntimeError naming the kernel and the instruction; returns the number of instantiations checked."""
    import re
    kernels = {m.group(1): m.group(2) for m in re.finditer(r"^(_ZN3dfx\d+k_partition_agg_lean\w*):.*?\n(.*?)\n\s*s_endpgm", asm, re.S | re.M)}
    if len(kernels) < 8:
        raise RuntimeError(f"build guard: expected every k_partition_agg_lean instantiation, found {len(kernels)}")
    meta = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n(?:(?!\.name:).*\n)*?\s+\.vgpr_count:\s+(\d+)", asm)}
    reserved = set(RESERVED_VGPRS)
    for k, body in kernels.items():
        if k not in meta:
            raise RuntimeError(f"build guard: no metadata for {k}")
        used = max(_asm_registers(body) & reserved, default=0)  # v118 with 12-byte rows (dwordx3 loads), v119 with 16-byte rows
        if used < 118:
            raise RuntimeError(f"build guard: {k}: the in-flight row registers v88..v119 are gone")
        if meta[k] <= used:
            raise RuntimeError(f"build guard: {k}: .vgpr_count = {meta[k]} does not cover v{used} (the in-flight row registers v88..v{used}): "
                               "a wave would be allocated fewer registers than the hand-written loads write")
        for line in body.split("\n"):
            ins = line.split(";")[0].strip()
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            op, _, rest = ins.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            if not (_asm_registers(rest) & reserved):
                continue
            if op in ("global_load_dwordx3", "global_load_dwordx4"):
                ok = _asm_registers(ops[0]) <= reserved and not (_asm_registers(",".join(ops[1:])) & reserved)
            elif op in ("v_mov_b32", "v_mov_b32_e32"):
                ok = not (_asm_registers(ops[0]) & reserved) and _asm_registers(ops[1]) <= reserved
            else:
                ok = False
            if not ok:
                raise RuntimeError(f"build guard: {k}: compiler-generated instruction touches a reserved register: {ins}")
    return len(kernels)


def pass2_guard_asm(hipcc: str, out_path: str) -> str:
    """device assembly of the pass-2 translation unit, with the flags the library is built with"""
    flags = [f for f in CXXFLAGS if f not in ("-fPIC",)]
    r = subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", "-o", out_path, os.path.join(CSRC, GUARD_SRC)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build guard: hipcc -S failed:\n" + r.stderr[-2000:])
    return open(out_path).read()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        depfile = obj + ".d"
        recorded = _depfile_deps(depfile)  # exact per-object prerequisites when known, else every header
        deps = ([sp] + recorded + [os.path.abspath(__file__)]) if recorded else ([sp] + hdrs)
        if force or _stale(obj, deps) or os.environ.get("DFX_EXTRA_CXXFLAGS", "") != _recorded_flags(obj):
            cmd = [hipcc] + CXXFLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-MD", "-MF", depfile, "-c", sp, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if "-c" in cmd:  # remember the extra flags this object was built with (a -D switch must rebuild it)
            with open(cmd[-1] + ".flags", "w") as fh:
                fh.write(os.environ.get("DFX_EXTRA_CXXFLAGS", ""))
        return r

    guard_obj = os.path.join(OBJDIR, os.path.splitext(GUARD_SRC)[0] + ".o")
    guard_stamp = guard_obj + ".guard"
    guard_due = any(cmd[-1] == guard_obj for cmd in jobs) or not os.path.exists(guard_stamp) or \
        (os.path.exists(guard_obj) and os.path.getmtime(guard_stamp) < os.path.getmtime(guard_obj))

    def guard(_unused=None):
        n = check_pass2_reserved_registers(pass2_guard_asm(hipcc, os.path.join(OBJDIR, "dfx_k_partition.guard.s")))
        if verbose:
            print(f"build guard: {n} instantiations of k_partition_agg_lean keep v88..v119 to the hand-written loads", flush=True)

    if jobs or guard_due:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs) + 1)) as ex:
            guard_future = ex.submit(guard) if guard_due else None  # (beside the compile jobs: ~15 s of its own)
            list(ex.map(run, jobs))
            if guard_future is not None:
                guard_future.result()  # raises: no library from a translation unit that fails the guard
        if guard_due:
            with open(guard_stamp, "w") as fh:
                fh.write("ok\n")
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
