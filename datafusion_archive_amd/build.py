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
SOURCES = ["dfx_k_table1.hip", "dfx_k_table2.hip", "dfx_k_table3.hip", "dfx_k_table4.hip", "dfx_k_core.hip",
           "dfx_k_reduce.hip", "dfx_k_partition.hip", "dfx_k_partition_v0.hip", "dfx_k_partition_v1.hip", "dfx_k_partition_v2.hip",
           "dfx_k_partition_v3.hip", "dfx_k_partition_v4.hip", "dfx_k_partition_v5.hip", "dfx_k_partition_v6.hip", "dfx_k_partition_v7.hip", "dfx_k_partition_v8.hip",
           "dfx_k_partition_v9.hip", "dfx_k_partition_v10.hip", "dfx_k_partition_v11.hip", "dfx_k_partition_v12.hip",
           "dfx_k_partition_v13.hip", "dfx_k_partition_v14.hip", "dfx_k_partition_v15.hip", "dfx_k_partition_v16.hip",
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

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
