#!/usr/bin/env python3
"""A/B builds of the library: recompile SOME translation units with extra -D switches and link them with the other, unchanged
objects into datafusion_archive_amd/lib/variants/libdfx_<name>.so (loaded with DFX_LIB=<path>, see _ffi.py).
usage: build_variant.py <name> <tu.hip>[,<tu.hip>...] -DFOO=1 [-DBAR ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from datafusion_archive_amd import build as b  # noqa: E402

name, tus, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
b.build()
vdir = os.path.join(b.LIBDIR, "variants")
odir = os.path.join(b.OBJDIR, "variant_" + name)
os.makedirs(vdir, exist_ok=True)
os.makedirs(odir, exist_ok=True)
objs, jobs = [], []
for src in b.SOURCES:
    obj = os.path.join(b.OBJDIR, os.path.splitext(src)[0] + ".o")
    if src in tus:
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        jobs.append([b._hipcc()] + b.CXXFLAGS + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(b.CSRC, src), "-o", obj])
    objs.append(obj)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
with ThreadPoolExecutor(max_workers=int(os.environ.get("DFX_BUILD_JOBS", "8"))) as ex:  # (the translation units compile side by side, as in build.py)
    list(ex.map(subprocess.check_call, jobs))
out = os.path.join(vdir, f"libdfx_{name}.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
