#!/usr/bin/env python3
"""Register / spill metadata and the instruction mix of the scan loop of the pass-1 kernels, from the device assembly of the
translation units as the library builds them (hipcc --cuda-device-only -S with build.py's flags; no GPU needed).
usage: kernel_meta.py [unit ...]   (default: dfx_k_partition_v0 = compile-time signatures, v9 = scan plan, v17 = scan plan with a 4-byte key)
Prints one line per kernel symbol matching k_partition_ws: SGPRs, spilled SGPRs, VGPRs, spilled VGPRs, scratch bytes, total
instructions, v_readlane / v_writelane counts (SGPR spill traffic).  profiles/r04_kernel_metadata.txt is its output."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from datafusion_archive_amd import build as B  # noqa: E402


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt or not names:
        return {n: n for n in names}
    out = subprocess.run([filt] + list(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    every = "--all" in sys.argv  # every kernel of the unit, not only k_partition_ws
    units = [a for a in sys.argv[1:] if not a.startswith("--")] or ["dfx_k_partition_v0", "dfx_k_partition_v9", "dfx_k_partition_v17"]
    hipcc = B._hipcc()
    flags = [f for f in B.CXXFLAGS if f != "-fPIC"]
    for u in units:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, u + ".s")
            r = subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", "-o", out, os.path.join(B.CSRC, u + ".hip")], capture_output=True, text=True)
            if r.returncode != 0:
                print(u, "failed:", r.stderr[-500:])
                continue
            txt = open(out).read()
        meta = {}
        for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
            body = m.group(2)
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", body).group(1))  # noqa: E731
            meta[m.group(1)] = (g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"))
        names = [n for n in meta if every or "k_partition_ws" in n]
        pretty = demangle(names)
        print(f"== {u}.hip")
        for n in names:
            body = re.search(r"^" + re.escape(n) + r":[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M)
            ins = [ln.split()[0] for ln in body.group(1).splitlines() if ln.startswith("\t") and not ln.strip().startswith((".", ";"))] if body else []
            rl = sum(1 for i in ins if i.startswith("v_readlane"))
            wl = sum(1 for i in ins if i.startswith("v_writelane"))
            s, ss, v, vs, scr = meta[n]
            short = re.sub(r"\(dfx::DevProgram.*", "", pretty[n]).replace("void dfx::", "")
            print(f"{short:78s} sgpr {s:3d} spilled_sgpr {ss:3d} vgpr {v:3d} spilled_vgpr {vs:2d} scratch {scr:4d} B  instructions {len(ins):5d}  v_readlane {rl:4d}  v_writelane {wl:4d}")


if __name__ == "__main__":
    main()
