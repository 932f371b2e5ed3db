#!/usr/bin/env python3
"""Per-kernel register / LDS / spill footprint from the CODE-OBJECT metadata (not from a profiler trace, whose
VGPR_Count column reports allocation granules on gfx950): every .hip of the library is compiled to device assembly
with the flags of mba-vo_amd/build.sh and the amdhsa.kernels notes are printed.

    python tools/kernel_resources.py [tag]      -> profiles/<tag>_kernel_resources.txt   (runs on CPU, no GPU needed)
"""
import os
import re
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mba-vo_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument".split()
SOURCES = [("engine.hip", ["-fno-slp-vectorize"]), ("ba_tracker.hip", []), ("image_ops.hip", []), ("keyframe_ops.hip", []),
           ("lm_batch.hip", [])]
KEYS = [".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
        ".group_segment_fixed_size", ".max_flat_workgroup_size"]


def device_asm(src, extra, out):
    subprocess.run([HIPCC] + FLAGS + extra + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels_of(asm):
    i = asm.index("amdhsa.kernels:")
    blk = asm[asm.rindex("---", 0, i):asm.index("...", i)]
    return yaml.safe_load(blk.replace("\t", "  "))["amdhsa.kernels"]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def spill_sites(asm, name):
    """v_writelane / v_readlane (scalar spills into VGPR lanes) and scratch_ accesses inside the kernel body, in all
    and inside its innermost loops (a label that a later s_cbranch jumps back to)."""
    m = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(name), asm, re.S | re.M)
    if not m:
        return None
    body = m.group(1).splitlines()
    return dict(v_writelane=sum("v_writelane" in l for l in body), v_readlane=sum("v_readlane" in l for l in body),
                scratch=sum(re.search(r"\bscratch_(load|store)", l) is not None for l in body), instructions=sum(
                    1 for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    lines = ["# kernel resources from the code-object metadata (hipcc %s; tools/kernel_resources.py)" % " ".join(FLAGS),
             "# dynamic LDS (extern __shared__) is not in .group_segment_fixed_size: k_fused<4,true,*> takes 12 x 12 800 B slabs "
             "+ 192 B = 153 792 B, k_fused<2,true,*> 16 x 6 656 + 256 B, k_fused_sp see SpLds in engine.hip",
             ""]
    with tempfile.TemporaryDirectory() as td:
        for src, extra in SOURCES:
            asm = device_asm(src, extra, os.path.join(td, src + ".s"))
            ks = kernels_of(asm)
            names = demangle([k[".name"] for k in ks])
            lines.append("## %s" % src)
            for k, n in zip(ks, names):
                sp = spill_sites(asm, k[".name"]) or {}
                n = re.sub(r"\(.*", "", n).replace("void ", "")
                lines.append("%-62s %s" % (n, "  ".join("%s=%s" % (x.strip("."), k.get(x)) for x in KEYS)
                                           + "  | v_writelane=%s v_readlane=%s scratch_ops=%s instrs=%s" % (
                                               sp.get("v_writelane"), sp.get("v_readlane"), sp.get("scratch"), sp.get("instructions"))))
            lines.append("")
    out = os.path.join(ROOT, "profiles", "%s_kernel_resources.txt" % tag)
    open(out, "w").write("\n".join(lines))
    print("\n".join(lines[:12]))
    print("... ->", out)


if __name__ == "__main__":
    main()
