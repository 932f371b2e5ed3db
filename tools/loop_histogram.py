#!/usr/bin/env python3
"""What the sample loop of the headline kernel issues, from the ISA ("41 % of the vector instructions are
not FP64 arithmetic -- list the top 10 by count and say which are removable").  CPU only: engine.hip is compiled for gfx950 with
build.sh's flags, the code object disassembled, the loops of k_fused<4, true, 0, true> found by their backward branches, and the
SAMPLE loop (the innermost loop that holds the image taps: global_load_ushort / dwordx4 and the fp32 blend) histogrammed by opcode
and by class.

    python tools/loop_histogram.py [tag]        -> profiles/<tag>_sample_loop_isa.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN5mbavo7k_fusedILi4ELb1ELi0ELb1E"  # k_fused<4, true, 0, true>: configs[1]'s kernel


def classify(op):
    if op.startswith("v_mfma"):
        return "f64 MFMA"
    if re.match(r"v_cvt", op):
        return "conversions (v_cvt_*)"
    if re.search(r"_f64(_e32|_e64|_dpp|_sdwa)?$", op) and op.startswith("v_"):
        return "f64 VALU"
    if re.search(r"_f(32|16)(_e32|_e64|_dpp|_sdwa)?$", op) and op.startswith("v_"):
        return "f32 VALU (the reference's fp32 bilinear blend)"
    if re.match(r"v_(mov|accvgpr|readlane|writelane|readfirstlane|swap|perm|bfi|permlane)", op):
        return "moves / lane traffic"
    if op.startswith("v_"):
        return "integer / logic VALU"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "scalar loads (pose entry through SGPRs)"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "waits / nops"
    if op.startswith("s_"):
        return "scalar ALU / branch"
    if op.startswith("ds_"):
        return "LDS"
    return "vector memory"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    with tempfile.TemporaryDirectory() as tmp:
        co, obj = os.path.join(tmp, "engine.co"), os.path.join(tmp, "engine.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--cuda-device-only", "-c",
                        os.path.join(ROOT, "mba-vo_amd", "csrc", "engine.hip"), "-o", co], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + co,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + obj], check=True)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], check=True, capture_output=True, text=True).stdout
    lines = dis.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <" + KERNEL, l))
    end = next(i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i]))
    ins = []
    for l in lines[start + 1:end]:
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            s = int(args.split()[0])
            if s > 32767 and a + 4 + (s - 65536) * 4 in idx:
                loops.append((idx[a + 4 + (s - 65536) * 4], i))
    # the sample loop: the SMALLEST loop that contains image taps and the fp32 blend but no MFMA (the outer product is per pixel)
    def has(lo, hi, pat):
        return any(re.match(pat, op) for _, op, _ in ins[lo:hi + 1])
    cands = [(hi - lo, lo, hi) for lo, hi in loops if has(lo, hi, r"global_load_ushort|global_load_dwordx4") and has(lo, hi, r"v_(pk_)?mul_f32")
             and not has(lo, hi, r"v_mfma")]
    n, lo, hi = min(cands)
    body = ins[lo:hi + 1]
    ops = collections.Counter(op for _, op, _ in body)
    cls = collections.Counter(classify(op) for _, op, _ in body)
    valu = sum(v for k, v in cls.items() if "VALU" in k or k.startswith("conversions") or k.startswith("moves") or "MFMA" in k)
    f64 = cls["f64 VALU"] + cls["f64 MFMA"]
    out = ["# %s_sample_loop_isa.txt -- the sample loop of k_fused<4, true, 0, true> (configs[1]'s kernel) by opcode; tools/loop_histogram.py" % tag,
           "# kernel: %d instructions, %d bytes; sample loop: %d instructions per iteration = ONE PAIR of blur samples of one pixel per lane"
           % (len(ins), ins[-1][0] - ins[0][0] + 4, len(body)),
           "# (the loop streams two pose entries through the scalar cache, taps keyframe intensity + gradient at 2 x 4 bilinear corners, blends in fp32 as the",
           "#  reference does, converts to fp64 and accumulates the 1 x 24 Jacobian row; the outer product runs once per pixel, outside this loop)",
           "",
           "class                                              instructions   share of the loop   share of its vector instructions"]
    for k, v in cls.most_common():
        is_v = "VALU" in k or k.startswith("conversions") or k.startswith("moves") or "MFMA" in k
        out.append("%-50s %8d        %5.1f %%            %s" % (k, v, 100.0 * v / len(body), ("%5.1f %%" % (100.0 * v / valu)) if is_v else "  -"))
    out += ["", "vector instructions %d, of them FP64 arithmetic %d = %.1f %% (not FP64: %.1f %%)" % (valu, f64, 100.0 * f64 / valu, 100.0 - 100.0 * f64 / valu),
            "", "top 25 opcodes:"]
    for op, v in ops.most_common(25):
        out.append("%6d  %-28s %s" % (v, op, classify(op)))
    path = os.path.join(ROOT, "profiles", "%s_sample_loop_isa.txt" % tag)
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
