#!/usr/bin/env python3
"""Build check for the hand-counted LDS rings: tok_linear.hip, tok_linear16.hip and hiera_mlp.hip keep several inline-asm
`ds_read_b128` in flight and release each MFMA with a COUNTED `s_waitcnt lgkmcnt(N)`.  That is only valid while nothing else of the wave
counts on lgkmcnt inside the window: scalar memory loads (s_load / s_buffer_load / s_memtime / s_memrealtime) share the counter and return
out of order, so a compiler that re-materialised a kernel argument inside the window would let a partial wait release before the LDS
operand has arrived -- silently.  This script compiles the device code to assembly and fails on any such instruction between the first
ring read of a sequence and the `lgkmcnt(0)` wait that ends it.

usage: check_ring_asm.py [file.hip ...]      (default: the three ring kernels; exit code 1 on a violation)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "circuitvision_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SMEM = re.compile(r"^\s*(s_load_|s_buffer_load_|s_memtime|s_memrealtime|s_scratch_load)")


def device_asm(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-S", "--cuda-device-only", *extra, src, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
        return open(out).read()


def check(asm):
    """-> (kernels seen, ring sequences seen, [violations])"""
    kernels = sequences = 0
    bad = []
    func, in_asm, open_ring, block = None, False, False, []
    for ln, line in enumerate(asm.split("\n"), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func, open_ring = m.group(1), False
            kernels += 1
            continue
        if t.startswith(".end_amdhsa_kernel") or t.startswith(".Lfunc_end"):
            open_ring = False
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm, block = True, []
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            body = " ".join(block)
            if "ds_read_b128" in body:
                if not open_ring:
                    sequences += 1
                open_ring = True
            elif re.search(r"s_waitcnt\s+lgkmcnt\(0\)", body):
                open_ring = False                                    # everything the wave had in flight on lgkmcnt is back
            continue
        if in_asm:
            block.append(t)
            continue
        if open_ring and SMEM.match(t):
            bad.append((func, ln, t))
    return kernels, sequences, bad


def main(files):
    rc = 0
    for f in files:
        for tag, extra in (("fp16", ()), ("bf16", ("-DCVMI_OPERAND_BF16",))):
            k, s, bad = check(device_asm(os.path.join(CSRC, f), extra))
            print(f"{f} [{tag}]: {k} functions, {s} ring sequences, {len(bad)} scalar-memory instructions inside a ring window")
            for func, ln, t in bad[:10]:
                print(f"  {func} line {ln}: {t}")
            rc |= 1 if bad or s == 0 else 0
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["tok_linear16.hip", "hiera_mlp.hip", "tok_linear.hip"]))
