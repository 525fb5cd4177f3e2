#!/usr/bin/env python3
"""Occupancy guard: kernels whose design counts on N resident waves per SIMD must stay inside that register budget, without spills.

The unified register file of a CDNA4 SIMD gives a wave 512 / (waves per SIMD) registers: 128 at four waves.  `__launch_bounds__` only tells
hipcc the MINIMUM occupancy to allow; a kernel that drifts from 127 to 139 registers still builds, runs correctly -- and silently drops to
half the resident workgroups (attn_dma72_kernel did exactly that in r03: 830 us per launch instead of 730).  This reads the kernel
descriptors out of the device assembly of a source file and checks (vgpr count, spill count) of the kernels listed in BUDGETS.

usage: tools/check_occupancy.py            (exit status 0 = all inside their budgets)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "circuitvision_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# source -> (extra flags, {substring of the mangled kernel name: max registers})
BUDGETS = {
    "attention.hip": (["-fno-honor-nans"], {
        "attn_dma72_kernelILi8ELb0ELb0EE": 128,   # long sequences: two 8-wave workgroups per CU
        "attn_dma72_kernelILi8ELb0ELb1EE": 128,   # the same with the running maximum on the matrix pipe (q_log2: Hiera global attention)
        "attn_res256_kernelILi8ELb0ELb0EE": 128,  # 16 x 16 windows: two 8-wave workgroups per CU (2 x 79 KiB of LDS)
        "attn_res256_kernelILi8ELb0ELb1EE": 128,  # the same on log2-prescaled q (Hiera stage 3)
        "attn_res64_kernelILi2EE": 128,           # 8 x 8 windows: launch bounds ask for 4 waves per SIMD
    }),
}


def kernel_table(src, extra):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-S", "--cuda-device-only", *extra, src, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
        text = open(out).read()
    table = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        table[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return table


def parse_table(text):
    """(for the self-test) the same extraction on a given metadata text"""
    return {m.group(1): (int(m.group(2)), int(m.group(3)))
            for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)}


def check(table, budgets):
    bad = []
    for key, cap in budgets.items():
        hits = [(n, v) for n, v in table.items() if key in n]
        if not hits:
            bad.append((key, "kernel not found"))
        for n, (regs, spills) in hits:
            if regs > cap or spills:
                bad.append((n, f"{regs} registers (budget {cap}), {spills} spilled"))
    return bad


def main():
    rc = 0
    for src, (extra, budgets) in BUDGETS.items():
        for bf16 in ([], ["-DCVMI_OPERAND_BF16"]):
            table = kernel_table(src, extra + bf16)
            bad = check(table, budgets)
            for n, why in bad:
                print(f"{src}{' (bf16)' if bf16 else ''}: {n}: {why}")
            rc |= bool(bad)
            if not bad:
                print(f"{src}{' (bf16)' if bf16 else ''}: " + ", ".join(f"{k} {v[0]}" for k, v in table.items() if any(b in k for b in budgets)))
    return rc


if __name__ == "__main__":
    sys.exit(main())
