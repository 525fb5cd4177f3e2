#!/usr/bin/env python3
"""Guard for the MFMAs that hiera_mlp_kernel<288, 2, 2> issues from inline asm (the software-pipelined chunk loop).  hipcc's hazard recogniser
inserts the wait states an MFMA needs around it -- for MFMAs it can see.  Inside an asm statement it sees nothing, so this check reads the device
assembly of the shipped kernel (both operand-type builds) and fails when
  (a) a VGPR that a VALU instruction wrote is read by an MFMA fewer than MIN_VALU_TO_MFMA wait states later.  Measured r04 on gfx950: a
      `v_cvt_pk` that hipcc had sunk to 0 - 1 wait states in front of the asm block consuming it gave wrong results in whole waves; builds whose
      closest pair was 2 wait states apart were bit-identical to the reference loop; the kernel source keeps these operands behind `s_nop 1`
      statements, which this check verifies survived the compiler;
  (b) a register an MFMA wrote is read by a non-MFMA instruction fewer than MIN_MFMA_TO_VALU wait states later (hipcc uses `s_nop 11` = 12 for
      v_mfma_f32_32x32x16 results read by VALU instructions on this target).
Wait states are counted as instructions in between (`s_nop N` = N + 1), which undercounts (an MFMA holds the issue port longer): the check errs
on the side of reporting.

usage: tools/check_asm_mfma.py     (exit status 0 = clean)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "circuitvision_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KERNEL = "hiera_mlp_kernelILi288ELi2ELi2ELi0"          # <288, VAR 2, SLOTS 2, DIAG 0>
MIN_VALU_TO_MFMA, MIN_MFMA_TO_VALU = 2, 12
# (source, kernel name fragment, MFMAs expected at least, wait states an MFMA result needs before a non-MFMA read: 32x32x16 = 8 passes -> 12,
#  16x16x32 = 4 passes -> 8 in what hipcc emits for the builtins)
TARGETS = [("hiera_mlp.hip", KERNEL, 148, 12)]


def device_asm(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-S", "--cuda-device-only", *extra, src, "-o", out],
                       check=True, capture_output=True, cwd=CSRC)
        return open(out).read()


def kernel_body(asm, name):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if name in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\ts_endpgm"))
    return [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith((";", "."))]


def regs(tok):
    tok = tok.strip()
    m = re.match(r"([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def wait_states(between):
    n = 0
    for z in between:
        m = re.match(r"s_nop (\d+)", z)
        n += int(m.group(1)) + 1 if m else 1
    return n


def check(body, min_mfma_to_valu=MIN_MFMA_TO_VALU):
    """-> (number of MFMAs, list of violations)"""
    MIN_MFMA_TO_VALU = min_mfma_to_valu
    bad, n_mfma = [], 0
    ops = [re.split(r"[ ,]+", l) for l in body]
    for i, p in enumerate(ops):
        if p[0].startswith("v_mfma"):
            n_mfma += 1
            src = set().union(*(regs(t) for t in p[2:5]))
            for back in range(1, MIN_VALU_TO_MFMA + 1):             # (a) a VALU write right in front of it
                if i - back < 0:
                    break
                q = ops[i - back]
                if q[0].startswith("v_") and not q[0].startswith("v_mfma") and len(q) > 1 and regs(q[1]) & src:
                    ws = wait_states(body[i - back + 1:i])
                    if ws < MIN_VALU_TO_MFMA:
                        bad.append(f"VALU -> MFMA, {ws} wait states: `{body[i - back]}` -> `{body[i]}`")
            dst = regs(p[1])                                            # (b) its result read too early
            ws = 0
            for j in range(i + 1, len(ops)):
                if ws >= MIN_MFMA_TO_VALU:
                    break
                q = ops[j]
                if not q[0].startswith("v_mfma") and q[0].startswith(("v_", "ds_", "global_", "buffer_", "scratch_")):
                    read = set().union(*(regs(t) for t in (q[1:] if q[0].startswith(("ds_write", "global_store", "buffer_store", "scratch_store")) else q[2:])))
                    if read & dst:
                        bad.append(f"MFMA -> read, {ws} wait states: `{body[i]}` -> `{body[j]}`")
                        break
                if q[0].startswith("v_mfma") and regs(q[1]) & dst:      # the accumulator chain continues: later readers are checked from there
                    break
                m = re.match(r"s_nop (\d+)", body[j])
                ws += int(m.group(1)) + 1 if m else 1
    return n_mfma, bad


def main():
    rc = 0
    for tag, extra in (("fp16", ()), ("bf16", ("-DCVMI_OPERAND_BF16",))):
        asm = {}
        for src, kernel, n_min, mv in TARGETS:
            if src not in asm:
                asm[src] = device_asm(src, extra)
            body = kernel_body(asm[src], kernel)
            n, bad = check(body, mv)
            print(f"{src} [{tag}] {kernel}: {n} MFMAs in {len(body)} instructions, {len(bad)} hazard(s)")
            for b in bad[:20]:
                print("   ", b)
            rc |= bool(bad) or n < n_min
    return rc


if __name__ == "__main__":
    sys.exit(main())
