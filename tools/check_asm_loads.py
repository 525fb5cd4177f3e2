#!/usr/bin/env python3
"""Guard for inline-asm global loads whose data arrives LATER (gemm256x192r_kernel: the residual / bias loads the K-loop's counted `vmcnt`
retires).  hipcc treats an asm statement's outputs as valid the moment the statement ends: if it decides to move them (a `v_mov` / `v_pk_mov` /
`v_accvgpr_write` out of the destination registers, a spill, any read) before the kernel's own wait, it copies registers the load has not
written yet -- silently, and only sometimes wrong (it did exactly that to a first attn_res256 variant in r03: `global_load_dwordx4 v[2:5]`
followed by four `v_mov_b32` out of v2..v5).  The loads are safe while the destination registers are the loop-carried home of the value
("+v" operands) and nothing touches them until the consuming add many K-tiles later.  This check disassembles the kernel and fails if any
instruction between an asm `global_load_dwordx4 v[a:b], ...` and the NEXT `s_waitcnt vmcnt(..)` reads or writes v[a:b].

usage: tools/check_asm_loads.py     (exit status 0 = clean)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "circuitvision_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
TARGETS = {"igemm.hip": ["gemm256x192r_kernel"]}


def device_asm(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-S", "--cuda-device-only", *extra, src, "-o", out],
                       check=True, capture_output=True, cwd=CSRC)
        return open(out).read()


def kernel_body(asm, name):
    m = re.search(r"^(_Z\w*%s\w*):\s*(?:;.*)?$" % re.escape(name), asm, re.M)
    if not m:
        return None
    end = asm.index("s_endpgm", m.end())
    return asm[m.end():end].splitlines()


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(lines):
    """-> (asm loads seen, [(line number, instruction, registers touched early)])"""
    bad, seen, in_asm, pending = [], 0, False, {}          # pending: register -> line of the load that owns it
    for i, raw in enumerate(lines):
        ln = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ln.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        if in_asm and ln.startswith("global_load_dwordx4"):
            dest = ln.split(",")[0]
            for r in regs_of(dest):
                pending[r] = i
            seen += 1
            continue
        if ln.startswith("s_waitcnt") and "vmcnt" in ln:
            pending = {}
            continue
        if pending:
            hit = regs_of(ln) & set(pending)
            if hit:
                bad.append((i, ln, sorted(hit)))
    return seen, bad


def main():
    rc = 0
    for src, kernels in TARGETS.items():
        for extra in ([], ["-DCVMI_OPERAND_BF16"]):
            asm = device_asm(src, extra)
            for k in kernels:
                body = kernel_body(asm, k)
                if body is None:
                    print(f"{src}: {k}: kernel not found"); rc = 1
                    continue
                seen, bad = check(body)
                tag = f"{src}{' (bf16)' if extra else ''}: {k}"
                if not seen:
                    print(f"{tag}: no asm loads found"); rc = 1
                for i, ln, hit in bad:
                    print(f"{tag}: line {i}: `{ln}` touches v{hit} before the wait that retires their load"); rc = 1
                if seen and not bad:
                    print(f"{tag}: {seen} asm loads, destinations untouched until the next vmcnt wait")
    return rc


if __name__ == "__main__":
    sys.exit(main())
