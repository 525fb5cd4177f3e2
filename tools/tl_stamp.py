#!/usr/bin/env python3
"""Where a tok_linear chunk interval spends its cycles (diagnostic build with s_memtime stamps; SAM 2.1-L B=16, stage-3 qkv / fc1 launches)."""
import ctypes as C
import os
import sys
os.environ.setdefault("CVMI_TOKLIN_STAMP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from circuitvision_amd import _lib
from circuitvision_amd.sam2 import HIERA_L, LORA_TARGETS_REFERENCE, Sam2Plan, Sam2Weights, SamSyntheticParams
lib = _lib.load()
wt = Sam2Weights(SamSyntheticParams(0, LORA_TARGETS_REFERENCE), HIERA_L, 1024, _lib.F16)
sp = Sam2Plan(wt, 16, torch.cuda.Stream())
sp.x_in.t.normal_()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 24)()
pp = os.environ.get("CVMI_TOKLIN_PP", "1") != "0"
for rep in range(2):
    sp.plan.run_eager(); sp.plan.stream.synchronize()
    _lib.check(lib.cvmi_debug_stamps(buf), "stamps")
    v = [int(buf[k]) for k in range(24)]
    if pp:
        for name, o, labels in (("wave 0 (leading)", 0, ("b1 wait", "MFMAs", "vmcnt wait", "b2 wait", "epilogue", "prefetch issue")),
                                ("wave 4 (trailing)", 12, ("b1 wait", "epilogue", "b2 wait", "MFMAs", "vmcnt wait", "prefetch issue"))):
            ch, n = v[o + 7], v[o + 8]
            if n:
                print(f"pass {rep} {name}: {n} launches, {ch} chunks; per chunk: " + ", ".join(f"{l} {v[o + k] / ch:.0f}" for k, l in enumerate(labels))
                      + f"; per launch: loop {sum(v[o:o + 6]) / n:.0f} of {v[o + 6] / n:.0f} cycles")
    else:
        w, b, i, m, tot, ch, n = v[:7]
        if n:
            print(f"pass {rep}: {n} stamped launches, {ch} chunks; per chunk: DMA-wait {w / ch:.0f}, barrier {b / ch:.0f}, issue (epilogue + prefetch) {i / ch:.0f}, "
                  f"MFMA sequence {m / ch:.0f} cycles; per launch: loop {(w + b + i + m) / n:.0f} of {tot / n:.0f} cycles (prologue = the rest)")
