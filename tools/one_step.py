#!/usr/bin/env python3
"""Run N eager (un-captured) steps of a workload's plan: the target of rocprofv3 --pmc / --kernel-trace passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from circuitvision_amd import _lib
workload = sys.argv[1] if len(sys.argv) > 1 else "yolo11n"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stream = torch.cuda.Stream()
if workload.startswith("yolo"):
    from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Plan, Yolo11Weights
    wt = Yolo11Weights(workload[-1], 62, SyntheticParams(0, 62), _lib.F16)
    yp = Yolo11Plan(wt, 32, 640, 640, stream, keep_scores=False)
    yp.set_input_nchw(torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(0)))
    plan = yp.plan
else:
    from circuitvision_amd.sam2 import HIERA_L, LORA_TARGETS_REFERENCE, Sam2Plan, Sam2Weights, SamSyntheticParams
    wt = Sam2Weights(SamSyntheticParams(0, LORA_TARGETS_REFERENCE), HIERA_L, 1024, _lib.F16)
    sp = Sam2Plan(wt, 16, stream)
    sp.x_in.t.normal_()
    plan = sp.plan
torch.cuda.synchronize()
for _ in range(steps):
    plan.run_eager()
stream.synchronize()
print("steps", steps)
