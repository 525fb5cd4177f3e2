#!/usr/bin/env python3
"""YOLO11-n B=32 as S concurrent sub-batches (S graphs on S streams) vs one graph: is the small-launch tail latency-bound?"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from circuitvision_amd import _lib
from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Plan, Yolo11Weights

wt = Yolo11Weights("n", 62, SyntheticParams(0, 62), _lib.F16)
B = 32
for S in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    plans = [Yolo11Plan(wt, B // S, 640, 640, st) for st in streams]
    for p in plans:
        p.set_input_nchw(torch.rand(B // S, 3, 640, 640))
    torch.cuda.synchronize()
    for _ in range(5):
        for p in plans: p.plan.run()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        for p in plans: p.plan.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"S={S}: {dt * 1e3:.3f} ms per {B} images -> {B / dt:.0f} images/s", flush=True)
    del plans
