#!/usr/bin/env python3
"""Time cvmi_hiera_mlp alone: stage-2 shape (262144 rows, C = 288) and stage-1 shape (1048576 rows, C = 144), 10 launches each."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from circuitvision_amd import _lib
from circuitvision_amd.engine import Buf, PackedHieraMlp, Plan, op_hiera_mlp
for C_, rows in ((288, 262144), (144, 1048576)):
    g = torch.Generator().manual_seed(0)
    pm = PackedHieraMlp(torch.randn(4 * C_, C_, generator=g) / C_ ** 0.5, torch.zeros(4 * C_), torch.randn(C_, 4 * C_, generator=g) / (4 * C_) ** 0.5, torch.zeros(C_), dtype=_lib.F16)
    xb = Buf(1, 1, rows, C_, _lib.F32); xb.t.normal_()
    gam, bet = torch.ones(C_).cuda(), torch.zeros(C_).cuda()
    plan = Plan(torch.cuda.Stream())
    for _ in range(10):
        op_hiera_mlp(plan, "mlp", pm, xb.view(), gam, bet, 1e-6)
    plan.run_eager(); plan.stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(plan.stream); plan.run_eager(); e1.record(plan.stream)
    plan.stream.synchronize()
    print(f"hiera_mlp C={C_} rows={rows}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us / launch", flush=True)
