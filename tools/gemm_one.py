#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from circuitvision_amd._lib import ACT_NONE, F16
from circuitvision_amd.engine import Buf, PackedConv, Plan, op_conv
M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
stream = torch.cuda.Stream()
x = Buf(1, 1, M, K, F16); x.t.normal_()
pc = PackedConv(torch.randn(N, K, 1, 1) / K ** 0.5, torch.zeros(N), F16)
y = Buf(1, 1, M, N, F16, zero=True)
plan = Plan(stream)
op_conv(plan, "g", pc, [(x.view(), 0)], y.view(), act=ACT_NONE)
torch.cuda.synchronize()
for _ in range(reps):
    plan.run_eager()
stream.synchronize()
print("done")
