#!/usr/bin/env python3
"""What the fused LayerNorm prologue of tok_linear costs: stage-3 fc1 shape (65536 x 576 -> 2304, GELU) with the f32 + LayerNorm input
against the same launch on a ready fp16 matrix, and stage-3 qkv (-> 1728).  Event pairs around 20 launches each."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from circuitvision_amd import _lib
from circuitvision_amd.engine import PackedTokLinear, Plan, Rows, op_tok_linear
rows, K = 65536, 576
g = torch.Generator().manual_seed(0)
gam, bet = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.1).cuda()
xf = torch.randn(rows, K, generator=g).cuda()
xh = xf.half()
for N, act, name in ((2304, _lib.ACT_GELU, "fc1"), (1728, _lib.ACT_NONE, "qkv")):
    w = torch.randn(N, K, generator=g) / K ** 0.5
    pt = PackedTokLinear(w, torch.zeros(N), dtype=_lib.F16)
    out = torch.empty(rows, N, dtype=torch.half, device="cuda")
    for ln in (True, False):
        plan = Plan(torch.cuda.Stream())
        for _ in range(20):
            op_tok_linear(plan, "t", pt, Rows(xf if ln else xh, rows, K), Rows(out, rows, N), ln=(gam, bet, 1e-6) if ln else None, act=act)
        plan.run_eager(); plan.stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(plan.stream):
            e0.record(plan.stream); plan.run_eager(); e1.record(plan.stream)
        plan.stream.synchronize()
        print(f"{name} N={N} LayerNorm-fused={ln}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us / launch", flush=True)
