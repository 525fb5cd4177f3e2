#!/usr/bin/env python3
"""K-sweep of one plain GEMM shape: slope = main-loop time per 64-deep K-tile, intercept = prologue + epilogue."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from circuitvision_amd._lib import ACT_GELU, ACT_NONE, F16, F32
from circuitvision_amd.engine import Buf, PackedConv, Plan, op_conv


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1728
    stream = torch.cuda.Stream()
    for name, act, f32o in (("f16", ACT_NONE, False), ("gelu", ACT_GELU, False), ("f32+res", ACT_NONE, True)):
        pts = []
        for K in (256, 576, 1152, 2304):
            x = Buf(1, 1, M, K, F16); x.t.normal_()
            pc = PackedConv(torch.randn(N, K, 1, 1) / K ** 0.5, torch.zeros(N), F16)
            y = Buf(1, 1, M, N, F32 if f32o else F16, zero=True)
            plan = Plan(stream)
            op_conv(plan, name, pc, [(x.view(), 0)], y.view(), act=act, res=y.view() if f32o else None)
            torch.cuda.synchronize()
            plan.run_eager(); stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                plan.run_eager()
            e1.record(stream); stream.synchronize()
            us = e0.elapsed_time(e1) * 100
            pts.append((K, us))
            print(f"{name:8s} M={M} N={N} K={K:5d} {us:8.1f} us {2 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
        (k0, t0), (k1, t1) = pts[1], pts[3]
        slope = (t1 - t0) / ((k1 - k0) / 64)
        print(f"{name:8s} per K-tile {slope:.2f} us  -> main-loop rate {2 * M * N * 64 / slope / 1e6:.0f} TF/s; intercept {t0 - slope * k0 / 64:.1f} us")


if __name__ == "__main__":
    main()
