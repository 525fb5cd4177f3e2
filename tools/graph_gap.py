#!/usr/bin/env python3
"""Cost of one dependent kernel node in a replayed HIP graph: N tiny casts chained on one lane (and the same work on
independent lanes), microseconds per node.  Tells how much of a latency-bound plan is launch floor + inter-node gap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from circuitvision_amd._lib import F16, F32
from circuitvision_amd.engine import Buf, Plan, op_cast

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
stream = torch.cuda.Stream()
a = Buf(1, 1, rows, 64, F16, zero=True)
b = Buf(1, 1, rows, 64, F32, zero=True)


def timed(plan, reps=20):
    plan.run(); stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.run()
    stream.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


plan = Plan(stream)
for i in range(N):
    op_cast(plan, f"c{i}", a.view() if i % 2 == 0 else b.view(), b.view() if i % 2 == 0 else a.view())
us = timed(plan)
print(f"chain of {N} nodes: {us:.1f} us per replay, {us / N:.2f} us per node")

plan2 = Plan(stream)
bufs = [(Buf(1, 1, rows, 64, F16, zero=True), Buf(1, 1, rows, 64, F32, zero=True)) for _ in range(4)]
plan2.fork()
for i in range(N):
    lane = i % 4
    plan2.lane(lane + 1)
    x, y = bufs[lane]
    op_cast(plan2, f"c{i}", x.view(), y.view())
plan2.lane(0)
plan2.join()
us2 = timed(plan2)
print(f"4 lanes x {N // 4} nodes: {us2:.1f} us per replay, {us2 / N:.2f} us per node")
