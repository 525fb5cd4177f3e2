#!/usr/bin/env python3
"""Does slicing a Hiera MLP by image keep the hidden activation in the Infinity Cache?  fc1 (+GELU) -> fc2 (+f32 residual) at a
stage shape, over the whole batch vs image slices that reuse ONE hidden buffer.  us per MLP."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from circuitvision_amd._lib import ACT_GELU, ACT_NONE, F16, F32
from circuitvision_amd.engine import Buf, PackedConv, Plan, op_conv

B, G, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stream = torch.cuda.Stream()
x = Buf(B, G, G, C, F16); x.t.normal_()
y = Buf(B, G, G, C, F32, zero=True)
pc1 = PackedConv(torch.randn(4 * C, C, 1, 1) / C ** 0.5, torch.zeros(4 * C), F16)
pc2 = PackedConv(torch.randn(C, 4 * C, 1, 1) / (4 * C) ** 0.5, torch.zeros(C), F16)


def timed(plan, reps=10):
    plan.run(); stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.run()
    stream.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for nb in [B] + [n for n in (4, 2, 1) if n < B]:
    hid = Buf(nb, G, G, 4 * C, F16)
    plan = Plan(stream)
    for b0 in range(0, B, nb):
        xs, ys = x.images(b0, nb), y.images(b0, nb)
        op_conv(plan, "fc1", pc1, [(xs.view(), 0)], hid.view(), act=ACT_GELU)
        op_conv(plan, "fc2", pc2, [(hid.view(), 0)], ys.view(), res=ys.view())
    print(f"B={B} grid={G} C={C}: {nb:2d} images per slice (hidden {hid.nbytes / 1e6:.0f} MB): {timed(plan):9.1f} us")
