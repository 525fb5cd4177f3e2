#!/usr/bin/env python3
"""Per-launch times of a plan at two batch sizes side by side (event pairs, eager passes): which launches lose efficiency at the smaller
batch -- e.g. one rank's share of an 8-GPU job (8 images) against the tuned per-GPU batch (16 / 64).
usage: tools/batch_compare.py --workload sam2l|yolo11l|yolo11n --batches 8 16 [--top 40]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from circuitvision_amd import _lib  # noqa: E402


def table(workload, B, reps, wt_cache):
    stream = torch.cuda.Stream()
    if workload.startswith("yolo"):
        from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Plan, Yolo11Weights
        wt = wt_cache.setdefault("w", Yolo11Weights(workload[-1], 62, SyntheticParams(0, 62), _lib.F16))
        yp = Yolo11Plan(wt, B, 640, 640, stream, lanes=0)
        yp.set_input_nchw(torch.rand(B, 3, 640, 640, generator=torch.Generator().manual_seed(0)))
        plan = yp.plan
    else:
        from circuitvision_amd.sam2 import HIERA_L, LORA_TARGETS_REFERENCE, Sam2Plan, Sam2Weights, SamSyntheticParams
        wt = wt_cache.setdefault("w", Sam2Weights(SamSyntheticParams(0, LORA_TARGETS_REFERENCE), HIERA_L, 1024, _lib.F16))
        sp = Sam2Plan(wt, B, stream)
        sp.x_in.t.normal_(0, 1)
        plan = sp.plan
    torch.cuda.synchronize()
    plan.timed_eager(with_kernels=True)
    acc = {}
    order = []
    for _ in range(reps):
        for i, (label, kind, ms, b, f, kn) in enumerate(plan.timed_eager(with_kernels=True)):
            key = (i, label)
            if key not in acc:
                acc[key] = [kind, 0.0, b, f, kn]
                order.append(key)
            acc[key][1] += ms / reps
    return order, acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sam2l")
    ap.add_argument("--batches", type=int, nargs=2, default=[8, 16])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    cache = {}
    b0, b1 = a.batches
    o0, t0 = table(a.workload, b0, a.reps, cache)
    o1, t1 = table(a.workload, b1, a.reps, cache)
    rows = []
    for k0, k1 in zip(o0, o1):
        assert k0[1] == k1[1], (k0, k1)
        kind, ms0, _, _, kn0 = t0[k0]
        _, ms1, _, f1, kn1 = t1[k1]
        loss = ms0 - ms1 * b0 / b1                         # ms this launch takes at b0 beyond its share of the b1 time
        rows.append((loss, k0[1], kind, ms0 * 1e3, ms1 * 1e3, (kn0 or "")[:60], (kn1 or "")[:60] if kn1 != kn0 else "="))
    tot0, tot1 = sum(t0[k][1] for k in o0), sum(t1[k][1] for k in o1)
    print(f"{a.workload}: B={b0} {tot0:.3f} ms, B={b1} {tot1:.3f} ms; per-image ratio {tot1 / b1 / (tot0 / b0):.3f}; loss at B={b0}: {tot0 - tot1 * b0 / b1:.3f} ms")
    agg = {}
    for loss, label, kind, u0, u1, kn0, kn1 in rows:
        g = agg.setdefault((kn0, kn1), [0.0, 0, 0.0, 0.0])
        g[0] += loss; g[1] += 1; g[2] += u0; g[3] += u1
    print(f"\n{'loss ms':>8s} {'n':>4s} {'us@'+str(b0):>9s} {'us@'+str(b1):>9s}  kernel at B={b0} / at B={b1}")
    for (kn0, kn1), (loss, n, u0, u1) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"{loss:8.3f} {n:4d} {u0 / n:9.1f} {u1 / n:9.1f}  {kn0} / {kn1}")
    print(f"\n{'loss us':>8s} {'us@'+str(b0):>9s} {'us@'+str(b1):>9s}  label")
    for loss, label, kind, u0, u1, kn0, kn1 in sorted(rows, key=lambda r: -r[0])[:a.top]:
        print(f"{loss * 1e3:8.1f} {u0:9.1f} {u1:9.1f}  {label[:40]:40s} {kind:10s} {kn0}")


if __name__ == "__main__":
    main()
