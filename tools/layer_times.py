#!/usr/bin/env python3
"""Per-launch timing table of a plan (event pairs on the engine stream): label, kind, us, plan bytes, GB/s, TFLOP/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from circuitvision_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="yolo11n")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--top", type=int, default=0)
    ap.add_argument("--prompts", type=int, default=0)
    a = ap.parse_args()
    stream = torch.cuda.Stream()
    if a.workload.startswith("yolo"):
        from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Plan, Yolo11Weights
        wt = Yolo11Weights(a.workload[-1], 62, SyntheticParams(0, 62), _lib.F16)
        yp = Yolo11Plan(wt, a.batch or 32, 640, 640, stream)
        yp.set_input_nchw(torch.rand(a.batch or 32, 3, 640, 640))
        plan = yp.plan
    else:
        from circuitvision_amd.sam2 import HIERA_L, LORA_TARGETS_REFERENCE, Sam2Plan, Sam2Weights, SamSyntheticParams
        wt = Sam2Weights(SamSyntheticParams(0, LORA_TARGETS_REFERENCE), HIERA_L, 1024, _lib.F16)
        sp = Sam2Plan(wt, a.batch or 16, stream, prompts=a.prompts)
        sp.x_in.t.normal_(0, 1)
        plan = sp.plan
    torch.cuda.synchronize()
    plan.timed_eager()
    acc = None
    for _ in range(a.reps):
        r = plan.timed_eager()
        acc = r if acc is None else [(l, k, m0 + m1, b, f) for (l, k, m0, b, f), (_, _, m1, _, _) in zip(acc, r)]
    rows = [(l, k, ms / a.reps, b, f) for l, k, ms, b, f in acc]
    tot = sum(r[2] for r in rows)
    if a.top:
        rows = sorted(rows, key=lambda r: -r[2])[:a.top]
    print(f"{'label':44s} {'kind':10s} {'us':>9s} {'MB':>9s} {'GB/s':>8s} {'TF/s':>7s}")
    for l, k, ms, b, f in rows:
        print(f"{l[:44]:44s} {k:10s} {ms * 1e3:9.1f} {b / 1e6:9.2f} {b / ms / 1e6 if ms > 0 else 0:8.0f} {f / ms / 1e9 if ms > 0 else 0:7.1f}")
    print(f"total {tot:.3f} ms")


if __name__ == "__main__":
    main()
