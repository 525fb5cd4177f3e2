#!/usr/bin/env python3
"""Micro-benchmark of cvmi_conv2d as a plain GEMM on representative Hiera / YOLO shapes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from circuitvision_amd import _lib
from circuitvision_amd._lib import ACT_GELU, ACT_NONE, ACT_SILU, F16, F32
from circuitvision_amd.engine import Buf, PackedConv, Plan, op_conv

SHAPES = [  # name, M, N, K, act, out_f32+res
    ("s1.fc1 ", 1048576, 576, 144, ACT_GELU, False),
    ("s1.qkv ", 1048576, 432, 144, ACT_NONE, False),
    ("s1.proj", 1048576, 144, 144, ACT_NONE, True),
    ("s1.fc2 ", 1048576, 144, 576, ACT_NONE, True),
    ("s3.qkv ", 65536, 1728, 576, ACT_NONE, False),
    ("s3.fc1 ", 65536, 2304, 576, ACT_GELU, False),
    ("s3.fc2 ", 65536, 576, 2304, ACT_NONE, True),
    ("s3.proj", 65536, 576, 576, ACT_NONE, True),
    ("s2.qkv ", 262144, 864, 288, ACT_NONE, False),
    ("s2.fc1 ", 262144, 1152, 288, ACT_GELU, False),
    ("s2.fc2 ", 262144, 288, 1152, ACT_NONE, True),
    ("s4.qkv ", 16384, 3456, 1152, ACT_NONE, False),
    ("s4.fc1 ", 16384, 4608, 1152, ACT_GELU, False),
    ("s4.fc2 ", 16384, 1152, 4608, ACT_NONE, True),
    ("sq4096 ", 4096, 4096, 4096, ACT_NONE, False),
    ("sq8192 ", 8192, 8192, 8192, ACT_NONE, False),
    ("y.2cv2 ", 819200, 64, 48, ACT_SILU, False),
    ("y.4cv2 ", 204800, 128, 96, ACT_SILU, False),
]


def main():
    stream = torch.cuda.Stream()
    for name, M, N, K, act, f32o in SHAPES:
        x = Buf(1, 1, M, K, F16); x.t.normal_()
        w = torch.randn(N, K, 1, 1) / K ** 0.5
        pc = PackedConv(w, torch.zeros(N), F16)
        y = Buf(1, 1, M, (N + 7) // 8 * 8, F32 if f32o else F16, zero=True)
        plan = Plan(stream)
        op_conv(plan, name, pc, [(x.view(), 0)], y.view(0, N), act=act, res=y.view(0, N) if f32o else None)
        torch.cuda.synchronize()
        plan.run_eager(); stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record(stream)
        for _ in range(reps):
            plan.run_eager()
        e1.record(stream); stream.synchronize()
        ms = e0.elapsed_time(e1) / reps
        byt = M * K * 2 + M * N * (12 if f32o else 2)
        print(f"{name} M={M:8d} N={N:5d} K={K:5d}  {ms * 1e3:8.1f} us  {2 * M * N * K / ms / 1e9:7.1f} TF/s  {byt / ms / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
