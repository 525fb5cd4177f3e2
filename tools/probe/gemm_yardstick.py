#!/usr/bin/env python3
"""Yardstick only (not product, not a baseline): what the vendor GEMM (hipBLASLt / rocBLAS through torch.matmul) reaches on this box for the four
Hiera stage-3 linear shapes, M = 65536 tokens -- plain GEMM, no LayerNorm / GELU / residual / f32 stream.  Tells how much of the gap between the
shipped kernels (0.34-0.39 of the nominal peak) and 1.0 is the chip's power-limited clock and how much is kernel design."""
import sys
import torch
M = 65536
shapes = [("fc1", 576, 2304), ("fc2", 2304, 576), ("qkv", 576, 1728), ("proj", 576, 576)]
for dt in (torch.float16, torch.bfloat16):
    for name, K, N in shapes:
        a = torch.randn(M, K, device="cuda", dtype=dt)
        w = torch.randn(N, K, device="cuda", dtype=dt)
        for _ in range(5):
            c = a @ w.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            c = a @ w.t()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{str(dt)[6:]:9s} {name:5s} M={M} K={K} N={N}: {us:7.1f} us  {2 * M * K * N / us / 1e6:7.1f} TFLOP/s", flush=True)
