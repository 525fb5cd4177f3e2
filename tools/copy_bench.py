#!/usr/bin/env python3
"""Calibration: achieved GB/s of a plain device copy (torch copy_) vs buffer size, timed like tools/layer_times.py."""
import torch
s = torch.cuda.Stream()
for mb in (4, 8, 16, 32, 64, 128, 256, 512, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.float16, device="cuda").normal_()
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(3):
            y.copy_(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record(s)
        for _ in range(reps):
            y.copy_(x)
        e1.record(s)
    s.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{mb:5d} MB read + {mb:5d} MB write: {ms * 1e3:8.1f} us  {2 * mb * 1.048576 / ms:8.0f} GB/s")
