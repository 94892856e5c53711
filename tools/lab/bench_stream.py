#!/usr/bin/env python3
"""Calibration: what plain streaming kernels reach on this GPU (torch elementwise / reduction over GB-sized fp32 tensors)."""
import torch

n = 3211264 * 96          # 1.23 GB per tensor, the largest EfficientNet activation of a 256-crop batch
a, b = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
c = torch.empty_like(a)


def timed(fn, bytes_, name, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:28s} {us:8.1f} us  {bytes_ / us * 1e6 / 1e9:7.0f} GB/s")


timed(lambda: torch.add(a, b, out=c), 3 * n * 4, "c = a + b (2R + 1W)")
timed(lambda: c.copy_(a), 2 * n * 4, "copy (1R + 1W)")
timed(lambda: a.sum(), n * 4, "sum (1R)")
timed(lambda: torch.dot(a, b), 2 * n * 4, "dot (2R)")
timed(lambda: c.fill_(1.0), n * 4, "fill (1W)")
small = a[: 64 * 1024 * 1024 // 4]
timed(lambda: small.sum(), small.numel() * 4, "sum 64 MB (cache resident)")
