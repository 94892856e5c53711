#!/usr/bin/env python3
"""Micro-benchmark of mt_conv1x1_bwd_fused on the three EfficientNet-B0 shapes of a 256-crop batch (HIP events, 20 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mintime_amd import lib as L

lib = L.get()
for rows, cout, cin, with_res in [(256 * 112 * 112, 96, 16, False), (256 * 56 * 56, 144, 24, True), (256 * 56 * 56, 144, 24, False)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    du = torch.randn(rows, cout, device="cuda", generator=g)
    x = torch.randn(rows, cin, device="cuda", generator=g)
    W = torch.randn(cout, cin, device="cuda", generator=g) * 0.2
    kabc = torch.randn(3, cout, device="cuda", generator=g) * 0.3
    res = torch.randn(rows, cin, device="cuda", generator=g) if with_res else None
    dx = torch.empty(rows, cin, device="cuda")
    dW = torch.zeros(cout, cin, device="cuda")
    def run():
        L.check(lib.mt_conv1x1_bwd_fused(L.ptr(du), L.ptr(kabc), L.ptr(x), L.ptr(W), L.ptr(res), L.ptr(dx), L.ptr(dW), rows, cout, cin,
                                         L.stream_ptr()), "fused")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gb = rows * (cout + cin * (3 if with_res else 2)) * 4 / 1e9
    print(f"rows={rows} {cout}->{cin} res={with_res}: {us:.0f} us  {gb / us * 1e6:.0f} GB/s algorithmic")
