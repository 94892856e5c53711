#!/usr/bin/env python3
"""Times mt_conv1x1_wgrad on the EfficientNet-B0 shapes of a 256-crop batch (HIP events) and prints achieved GB/s against
the algorithmic bytes rows*(2*Cout+Cin)*4.  Env MT_WGRAD_BPC / MT_WGRAD_DIAG are the kernel's tuning aids."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import lib as L

SHAPES = [  # (name, rows, Cout, Cin, gated)
    ("b0 project", 256 * 112 * 112, 16, 32, True), ("b1 expand", 256 * 112 * 112, 96, 16, False),
    ("b1 project", 256 * 56 * 56, 24, 96, True), ("b2 expand", 256 * 56 * 56, 144, 24, False),
    ("b2 project", 256 * 56 * 56, 24, 144, True), ("b3 project", 256 * 28 * 28, 40, 144, True),
    ("b4 expand", 256 * 28 * 28, 240, 40, False), ("b4 project", 256 * 28 * 28, 40, 240, True)]
lib = L.get()
only = sys.argv[1:] or None
for name, rows, co, ci, gated in SHAPES:
    if only and not any(o in name for o in only):
        continue
    du, z = torch.randn(rows, co, device="cuda"), torch.randn(rows, co, device="cuda")
    x = torch.randn(rows, ci, device="cuda")
    kabc = torch.randn(3, co, device="cuda")
    sc, sh = torch.randn(ci, device="cuda"), torch.randn(ci, device="cuda")
    hw = rows // 256
    gate = torch.rand(256, ci, device="cuda")
    dw = torch.zeros(co, ci, device="cuda")

    def run():
        L.check(lib.mt_conv1x1_wgrad(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x), L.ptr(sc) if gated else None,
                                     L.ptr(sh) if gated else None, L.ptr(gate) if gated else None, hw, L.ptr(dw), rows, co, ci,
                                     L.stream_ptr()), "wgrad")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = rows * (2 * co + ci) * 4 / 1e9
    print(f"{name:11s} {co:4d}x{ci:<4d} rows {rows:8d}: {us:7.1f} us  {gb / us * 1e6:7.0f} GB/s  ({gb * 1e3:.0f} MB)")
