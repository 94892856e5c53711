#!/usr/bin/env python3
"""Split-K sweep of the TN weight-gradient GEMM (BN-backward prologue, atomic epilogue) on EfficientNet-B0's late-layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import lib as L
L.get()
SHAPES = [(1152, 192, 12544), (192, 1152, 12544), (1280, 320, 12544), (320, 1152, 12544), (672, 112, 50176), (112, 672, 50176),
          (480, 80, 50176), (80, 480, 50176), (192, 672, 12544)]
for (co, ci, rows) in SHAPES:
    du, z = torch.randn(rows, co, device="cuda"), torch.randn(rows, co, device="cuda")
    x = torch.randn(rows, ci, device="cuda")
    kabc = torch.randn(3, co, device="cuda")
    dw = torch.zeros(co, ci, device="cuda")
    line = f"{co:5d}x{ci:<5d} K={rows:6d}: "
    for sk in (0, 2, 4, 7, 14, 28, 49):
        def run():
            L.gemm(L.OP_TN, du, x, dw, co, ci, rows, co, ci, ci, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=sk, A2=z,
                   scale=kabc[0], shift=kabc[1], gate=kabc[2])
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        line += f" sk{sk}={e0.elapsed_time(e1) / 20 * 1e3:6.1f}"
    print(line + "  us")
