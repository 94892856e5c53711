#!/usr/bin/env python3
"""Split-K sweep of the skinny NN dgrad GEMMs (N = 512): out[M,512] = dY[M,K] . W[K,512] with fp32 atomics onto a zeroed output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import lib as L
L.get()
M, D = 32 * 393, 512
for K in (4096, 1536, 512):
    dY, W = torch.randn(M, K, device="cuda"), torch.randn(K, D, device="cuda") * 0.05
    out = torch.zeros(M, D, device="cuda")
    line = f"NN {M}x{D}x{K}: "
    for sk in (1, 2, 3, 4, 5, 6, 8, 10, 13):
        if K // sk < 128:
            continue
        def run():
            if sk == 1:
                L.gemm(L.OP_NN, dY, W, out, M, D, K, K, D, D)
            else:
                out.zero_()
                L.gemm(L.OP_NN, dY, W, out, M, D, K, K, D, D, epilogue=L.EPI_ATOMIC, split_k=sk)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f" sk{sk}={us:6.1f}({2.0 * M * D * K / us / 1e6 / 157.3 * 100:3.0f}%)"
    print(line)
