#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import lib as L
L.get()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def bench(fn, flops, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"{name:40s} {t*1e6:8.1f} us {flops/t/1e12:6.1f} TF ({flops/t/157.3e12*100:4.1f}%)")
for (M, N, K, nm) in ((12576, 512, 2048, "ff2"), (12576, 512, 512, "outproj"), (12576, 512, 4096, "dgrad-like NT")):
    A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) * 0.05
    C = torch.zeros(M, N, device=dev); R = torch.randn(M, N, device=dev, generator=g); b = torch.randn(N, device=dev, generator=g)
    fl = 2.0 * M * N * K
    bench(lambda: L.gemm(L.OP_NT, A, W, C, M, N, K, K, K, N, epilogue=L.EPI_BIAS_RES, bias=b, R=R, ldr=N), fl, nm + " plain bias+res")
    for S in (2, 3, 4, 5, 8):
        if K // S < 64: continue
        def run():
            torch.add(R, b, out=C)
            L.gemm(L.OP_NT, A, W, C, M, N, K, K, K, N, epilogue=L.EPI_ATOMIC, split_k=S)
        bench(run, fl, nm + f" init + split-K {S} atomics")
