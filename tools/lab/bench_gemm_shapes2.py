#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import lib as L
L.get()
dev = "cuda"
def run(M, N, K):
    g = torch.Generator(device=dev).manual_seed(0)
    A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    C = torch.zeros(M, N, device=dev)
    fn = lambda: L.gemm(L.OP_NT, A, B, C, M, N, K, K, K, N)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    fl = 2.0 * M * N * K
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"NT {M:6d}x{N:5d}x{K:5d} tiles {tiles:5d} ({tiles/256:5.2f}/CU) {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF ({fl/t/157.3e12*100:4.1f}%)")
for M, N, K in ((12576, 1536, 512), (12544, 1536, 512), (16384, 1536, 512), (8192, 1536, 512), (8192, 2048, 512), (8192, 8192, 512),
                (32768, 1536, 512), (12576, 4096, 512), (16384, 4096, 512), (12576, 512, 512), (16384, 512, 512)):
    run(M, N, K)
