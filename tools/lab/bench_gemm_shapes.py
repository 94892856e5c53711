#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import lib as L
if os.environ.get("MT_LIB"):
    L.LIB_PATH = os.environ["MT_LIB"]
L.get()
dev = "cuda"
def run(op, M, N, K, name):
    g = torch.Generator(device=dev).manual_seed(0)
    if op == L.OP_NT:
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
        fn = lambda: L.gemm(op, A, B, C, M, N, K, K, K, N)
    elif op == L.OP_NN:
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
        fn = lambda: L.gemm(op, A, B, C, M, N, K, K, N, N)
    else:
        A, B = torch.randn(K, M, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
        fn = lambda: L.gemm(op, A, B, C, M, N, K, M, N, N, epilogue=L.EPI_ATOMIC, split_k=0)
    C = torch.zeros(M, N, device=dev)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    fl = 2.0 * M * N * K
    print(f"{name:14s} {M:7d}x{N:5d}x{K:5d} {t*1e6:9.1f} us {fl/t/1e12:6.1f} TF ({fl/t/157.3e12*100:4.1f}%)")
run(L.OP_NT, 4096, 4096, 4096, "NT cube")
run(L.OP_NN, 4096, 4096, 4096, "NN cube")
run(L.OP_TN, 4096, 4096, 4096, "TN cube")
run(L.OP_NT, 8192, 8192, 512, "NT K512 big")
run(L.OP_NT, 12544, 512, 2048, "NT exact98")
run(L.OP_NT, 16384, 512, 2048, "NT 128tiles")     # 128*4 = 512 tiles = exactly 2/CU
run(L.OP_NT, 32768, 512, 2048, "NT 256tiles")     # 1024 tiles = 4/CU
run(L.OP_NT, 100608, 512, 2048, "NT 8xM")
run(L.OP_NT, 12576, 512, 2048, "NT ff2")
