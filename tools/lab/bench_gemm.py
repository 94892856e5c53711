#!/usr/bin/env python3
"""GEMM micro-benchmark on the TimeSformer shapes; optional alternative .so for A/B runs (MT_LIB=path)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import lib as L
if os.environ.get("MT_LIB"):
    L.LIB_PATH = os.environ["MT_LIB"]
L.get()
M, D = 32 * 393, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
x, w1, b1 = r(M, D), r(8 * D, D) * 0.05, r(8 * D)
h, u = torch.empty(M, 4 * D, device=dev), torch.empty(M, 8 * D, device=dev)
wqkv, qkv = r(3 * D, D) * 0.05, torch.empty(M, 3 * D, device=dev)
w2, xo = r(D, 4 * D) * 0.05, torch.empty(M, D, device=dev)
dx, dW1 = r(M, D), torch.zeros(8 * D, D, device=dev)
du = r(M, 8 * D)
cases = {
    "ff1_geglu NT 12576x4096x512": (lambda: L.gemm(L.OP_NT, x, w1, h, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b1, C2=u, ldc2=8 * D, n_half=4 * D), 2 * M * 8 * D * D),
    "qkv NT 12576x1536x512": (lambda: L.gemm(L.OP_NT, x, wqkv, qkv, M, 3 * D, D, D, D, 3 * D), 2 * M * 3 * D * D),
    "ff2 NT+res 12576x512x2048": (lambda: L.gemm(L.OP_NT, h, w2, xo, M, D, 4 * D, 4 * D, 4 * D, D, epilogue=L.EPI_BIAS_RES, bias=b1, R=x, ldr=D), 2 * M * D * 4 * D),
    "dgrad NN 12576x512x4096": (lambda: L.gemm(L.OP_NN, du, w1, xo, M, D, 8 * D, 8 * D, D, D), 2 * M * D * 8 * D),
    "wgrad TN 4096x512x12576": (lambda: L.gemm(L.OP_TN, du, x, dW1, 8 * D, D, M, 8 * D, D, D, epilogue=L.EPI_ATOMIC, split_k=0), 2 * M * D * 8 * D),
}
for name, (fn, flops) in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"{name:32s} {t*1e6:8.1f} us  {flops/t/1e12:6.1f} TF  ({flops/t/157.3e12*100:4.1f}% of fp32 MFMA peak)")
