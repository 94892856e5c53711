#!/usr/bin/env python3
"""Per-block phase timestamps of one mt_gemm launch (tuning aid mt_debug_gemm_trace): where does a tile's time go?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import mintime_amd
from mintime_amd import lib as L
lib = L.get()
lib.mt_debug_gemm_trace.argtypes = [ctypes.c_void_p]
lib.mt_debug_gemm_trace.restype = None
M, D = 32 * 393, 512
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x, w1, b1 = r(M, D), r(8 * D, D) * 0.05, r(8 * D)
h, u = torch.empty(M, 4 * D, device="cuda"), torch.empty(M, 8 * D, device="cuda")
wqkv, qkv = r(3 * D, D) * 0.05, torch.empty(M, 3 * D, device="cuda")
cases = {
    "qkv NT 12576x1536x512": lambda: L.gemm(L.OP_NT, x, wqkv, qkv, M, 3 * D, D, D, D, 3 * D),
    "ff1_geglu NT 12576x4096x512": lambda: L.gemm(L.OP_NT, x, w1, h, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b1, C2=u, ldc2=8 * D, n_half=4 * D),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.mt_debug_gemm_trace(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.mt_debug_gemm_trace(None)
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    us = lambda v: v / 100.0                      # s_memrealtime ticks at 100 MHz
    pro, loop, epi = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
    start, end = us(t[:, 0] - t0), us(t[:, 3] - t0)
    print(f"{name}: {len(t)} blocks, span {end.max():.1f} us")
    for nm, v in (("prologue", pro), ("main loop", loop), ("epilogue", epi), ("block total", end - start)):
        print(f"   {nm:12s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
    # residency: how many blocks are alive over time (all CUs), and how start times cluster
    order = np.argsort(start)
    print("   start-time deciles (us):", np.round(np.percentile(start, [0, 10, 25, 50, 75, 90, 100]), 1))
    alive = [(int(((start <= q) & (end > q)).sum())) for q in np.linspace(0, end.max(), 11)[1:-1]]
    print("   blocks alive at 10%..90% of the span:", alive)
    busy = (pro.sum() + loop.sum() + epi.sum())
    print(f"   sum of block time / (span x 768 slots) = {busy / (end.max() * 768):.2f};  loop share of block time {loop.sum() / busy:.2f}")
