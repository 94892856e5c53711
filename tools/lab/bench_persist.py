"""Does a persistent grid by itself help the K = 512 shapes?  Stream-K launches whose units divide into whole tiles (no slabs)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import mintime_amd
from mintime_amd import lib as L


def timeit(f, reps=30):
    for _ in range(25):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for m, n, k in [(16384, 512, 512), (16384, 1024, 512), (16384, 2048, 512), (16384, 4096, 512), (16384, 1024, 2048), (12576, 1536, 512)]:
    A, W = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda") * 0.05
    a_p, w_p = L.split_planes_blk(A), L.split_planes_blk(W)
    c = torch.empty(m, n, device="cuda")
    t0 = timeit(lambda: L.gemm_planes(L.OP_NT, a_p, w_p, m, n, k, Cout=c, ldc=n, streamk=False))
    t1 = timeit(lambda: L.gemm_planes(L.OP_NT, a_p, w_p, m, n, k, Cout=c, ldc=n, streamk=True))
    fl = 2.0 * m * n * k
    print(f"{m} x {n} x {k}: {m // 128 * (n // 128) if m % 128 == 0 else -1} tiles | block per tile {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF-eq) | persistent (stream-K units) {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF-eq)")
