"""Timing of the attention core kernels (mt_attn_fwd / mt_attn_bwd, time and space) at B = 32 with plane outputs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import mintime_amd
from mintime_amd import lib as L

B, F, H, n = 32, 8, 8, 49
N, inner = 1 + F * n, 512
M = B * N
lib = L.get()
qkv = torch.randn(M, 3 * inner, device="cuda") * 0.5
do = torch.randn(M, inner, device="cuda")
mask = torch.ones(B, F, dtype=torch.uint8, device="cuda")
ident = torch.ones(B, F, F, dtype=torch.uint8, device="cuda")
o = torch.empty(M, inner, device="cuda")
dqkv = torch.empty(M, 3 * inner, device="cuda")
o_p, d_p = L.planes_empty(M, inner, "cuda"), L.planes_empty(M, 3 * inner, "cuda")


def timeit(f, reps=30):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for mode in (0, 1):
    for planes in (False, True):
        f = lambda: L.check(lib.mt_attn_fwd(L.ptr(qkv), None if planes else L.ptr(o), None, L.ptr(mask), L.ptr(ident), B, H, F, n, mode, 0.125,
                                            L.ptr(o_p) if planes else None, L.stream_ptr()), "f")
        b = lambda: L.check(lib.mt_attn_bwd(L.ptr(qkv), L.ptr(do), L.ptr(dqkv), L.ptr(mask), L.ptr(ident), B, H, F, n, mode, 0.125,
                                            L.ptr(d_p) if planes else None, L.stream_ptr()), "b")
        print(f"mode {mode} ({'time' if mode == 0 else 'space'}) planes={planes}: fwd {timeit(f):7.1f} us   bwd {timeit(b):7.1f} us")
