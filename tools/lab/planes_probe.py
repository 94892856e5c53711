"""mt_gemm with the weight operand pre-split into bf16 planes (b_planes) against the in-kernel split: error vs fp64 and time."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mintime_amd import lib as L

g = torch.Generator().manual_seed(0)
M = 12576


def timeit(fn, reps=20, warm=25):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, N, K, kw in [("qkv store", 1536, 512, {}), ("out-proj bias+res", 512, 512, {"res": True}), ("ff2 bias+res", 512, 2048, {"res": True}),
                       ("ff1 dgrad as NT", 512, 4096, {}), ("qkv dgrad as NT", 512, 1536, {}), ("4096^3", 4096, 4096, {"M": 4096})]:
    m = kw.get("M", M)
    A = torch.randn(m, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(m, N, generator=g).cuda() if kw.get("res") else None
    P = L.split_planes(W)
    assert torch.equal(P.float().sum(0), W), "planes do not sum back to the weight"
    ref = A.double().cpu() @ W.double().cpu().T + b.double().cpu() + (R.double().cpu() if R is not None else 0)
    out = {}
    for tag, planes in (("in-kernel split", None), ("pre-split planes", P)):
        C = torch.empty(m, N, device="cuda")
        def run():
            if R is not None:
                L.gemm(L.OP_NT, A, W, C, m, N, K, K, K, N, epilogue=L.EPI_BIAS_RES, bias=b, R=R, ldr=N, b_planes=planes)
            else:
                L.gemm(L.OP_NT, A, W, C, m, N, K, K, K, N, bias=b, b_planes=planes)
        run(); torch.cuda.synchronize()
        err = float((C.double().cpu() - ref).abs().max() / ref.abs().max())
        us = timeit(run)
        out[tag] = C.clone()
        print(f"{name:20s} {tag:18s} {us:8.1f} us  {2.0 * m * N * K / us / 1e6:6.1f} TF   max err vs fp64 {err:.2e}")
    print(f"{'':20s} bitwise equal: {torch.equal(out['in-kernel split'], out['pre-split planes'])}")
