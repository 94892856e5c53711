"""Per-shape timing of mt_gemm_planes on the TimeSformer's GEMMs at B = 32 (stream-K on / off) next to mt_gemm's in-kernel split."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import mintime_amd
from mintime_amd import lib as L

M = 32 * 393
D = 512
dev = "cuda"


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


def r(*s, scale=1.0):
    return torch.randn(*s, device=dev) * scale


shapes = [("QKV", L.OP_NT, M, 1536, 512, L.EPI_STORE), ("out-proj", L.OP_NT, M, 512, 512, L.EPI_BIAS_RES),
          ("FF1+GEGLU", L.OP_NT, M, 4096, 512, L.EPI_GEGLU), ("FF2", L.OP_NT, M, 512, 2048, L.EPI_BIAS_RES),
          ("FF1 dgrad", L.OP_NN, M, 512, 4096, L.EPI_STORE), ("FF2 dgrad+GEGLU'", L.OP_NN, M, 2048, 512, L.EPI_GEGLU_BWD),
          ("QKV dgrad", L.OP_NN, M, 512, 1536, L.EPI_STORE), ("out dgrad", L.OP_NN, M, 512, 512, L.EPI_STORE)]
tot = {"old": 0.0, "tile": 0.0, "sk": 0.0}
for name, op, m, n, k, epi in shapes:
    A = r(m, k)
    Bm = r(n, k, scale=0.05) if op == L.OP_NT else r(k, n, scale=0.05)
    a_p, b_p = L.split_planes_blk(A), L.split_planes_blk(Bm)
    bias = r(n if epi != L.EPI_GEGLU_BWD else 2 * n)
    kw, kwo = {}, {}
    if epi == L.EPI_GEGLU:
        out = None
        u = torch.empty(m, n, device=dev)
        hp = L.planes_empty(m, n // 2, dev)
        kw = dict(epilogue=epi, bias=bias, C2=u, ldc2=n, n_half=n // 2, c_planes=hp)
        h = torch.empty(m, n // 2, device=dev)
        old = lambda: L.gemm(op, A, Bm, h, m, n, k, k, k, n // 2, epilogue=epi, bias=bias, C2=u, ldc2=n, n_half=n // 2)
    elif epi == L.EPI_GEGLU_BWD:
        u = r(m, 2 * n)
        dup = L.planes_empty(m, 2 * n, dev)
        cs = torch.zeros(2 * n, device=dev)
        kw = dict(epilogue=epi, C2=u, ldc2=2 * n, n_half=n, col_sum=cs, c_planes=dup)
        du = torch.empty(m, 2 * n, device=dev)
        old = lambda: L.gemm(op, A, Bm, du, m, n, k, k, n, 2 * n, epilogue=epi, C2=u, ldc2=2 * n, n_half=n, col_sum=cs)
    else:
        c = torch.empty(m, n, device=dev)
        R = r(m, n)
        kw = dict(Cout=c, ldc=n, epilogue=epi, bias=bias)
        if epi == L.EPI_BIAS_RES:
            kw.update(R=R, ldr=n)
        ldb = k if op == L.OP_NT else n
        old = lambda: L.gemm(op, A, Bm, c, m, n, k, k, ldb, n, epilogue=epi, bias=bias, R=R if epi == L.EPI_BIAS_RES else None, ldr=n)
    t_old = timeit(old)
    t_tile = timeit(lambda: L.gemm_planes(op, a_p, b_p, m, n, k, streamk=False, **kw))
    t_sk = timeit(lambda: L.gemm_planes(op, a_p, b_p, m, n, k, streamk=True, **kw))
    fl = 2.0 * m * n * k
    print(f"{name:18s} in-kernel split {t_old:7.1f} us | planes, block per tile {t_tile:7.1f} us ({fl / t_tile / 1e6:6.1f} TF-eq) | stream-K {t_sk:7.1f} us "
          f"({fl / t_sk / 1e6:6.1f} TF-eq)")
    mult = 2 if name in ("QKV", "out-proj", "QKV dgrad", "out dgrad") else 1
    tot["old"] += mult * t_old; tot["tile"] += mult * t_tile; tot["sk"] += mult * t_sk
print("per layer (fwd + data gradients):", {k: round(v, 1) for k, v in tot.items()}, "us")
