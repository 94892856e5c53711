"""Lab: the TimeSformer's forward / data-gradient GEMM shapes (M = 12576 token rows) through mt_gemm_planes, us per launch and
fp32-equivalent TFLOP/s; run under different MT_PLANES_* knobs to compare (e.g. MT_PLANES_STAGGER=2,2)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mintime_amd
from mintime_amd import lib as L

dev = "cuda"
torch.manual_seed(0)
M = int(os.environ.get("M", 12576))


def timeit(fn, n=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


out = []
for name, op, N, K in (("qkv", L.OP_NT, 1536, 512), ("outproj", L.OP_NT, 512, 512), ("ff2", L.OP_NT, 512, 2048),
                       ("dqkv", L.OP_NN, 512, 1536), ("dff1", L.OP_NN, 512, 4096), ("dout", L.OP_NN, 512, 512), ("4096^3", L.OP_NT, 4096, 4096)):
    m = 4096 if name == "4096^3" else M
    a = L.split_planes_blk(torch.randn(m, K, device=dev))
    b = L.split_planes_blk(torch.randn(N, K, device=dev) * 0.05) if op == L.OP_NT else L.split_planes_blk(torch.randn(K, N, device=dev) * 0.05)
    c = torch.empty(m, N, device=dev)
    t = timeit(lambda: L.gemm_planes(op, a, b, m, N, K, Cout=c, ldc=N, streamk=False))
    out.append(f"{name} {t:6.1f} us {2.0 * m * N * K / t / 1e6:6.1f} TF")
print(os.environ.get("MT_PLANES_STAGGER", "-"), " | ".join(out))
