"""Lab: Xception max-pool adjoint, arg-max scatter with fp32 atomics (+ the zero fill of du) against the gather form (deterministic mode's
kernel: one writer per du element).  Sizes = the three stride-2 blocks of config 5 (512 crops)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mintime_amd
from mintime_amd import lib as L

lib = L.get()
N = int(os.environ.get("N", 512))


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for H, C in ((109, 128), (55, 256), (28, 728), (14, 1024)):
    Ho = (H - 1) // 2 + 1
    z = torch.randn(N * H * H, C, device="cuda")
    dy = torch.randn(N * Ho * Ho, C, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    du = torch.empty(N * H * H, C, device="cuda")

    def scatter():
        du.zero_()
        L.check(lib.mt_maxpool_bwd(L.ptr(dy), L.ptr(z), L.ptr(sc), L.ptr(sh), L.ptr(du), N, H, H, C, L.stream_ptr()), "mp")

    L.set_deterministic(False)
    t_s = timeit(scatter)
    ref = du.clone()
    L.set_deterministic(True)
    t_g = timeit(scatter)
    same = torch.equal(ref, du)
    L.set_deterministic(False)
    print(f"H={H} C={C}: scatter+zero {t_s:.3f} ms   gather(+zero) {t_g:.3f} ms   identical {same}  max diff {float((ref-du).abs().max()):.2e}")
