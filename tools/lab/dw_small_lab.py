"""Lab: the depthwise kernels on SMALL images (a tile is a whole image): Xception's 728-channel middle flow at 14 x 14 (512 crops, relu)
and EfficientNet-B0's 14 x 14 / 7 x 7 stages (256 crops, swish).  us per launch and GB/s of algorithmic bytes for the forward, the data
gradient (parts = 2) and the weight gradient (parts = 1).  MT_LIB selects a library variant (e.g. -DMT_DW_HOIST=1)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mintime_amd
from mintime_amd import lib as L

lib = L.get()
SLOTS = 32


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def one(tag, N, H, C, k, stride, act, res=False):
    Ho = (H + stride - 1) // stride
    Mi, Mo = N * H * H, N * Ho * Ho
    zin = torch.randn(Mi, C, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    mi = torch.randn(2, C, device="cuda")
    w = torch.randn(C, 1, k, k, device="cuda") * 0.2
    zout = torch.empty(Mo, C, device="cuda")
    st = torch.zeros(SLOTS * 2 * C, dtype=torch.float64, device="cuda")
    s = L.stream_ptr()
    t_f = timeit(lambda: L.check(lib.mt_dwconv_fwd(L.ptr(zin), L.ptr(sc), L.ptr(sh), L.ptr(w), L.ptr(zout), L.ptr(st), SLOTS, N, H, H, C, k,
                                                   stride, act, s), "fwd"))
    t_fp = t_sp = float("nan")
    if stride == 1:
        pl = L.planes_empty(Mo, C, "cuda")
        t_fp = timeit(lambda: L.check(lib.mt_dwconv_fwd_planes(L.ptr(zin), L.ptr(sc), L.ptr(sh), L.ptr(w), L.ptr(pl), N, H, H, C, k, stride, act,
                                                               s), "fwd_planes"))
        t_sp = timeit(lambda: L.split_planes_blk(zout, Mo, C, out=pl))
    du = torch.randn(Mo, C, device="cuda")
    kabc = torch.randn(3, C, device="cuda")
    du_in = torch.empty(Mi, C, device="cuda")
    dw = torch.zeros(C, 1, k, k, device="cuda")
    rp = torch.randn(Mi, C, device="cuda") if res else None

    def bwd(parts):
        L.check(lib.mt_dwconv_bwd(L.ptr(du), L.ptr(zout), L.ptr(kabc), L.ptr(w), L.ptr(zin), L.ptr(sc), L.ptr(sh), L.ptr(mi), L.ptr(du_in),
                                  L.ptr(st), SLOTS, L.ptr(dw), N, H, H, C, k, stride, parts, act, None, L.ptr(rp), s), "bwd")
    t_d = timeit(lambda: bwd(2))
    t_w = timeit(lambda: bwd(1))
    b_i, b_o = Mi * C * 4, Mo * C * 4
    gb = lambda b, t: b / t / 1e3
    print(f"{tag:28s} fwd {t_f:7.1f} us {gb(b_i + b_o, t_f):6.0f} GB/s (+ split {t_sp:6.1f}; as planes {t_fp:6.1f}) | dgrad {t_d:7.1f} us {gb(2 * b_o + (3 if res else 2) * b_i, t_d):6.0f} GB/s | "
          f"wgrad {t_w:7.1f} us {gb(2 * b_o + b_i, t_w):6.0f} GB/s")


if __name__ == "__main__":
    print("library:", L.LIB_PATH, " MT_DW_DGRAD_CAP =", os.environ.get("MT_DW_DGRAD_CAP"))
    one("xception 14x14 728 k3 relu", 512, 14, 728, 3, 1, 2)
    one("xception 14x14 728 +res", 512, 14, 728, 3, 1, 2, res=True)
    one("xception 28x28 256 k3 relu", 512, 28, 256, 3, 1, 2)
    one("xception 7x7 1536 k3", 512, 7, 1536, 3, 1, 0)
    one("effnet 14x14 480 k3 swish", 256, 14, 480, 3, 1, 1)
    one("effnet 14x14 672 k5 swish", 256, 14, 672, 5, 1, 1)
    one("effnet 14x14 672 k5 s2", 256, 14, 672, 5, 2, 1)
    one("effnet 7x7 1152 k5 swish", 256, 7, 1152, 5, 1, 1)
    one("effnet 7x7 1152 k3 swish", 256, 7, 1152, 3, 1, 1)
    one("effnet 28x28 240 k5 swish", 256, 28, 240, 5, 1, 1)
    one("effnet 56x56 144 k3 swish", 256, 56, 144, 3, 1, 1)
