"""Lab: EfficientNet-B0's late-stage 1x1 convolutions (14x14 and 7x7 stages of a 256-crop batch) -- the current launches (fp32 MFMA
LDS-DMA GEMM for the expand convs, in-kernel-split GEMMs with operand prologues for the project convs and the backward) against
"operand written once as planes + plane-operand GEMMs" (the Xception recipe, csrc/gemm_planes.hpp).  Prints us per launch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mintime_amd
from mintime_amd import lib as L

dev = "cuda"
torch.manual_seed(0)
lib = L.get()
SLOTS = 32


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def one(rows, cin, cexp, hw, tag):
    """MBConv block with `cin` block channels, `cexp` expanded channels, `rows` = N*H*W pixels."""
    N = rows // hw
    y = torch.randn(rows, cin, device=dev)
    w_e = torch.randn(cexp, cin, device=dev) * 0.1
    z_e = torch.empty(rows, cexp, device=dev)
    stats = torch.zeros(SLOTS * 2 * cexp, dtype=torch.float64, device=dev)
    # ---- expand forward
    t_cur = timeit(lambda: L.gemm(L.OP_NT, y, w_e, z_e, rows, cexp, cin, cin, cin, cexp, epilogue=L.EPI_STATS, stats=stats, stats_slots=SLOTS))
    y_p = L.planes_empty(rows, cin, dev)
    t_split_y = timeit(lambda: L.split_planes_blk(y, rows, cin, out=y_p))
    we_p = L.split_planes_blk(w_e, cexp, cin)
    z2 = torch.empty(rows, cexp, device=dev)
    t_pl = timeit(lambda: L.gemm_planes(L.OP_NT, y_p, we_p, rows, cexp, cin, Cout=z2, ldc=cexp, epilogue=L.EPI_STATS, stats=stats, stats_slots=SLOTS))
    err = float((z2 - z_e).abs().max() / z_e.abs().max())
    print(f"{tag} expand fwd  [{rows} x {cin} -> {cexp}]: current {t_cur:6.1f} us | planes GEMM {t_pl:6.1f} (+ y split {t_split_y:5.1f})  diff {err:.1e}")
    # ---- project forward: A' = swish(bn(z_d)) * gate
    z_d = torch.randn(rows, cexp, device=dev)
    sc, sh = torch.rand(cexp, device=dev) + 0.5, torch.randn(cexp, device=dev) * 0.1
    gate = torch.rand(N, cexp, device=dev)
    w_p = torch.randn(cin, cexp, device=dev) * 0.05
    z_p = torch.empty(rows, cin, device=dev)
    st2 = torch.zeros(SLOTS * 2 * cin, dtype=torch.float64, device=dev)
    t_cur = timeit(lambda: L.gemm(L.OP_NT, z_d, w_p, z_p, rows, cin, cexp, cexp, cexp, cin, prologue=L.PRO_BN_SWISH_GATE, epilogue=L.EPI_STATS,
                                  scale=sc, shift=sh, gate=gate, hw=hw, stats=st2, stats_slots=SLOTS))
    a_p = L.planes_empty(rows, cexp, dev)
    a_f = torch.empty(rows, cexp, device=dev)
    # stand-in producer pass of the same traffic: read z_d, write planes (mt_bn_bwd_apply_planes reads two tensors: an upper bound)
    t_prod = timeit(lambda: L.split_planes_blk(z_d, rows, cexp, out=a_p))
    wp_p = L.split_planes_blk(w_p, cin, cexp)
    t_pl = timeit(lambda: L.gemm_planes(L.OP_NT, a_p, wp_p, rows, cin, cexp, Cout=z_p, ldc=cin, epilogue=L.EPI_STATS, stats=st2, stats_slots=SLOTS))
    print(f"{tag} project fwd [{rows} x {cexp} -> {cin}]: current {t_cur:6.1f} us | planes GEMM {t_pl:6.1f} + producer pass {t_prod:5.1f}")
    # ---- expand backward: dz = ka du + kb z + kc; dX = dz . We (+ res); dWe = dz^T y
    du = torch.randn(rows, cexp, device=dev)
    kabc = torch.randn(3, cexp, device=dev)
    dx = torch.empty(rows, cin, device=dev)
    res = torch.randn(rows, cin, device=dev)
    dwe = torch.zeros(cexp, cin, device=dev)
    t_dg = timeit(lambda: L.gemm(L.OP_NN, du, w_e, dx, rows, cin, cexp, cexp, cin, cin, prologue=L.PRO_BN_BWD, epilogue=L.EPI_BIAS_RES, A2=z_e,
                                 scale=kabc[0], shift=kabc[1], gate=kabc[2], R=res, ldr=cin))
    t_wg = timeit(lambda: L.gemm(L.OP_TN, du, y, dwe, cexp, cin, rows, cexp, cin, cin, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0,
                                 A2=z_e, scale=kabc[0], shift=kabc[1], gate=kabc[2]))
    dz_p = L.planes_empty(rows, cexp, dev)
    t_ap = timeit(lambda: L.check(lib.mt_bn_bwd_apply_planes(L.ptr(du), L.ptr(z_e), L.ptr(kabc), L.ptr(dz_p), rows, cexp, L.stream_ptr()), "apply"))
    dx2 = torch.empty(rows, cin, device=dev)
    t_pdg = timeit(lambda: L.gemm_planes(L.OP_NN, dz_p, we_p, rows, cin, cexp, Cout=dx2, ldc=cin))
    dwe2 = torch.zeros(cexp, cin, device=dev)
    t_pwg = timeit(lambda: L.gemm_planes(L.OP_TN, dz_p, y_p, cexp, cin, rows, Cout=dwe2, ldc=cin, epilogue=L.EPI_ATOMIC))
    print(f"{tag} expand bwd: current dgrad {t_dg:6.1f} + wgrad {t_wg:6.1f} us | planes: dz pass {t_ap:5.1f} + dgrad {t_pdg:5.1f} + wgrad {t_pwg:5.1f}")
    # ---- project backward: dzp = ka dy + kb z_p + kc (narrow); da = dzp . Wp; dWp = dzp^T A'
    dy = torch.randn(rows, cin, device=dev)
    kp = torch.randn(3, cin, device=dev)
    da = torch.empty(rows, cexp, device=dev)
    dwp = torch.zeros(cin, cexp, device=dev)
    t_dg = timeit(lambda: L.gemm(L.OP_NN, dy, w_p, da, rows, cexp, cin, cin, cexp, cexp, prologue=L.PRO_BN_BWD, A2=z_p, scale=kp[0], shift=kp[1],
                                 gate=kp[2]))
    if lib.mt_conv1x1_wgrad_wide_supported(cin, cexp):
        t_wg = timeit(lambda: L.check(lib.mt_conv1x1_wgrad_wide(L.ptr(dy), L.ptr(z_p), L.ptr(kp), L.ptr(z_d), L.ptr(sc), L.ptr(sh), L.ptr(gate), hw,
                                                                L.ptr(dwp), rows, cin, cexp, L.stream_ptr()), "wide"))
    else:
        t_wg = float("nan")
    dzp_p = L.planes_empty(rows, cin, dev)
    t_ap = timeit(lambda: L.check(lib.mt_bn_bwd_apply_planes(L.ptr(dy), L.ptr(z_p), L.ptr(kp), L.ptr(dzp_p), rows, cin, L.stream_ptr()), "apply"))
    da2 = torch.empty(rows, cexp, device=dev)
    t_pdg = timeit(lambda: L.gemm_planes(L.OP_NN, dzp_p, wp_p, rows, cexp, cin, Cout=da2, ldc=cexp))
    dwp2 = torch.zeros(cin, cexp, device=dev)
    t_pwg = timeit(lambda: L.gemm_planes(L.OP_TN, dzp_p, a_p, cin, cexp, rows, Cout=dwp2, ldc=cexp, epilogue=L.EPI_ATOMIC))
    print(f"{tag} project bwd: current dgrad {t_dg:6.1f} + wgrad(wide) {t_wg:6.1f} us | planes: dzp pass {t_ap:5.1f} + dgrad {t_pdg:5.1f} + wgrad {t_pwg:5.1f}")


if __name__ == "__main__":
    one(256 * 196, 80, 480, 196, "14x14/480 ")
    one(256 * 196, 112, 672, 196, "14x14/672 ")
    one(256 * 49, 192, 1152, 49, "7x7/1152  ")
    one(256 * 784, 40, 240, 784, "28x28/240 ")
    # head: 320 -> 1280 at 7x7
    rows, cin, cout = 256 * 49, 320, 1280
    y = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev) * 0.05; z = torch.empty(rows, cout, device=dev)
    st = torch.zeros(SLOTS * 2 * cout, dtype=torch.float64, device=dev)
    t_cur = timeit(lambda: L.gemm(L.OP_NT, y, w, z, rows, cout, cin, cin, cin, cout, epilogue=L.EPI_STATS, stats=st, stats_slots=SLOTS))
    y_p, w_p = L.split_planes_blk(y, rows, cin), L.split_planes_blk(w, cout, cin)
    t_pl = timeit(lambda: L.gemm_planes(L.OP_NT, y_p, w_p, rows, cout, cin, Cout=z, ldc=cout, epilogue=L.EPI_STATS, stats=st, stats_slots=SLOTS))
    print(f"head fwd [{rows} x {cin} -> {cout}]: current {t_cur:6.1f} us | planes GEMM {t_pl:6.1f}")
