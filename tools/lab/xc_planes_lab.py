"""Lab: Xception middle-flow pointwise convolutions (728 -> 728 over 512 x 14 x 14 rows) -- the BatchNorm-backward-prologue GEMMs on the
in-kernel-split loop against "dz written once as planes + plane-operand GEMMs".  Prints ms per launch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mintime_amd
from mintime_amd import lib as L

M, C = int(os.environ.get("M", 100352)), 728
dev = "cuda"
torch.manual_seed(0)
g, z, d = (torch.randn(M, C, device=dev) for _ in range(3))
w = torch.randn(C, C, device=dev) * 0.05
kabc = torch.randn(3, C, device=dev)
lib = L.get()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dd = torch.empty(M, C, device=dev)
dw = torch.zeros(C, C, device=dev)
zz = torch.empty(M, C, device=dev)
t_fwd = timeit(lambda: L.gemm(L.OP_NT, d, w, zz, M, C, C, C, C, C))
t_dg = timeit(lambda: L.gemm(L.OP_NN, g, w, dd, M, C, C, C, C, C, prologue=L.PRO_BN_BWD, A2=z, scale=kabc[0], shift=kabc[1], gate=kabc[2]))
t_wg = timeit(lambda: L.gemm(L.OP_TN, g, d, dw, C, C, M, C, C, C, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0, A2=z,
                             scale=kabc[0], shift=kabc[1], gate=kabc[2]))
ref_dd = dd.clone()
print(f"current: fwd NT {t_fwd:.3f}  dgrad NN+prologue {t_dg:.3f}  wgrad TN+prologue {t_wg:.3f} ms  (2MNK = {2*M*C*C/1e9:.1f} GF)")

dz = torch.empty(M, C, device=dev)
t_apply = timeit(lambda: L.check(lib.mt_bn_bwd_apply(L.ptr(g), L.ptr(z), L.ptr(kabc), L.ptr(dz), M, C, L.stream_ptr()), "apply"))
dz_p = L.planes_empty(M, C, dev)
t_split = timeit(lambda: L.split_planes_blk(dz, M, C, out=dz_p))
d_p = L.split_planes_blk(d, M, C)
w_p = L.split_planes_blk(w, C, C)
dd2 = torch.empty(M, C, device=dev)
t_pfwd = timeit(lambda: L.gemm_planes(L.OP_NT, d_p, w_p, M, C, C, Cout=zz, ldc=C))
t_pdg = timeit(lambda: L.gemm_planes(L.OP_NN, dz_p, w_p, M, C, C, Cout=dd2, ldc=C))
dw2 = torch.zeros(C, C, device=dev)
t_pwg = timeit(lambda: L.gemm_planes(L.OP_TN, dz_p, d_p, C, C, M, Cout=dw2, ldc=C, epilogue=L.EPI_ATOMIC))
print(f"planes:  apply {t_apply:.3f} + split {t_split:.3f}; fwd NT {t_pfwd:.3f}  dgrad NN {t_pdg:.3f}  wgrad TN {t_pwg:.3f} ms")
print("dgrad max rel diff", float((dd2 - ref_dd).abs().max() / ref_dd.abs().max()))
# both backward GEMMs concurrently (main + side stream), as in the step
side = torch.cuda.Stream()
def both_cur():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        L.gemm(L.OP_TN, g, d, dw, C, C, M, C, C, C, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0, A2=z, scale=kabc[0], shift=kabc[1], gate=kabc[2])
    L.gemm(L.OP_NN, g, w, dd, M, C, C, C, C, C, prologue=L.PRO_BN_BWD, A2=z, scale=kabc[0], shift=kabc[1], gate=kabc[2])
    torch.cuda.current_stream().wait_stream(side)
def both_pl():
    L.check(lib.mt_bn_bwd_apply(L.ptr(g), L.ptr(z), L.ptr(kabc), L.ptr(dz), M, C, L.stream_ptr()), "apply")
    L.split_planes_blk(dz, M, C, out=dz_p)
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        L.gemm_planes(L.OP_TN, dz_p, d_p, C, C, M, Cout=dw2, ldc=C, epilogue=L.EPI_ATOMIC)
    L.gemm_planes(L.OP_NN, dz_p, w_p, M, C, C, Cout=dd2, ldc=C)
    torch.cuda.current_stream().wait_stream(side)
print(f"backward pair, two streams: current {timeit(both_cur):.3f} ms   planes (incl. producer passes) {timeit(both_pl):.3f} ms")
