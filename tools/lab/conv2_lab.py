"""Lab: Xception conv2's three GEMMs alone at config-5 size (512 crops): forward (im2col K=288 -> 64), data gradient (im2col of dz2, pad 2,
K=576 -> 32) and weight gradient (64 x 288 over 11 M rows), each timed with events over a few repeats.  MT_CONV_WG64 / MT_FORCE_CFG are
read by the library at its first launch, so variants are separate processes (tools/lab/conv2_lab.sh)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd")
L = importlib.import_module(pkg.__name__ + ".lib")
lib = L.get()
dev = torch.device("cuda:0")
N = int(os.environ.get("LAB_N", "512")); H1, H2 = 149, 147
M1, M2 = N * H1 * H1, N * H2 * H2
g = torch.Generator(device=dev); g.manual_seed(0)
z1 = torch.randn(M1, 32, device=dev, generator=g)
dy = torch.randn(M2, 64, device=dev, generator=g) * 1e-2
z2 = torch.randn(M2, 64, device=dev, generator=g)
k2 = torch.randn(3, 64, device=dev, generator=g) * 0.1
sc1 = torch.rand(32, device=dev, generator=g) + 0.5; sh1 = torch.randn(32, device=dev, generator=g) * 0.1
wp2 = torch.randn(64, 288, device=dev, generator=g) * 0.05
wp2t = torch.randn(32, 576, device=dev, generator=g) * 0.05
RELU, NONE = 2, 0

def timed(fn, reps=int(os.environ.get("LAB_REPS", "5"))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

which = os.environ.get("LAB_WHICH", "fwd,dgrad,wgrad").split(",")
out = {}
if "fwd" in which:
    z = torch.empty(M2, 64, device=dev)
    out["fwd_ms"] = timed(lambda: L.gemm(L.OP_NT, z1, wp2, z, M2, 64, 288, 288, 288, 64, prologue=L.PRO_IM2COL, scale=sc1, shift=sh1,
                                         conv=(H1, H1, 32, H2, H2, 3, 1, 0, RELU, 0)))
if "dgrad" in which:
    da1 = torch.empty(M1, 32, device=dev)
    out["dgrad_ms"] = timed(lambda: L.gemm(L.OP_NT, z2, wp2t, da1, M1, 32, 576, 576, 576, 32, prologue=L.PRO_IM2COL,
                                           conv=(H2, H2, 64, H1, H1, 3, 1, 2, NONE)))
if "wgrad" in which:
    dw = torch.zeros(64, 288, device=dev)
    def wg():
        L.gemm(L.OP_TN, dy, z1, dw, 64, 288, M2, 64, 288, 288, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0, A2=z2,
               scale=k2[0], shift=k2[1], gate=k2[2], b_prologue=L.BPRO_IM2COL, b_scale=sc1, b_shift=sh1, conv=(H1, H1, 32, H2, H2, 3, 1, 0, RELU))
    out["wgrad_ms"] = timed(wg)
    dw.zero_(); wg(); torch.cuda.synchronize()
    out["wgrad_sum"] = float(dw.double().sum()); out["wgrad_abs"] = float(dw.double().abs().sum())
    if os.environ.get("LAB_SAVE"):
        torch.save(dw.cpu(), os.environ["LAB_SAVE"])
    if os.environ.get("LAB_CMP") and os.path.exists(os.environ["LAB_CMP"]):
        ref = torch.load(os.environ["LAB_CMP"]).double()
        out["wgrad_rel_vs_old_tile"] = float((dw.cpu().double() - ref).norm() / ref.norm())
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}, "env", {k: os.environ[k] for k in ("MT_CONV_WG64", "MT_FORCE_CFG") if k in os.environ})
