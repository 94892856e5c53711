"""Accuracy probe of the two matrix pipes on data shaped like the Xception -> patch-embedding contraction."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mintime_amd import lib as L


def run(A, W, b, split):
    prev = L.set_gemm_split(split)
    M, K = A.shape
    N = W.shape[0]
    C = torch.zeros(M, N, device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), C, M, N, K, K, K, N, bias=None if b is None else b.cuda())
    torch.cuda.synchronize()
    L.set_gemm_split(prev)
    return C.cpu().double()


g = torch.Generator().manual_seed(0)
for name, M, N, K, mk in [
    ("randn", 784, 512, 2048, lambda s: torch.randn(*s, generator=g)),
    ("relu(randn)", 784, 512, 2048, lambda s: torch.relu(torch.randn(*s, generator=g))),
    ("relu heavy tail", 784, 512, 2048, lambda s: torch.relu(torch.randn(*s, generator=g)) * torch.exp(2 * torch.randn(*s, generator=g))),
    ("sparse relu x100", 784, 512, 2048, lambda s: torch.relu(torch.randn(*s, generator=g) - 1.0) * 100),
    ("M=785", 785, 512, 2048, lambda s: torch.relu(torch.randn(*s, generator=g))),
    ("K=1280", 786, 512, 1280, lambda s: torch.randn(*s, generator=g)),
    ("K=512 randn", 4096, 512, 512, lambda s: torch.randn(*s, generator=g)),
    ("K=256 relu", 4096, 256, 256, lambda s: torch.relu(torch.randn(*s, generator=g))),
    ("K=128 relu", 4096, 256, 128, lambda s: torch.relu(torch.randn(*s, generator=g))),
    ("K=128 randn", 4096, 256, 128, lambda s: torch.randn(*s, generator=g)),
    ("K=64 relu", 8192, 128, 64, lambda s: torch.relu(torch.randn(*s, generator=g))),
    ("K=16 randn", 8192, 128, 16, lambda s: torch.randn(*s, generator=g)),
]:
    A = mk((M, K)).float()
    W = (torch.randn(N, K, generator=g) * 0.02).float()
    ref = A.double() @ W.double().T
    bound = A.double().abs() @ W.double().abs().T
    for split in (True, False):
        C = run(A, W, None, split)
        e_max = float((C - ref).abs().max() / ref.abs().max())
        e_cond = float(((C - ref).abs() / bound.clamp_min(1e-30)).max())
        bias = float(((C - ref) / bound.clamp_min(1e-30)).mean())
        rms = float((((C - ref) / bound.clamp_min(1e-30)) ** 2).mean().sqrt())
        shrink = float((((C - ref) * ref.sign()) / bound.clamp_min(1e-30)).mean())      # < 0: results pulled toward zero
        print(f"{name:18s} split={int(split)}  max|err|/max|ref| {e_max:.2e}   err/sum|a||b|: max {e_cond:.2e} rms {rms:.2e} mean {bias:+.2e} toward-zero {shrink:+.2e}")
