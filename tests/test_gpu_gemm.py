"""GPU parity of the GEMM family (through the C ABI) against fp64 CPU matmuls.  Every test runs on both matrix pipes:
the split-operand fp32 loop (bf16 pipe, default) and the fp32 MFMA kernels (mt_gemm_set_split)."""
import numpy as np
import pytest
import torch

import mintime_amd
from mintime_amd import lib as L
from tests.util import assert_close

pytestmark = pytest.mark.gpu
TOL = 2e-5   # fp32 products and accumulation vs fp64: rounding only (same bound for both pipes)


@pytest.fixture(autouse=True, params=["split", "fp32"])
def matrix_pipe(request):
    prev = L.set_gemm_split(request.param == "split")
    yield request.param
    L.set_gemm_split(prev)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("M,N,K", [(300, 512, 512), (1000, 1536, 512), (257, 96, 16), (4000, 16, 32), (777, 40, 144),
                                   (513, 320, 1152), (130, 1280, 320), (64, 24, 96)])
def test_nt_store_bias(M, N, K):
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    Cd = torch.full((M, N), float("nan"), device="cuda")
    L.gemm(L.OP_NT, Ad, Wd, Cd, M, N, K, K, K, N, bias=bd)
    ref = A.double() @ W.double().T + b.double()
    assert_close(Cd, ref, TOL, f"NT {M}x{N}x{K}")


def test_nt_bias_res_inplace_and_rowmap():
    M, N, K = 2 * 392, 512, 1280
    A, W, b = _rand(M, K, seed=1, scale=0.3), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3)
    X = torch.zeros(2, 393, N)
    Xd = X.cuda()
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Xd, M, N, K, K, K, N, bias=b.cuda(), c_map=(392, 393, 1))
    ref = (A.double() @ W.double().T + b.double()).reshape(2, 392, N)
    assert_close(Xd[:, 1:], ref, TOL, "row-mapped store")
    assert float(Xd[:, 0].abs().max()) == 0.0
    # residual, in place
    R = _rand(M, N, seed=5)
    Rd = R.cuda()
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Rd, M, N, K, K, K, N, epilogue=L.EPI_BIAS_RES, bias=b.cuda(), R=Rd, ldr=N)
    assert_close(Rd, A.double() @ W.double().T + b.double() + R.double(), TOL, "bias+residual in place")


@pytest.mark.parametrize("M", [393 * 2, 1000])
def test_nt_geglu(M):
    D = 512
    A, W, b = _rand(M, D, seed=1), _rand(8 * D, D, seed=2, scale=0.05), _rand(8 * D, seed=3, scale=0.1)
    h = torch.full((M, 4 * D), float("nan"), device="cuda")
    u = torch.full((M, 8 * D), float("nan"), device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), h, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b.cuda(), C2=u, ldc2=8 * D,
           n_half=4 * D)
    uref = A.double() @ W.double().T + b.double()
    a, g = uref.chunk(2, dim=-1)
    assert_close(u, torch.stack((a, g), dim=-1).reshape(M, 8 * D), TOL, "GEGLU pre-activations (stored interleaved a_j, g_j)")
    assert_close(h, a * torch.nn.functional.gelu(g), TOL, "GEGLU output")


def test_nt_stats_and_gate_prologue():
    n_img, hw, K, N = 3, 49, 96, 24
    M = n_img * hw
    Z, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.2)
    sc, sh, gate = _rand(K, seed=3).abs() + 0.5, _rand(K, seed=4, scale=0.2), torch.sigmoid(_rand(n_img, K, seed=5))
    slots = 4
    stats = torch.zeros(slots, 2, N, dtype=torch.float64, device="cuda")
    Cd = torch.empty(M, N, device="cuda")
    L.gemm(L.OP_NT, Z.cuda(), W.cuda(), Cd, M, N, K, K, K, N, prologue=L.PRO_BN_SWISH_GATE, epilogue=L.EPI_STATS,
           scale=sc.cuda(), shift=sh.cuda(), gate=gate.cuda(), hw=hw, stats=stats, stats_slots=slots)
    a = Z.double() * sc.double() + sh.double()
    a = a * torch.sigmoid(a) * gate.double().repeat_interleave(hw, 0)
    ref = a @ W.double().T
    assert_close(Cd, ref, 5e-5, "gated project conv")
    st = stats.sum(0).cpu()
    assert_close(st[0], ref.sum(0), 5e-5, "column sums")
    assert_close(st[1], (ref * ref).sum(0), 5e-5, "column sums of squares")


@pytest.mark.parametrize("M,N,K", [(500, 512, 2048), (786, 1280, 512), (300, 16, 96)])
def test_nn_dgrad(M, N, K):
    dY, W = _rand(M, K, seed=1), _rand(K, N, seed=2, scale=0.1)
    Cd = torch.empty(M, N, device="cuda")
    L.gemm(L.OP_NN, dY.cuda(), W.cuda(), Cd, M, N, K, K, N, N)
    assert_close(Cd, dY.double() @ W.double(), TOL, "NN")
    acc = _rand(M, N, seed=3)
    accd = acc.cuda()
    L.gemm(L.OP_NN, dY.cuda(), W.cuda(), accd, M, N, K, K, N, N, epilogue=L.EPI_ACCUM)
    assert_close(accd, acc.double() + dY.double() @ W.double(), TOL, "NN accumulate")


def test_nn_geglu_bwd():
    M, D = 700, 512
    dy, W2 = _rand(M, D, seed=1), _rand(D, 4 * D, seed=2, scale=0.05)
    u = _rand(M, 8 * D, seed=3)
    du = torch.empty(M, 8 * D, device="cuda")
    u_il = torch.stack(u.chunk(2, dim=-1), dim=-1).reshape(M, 8 * D).contiguous()      # the layout EPI_GEGLU stores: (a_j, g_j) pairs
    L.gemm(L.OP_NN, dy.cuda(), W2.cuda(), du, M, 4 * D, D, D, 4 * D, 8 * D, epilogue=L.EPI_GEGLU_BWD, C2=u_il.cuda(), ldc2=8 * D,
           n_half=4 * D)
    ud = u.double().requires_grad_(True)
    a, g = ud.chunk(2, dim=-1)
    (a * torch.nn.functional.gelu(g) * (dy.double() @ W2.double())).sum().backward()
    assert_close(du, ud.grad, 5e-5, "GEGLU backward")


@pytest.mark.parametrize("Mrows,N1,N2,split", [(2 * 393, 512, 1536, 1), (5000, 96, 16, 8), (1234, 2048, 512, 4)])
def test_tn_wgrad(Mrows, N1, N2, split):
    dY, X = _rand(Mrows, N1, seed=1), _rand(Mrows, N2, seed=2)
    dW = torch.zeros(N1, N2, device="cuda")
    L.gemm(L.OP_TN, dY.cuda(), X.cuda(), dW, N1, N2, Mrows, N1, N2, N2, epilogue=L.EPI_ATOMIC, split_k=split)
    assert_close(dW, dY.double().T @ X.double(), 5e-5, "TN")


@pytest.mark.parametrize("op", ["NT", "NN"])
def test_split_k_atomic_onto_initialised_output(op):
    """The skinny-GEMM path of the engines: K-slices accumulated with fp32 atomics onto a residual (bias on slice 0)."""
    M, N, K = 1500, 512, 2048
    A, b, R = _rand(M, K, seed=1), _rand(N, seed=3), _rand(M, N, seed=4)
    if op == "NT":
        W = _rand(N, K, seed=2, scale=0.05)
        Cd = R.cuda().clone()
        L.gemm(L.OP_NT, A.cuda(), W.cuda(), Cd, M, N, K, K, K, N, epilogue=L.EPI_ATOMIC, bias=b.cuda(), split_k=5)
        ref = R.double() + b.double() + A.double() @ W.double().T
    else:
        W = _rand(K, N, seed=2, scale=0.05)
        Cd = torch.zeros(M, N, device="cuda")
        L.gemm(L.OP_NN, A.cuda(), W.cuda(), Cd, M, N, K, K, N, N, epilogue=L.EPI_ATOMIC, split_k=3)
        ref = A.double() @ W.double()
    assert_close(Cd, ref, TOL, "split-K " + op)


def test_im2col_conv_matches_torch_conv():
    """Dense convolution as a GEMM with the im2col prologue (Xception conv1/conv2, EfficientNet stem, strided 1x1 skip)."""
    import torch.nn.functional as F
    lib = L.get()
    for (Cin, Cout, k, s, p, H, relu) in ((3, 32, 3, 2, 0, 37, False), (32, 64, 3, 1, 0, 21, True), (64, 128, 1, 2, 0, 19, False),
                                          (64, 32, 3, 1, 2, 11, False), (8, 24, 5, 2, 2, 13, True), (16, 32, 3, 2, 1, 10, True)):
        n = 2
        x = _rand(n, H, H, Cin, seed=1)
        w = _rand(Cout, Cin, k, k, seed=2, scale=0.2)
        sc, sh = _rand(Cin, seed=3).abs() + 0.5, _rand(Cin, seed=4, scale=0.3)
        Ho = (H + 2 * p - k) // s + 1
        K = (k * k * Cin + 3) // 4 * 4
        wp = torch.empty(Cout, K, device="cuda")
        L.check(lib.mt_conv_weight_pack(L.ptr(w.cuda()), L.ptr(wp), Cout, Cin, k, K, 0, L.stream_ptr()), "pack")
        out = torch.empty(n * Ho * Ho, Cout, device="cuda")
        L.gemm(L.OP_NT, x.cuda(), wp, out, n * Ho * Ho, Cout, K, K, K, Cout, prologue=L.PRO_IM2COL, scale=sc.cuda(), shift=sh.cuda(),
               conv=(H, H, Cin, Ho, Ho, k, s, p, 2 if relu else 0))
        a = x.double() * sc.double() + sh.double()
        if relu:
            a = a.clamp_min(0)
        ref = F.conv2d(a.permute(0, 3, 1, 2), w.double(), None, s, p).permute(0, 2, 3, 1).reshape(n * Ho * Ho, Cout)
        assert_close(out, ref, 5e-5, f"im2col conv {Cin}->{Cout} k{k} s{s} p{p}")


def test_im2col_weight_gradient_matches_torch_conv():
    """Weight gradient of a dense convolution as a TN GEMM whose B operand is gathered (BPRO_IM2COL) and whose A operand carries the
    BatchNorm-backward prologue: dW[co][(kh,kw,ci)] = sum_pixels (ka*dy + kb*z + kc)[pix][co] * act(x*sc+sh)[window(pix)][kh,kw,ci].
    Cases: Xception conv2's 64 x 288 (the one-tile-per-K-range configuration, gemm.hip CFG_WG64) over image rows that wrap inside a
    k-step, a padded 3x3 (window tests on the B side), a strided 1x1, and the 3-channel per-element gather."""
    import torch.nn.functional as F
    for (Cin, Cout, k, s, p, H, relu) in ((32, 64, 3, 1, 0, 23, True), (32, 48, 3, 1, 1, 9, False), (64, 128, 1, 2, 0, 19, True),
                                          (3, 32, 3, 2, 0, 21, False), (8, 64, 5, 2, 2, 13, True)):
        n = 3
        x = _rand(n, H, H, Cin, seed=1)
        Ho = (H + 2 * p - k) // s + 1
        M = n * Ho * Ho
        dy, z = _rand(M, Cout, seed=2), _rand(M, Cout, seed=3)
        ka, kb, kc = _rand(Cout, seed=4).abs() + 0.5, _rand(Cout, seed=5, scale=0.2), _rand(Cout, seed=6, scale=0.1)
        sc, sh = _rand(Cin, seed=7).abs() + 0.5, _rand(Cin, seed=8, scale=0.3)
        K = (k * k * Cin + 3) // 4 * 4
        dw = torch.zeros(Cout, K, device="cuda")
        L.gemm(L.OP_TN, dy.cuda(), x.cuda(), dw, Cout, K, M, Cout, K, K, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0,
               A2=z.cuda(), scale=ka.cuda(), shift=kb.cuda(), gate=kc.cuda(), b_prologue=L.BPRO_IM2COL, b_scale=sc.cuda(),
               b_shift=sh.cuda(), conv=(H, H, Cin, Ho, Ho, k, s, p, 2 if relu else 0))
        a = x.double() * sc.double() + sh.double()
        if relu:
            a = a.clamp_min(0)
        g = (ka.double() * dy.double() + kb.double() * z.double() + kc.double()).reshape(n, Ho, Ho, Cout).permute(0, 3, 1, 2)
        a_nchw = a.permute(0, 3, 1, 2).requires_grad_(False)
        w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
        F.conv2d(a_nchw, w, None, s, p).backward(g)
        ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin)          # columns in (kh, kw, ci) order
        assert_close(dw[:, :k * k * Cin], ref, 5e-5, f"im2col wgrad {Cin}->{Cout} k{k} s{s} p{p}")
        assert float(dw[:, k * k * Cin:].abs().sum()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# The shapes bench.py actually times (config 3: B = 32 -> M = 32*393 = 12576 token rows).  At this size the host and the
# library take different branches than at fixture size: 64x64 tiles for M >= 4096 (gemm.hip pick_cfg), the L2-blocked tile
# order with its padding-block early return (m_tiles >= 32 && n_tiles >= 2), split-K + fp32 atomics for FF2 and the skinny
# N = 512 data gradients, auto split-K for the weight gradients.
# ---------------------------------------------------------------------------------------------------------------------
M_FULL = 32 * 393
ATOL_SPLITK = 5e-5


def _dmm(a, b):
    return a.double() @ b.double()


def test_full_size_ff1_geglu_l2_blocked():
    D = 512
    A, W, b = _rand(M_FULL, D, seed=1), _rand(8 * D, D, seed=2, scale=0.05), _rand(8 * D, seed=3, scale=0.1)
    h = torch.full((M_FULL, 4 * D), float("nan"), device="cuda")
    u = torch.full((M_FULL, 8 * D), float("nan"), device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), h, M_FULL, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b.cuda(), C2=u, ldc2=8 * D,
           n_half=4 * D)
    uref = _dmm(A, W.T) + b.double()
    a, g = uref.chunk(2, dim=-1)
    assert_close(u, torch.stack((a, g), dim=-1).reshape(M_FULL, 8 * D), TOL, "FF1 pre-activations at M=12576")
    assert_close(h, a * torch.nn.functional.gelu(g), TOL, "FF1 + GEGLU at M=12576")
    assert not bool(torch.isnan(h).any()) and not bool(torch.isnan(u).any())       # every tile was written (no skipped block)


def test_full_size_geglu_backward_dgrad():
    D = 512
    dy, W2 = _rand(M_FULL, D, seed=1), _rand(D, 4 * D, seed=2, scale=0.05)
    u = _rand(M_FULL, 8 * D, seed=3)
    du = torch.full((M_FULL, 8 * D), float("nan"), device="cuda")
    u_il = torch.stack(u.chunk(2, dim=-1), dim=-1).reshape(M_FULL, 8 * D).contiguous()
    db1 = torch.zeros(8 * D, device="cuda")
    L.gemm(L.OP_NN, dy.cuda(), W2.cuda(), du, M_FULL, 4 * D, D, D, 4 * D, 8 * D, epilogue=L.EPI_GEGLU_BWD, C2=u_il.cuda(),
           ldc2=8 * D, n_half=4 * D, col_sum=db1)
    ud = u.double().requires_grad_(True)
    a, g = ud.chunk(2, dim=-1)
    (a * torch.nn.functional.gelu(g) * _dmm(dy, W2)).sum().backward()
    assert_close(du, ud.grad, 5e-5, "GEGLU backward at M=12576")
    assert_close(db1, ud.grad.sum(0), 2e-4, "net.0.bias gradient from the epilogue (column sums of du)")


@pytest.mark.parametrize("N,K", [(1536, 512), (512, 512), (512, 1280)])
def test_full_size_nt_store_and_bias_residual(N, K):
    A, W, b, R = _rand(M_FULL, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3), _rand(M_FULL, N, seed=4)
    Cd = torch.full((M_FULL, N), float("nan"), device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Cd, M_FULL, N, K, K, K, N)
    ref = _dmm(A, W.T)
    assert_close(Cd, ref, TOL, f"NT store {N}x{K} at M=12576")
    Rd = R.cuda()
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Rd, M_FULL, N, K, K, K, N, epilogue=L.EPI_BIAS_RES, bias=b.cuda(), R=Rd, ldr=N)
    assert_close(Rd, ref + b.double() + R.double(), TOL, f"NT bias+residual in place {N}x{K} at M=12576")


def test_full_size_patch_embedding_rowmap():
    B, Fn, N, K = 32, 392, 512, 1280
    A, W, b = _rand(B * Fn, K, seed=1, scale=0.3), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3)
    Xd = torch.zeros(B, Fn + 1, N, device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Xd, B * Fn, N, K, K, K, N, bias=b.cuda(), c_map=(Fn, Fn + 1, 1))
    assert_close(Xd[:, 1:], (_dmm(A, W.T) + b.double()).reshape(B, Fn, N), TOL, "patch embedding at B=32")
    assert float(Xd[:, 0].abs().max()) == 0.0


def test_full_size_ff2_split_k_atomic():
    """tsf_engine.py: FF2 for M >= 4096 = 5 K-slices, fp32 atomics onto the residual, bias on slice 0."""
    N, K = 512, 2048
    A, W, b, R = _rand(M_FULL, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3), _rand(M_FULL, N, seed=4)
    Cd = R.cuda().clone()
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Cd, M_FULL, N, K, K, K, N, epilogue=L.EPI_ATOMIC, bias=b.cuda(), split_k=5)
    assert_close(Cd, R.double() + b.double() + _dmm(A, W.T), ATOL_SPLITK, "FF2 split-K at M=12576")


@pytest.mark.parametrize("K,split", [(4096, 4), (1536, 3), (512, 1)])
def test_full_size_skinny_dgrad(K, split):
    """tsf_backward.dgrad_skinny: dxn[M,512] = dY[M,K] . W[K,512]; K-slices + atomics onto zeros for K >= 1024."""
    N = 512
    dY, W = _rand(M_FULL, K, seed=1), _rand(K, N, seed=2, scale=0.05)
    if split > 1:
        Cd = torch.zeros(M_FULL, N, device="cuda")
        L.gemm(L.OP_NN, dY.cuda(), W.cuda(), Cd, M_FULL, N, K, K, N, N, epilogue=L.EPI_ATOMIC, split_k=split)
    else:
        Cd = torch.full((M_FULL, N), float("nan"), device="cuda")
        L.gemm(L.OP_NN, dY.cuda(), W.cuda(), Cd, M_FULL, N, K, K, N, N)
    assert_close(Cd, _dmm(dY, W), ATOL_SPLITK, f"skinny dgrad K={K} at M=12576")


@pytest.mark.parametrize("N1,N2", [(512, 2048), (4096, 512), (1536, 512), (512, 512)])
def test_full_size_wgrad_auto_split(N1, N2):
    """tsf_backward.wgrad: dW[N1,N2] = dY[M,N1]^T . X[M,N2] with split_k=0 (auto) + atomics onto zero-filled grads."""
    dY, X = _rand(M_FULL, N1, seed=1), _rand(M_FULL, N2, seed=2)
    dW = torch.zeros(N1, N2, device="cuda")
    L.gemm(L.OP_TN, dY.cuda(), X.cuda(), dW, N1, N2, M_FULL, N1, N2, N2, epilogue=L.EPI_ATOMIC, split_k=0)
    assert_close(dW, _dmm(dY.T, X), 5e-5, f"wgrad {N1}x{N2} over 12576 rows")


def test_full_size_wgrad_with_token_rowmap():
    """Patch-embedding weight gradient: A rows skip the cls slot (a_map), K = 32*392 rows."""
    B, Fn, D, Cin = 32, 392, 512, 1280
    dx = _rand(B, Fn + 1, D, seed=1)
    feat = _rand(B * Fn, Cin, seed=2)
    dW = torch.zeros(D, Cin, device="cuda")
    L.gemm(L.OP_TN, dx.cuda(), feat.cuda(), dW, D, Cin, B * Fn, D, Cin, Cin, epilogue=L.EPI_ATOMIC, split_k=0, a_map=(Fn, Fn + 1, 1))
    assert_close(dW, _dmm(dx[:, 1:].reshape(B * Fn, D).T, feat), 5e-5, "patch-embedding wgrad at B=32")


# ---------------------------------------------------------------------------------------------------------------------
# LDS-DMA main loop (csrc/gemm_dma.hpp): every variant (tile x BK x ring depth), every layout, ragged M / N tails, split-K,
# row maps -- forced through MT_DMA_VARIANT so that the shapes below do not depend on the dispatch heuristics.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def dma_variant(monkeypatch, matrix_pipe):
    if matrix_pipe == "split":
        pytest.skip("the LDS-DMA loop is the fp32 pipe's")
    def force(v):
        monkeypatch.setenv("MT_DMA_VARIANT", str(v))
    return force


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_dma_variants_all_layouts(variant, dma_variant):
    dma_variant(variant)
    # NT: ragged M (not a tile multiple), N with a partial tile, bias
    M, N, K = 1000 + 37, 320, 288 if variant in (1, 2, 4) else 320
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(N, seed=3)
    Cd = torch.full((M, N), float("nan"), device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Cd, M, N, K, K, K, N, bias=b.cuda())
    assert_close(Cd, _dmm(A, W.T) + b.double(), TOL, f"DMA v{variant} NT")
    # NN: k-major B with a ragged N tail (N % 4 == 0 only)
    M, N, K = 777, 196, 256
    dY, W = _rand(M, K, seed=4), _rand(K, N, seed=5, scale=0.1)
    Cd = torch.full((M, N), float("nan"), device="cuda")
    L.gemm(L.OP_NN, dY.cuda(), W.cuda(), Cd, M, N, K, K, N, N)
    assert_close(Cd, _dmm(dY, W), TOL, f"DMA v{variant} NN")
    # TN with split-K + atomics, ragged M and N, K a multiple of 32 but not of the chunk
    Kr, N1, N2 = 4000 * 0 + 4128, 200, 332
    dY, X = _rand(Kr, N1, seed=6), _rand(Kr, N2, seed=7)
    dW = torch.zeros(N1, N2, device="cuda")
    L.gemm(L.OP_TN, dY.cuda(), X.cuda(), dW, N1, N2, Kr, N1, N2, N2, epilogue=L.EPI_ATOMIC, split_k=5)
    assert_close(dW, _dmm(dY.T, X), 5e-5, f"DMA v{variant} TN split-K")
    # NT split-K onto a residual (FF2 form) and the row-mapped store (patch-embedding form)
    M, N, K = 600, 128, 640
    A, W, b, R = _rand(M, K, seed=8), _rand(N, K, seed=9, scale=0.05), _rand(N, seed=10), _rand(M, N, seed=11)
    Cd = R.cuda().clone()
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Cd, M, N, K, K, K, N, epilogue=L.EPI_ATOMIC, bias=b.cuda(), split_k=3)
    assert_close(Cd, R.double() + b.double() + _dmm(A, W.T), ATOL_SPLITK, f"DMA v{variant} NT split-K")
    Xd = torch.zeros(3, 201, N, device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Xd, M, N, K, K, K, N, bias=b.cuda(), c_map=(200, 201, 1))
    assert_close(Xd[:, 1:], (_dmm(A, W.T) + b.double()).reshape(3, 200, N), TOL, f"DMA v{variant} row-mapped store")
    assert float(Xd[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("variant", [0, 1])
def test_dma_geglu_pair(variant, dma_variant):
    dma_variant(variant)
    M, D = 650, 256
    A, W, b = _rand(M, D, seed=1), _rand(8 * D, D, seed=2, scale=0.05), _rand(8 * D, seed=3, scale=0.1)
    h = torch.full((M, 4 * D), float("nan"), device="cuda")
    u = torch.full((M, 8 * D), float("nan"), device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), h, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b.cuda(), C2=u, ldc2=8 * D, n_half=4 * D)
    uref = _dmm(A, W.T) + b.double()
    a, g = uref.chunk(2, dim=-1)
    assert_close(u, torch.stack((a, g), dim=-1).reshape(M, 8 * D), TOL, "DMA GEGLU pre-activations")
    assert_close(h, a * torch.nn.functional.gelu(g), TOL, "DMA GEGLU output")


def test_dma_wgrad_with_token_rowmap_small(dma_variant):
    """k-major A with a row map on k (patch-embedding weight gradient form), every small/mid variant."""
    for v in (2, 3, 4):
        dma_variant(v)
        B, Fn, D, Cin = 3, 64, 128, 192
        dx = _rand(B, Fn + 1, D, seed=1)
        feat = _rand(B * Fn, Cin, seed=2)
        dW = torch.zeros(D, Cin, device="cuda")
        L.gemm(L.OP_TN, dx.cuda(), feat.cuda(), dW, D, Cin, B * Fn, D, Cin, Cin, epilogue=L.EPI_ATOMIC, split_k=2, a_map=(Fn, Fn + 1, 1))
        assert_close(dW, _dmm(dx[:, 1:].reshape(B * Fn, D).T, feat), 5e-5, f"DMA v{v} row-mapped wgrad")


# ---------------------------------------------------------------------------------------------------------------------
# Operand prologues on the LDS-DMA pipeline (transform applied when the fragment is read out of LDS): the extractor's project
# convs (BN + swish + per-image gate) and 1x1-conv data gradients (BatchNorm backward folded in) at M >= 4096.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [2, 3, 4])
@pytest.mark.parametrize("M,hw,K,N", [(5000 + 39, 49, 96, 80), (12544, 49, 1152, 192), (4096 + 196 * 3, 196, 480, 112)])
def test_dma_gate_prologue_with_stats(variant, M, hw, K, N, monkeypatch):
    monkeypatch.setenv("MT_DMA_PRO_VARIANT", str(variant))
    monkeypatch.setenv("MT_DMA_PRO_GATE", "1")          # opt-in form (the register-staged kernel is the default for this prologue)
    n_img = (M + hw - 1) // hw
    Z, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.2)
    sc, sh, gate = _rand(K, seed=3).abs() + 0.5, _rand(K, seed=4, scale=0.2), torch.sigmoid(_rand(n_img, K, seed=5))
    slots = 8
    stats = torch.zeros(slots, 2, N, dtype=torch.float64, device="cuda")
    Cd = torch.full((M, N), float("nan"), device="cuda")
    L.gemm(L.OP_NT, Z.cuda(), W.cuda(), Cd, M, N, K, K, K, N, prologue=L.PRO_BN_SWISH_GATE, epilogue=L.EPI_STATS,
           scale=sc.cuda(), shift=sh.cuda(), gate=gate.cuda(), hw=hw, stats=stats, stats_slots=slots)
    a = Z.double() * sc.double() + sh.double()
    a = a * torch.sigmoid(a) * gate.double().repeat_interleave(hw, 0)[:M]
    ref = a @ W.double().T
    assert_close(Cd, ref, 5e-5, f"gated project conv (DMA v{variant})")
    st = stats.sum(0).cpu()
    assert_close(st[0], ref.sum(0), 1e-4, "column sums")
    assert_close(st[1], (ref * ref).sum(0), 1e-4, "column sums of squares")


@pytest.mark.parametrize("variant", [2, 3, 4])
@pytest.mark.parametrize("M,K,N,with_res", [(5000 + 39, 80, 480, False), (12544, 1152, 192, True), (50176, 480, 80, True)])
def test_dma_bn_backward_prologue_dgrad(variant, M, K, N, with_res, monkeypatch):
    monkeypatch.setenv("MT_DMA_PRO_VARIANT", str(variant))
    du, z, W = _rand(M, K, seed=1), _rand(M, K, seed=2), _rand(K, N, seed=3, scale=0.1)
    ka, kb, kc = _rand(K, seed=4), _rand(K, seed=5, scale=0.3), _rand(K, seed=6, scale=0.1)
    res = _rand(M, N, seed=7) if with_res else None
    Cd = torch.full((M, N), float("nan"), device="cuda")
    kw = dict(epilogue=L.EPI_BIAS_RES, R=res.cuda(), ldr=N) if with_res else {}
    L.gemm(L.OP_NN, du.cuda(), W.cuda(), Cd, M, N, K, K, N, N, prologue=L.PRO_BN_BWD, A2=z.cuda(), scale=ka.cuda(), shift=kb.cuda(),
           gate=kc.cuda(), **kw)
    dz = ka.double() * du.double() + kb.double() * z.double() + kc.double()
    ref = dz @ W.double()
    if with_res:
        ref = ref + res.double()
    assert_close(Cd, ref, 5e-5, f"BN-backward data gradient (DMA v{variant})")


@pytest.mark.parametrize("rows,cout,cin", [(12544, 1152, 192), (50176 + 24, 480, 80), (12544, 728, 728), (9000, 672, 112)])
def test_bn_backward_prologue_wgrad_both_pipes(rows, cout, cin, matrix_pipe):
    """dW[co,ci] = sum_r (ka*du + kb*z + kc)[r,co] * x[r,ci]: the expand-conv / pointwise-conv weight gradient with BatchNorm
    backward folded into the operand load (TN + PRO_BN_BWD), on the fp32 MFMA kernels and -- for long contractions -- on the
    split-operand loop (K-range-major split-K, fp32 atomics)."""
    du, z, x = _rand(rows, cout, seed=1), _rand(rows, cout, seed=2), _rand(rows, cin, seed=3)
    ka, kb, kc = _rand(cout, seed=4), _rand(cout, seed=5, scale=0.3), _rand(cout, seed=6, scale=0.1)
    dW = torch.zeros(cout, cin, device="cuda")
    L.gemm(L.OP_TN, du.cuda(), x.cuda(), dW, cout, cin, rows, cout, cin, cin, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0,
           A2=z.cuda(), scale=ka.cuda(), shift=kb.cuda(), gate=kc.cuda())
    dz = ka.double() * du.double() + kb.double() * z.double() + kc.double()
    assert_close(dW, dz.T @ x.double(), 5e-5, f"BN-backward weight gradient ({matrix_pipe} pipe)")


@pytest.mark.parametrize("M,K,N,with_res", [(12544, 728, 728, False), (3000 + 17, 1288, 320, True), (12544, 672, 112, True)])
def test_bn_backward_prologue_dgrad_both_pipes_k_tail(M, K, N, with_res, matrix_pipe):
    """NN + PRO_BN_BWD with K % 16 == 8 (Xception's 728 channels: the half-filled last k-tile must stay zero THROUGH the prologue,
    whose constant term kc would otherwise leak into it) and K % 16 == 0, with and without the residual epilogue."""
    du, z, W = _rand(M, K, seed=1), _rand(M, K, seed=2), _rand(K, N, seed=3, scale=0.1)
    ka, kb, kc = _rand(K, seed=4), _rand(K, seed=5, scale=0.3), _rand(K, seed=6, scale=0.1)
    res = _rand(M, N, seed=7) if with_res else None
    Cd = torch.full((M, N), float("nan"), device="cuda")
    kw = dict(epilogue=L.EPI_BIAS_RES, R=res.cuda(), ldr=N) if with_res else {}
    L.gemm(L.OP_NN, du.cuda(), W.cuda(), Cd, M, N, K, K, N, N, prologue=L.PRO_BN_BWD, A2=z.cuda(), scale=ka.cuda(), shift=kb.cuda(),
           gate=kc.cuda(), **kw)
    dz = ka.double() * du.double() + kb.double() * z.double() + kc.double()
    ref = dz @ W.double() + (res.double() if with_res else 0)
    assert_close(Cd, ref, 5e-5, f"BN-backward data gradient ({matrix_pipe} pipe)")


# ---------------------------------------------------------------------------------------------------------------------
# Split-operand loop (csrc/gemm_split.hpp): fp32 in / fp32 accumulate with the products on the bf16 pipe.  Its claim is
# "the same error against fp64 as the fp32 MFMA pipe", checked here on well-scaled, wide-dynamic-range and cancelling data.
# ---------------------------------------------------------------------------------------------------------------------
def _both_pipes(op, A, B, M, N, K, split_k=1):
    """(C_split, C_fp32) of the same problem through mt_gemm."""
    out = []
    for on in (True, False):
        prev = L.set_gemm_split(on)
        try:
            C = torch.zeros(M, N, device="cuda")
            if op == "NT":
                L.gemm(L.OP_NT, A, B, C, M, N, K, K, K, N)
            elif op == "NN":
                L.gemm(L.OP_NN, A, B, C, M, N, K, K, N, N, epilogue=L.EPI_ATOMIC, split_k=split_k)
            else:
                L.gemm(L.OP_TN, A, B, C, M, N, K, M, N, N, epilogue=L.EPI_ATOMIC, split_k=split_k)
            torch.cuda.synchronize()
            out.append(C.cpu().double())
        finally:
            L.set_gemm_split(prev)
    return out


def _operands(op, M, N, K, kind):
    g = torch.Generator().manual_seed(11)
    a_shape = (K, M) if op == "TN" else (M, K)
    b_shape = (N, K) if op == "NT" else (K, N)
    A = torch.randn(*a_shape, generator=g)
    B = torch.randn(*b_shape, generator=g)
    if kind == "wide":          # 16 decades of dynamic range inside every dot product
        A = A * torch.pow(10.0, torch.rand(*a_shape, generator=g) * 8 - 4)
        B = B * torch.pow(10.0, torch.rand(*b_shape, generator=g) * 8 - 4)
    if kind == "cancel":        # dot products that cancel to ~1e-4 of their terms
        A = A.abs() + 1.0
        sign = torch.ones(K)
        sign[1::2] = -1.0
        B = (1.0 + 1e-4 * B)
        B = B * (sign[None, :] if op == "NT" else sign[:, None])
    return A.float().contiguous(), B.float().contiguous()


@pytest.mark.parametrize("kind", ["normal", "wide", "cancel"])
@pytest.mark.parametrize("op,M,N,K,split_k", [("NT", 1024, 512, 2048, 1), ("NN", 1024, 512, 1024, 2), ("TN", 1024, 512, 4096, 4)])
def test_split_pipe_error_equals_fp32_pipe_error(op, M, N, K, split_k, kind):
    A, B = _operands(op, M, N, K, kind)
    Cs, Cf = _both_pipes(op, A.cuda(), B.cuda(), M, N, K, split_k)
    Ad, Bd = A.double(), B.double()
    Am = Ad.T if op == "TN" else Ad
    Bm = Bd.T if op == "NT" else Bd
    ref = Am @ Bm
    bound = Am.abs() @ Bm.abs()            # sum_k |a||b|: what rounding errors scale with
    e_split = float(((Cs - ref).abs() / bound).max())
    e_fp32 = float(((Cf - ref).abs() / bound).max())
    assert not torch.equal(Cs, Cf), "both settings produced bit-identical results: the split loop did not run"
    # condition-aware bound: the random-walk growth of K fp32 roundings per unit of sum |a||b|, for either pipe ...
    walk = 2.0 * K ** 0.5 * 2.0 ** -24
    assert e_fp32 < walk and e_split < walk, (e_split, e_fp32, walk)
    # ... and the split pipe within a small factor of the fp32 pipe (measured: 0.85 - 1.2x)
    assert e_split <= 2.0 * e_fp32 + 2.0 ** -26, (e_split, e_fp32)


def test_split_pipe_is_exact_on_bf16_representable_products():
    """Operands that are exact in bf16 and small integer sums: both pipes must return the exact integers."""
    M, N, K = 1024, 512, 512
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-8, 9, (M, K), generator=g).float()
    B = torch.randint(-8, 9, (N, K), generator=g).float()
    Cs, Cf = _both_pipes("NT", A.cuda(), B.cuda(), M, N, K)
    ref = A.double() @ B.double().T
    assert torch.equal(Cs, ref) and torch.equal(Cf, ref)


def test_split_pipe_handles_fp32_only_values():
    """Values with all 24 mantissa bits set (nothing bf16 can hold alone) times powers of two: the three-piece split is exact,
    so a K = 16 dot product of one non-zero term must come back bit-exact."""
    M, N, K = 1024, 512, 16
    A = torch.zeros(M, K)
    B = torch.zeros(N, K)
    A[:, 3] = torch.tensor([1.0 + (2 ** 23 - 1 - i) * 2.0 ** -23 for i in range(M)], dtype=torch.float64).float()      # 1.111...1b and neighbours
    B[:, 3] = torch.tensor([2.0 ** (j % 20 - 10) for j in range(N)])
    Cs, Cf = _both_pipes("NT", A.cuda(), B.cuda(), M, N, K)
    ref = A.double() @ B.double().T
    assert L.gemm_split_enabled() in (True, False)
    assert torch.equal(Cf, ref)
    assert torch.equal(Cs, ref)


@pytest.mark.parametrize("epi", ["stats", "store"])
@pytest.mark.parametrize("M,hw,K,N", [(12544, 49, 1152, 192), (12544, 49, 1152, 320), (4096 + 196 * 3 + 5, 196, 480, 80),
                                      (50176, 196, 672, 112), (4100, 49, 256, 64)])
def test_full_size_gated_project_conv_both_pipes(M, hw, K, N, epi):
    """MBConv project convolution at the benchmarked sizes (7x7 and 14x14 stages, ragged last row tile, an image boundary inside
    every row tile): BN + swish + squeeze-excite gate folded into the A operand, BatchNorm statistics in the epilogue.  On the
    split pipe this is gemm_split.hpp's PRO_BN_SWISH_GATE (transform in the staging registers), on the fp32 pipe the
    register-staged kernel; the autouse fixture runs both."""
    n_img = (M + hw - 1) // hw
    Z, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.2)
    sc, sh, gate = _rand(K, seed=3).abs() + 0.5, _rand(K, seed=4, scale=0.2), torch.sigmoid(_rand(n_img, K, seed=5))
    slots = 8
    stats = torch.zeros(slots, 2, N, dtype=torch.float64, device="cuda")
    Cd = torch.full((M, N), float("nan"), device="cuda")
    kw = dict(epilogue=L.EPI_STATS, stats=stats, stats_slots=slots) if epi == "stats" else {}
    L.gemm(L.OP_NT, Z.cuda(), W.cuda(), Cd, M, N, K, K, K, N, prologue=L.PRO_BN_SWISH_GATE, scale=sc.cuda(), shift=sh.cuda(),
           gate=gate.cuda(), hw=hw, **kw)
    a = Z.double() * sc.double() + sh.double()
    a = a * torch.sigmoid(a) * gate.double().repeat_interleave(hw, 0)[:M]
    ref = a @ W.double().T
    assert_close(Cd, ref, 5e-5, "gated project conv")
    if epi == "stats":
        st = stats.sum(0).cpu()
        assert_close(st[0], ref.sum(0), 1e-4, "column sums")
        assert_close(st[1], (ref * ref).sum(0), 1e-4, "column sums of squares")


@pytest.mark.parametrize("M,N,K", [(12544, 1152, 192), (50176, 480, 80), (12544, 1280, 320)])
def test_full_size_expand_conv_stats_both_pipes(M, N, K):
    """MBConv expand / head convolution shapes: plain operands, BatchNorm statistics in the epilogue."""
    A, W = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.2)
    slots = 8
    stats = torch.zeros(slots, 2, N, dtype=torch.float64, device="cuda")
    Cd = torch.full((M, N), float("nan"), device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), Cd, M, N, K, K, K, N, epilogue=L.EPI_STATS, stats=stats, stats_slots=slots)
    ref = A.double() @ W.double().T
    assert_close(Cd, ref, TOL, "expand conv")
    st = stats.sum(0).cpu()
    assert_close(st[0], ref.sum(0), 1e-4, "column sums")
    assert_close(st[1], (ref * ref).sum(0), 1e-4, "column sums of squares")


@pytest.mark.parametrize("epi", ["store", "bias_res"])
def test_presplit_weight_planes_are_bit_identical_to_the_in_kernel_split(epi, monkeypatch, matrix_pipe):
    """mt_split_planes + mt_gemm(b_planes=...) (opt-in: MT_SPLIT_PLANES=1): the planes sum back to the weight exactly and the DMA-fed
    loop returns the same bits as the loop that splits B per tile."""
    if matrix_pipe != "split":
        pytest.skip("b_planes is read by the split-operand loop only")
    M, N, K = 2048 + 37, 512, 1024
    A, W, b, R = _rand(M, K, seed=1).cuda(), _rand(N, K, seed=2, scale=0.05).cuda(), _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    P = L.split_planes(W)
    assert torch.equal(P.float().sum(0), W)
    outs = []
    for planes, env in ((None, "0"), (P, "1")):
        monkeypatch.setenv("MT_SPLIT_PLANES", env)
        C = torch.full((M, N), float("nan"), device="cuda")
        if epi == "store":
            L.gemm(L.OP_NT, A, W, C, M, N, K, K, K, N, bias=b, b_planes=planes)
        else:
            L.gemm(L.OP_NT, A, W, C, M, N, K, K, K, N, epilogue=L.EPI_BIAS_RES, bias=b, R=R, ldr=N, b_planes=planes)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    ref = A.double().cpu() @ W.double().cpu().T + b.double().cpu() + (R.double().cpu() if epi == "bias_res" else 0)
    assert_close(outs[1], ref, TOL, "pre-split planes")


@pytest.mark.parametrize("op,M,N,K,split_k", [("NT", 3136, 728, 728, 1), ("NN", 3136, 728, 728, 1), ("NT", 1000, 512, 520, 1),
                                              ("TN", 1024, 512, 4104, 4), ("NN", 2048, 512, 1544, 3)])
def test_k_tail_of_eight(op, M, N, K, split_k):
    """K % 16 == 8 (Xception's 728-channel middle flow): the split loop's last k-tile holds one 8-k granule and zeros."""
    A, B = _operands(op, M, N, K, "normal")
    Cs, Cf = _both_pipes(op, A.cuda(), B.cuda(), M, N, K, split_k) if op != "NN" or split_k > 1 else (None, None)
    Ad, Bd = A.double(), B.double()
    ref = (Ad.T if op == "TN" else Ad) @ (Bd.T if op == "NT" else Bd)
    if Cs is None:                                   # NN store form
        outs = []
        for on in (True, False):
            prev = L.set_gemm_split(on)
            C = torch.full((M, N), float("nan"), device="cuda")
            L.gemm(L.OP_NN, A.cuda(), B.cuda(), C, M, N, K, K, N, N)
            L.set_gemm_split(prev)
            outs.append(C.cpu().double())
        Cs, Cf = outs
    assert not torch.equal(Cs, Cf), "the split loop did not take this shape"
    assert_close(Cs, ref, TOL, "split pipe, K tail")
    assert_close(Cf, ref, TOL, "fp32 pipe")
