"""GPU parity of the plane-operand GEMM path (csrc/gemm_planes.hpp, through the C ABI): the blocked bf16 plane format, its
producers, and mt_gemm_planes against (a) mt_gemm's split-operand loop on the same fp32 operands -- bit for bit where both use
the same summation order -- and (b) fp64 CPU matmuls."""
import pytest
import torch

import mintime_amd
from mintime_amd import lib as L
from tests.util import assert_close

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("rows,cols", [(64, 32), (77, 40), (393 * 2, 512), (1, 8), (100, 2048)])
def test_blocked_planes_are_the_exact_split_with_zero_padding(rows, cols):
    x = _rand(rows, cols, seed=rows + cols)
    x[0, 0] = 1.0 + 2.0 ** -20                           # needs all three pieces
    xd = x.cuda()
    p = L.split_planes_blk(xd)
    assert p.shape == L.planes_shape(rows, cols) and p.numel() == 3 * L.get().mt_planes_elems(rows, cols)
    back = L.planes_to_float(p, rows, cols)
    assert torch.equal(back, xd), "p0 + p1 + p2 must give the fp32 value back exactly"
    # the pieces are the ones the in-kernel split computes (round-to-nearest bf16 at each level), in the blocked order
    p0 = x.bfloat16()
    r1 = x - p0.float()
    p1 = r1.bfloat16()
    p2 = (r1 - p1.float()).bfloat16()
    full = p.permute(0, 1, 3, 2, 4).reshape(3, p.shape[1] * 32, p.shape[2] * 16)
    for k, piece in enumerate((p0, p1, p2)):
        assert torch.equal(full[k, :rows, :cols].cpu(), piece), f"plane {k}"
    assert float(full[:, rows:, :].float().abs().max() if full.shape[1] > rows else 0.0) == 0.0
    assert float(full[:, :, cols:].float().abs().max() if full.shape[2] > cols else 0.0) == 0.0
    # strided source (leading dimension > cols)
    wide = _rand(rows, cols + 24, seed=5).cuda()
    p_ld = L.split_planes_blk(wide, rows, cols, ld=cols + 24)
    assert torch.equal(L.planes_to_float(p_ld, rows, cols), wide[:, :cols])


@pytest.mark.parametrize("M,N,K", [(786, 1536, 512), (300, 192, 136), (1000, 512, 2048), (129, 128, 16)])
def test_nt_planes_bit_identical_to_in_kernel_split(M, N, K):
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    ref = torch.full((M, N), float("nan"), device="cuda")
    prev = L.set_gemm_split(True)
    try:
        L.gemm(L.OP_NT, Ad, Wd, ref, M, N, K, K, K, N, bias=bd)
    finally:
        L.set_gemm_split(prev)
    got = torch.full((M, N), float("nan"), device="cuda")
    L.gemm_planes(L.OP_NT, L.split_planes_blk(Ad), L.split_planes_blk(Wd), M, N, K, Cout=got, ldc=N, bias=bd, streamk=False)
    assert_close(got, A.double() @ W.double().T + b.double(), TOL, "NT planes vs fp64")
    if K >= 512:                                          # (below MT_SPLIT_MIN_K's cached default mt_gemm ran the fp32 pipe)
        assert torch.equal(got, ref), "same pieces, same products, same order: the results must be bit-identical"
    # bias + residual
    R = _rand(M, N, seed=5).cuda()
    out = torch.empty(M, N, device="cuda")
    L.gemm_planes(L.OP_NT, L.split_planes_blk(Ad), L.split_planes_blk(Wd), M, N, K, Cout=out, ldc=N, epilogue=L.EPI_BIAS_RES, bias=bd,
                  R=R, ldr=N)
    assert_close(out, A.double() @ W.double().T + b.double() + R.cpu().double(), TOL, "NT planes bias + residual")


@pytest.mark.parametrize("M", [393 * 2, 1000, 12576])
def test_geglu_pair_emits_planes(M):
    D = 512
    A, W, b = _rand(M, D, seed=1), _rand(8 * D, D, seed=2, scale=0.05), _rand(8 * D, seed=3, scale=0.1)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    h_ref = torch.empty(M, 4 * D, device="cuda")
    u_ref = torch.empty(M, 8 * D, device="cuda")
    prev = L.set_gemm_split(True)
    try:
        L.gemm(L.OP_NT, Ad, Wd, h_ref, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=bd, C2=u_ref, ldc2=8 * D, n_half=4 * D)
    finally:
        L.set_gemm_split(prev)
    a_p, w_p = L.split_planes_blk(Ad), L.split_planes_blk(Wd)
    h_p = L.planes_empty(M, 4 * D, "cuda")
    h_p.fill_(float("nan"))
    u = torch.full((M, 8 * D), float("nan"), device="cuda")
    L.gemm_planes(L.OP_NT, a_p, w_p, M, 8 * D, D, epilogue=L.EPI_GEGLU, bias=bd, C2=u, ldc2=8 * D, n_half=4 * D, c_planes=h_p, streamk=False)
    assert torch.equal(u, u_ref), "pre-activations"
    assert torch.equal(L.planes_to_float(h_p, M, 4 * D), h_ref), "h planes = the exact split of the fp32 h"
    assert not torch.isnan(h_p.float()).any() and float(L.planes_to_float(h_p, h_p.shape[1] * 32, 4 * D)[M:].abs().sum()) == 0.0
    # fp32 h next to the planes
    h32 = torch.full((M, 4 * D), float("nan"), device="cuda")
    L.gemm_planes(L.OP_NT, a_p, w_p, M, 8 * D, D, Cout=h32, ldc=4 * D, epilogue=L.EPI_GEGLU, bias=bd, n_half=4 * D, streamk=False)
    assert torch.equal(h32, h_ref)
    # ---- GEGLU backward: du = [dh * gelu(g) | dh * a * gelu'(g)] from dh = dx . W2, W2 [D, 4D] read along its rows (NN)
    dx, W2 = _rand(M, D, seed=7), _rand(D, 4 * D, seed=8, scale=0.05)
    dxd, W2d = dx.cuda(), W2.cuda()
    du_ref = torch.empty(M, 8 * D, device="cuda")
    cs_ref = torch.zeros(8 * D, device="cuda")
    prev = L.set_gemm_split(True)
    try:
        L.gemm(L.OP_NN, dxd, W2d, du_ref, M, 4 * D, D, D, 4 * D, 8 * D, epilogue=L.EPI_GEGLU_BWD, C2=u_ref, ldc2=8 * D, n_half=4 * D,
               col_sum=cs_ref)
    finally:
        L.set_gemm_split(prev)
    du_p = L.planes_empty(M, 8 * D, "cuda")
    du_p.fill_(float("nan"))
    cs = torch.zeros(8 * D, device="cuda")
    L.gemm_planes(L.OP_NN, L.split_planes_blk(dxd), L.split_planes_blk(W2d), M, 4 * D, D, epilogue=L.EPI_GEGLU_BWD, C2=u_ref, ldc2=8 * D,
                  n_half=4 * D, col_sum=cs, c_planes=du_p, streamk=False)
    assert torch.equal(L.planes_to_float(du_p, M, 8 * D), du_ref), "du planes = the exact split of the fp32 du"
    assert_close(cs, du_ref.double().sum(0), 1e-4, "column sums of du (bias gradient)")
    assert float(L.planes_to_float(du_p, du_p.shape[1] * 32, 8 * D)[M:].abs().sum()) == 0.0


@pytest.mark.parametrize("D", [48, 80, 144])
def test_geglu_backward_planes_when_the_half_width_is_not_a_multiple_of_the_block_tile(D):
    """n_half = 4 D with D % 32 == 16 is not a multiple of the 128-column block tile: the last tile's upper waves own no columns and
    must not touch the dg half of the planes (they once wrote zeros over columns another block owns)."""
    M = 786
    A, W, b = _rand(M, D, seed=1), _rand(8 * D, D, seed=2, scale=0.2), _rand(8 * D, seed=3, scale=0.1)
    u = torch.empty(M, 8 * D, device="cuda")
    h = torch.empty(M, 4 * D, device="cuda")
    L.gemm(L.OP_NT, A.cuda(), W.cuda(), h, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b.cuda(), C2=u, ldc2=8 * D, n_half=4 * D)
    dx, W2 = _rand(M, D, seed=7), _rand(D, 4 * D, seed=8, scale=0.2)
    du_ref = torch.empty(M, 8 * D, device="cuda")
    cs_ref = torch.zeros(8 * D, device="cuda")
    L.gemm(L.OP_NN, dx.cuda(), W2.cuda(), du_ref, M, 4 * D, D, D, 4 * D, 8 * D, epilogue=L.EPI_GEGLU_BWD, C2=u, ldc2=8 * D, n_half=4 * D,
           col_sum=cs_ref)
    for rep in range(3):                                  # (the overwrite was a race between blocks: give it a few chances)
        du_p = L.planes_empty(M, 8 * D, "cuda")
        du_p.fill_(float("nan"))
        cs = torch.zeros(8 * D, device="cuda")
        L.gemm_planes(L.OP_NN, L.split_planes_blk(dx.cuda()), L.split_planes_blk(W2.cuda()), M, 4 * D, D, epilogue=L.EPI_GEGLU_BWD,
                      C2=u, ldc2=8 * D, n_half=4 * D, col_sum=cs, c_planes=du_p, streamk=False)
        got = L.planes_to_float(du_p, M, 8 * D)
        assert not torch.isnan(got).any()
        assert_close(got, du_ref, TOL, f"du planes, D = {D}")
        assert_close(got[:, 4 * D:], du_ref[:, 4 * D:], TOL, f"dg half of du, D = {D}")
        assert_close(cs, du_ref.double().sum(0), 1e-4, "column sums of du")


@pytest.mark.parametrize("M,N,K", [(786, 512, 1536), (1000, 512, 4096), (300, 136, 512), (786, 512, 520)])
def test_nn_planes_reads_the_weight_along_its_rows(M, N, K):
    """Data gradient dX[M,N] = dY[M,K] . W[K,N] with W as stored: no transposed copy, LDS transpose-reads."""
    dY, W = _rand(M, K, seed=1), _rand(K, N, seed=2, scale=0.05)
    dYd, Wd = dY.cuda(), W.cuda()
    got = torch.full((M, N), float("nan"), device="cuda")
    L.gemm_planes(L.OP_NN, L.split_planes_blk(dYd), L.split_planes_blk(Wd), M, N, K, Cout=got, ldc=N, streamk=False)
    assert_close(got, dY.double() @ W.double(), TOL, "NN planes vs fp64")
    if K % 16 == 0 and N % 4 == 0 and M * N >= 1 << 18:     # (smaller problems: mt_gemm runs them on the fp32 pipe)
        ref = torch.empty(M, N, device="cuda")
        prev = L.set_gemm_split(True)
        try:
            L.gemm(L.OP_NN, dYd, Wd, ref, M, N, K, K, N, N)
        finally:
            L.set_gemm_split(prev)
        assert torch.equal(got, ref), "bit-identical to the in-kernel split"


@pytest.mark.parametrize("Mo,No,K", [(1536, 512, 786), (512, 2048, 1000), (4096, 512, 12576), (136, 200, 3 * 393)])
def test_tn_planes_weight_gradient(Mo, No, K):
    """dW[Mo,No] = dY[K,Mo]^T X[K,No]: both operands read along their rows from the planes the forward / data-gradient GEMMs use."""
    dY, X = _rand(K, Mo, seed=1, scale=0.1), _rand(K, No, seed=2)
    dW = torch.zeros(Mo, No, device="cuda")
    L.gemm_planes(L.OP_TN, L.split_planes_blk(dY.cuda()), L.split_planes_blk(X.cuda()), Mo, No, K, Cout=dW, ldc=No, epilogue=L.EPI_ATOMIC)
    assert_close(dW, dY.double().T @ X.double(), TOL, "TN planes vs fp64")
    # accumulates onto what is there
    L.gemm_planes(L.OP_TN, L.split_planes_blk(dY.cuda()), L.split_planes_blk(X.cuda()), Mo, No, K, Cout=dW, ldc=No, epilogue=L.EPI_ATOMIC)
    assert_close(dW, 2 * (dY.double().T @ X.double()), TOL, "TN planes accumulate")


@pytest.mark.parametrize("rows", [393 * 2, 77, 12576])
def test_layernorm_kernels_emit_planes(rows):
    D = 512
    x, g, b = _rand(rows, D, seed=1), _rand(D, seed=2), _rand(D, seed=3)
    xd, gd, bd = x.cuda(), g.cuda(), b.cuda()
    y = torch.empty(rows, D, device="cuda")
    stats = torch.empty(rows, 2, device="cuda")
    y_p = L.planes_empty(rows, D, "cuda")
    y_p.fill_(float("nan"))
    L.check(L.get().mt_layernorm_fwd(L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(y), L.ptr(stats), rows, D, 1e-5, L.ptr(y_p), L.stream_ptr()), "ln fwd")
    assert_close(y, torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-5), 1e-5, "LayerNorm")
    assert torch.equal(L.planes_to_float(y_p, rows, D), y)
    assert float(L.planes_to_float(y_p, y_p.shape[1] * 32, D)[rows:].abs().sum()) == 0.0 and not torch.isnan(y_p.float()).any()
    # planes only (no fp32 output)
    y_p2 = L.planes_empty(rows, D, "cuda")
    L.check(L.get().mt_layernorm_fwd(L.ptr(xd), L.ptr(gd), L.ptr(bd), None, None, rows, D, 1e-5, L.ptr(y_p2), L.stream_ptr()), "ln fwd planes only")
    assert torch.equal(y_p2, y_p)
    # backward rows kernel: dx fp32 and its planes
    dy, dx_in = _rand(rows, D, seed=4).cuda(), _rand(rows, D, seed=5).cuda()
    dx = torch.empty(rows, D, device="cuda")
    dx_ref = torch.empty(rows, D, device="cuda")
    dx_p = L.planes_empty(rows, D, "cuda")
    dx_p.fill_(float("nan"))
    L.check(L.get().mt_layernorm_bwd_rows(L.ptr(dy), L.ptr(xd), L.ptr(stats), L.ptr(gd), L.ptr(dx_ref), L.ptr(dx_in), rows, D, None,
                                          L.stream_ptr()), "ln bwd rows")
    L.check(L.get().mt_layernorm_bwd_rows(L.ptr(dy), L.ptr(xd), L.ptr(stats), L.ptr(gd), L.ptr(dx), L.ptr(dx_in), rows, D, L.ptr(dx_p),
                                          L.stream_ptr()), "ln bwd rows + planes")
    assert torch.equal(dx, dx_ref) and torch.equal(L.planes_to_float(dx_p, rows, D), dx)
    assert float(L.planes_to_float(dx_p, dx_p.shape[1] * 32, D)[rows:].abs().sum()) == 0.0 and not torch.isnan(dx_p.float()).any()


def test_multi_tensor_split_matches_single():
    ws = [_rand(1536, 512, seed=1).cuda(), _rand(512, 512, seed=2).cuda(), _rand(512, 2048, seed=3).cuda(), _rand(40, 24, seed=4).cuda()]
    outs = [L.planes_empty(w.shape[0], w.shape[1], "cuda") for w in ws]
    rows, first = [], 0
    for w, o in zip(ws, outs):
        rows.append((w.data_ptr(), o.data_ptr(), w.shape[0], w.shape[1], first))
        first += o.shape[1] * o.shape[2]
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    L.check(L.get().mt_split_planes_blk_multi(table.data_ptr(), len(rows), first, L.stream_ptr()), "multi split")
    for w, o in zip(ws, outs):
        assert torch.equal(o, L.split_planes_blk(w))


@pytest.mark.parametrize("op,M,N,K,epi", [(L.OP_NT, 12576, 1536, 512, L.EPI_STORE), (L.OP_NT, 12576, 512, 2048, L.EPI_BIAS_RES),
                                           (L.OP_NN, 12576, 512, 4096, L.EPI_STORE), (L.OP_NT, 6288, 512, 512, L.EPI_BIAS_RES),
                                           (L.OP_NN, 3000, 200, 520, L.EPI_STORE)])
def test_stream_k_matches_one_block_per_tile_and_is_reproducible(op, M, N, K, epi):
    """Stream-K (persistent grid sharing the (tile, k-step) list) against one block per tile: same sums up to the association of a
    split tile's two or three partial sums, identical from run to run, and the lent workspace comes back zero-filled."""
    A = _rand(M, K, seed=1)
    Bm = _rand(N, K, seed=2, scale=0.05) if op == L.OP_NT else _rand(K, N, seed=2, scale=0.05)
    b, R = _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    a_p, b_p = L.split_planes_blk(A.cuda()), L.split_planes_blk(Bm.cuda())
    kw = dict(Cout=None, ldc=N, epilogue=epi, bias=b)
    if epi == L.EPI_BIAS_RES:
        kw.update(R=R, ldr=N)
    outs = []
    for sk in (False, True, True):
        c = torch.full((M, N), float("nan"), device="cuda")
        kw["Cout"] = c
        L.gemm_planes(op, a_p, b_p, M, N, K, streamk=sk, **kw)
        outs.append(c)
    ref = (A.double() @ (Bm.double().T if op == L.OP_NT else Bm.double())) + b.cpu().double()
    if epi == L.EPI_BIAS_RES:
        ref = ref + R.cpu().double()
    assert_close(outs[1], ref, TOL, "stream-K vs fp64")
    assert_close(outs[1], outs[0], 2e-6, "stream-K vs one block per tile")
    assert torch.equal(outs[1], outs[2]), "stream-K must be bit-reproducible"
    ws = L.streamk_workspace("cuda")
    assert ws is not None and int(ws[:4096].view(torch.int32).abs().sum()) == 0, "flags must be cleared by their consumers"


@pytest.mark.parametrize("op,M,N,K,epi", [(L.OP_NT, 12576, 1536, 512, L.EPI_STORE), (L.OP_NT, 12576, 1536, 520, L.EPI_BIAS_RES),
                                           (L.OP_NN, 12576, 1024, 1536, L.EPI_STORE), (L.OP_NT, 9000, 640, 48, L.EPI_STORE)])
def test_persistent_blocks_give_the_same_bits(op, M, N, K, epi):
    """mt_gemm_planes_set_persist(2): 2 blocks per CU walk the tile list and prefetch the next tile's first stage under the epilogue.
    Every tile is still summed by one block in the same order: the results are the one-block-per-tile bits (also with a ragged K and
    with more XCD bands than tiles per band)."""
    A = _rand(M, K, seed=1)
    Bm = _rand(N, K, seed=2, scale=0.05) if op == L.OP_NT else _rand(K, N, seed=2, scale=0.05)
    b, R = _rand(N, seed=3).cuda(), _rand(M, N, seed=4).cuda()
    a_p, b_p = L.split_planes_blk(A.cuda()), L.split_planes_blk(Bm.cuda())
    kw = dict(ldc=N, epilogue=epi, bias=b, streamk=False)
    if epi == L.EPI_BIAS_RES:
        kw.update(R=R, ldr=N)
    outs = []
    lib = L.get()
    prev = lib.mt_gemm_planes_set_persist(0)
    try:
        for per_cu in (0, 2, 1):
            lib.mt_gemm_planes_set_persist(per_cu)
            c = torch.full((M, N), float("nan"), device="cuda")
            L.gemm_planes(op, a_p, b_p, M, N, K, Cout=c, **kw)
            outs.append(c)
    finally:
        lib.mt_gemm_planes_set_persist(prev)
    ref = (A.double() @ (Bm.double().T if op == L.OP_NT else Bm.double())) + b.cpu().double()
    if epi == L.EPI_BIAS_RES:
        ref = ref + R.cpu().double()
    assert_close(outs[0], ref, TOL, "one block per tile vs fp64")
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])


def test_stream_k_under_a_co_running_kernel():
    """The persistent blocks are not all resident when another stream holds CUs: a block only waits for slabs its successors write
    FIRST, so the launch still completes (and gives the same bits)."""
    M, N, K = 12576, 512, 2048
    a_p, b_p = L.split_planes_blk(_rand(M, K, seed=1).cuda()), L.split_planes_blk(_rand(N, K, seed=2, scale=0.05).cuda())
    quiet = torch.empty(M, N, device="cuda")
    L.gemm_planes(L.OP_NT, a_p, b_p, M, N, K, Cout=quiet, ldc=N, streamk=True)
    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    xa, xb = _rand(4096, 4096, seed=5).cuda(), _rand(4096, 4096, seed=6).cuda()
    xc = torch.empty(4096, 4096, device="cuda")
    busy = torch.empty(M, N, device="cuda")
    with torch.cuda.stream(other):
        for _ in range(6):
            L.gemm(L.OP_NT, xa, xb, xc, 4096, 4096, 4096, 4096, 4096, 4096)      # ~0.7 ms each, three blocks per CU
    for _ in range(4):
        L.gemm_planes(L.OP_NT, a_p, b_p, M, N, K, Cout=busy, ldc=N, streamk=True)
    torch.cuda.synchronize()
    assert torch.equal(busy, quiet)


@pytest.mark.parametrize("B,F,mode", [(2, 8, 0), (2, 8, 1), (3, 16, 0), (3, 16, 1)])
def test_attention_kernels_emit_planes(B, F, mode):
    H, n, dh = 8, 49, 64
    inner, N = H * dh, 1 + F * n
    M = B * N
    qkv = _rand(M, 3 * inner, seed=1, scale=0.5).cuda()
    mask = torch.ones(B, F, dtype=torch.uint8)
    mask[0, F - 1] = 0
    ident = torch.ones(B, F, F, dtype=torch.uint8)
    ident[0, : F // 2, F // 2:] = 0
    ident[0, F // 2:, : F // 2] = 0
    mask_d, ident_d = mask.cuda(), ident.cuda()
    lib = L.get()
    o = torch.full((M, inner), float("nan"), device="cuda")
    L.check(lib.mt_attn_fwd(L.ptr(qkv), L.ptr(o), None, L.ptr(mask_d), L.ptr(ident_d), B, H, F, n, mode, 0.125, None, L.stream_ptr()), "attn fwd")
    o_p = L.planes_empty(M, inner, "cuda")
    o_p.fill_(float("nan"))
    L.check(lib.mt_attn_fwd(L.ptr(qkv), None, None, L.ptr(mask_d), L.ptr(ident_d), B, H, F, n, mode, 0.125, L.ptr(o_p), L.stream_ptr()), "attn fwd planes")
    assert torch.equal(L.planes_to_float(o_p, M, inner), o)
    assert not torch.isnan(o_p.float()).any() and float(L.planes_to_float(o_p, o_p.shape[1] * 32, inner)[M:].abs().sum()) == 0.0
    # backward
    do = _rand(M, inner, seed=2).cuda()
    dqkv = torch.full((M, 3 * inner), float("nan"), device="cuda")
    L.check(lib.mt_attn_bwd(L.ptr(qkv), L.ptr(do), L.ptr(dqkv), L.ptr(mask_d), L.ptr(ident_d), B, H, F, n, mode, 0.125, None, L.stream_ptr()), "attn bwd")
    work = torch.full((M, 3 * inner), float("nan"), device="cuda")
    d_p = L.planes_empty(M, 3 * inner, "cuda")
    d_p.fill_(float("nan"))
    L.check(lib.mt_attn_bwd(L.ptr(qkv), L.ptr(do), L.ptr(work), L.ptr(mask_d), L.ptr(ident_d), B, H, F, n, mode, 0.125, L.ptr(d_p), L.stream_ptr()), "attn bwd planes")
    got = L.planes_to_float(d_p, M, 3 * inner)
    assert not torch.isnan(d_p.float()).any()
    patch = torch.ones(M, dtype=torch.bool, device="cuda")
    patch[::N] = False                                   # cls rows: dk / dv are sums of fp32 atomics (order varies)
    assert torch.equal(got[patch], dqkv[patch]), "patch rows are written by their single owner: identical values"
    assert_close(got[~patch], dqkv[~patch], 1e-5, "cls rows")
    assert float(L.planes_to_float(d_p, d_p.shape[1] * 32, 3 * inner)[M:].abs().sum()) == 0.0


@pytest.mark.parametrize("rows,C", [(200, 728), (96, 64), (1000, 1536)])
def test_bn_bwd_apply_planes_is_the_split_of_the_fp32_kernel(rows, C):
    """mt_bn_bwd_apply_planes = the planes of exactly what mt_bn_bwd_apply stores (dz = ka du + kb z + kc), zero padded
    (728 columns: the last 16-column block is half padding)."""
    lib = L.get()
    du, z = _rand(rows, C, seed=1).cuda(), _rand(rows, C, seed=2).cuda()
    kabc = _rand(3, C, seed=3).cuda()
    dz = torch.empty(rows, C, device="cuda")
    L.check(lib.mt_bn_bwd_apply(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(dz), rows, C, L.stream_ptr()), "mt_bn_bwd_apply")
    p = L.planes_empty(rows, C, "cuda")
    p.fill_(7.0)                                          # the kernel must overwrite everything, padding included
    L.check(lib.mt_bn_bwd_apply_planes(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(p), rows, C, L.stream_ptr()), "mt_bn_bwd_apply_planes")
    assert torch.equal(L.planes_to_float(p, rows, C), dz)
    assert torch.equal(p, L.split_planes_blk(dz, rows, C)), "same pieces and zero padding as the converter"
    ref = kabc[0].double() * du.double() + kabc[1].double() * z.double() + kabc[2].double()
    assert_close(dz, ref, 1e-6, "dz vs fp64")


@pytest.mark.parametrize("M,N,K", [(1000, 728, 728), (512, 256, 1024)])
def test_plane_gemm_stats_epilogue(M, N, K):
    """MT_EPI_STATS on the plane loop (a 1x1 convolution under train-mode BatchNorm): the stored result is the STORE epilogue's,
    the [slots][2][N] fp64 accumulators sum to the column sums / sums of squares of what was stored."""
    a, w = _rand(M, K, seed=4).cuda(), _rand(N, K, seed=5, scale=0.05).cuda()
    ap, wp = L.split_planes_blk(a, M, K), L.split_planes_blk(w, N, K)
    ref = torch.empty(M, N, device="cuda")
    L.gemm_planes(L.OP_NT, ap, wp, M, N, K, Cout=ref, ldc=N)
    slots = 32
    out = torch.empty(M, N, device="cuda")
    stats = torch.zeros(slots, 2, N, dtype=torch.float64, device="cuda")
    L.gemm_planes(L.OP_NT, ap, wp, M, N, K, Cout=out, ldc=N, epilogue=L.EPI_STATS, stats=stats, stats_slots=slots)
    assert torch.equal(out, ref)
    s = stats.sum(0)
    assert_close(s[0], out.double().sum(0), 1e-6, "column sums")
    assert_close(s[1], (out.double() ** 2).sum(0), 1e-6, "column sums of squares")
    assert_close(out, a.double() @ w.double().t(), TOL, "result vs fp64")


def test_dropout_passes_match_torch():
    """csrc/dropout.hip through the C ABI: planes of x * m, out = r + y * m, and the GEGLU backward pass (with and without a
    multiplier) against torch autograd of a * gelu(g)."""
    lib = L.get()
    rows, nh = 100, 72
    x, m = _rand(rows, nh, seed=1).cuda(), (torch.rand(rows, nh, generator=torch.Generator().manual_seed(2)) > 0.3).float().cuda() / 0.7
    p, out = L.planes_empty(rows, nh, "cuda"), torch.empty(rows, nh, device="cuda")
    L.check(lib.mt_mul_planes(L.ptr(x), L.ptr(m), L.ptr(p), L.ptr(out), rows, nh, L.stream_ptr()), "mt_mul_planes")
    assert torch.equal(out, x * m) and torch.equal(p, L.split_planes_blk(x * m, rows, nh))
    r = _rand(rows, nh, seed=3).cuda()
    y = torch.empty_like(r)
    L.check(lib.mt_mul_add(L.ptr(x), L.ptr(m), L.ptr(r), L.ptr(y), rows * nh, L.stream_ptr()), "mt_mul_add")
    assert_close(y, r.double() + x.double() * m.double(), 1e-6, "r + y m")
    # GEGLU backward
    a = _rand(rows, nh, seed=4).double().requires_grad_(True)
    g = _rand(rows, nh, seed=5).double().requires_grad_(True)
    dh = _rand(rows, nh, seed=6)
    for mult in (None, m):
        a.grad = g.grad = None
        h = a * torch.nn.functional.gelu(g)
        h.backward(dh.double() * (mult.cpu().double() if mult is not None else 1.0))
        u = torch.stack([a.detach().float(), g.detach().float()], dim=-1).reshape(rows, 2 * nh).contiguous().cuda()    # (a_0, g_0, a_1, g_1, ...)
        du_p, du = L.planes_empty(rows, 2 * nh, "cuda"), torch.empty(rows, 2 * nh, device="cuda")
        L.check(lib.mt_geglu_bwd(L.ptr(dh.cuda()), L.ptr(mult), L.ptr(u), L.ptr(du_p), L.ptr(du), rows, nh, L.stream_ptr()), "mt_geglu_bwd")
        assert_close(du[:, :nh], a.grad, 1e-5, "da")
        assert_close(du[:, nh:], g.grad, 1e-5, "dg")
        assert torch.equal(L.planes_to_float(du_p, rows, 2 * nh), du)


def test_operands_beyond_four_gigabytes_are_refused():
    """The plane loop reaches an operand's three planes by 32-bit byte offsets from one base: a 6.08 M x 128 activation (Xception's
    first block at 512 crops) spans 4.67 GB and must be refused loudly, not wrapped (lib.planes_fit is what the engines ask)."""
    assert L.planes_fit(1548800, 256) and L.planes_fit(25120, 4096)
    assert not L.planes_fit(6083072, 128)
    a = torch.zeros(4096, dtype=torch.bfloat16, device="cuda")          # never read: the shape check comes first
    out = torch.zeros(128, 128, device="cuda")
    with pytest.raises(L.MintimeHipError, match="4 GB"):
        L.gemm_planes(L.OP_NT, a, a, 6083072, 128, 128, Cout=out, ldc=128)
