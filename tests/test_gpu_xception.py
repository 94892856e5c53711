"""GPU parity of the HIP Xception (config 5 extractor) against the reference-generated fixtures and the CPU oracle."""
import numpy as np
import pytest
import torch

import mintime_amd
from mintime_amd import synth, xception
from oracle import mintime_oracle as O
from tests.util import REL_TOL, assert_close, golden, rel_err

pytestmark = pytest.mark.gpu


def _model(seed, training):
    m = xception(num_classes=1, pretrain_path=None)
    sd = synth.xception_state(seed)
    m.load_state_dict(sd, strict=True)
    m.train(training)
    return m.cuda(), sd


def _input(n, seed):
    return synth.clip_inputs(1, n, 1, seed)["videos"].reshape(n, 224, 224, 3).permute(0, 3, 1, 2)


@pytest.mark.parametrize("name", ["xc_eval", "xc_train"])
def test_forward_and_backward_match_reference_fixture(name):
    g = golden(name)
    n, training, seed = int(g["n_img"]), bool(g["training"]), int(g["seed"])
    model, sd = _model(seed, training)
    x = _input(n, seed)
    feats = model(x.cuda())
    assert feats.shape == (n, 2048, 7, 7)
    assert_close(feats, g["features"], REL_TOL, "features vs reference (fp32)")
    assert_close(feats[:, :256], g["feat64_slice"], REL_TOL, "features vs reference (fp64)")
    if training:
        msd = model.state_dict()
        for k in g.files:
            if k.startswith("stat."):
                assert_close(msd[k[5:]], g[k], REL_TOL, k)
    gw = torch.from_numpy(np.random.Generator(np.random.Philox(key=[seed, 4242])).standard_normal((n, 2048, 7, 7)) * 0.1).float()
    (feats * gw.cuda()).sum().backward()
    named = dict(model.named_parameters())
    # ReLU / max-pool networks are piecewise linear: a rounding-level change flips a few masks and moves upstream gradients
    # by ~1e-2.  The yardstick is therefore the reference arithmetic's OWN fp32-vs-fp64 deviation (CPU oracle, same ops):
    # the HIP path must stay within 3e-3 + 2x that.
    s32 = {k: (v.clone().requires_grad_("running_" not in k and not k.startswith("fc")) if v.is_floating_point() else v)
           for k, v in sd.items()}
    (O.xception_forward(s32, x, training=training) * gw).sum().backward()
    worst = (0.0, 0.0)
    for k in g.files:
        if k.startswith("gnorm64."):
            key = k[len("gnorm64."):]
            assert named[key].grad is not None, key
            ref64 = g["gslice64." + key]
            floor = rel_err(s32[key].grad.reshape(-1)[:256], ref64)
            ours = rel_err(named[key].grad.reshape(-1)[:256], ref64)
            assert ours <= 3 * REL_TOL + 2 * floor, f"{key}: ours {ours:.2e} vs fp32-reference floor {floor:.2e}"
            nfloor = rel_err(s32[key].grad.norm(), g[k])
            nours = rel_err(named[key].grad.norm(), g[k])
            assert nours <= 3 * REL_TOL + 2 * nfloor, f"{k}: ours {nours:.2e} vs fp32-reference floor {nfloor:.2e}"
            worst = max(worst, (ours, floor))
    print(f"{name}: worst gradient-slice error vs fp64 reference: ours {worst[0]:.2e}, reference-fp32 {worst[1]:.2e}")


@pytest.mark.parametrize("training", [False, True])
def test_all_parameter_gradients_vs_oracle(training):
    seed, n = 4, 2
    model, sd = _model(seed, training)
    x = _input(n, seed)
    gen = torch.Generator().manual_seed(3)
    wgt = torch.randn(n, 2048, 7, 7, generator=gen) * 0.1
    feat = model(x.cuda())
    (feat * wgt.cuda()).sum().backward()
    osd = {k: (v.double().requires_grad_("running_" not in k and not k.startswith("fc")) if v.is_floating_point() else v)
           for k, v in sd.items()}
    ofeat = O.xception_forward(osd, x.double(), training=training)
    (ofeat * wgt.double()).sum().backward()
    o32 = {k: (v.clone().requires_grad_("running_" not in k and not k.startswith("fc")) if v.is_floating_point() else v)
           for k, v in sd.items()}
    (O.xception_forward(o32, x, training=training) * wgt).sum().backward()
    assert_close(feat, ofeat, REL_TOL, "features")
    worst = 0.0
    for k, p in model.named_parameters():
        if k.startswith("fc"):
            assert p.grad is None
            continue
        ref = osd[k].grad
        if training and k.endswith(".bias") and float(ref.norm()) < 1e-9 * float(osd[k.replace(".bias", ".weight")].grad.norm() + 1e-30):
            continue
        # whole-tensor comparison in relative L2 (mask-flip noise is heavy-tailed, max-norm over 500k entries is not robust)
        rl2 = lambda a: float((a.detach().cpu().double() - ref).norm() / ref.norm())
        floor, ours = rl2(o32[k].grad), rl2(p.grad)
        assert ours <= 3 * REL_TOL + 3 * floor, f"grad {k}: ours {ours:.2e} vs fp32-reference floor {floor:.2e}"
        worst = max(worst, ours)
    print("worst relative gradient error vs fp64", worst)


def test_uint8_crops_are_ingested_directly():
    """Next-row f2: conv1's im2col gather reads uint8 crops; features and conv1's weight gradient equal the fp32-input run."""
    outs = []
    for as_u8 in (False, True):
        model, _ = _model(4, True)
        x = _input(2, 6)
        x = (x.to(torch.uint8) if as_u8 else x).cuda()
        f = model(x)
        f.sum().backward()
        outs.append((f.detach().clone(), model.conv1.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert_close(outs[1][1], outs[0][1], 1e-5, "conv1 weight gradient (split-K atomics order only)")


@pytest.mark.parametrize("N,H,C", [(3, 19, 128), (2, 28, 728), (2, 7, 64)])
def test_max_pool_adjoint_from_the_recorded_arg_max(N, H, C):
    """Block tail (xception.py:64-79) with the routing recorded in the forward: y is the plain kernel's, arg / zmax are torch's
    max_pool2d indices and the raw z there, the routed gradient is torch's max_pool2d backward (one writer per element: the same bits
    on every launch), the pooled BatchNorm sums equal the full-resolution ones, and the fused BatchNorm-backward + routing writes the
    planes of ka du + kb z + kc."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator().manual_seed(N * 1000 + H)
    Ho = (H - 1) // 2 + 1
    z = torch.randn(N, H, H, C, generator=g).cuda()
    z[0, :4, :4] = z[0, 0, 0]                                            # ties: the first maximum in window order wins
    # some negative gammas; powers of two, so that z * scale is exact and torch's (unfused) affine rounds like the kernels' fma
    sc = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (C,), generator=g)] * torch.where(torch.rand(C, generator=g) < 0.2, -1.0, 1.0)
    sc, sh = sc.cuda(), torch.randn(C, generator=g).cuda()
    zs, ss, hs = torch.randn(N, Ho, Ho, C, generator=g).cuda(), torch.rand(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    y0, y1, zmax = (torch.empty(N, Ho, Ho, C, device="cuda") for _ in range(3))
    arg = torch.empty(N, Ho, Ho, C, dtype=torch.uint8, device="cuda")
    st = L.stream_ptr()
    L.check(lib.mt_maxpool_add_fwd(L.ptr(z), L.ptr(sc), L.ptr(sh), L.ptr(zs), L.ptr(ss), L.ptr(hs), L.ptr(y0), N, H, H, C, st), "fwd")
    L.check(lib.mt_maxpool_add_fwd_arg(L.ptr(z), L.ptr(sc), L.ptr(sh), L.ptr(zs), L.ptr(ss), L.ptr(hs), L.ptr(y1), L.ptr(arg), L.ptr(zmax),
                                       N, H, H, C, st), "fwd_arg")
    assert torch.equal(y0, y1)
    u = (z * sc + sh).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pooled, idx = torch.nn.functional.max_pool2d(u, 3, 2, 1, return_indices=True)
    ih, iw = idx // H, idx % H
    oh = torch.arange(Ho, device="cuda").view(1, 1, Ho, 1)
    ow = torch.arange(Ho, device="cuda").view(1, 1, 1, Ho)
    k_ref = (ih - 2 * oh + 1) * 3 + (iw - 2 * ow + 1)
    assert torch.equal(arg.permute(0, 3, 1, 2).long(), k_ref)
    assert torch.equal(zmax.permute(0, 3, 1, 2), z.permute(0, 3, 1, 2).reshape(N, C, H * H).gather(2, idx.view(N, C, -1)).view(N, C, Ho, Ho))
    dy = torch.randn(N, Ho, Ho, C, generator=g).cuda()
    pooled.backward(dy.permute(0, 3, 1, 2))
    du_ref = u.grad.permute(0, 2, 3, 1).contiguous()
    du = torch.full((N, H, H, C), 7.0, device="cuda")
    L.check(lib.mt_maxpool_bwd_arg(L.ptr(dy), L.ptr(arg), L.ptr(du), N, H, H, C, st), "bwd_arg")
    assert_close(du, du_ref, 1e-6, "routed gradient vs torch")
    du2 = torch.full((N, H, H, C), 7.0, device="cuda")
    L.check(lib.mt_maxpool_bwd_arg(L.ptr(dy), L.ptr(arg), L.ptr(du2), N, H, H, C, st), "bwd_arg")
    assert torch.equal(du, du2)
    old = torch.zeros(N, H, H, C, device="cuda")
    L.check(lib.mt_maxpool_bwd(L.ptr(dy), L.ptr(z), L.ptr(sc), L.ptr(sh), L.ptr(old), N, H, H, C, st), "bwd")
    assert_close(du, old, 1e-6, "routed gradient vs the arg-max scatter")
    # BatchNorm-backward sums: pooled tensors vs full resolution
    mi = torch.stack([z.mean((0, 1, 2)), 1.0 / z.var((0, 1, 2), unbiased=False).add(1e-5).sqrt()]).contiguous()
    sums = []
    for d_, z_, rows in ((dy, zmax, N * Ho * Ho), (du, z, N * H * H)):
        s_ = torch.zeros(32, 2, C, dtype=torch.float64, device="cuda")
        L.check(lib.mt_bn_act_bwd(L.ptr(d_), L.ptr(z_), L.ptr(sc), L.ptr(sh), L.ptr(mi), None, None, None, None, L.ptr(s_), 32, rows, C, 1, 0,
                                  st), "sums")
        sums.append(s_.sum(0))
    assert_close(sums[0], sums[1], 1e-5, "sum du, sum du xhat: pooled vs full resolution")
    # fused BatchNorm-backward affine + routing, as planes
    kabc = torch.randn(3, C, generator=g).cuda()
    dz = torch.empty(N * H * H, C, device="cuda")
    L.check(lib.mt_bn_bwd_apply(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(dz), N * H * H, C, st), "apply")
    p = L.planes_empty(N * H * H, C, "cuda")
    p.fill_(7.0)
    L.check(lib.mt_maxpool_bn_bwd_apply_planes(L.ptr(dy), L.ptr(arg), L.ptr(z), L.ptr(kabc), L.ptr(p), N, H, H, C, st), "apply_planes")
    assert torch.equal(p, L.split_planes_blk(dz, N * H * H, C))


@pytest.mark.parametrize("knob", ["POOL_ARG", "DW_PLANES", "SKIP_HALF"])
def test_training_step_is_the_same_with_the_round_4_form(monkeypatch, knob):
    """MT_XC_POOL_ARG=0 (max-pool adjoint: zero fill + atomics + full-resolution sums), MT_XC_DW_PLANES=0 (depthwise output as fp32 +
    a split pass), MT_XC_SKIP_HALF=0 (skip-path gradient scattered into a zeroed full-size tensor) give the default's gradients."""
    from mintime_amd import xception_engine as XE
    grads = []
    for on in (True, False):
        monkeypatch.setattr(XE, knob, on)
        model, _ = _model(5, True)
        f = model(_input(2, 7).cuda())
        (f * torch.linspace(-1, 1, f.numel(), device="cuda").view_as(f)).sum().backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    for k in grads[0]:
        assert_close(grads[0][k], grads[1][k], 2e-4, k)


@pytest.mark.parametrize("N,H,C,act", [(3, 14, 728, 2), (2, 7, 1536, 0), (5, 28, 256, 2), (1, 19, 264, 2)])
def test_depthwise_output_as_planes(N, H, C, act):
    """mt_dwconv_fwd_planes writes exactly the planes mt_split_planes_blk makes of mt_dwconv_fwd's output, padding rows (N*H*W not
    a multiple of 32) and padding columns (728 = 45.5 sixteen-column blocks) zeroed."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator().manual_seed(C + H)
    zin = torch.randn(N * H * H, C, generator=g).cuda()
    sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).cuda()
    d = torch.empty(N * H * H, C, device="cuda")
    st = L.stream_ptr()
    L.check(lib.mt_dwconv_fwd(L.ptr(zin), L.ptr(sc), L.ptr(sh), L.ptr(w), L.ptr(d), None, 32, N, H, H, C, 3, 1, act, st), "fwd")
    p = L.planes_empty(N * H * H, C, "cuda")
    p.fill_(7.0)
    L.check(lib.mt_dwconv_fwd_planes(L.ptr(zin), L.ptr(sc), L.ptr(sh), L.ptr(w), L.ptr(p), N, H, H, C, 3, 1, act, st), "fwd_planes")
    assert torch.equal(p, L.split_planes_blk(d, N * H * H, C))


def test_side_stream_hold_mode_frees_in_host_order():
    """SideStream(hold=True) (the Xception backward): what a side launch reads is kept alive by the stream object, not by
    record_stream, and is dropped two release points later / at the join -- memory use does not depend on how far the host runs ahead
    of the device (config 5 grew from 127 to 253 GiB reserved in 12 un-synchronised steps with record_stream)."""
    import weakref
    from mintime_amd import lib as L
    side = L.SideStream(torch.device("cuda:0"), hold=True)
    if not side.enabled:
        pytest.skip("MT_SIDE_STREAM=0")
    out = torch.zeros(4, device="cuda")
    t = torch.ones(4, device="cuda")
    w = weakref.ref(t)
    side.launch(lambda: out.add_(t), reads=(t,))
    del t
    side.release_point()
    side.release_point()
    assert w() is not None, "held until the main stream has been ordered behind the side stream's use"
    side.release_point()
    assert w() is None
    u = torch.ones(4, device="cuda")
    wu = weakref.ref(u)
    side.launch(lambda: out.add_(u), reads=(u,))
    del u
    assert wu() is not None
    side.wait()
    assert wu() is None
    torch.cuda.synchronize()
    assert out.tolist() == [2.0] * 4


@pytest.mark.parametrize("N,H,u8", [(3, 224, False), (2, 37, True), (2, 64, False)])
def test_conv1_on_the_stem_kernels_without_padding(N, H, u8):
    """Xception's conv1 = Conv2d(3, 32, 3, 2, 0) (xception.py:135) on the EfficientNet stem's kernels with no padding: output, its
    BatchNorm sums and the weight gradient of dz = ka du + kb z + kc against torch in float64 (odd and even sizes, uint8 crops)."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator().manual_seed(H)
    x = torch.randint(0, 256, (N, H, H, 3), generator=g).to(torch.uint8 if u8 else torch.float32).cuda()
    w = (torch.randn(32, 3, 3, 3, generator=g) * 0.2).cuda()
    Ho = (H - 3) // 2 + 1
    z = torch.full((N, Ho, Ho, 32), 7.0, device="cuda")
    stats = torch.zeros(32, 2, 32, dtype=torch.float64, device="cuda")
    st = L.stream_ptr()
    L.check(lib.mt_stem_conv_fwd_valid(L.ptr(x), int(u8), L.ptr(w), L.ptr(z), L.ptr(stats), 32, N, H, H, st), "fwd")
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=2).permute(0, 2, 3, 1)
    assert_close(z, ref, 1e-5, "conv1 output")
    assert_close(stats.sum(0)[0], ref.sum((0, 1, 2)), 1e-5, "sum z")
    assert_close(stats.sum(0)[1], (ref * ref).sum((0, 1, 2)), 1e-5, "sum z^2")
    du = torch.randn(N, Ho, Ho, 32, generator=g).cuda()
    kabc = torch.randn(3, 32, generator=g).cuda()
    dw = torch.zeros(32, 3, 3, 3, device="cuda")
    L.check(lib.mt_stem_conv_wgrad_valid(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x), int(u8), L.ptr(dw), N, H, H, st), "wgrad")
    dz = kabc[0].double() * du.double() + kabc[1].double() * z.double() + kabc[2].double()
    wd = w.double().requires_grad_(True)
    (torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wd, stride=2) * dz.permute(0, 3, 1, 2)).sum().backward()
    assert_close(dw, wd.grad, 2e-5, "conv1 weight gradient")
