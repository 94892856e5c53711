"""GPU parity of the HIP EfficientNet-B0 against the CPU oracle and the reference-generated fixtures."""
import pytest
import torch

import mintime_amd
from mintime_amd import arch, synth, EfficientNet
from oracle import mintime_oracle as O
from tests.util import GRAD_TOL_UNIT, REL_TOL, assert_close, golden

pytestmark = pytest.mark.gpu


def _model(seed, training):
    m = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=0.0)
    sd = synth.effnet_b0_state(seed)
    m.load_state_dict(sd, strict=True)
    m.train(training)
    return m.cuda(), sd


def _input(n, seed):
    vid = synth.clip_inputs(1, n, 1, seed)["videos"]
    return vid.reshape(n, 224, 224, 3).permute(0, 3, 1, 2)      # NHWC-strided NCHW view (train.py:341)


@pytest.mark.parametrize("name", ["ef_eval", "ef_train"])
def test_features_match_reference_fixture(name):
    g = golden(name)
    n, training, seed = int(g["n_img"]), bool(g["training"]), int(g["seed"])
    model, sd = _model(seed, training)
    x = _input(n, seed)
    with torch.no_grad():
        feats = model(x.cuda())
    assert feats.shape == (n, 1280, 7, 7)
    assert_close(feats, g["features"], REL_TOL, "features vs reference")
    with torch.no_grad():
        ends = model.extract_endpoints(x.cuda()) if not training else None
    if ends is not None:
        assert [tuple(v.shape[1:]) for v in ends.values()] == [(16, 112, 112), (24, 56, 56), (40, 28, 28), (112, 14, 14),
                                                                (320, 7, 7), (1280, 7, 7)]
    if training:
        msd = model.state_dict()
        for k in g.files:
            if k.startswith("stat."):
                assert_close(msd[k[5:]], g[k], REL_TOL, k)
        assert int(msd["_bn0.num_batches_tracked"]) == int(g["nbt"])


@pytest.mark.parametrize("training", [False, True])
def test_block_outputs_match_oracle(training):
    seed, n = 3, 3
    model, sd = _model(seed, training)
    x = _input(n, seed)
    taps = {}
    with torch.no_grad():
        ref = O.effnet_b0_forward(sd, x, training=training, taps=taps)
        from mintime_amd import effnet_engine
        feat, ys = effnet_engine.effnet_apply(model, x.cuda(), want_blocks=True)
    for i, y in enumerate(ys):
        assert_close(y, taps[f"block{i}"], REL_TOL, f"block {i} output")
    assert_close(feat, ref, REL_TOL, "features vs oracle")


@pytest.mark.parametrize("n,training", [(3, True), (9, True), (2, False)])
def test_expand_recompute_matches_stored_expansion(n, training, monkeypatch):
    """Blocks 1-3 with the expand convolution rebuilt inside the depthwise kernels (csrc/rc.hpp: the expanded tensor is never in
    memory; BatchNorm statistics from a statistics-only pass) against the same network with the expanded tensor stored: features,
    every parameter gradient and the running statistics.  n = 9 puts block 1 above the row count at which the expand convolution's
    backward runs as the fused streaming kernel (no z needed); below it the backward rebuilds z with one GEMM -- both are covered.
    The two forms differ by the summation order of a 16- / 24-term dot product (1 ulp of z), carried through 16 blocks of
    train-mode BatchNorm: 1e-4 is 10x below the parity gate."""
    from mintime_amd import effnet_engine as E
    monkeypatch.setattr(E, "EF_RC_MIN_ROWS", 0)
    x = _input(n, 5).cuda()
    gen = torch.Generator().manual_seed(11)
    wts = torch.randn(n, 1280, 7, 7, generator=gen).cuda()
    out = {}
    for rc in (False, True):
        monkeypatch.setattr(E, "EF_RC", rc)
        model, _ = _model(5, training)
        if training:
            feat = model(x)
            (feat * wts).sum().backward()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        else:
            with torch.no_grad():
                feat = model(x)
            grads = {}
        stats = {k: v.detach().clone() for k, v in model.state_dict().items() if "running_" in k}
        out[rc] = (feat.detach().clone(), grads, stats)
    ran = [b for b in model._blocks if b.spec.has_expand and E.L.get().mt_dwconv_rc_supported(b.spec.cin, b.spec.cexp, b.spec.k, b.spec.s, b.spec.hin)]
    assert len(ran) == 3                      # blocks 1, 2, 3
    assert_close(out[True][0], out[False][0], 1e-4, "features, recompute vs stored")
    assert set(out[True][1]) == set(out[False][1])
    for k in out[False][1]:
        ref = out[False][1][k]
        if k.endswith("_bn2.bias") and float(ref.norm()) < 1e-3 * float(out[False][1][k.replace(".bias", ".weight")].norm()):
            continue      # analytically zero (the train-mode BatchNorm behind the next 1x1 conv removes a per-channel shift): rounding noise
        assert_close(out[True][1][k], ref, 5e-4, f"grad {k}, recompute vs stored")      # (worst seen: 2.0e-4, a squeeze-excite bias: a sum that cancels)
    for k in out[False][2]:
        assert_close(out[True][2][k], out[False][2][k], 1e-4, f"{k}, recompute vs stored")


@pytest.mark.parametrize("k,s,H,C,N", [(3, 1, 28, 48, 3), (3, 2, 28, 32, 2), (5, 1, 14, 32, 3), (5, 2, 28, 48, 2), (5, 1, 7, 64, 5),
                                       (3, 1, 7, 24, 3), (3, 1, 112, 32, 1)])
def test_fused_depthwise_backward_equals_the_two_kernels(k, s, H, C, N):
    """mt_dwconv_bwd parts = 3 (data AND weight gradient from one pass over du / z / the depthwise input: the gather leaves the
    activated input of its tile in LDS, a second phase walks it against the dz tile) against parts = 2 + parts = 1 (the two
    stand-alone kernels): du_in bit-identical (same arithmetic), BatchNorm-backward sums and the weight gradient within summation
    order.  Odd tile counts, image edges (H = 7: one partial tile; 28 / 14: 2 x 2 and 1 x 1 tiles of 14) and 8-channel chunks (C = 24)."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator().manual_seed(k * 100 + s * 10 + H)
    Ho = (H + s - 1) // s
    dev = "cuda"
    du = torch.randn(N * Ho * Ho, C, generator=g).to(dev)
    z = torch.randn(N * Ho * Ho, C, generator=g).to(dev)
    zin = torch.randn(N * H * H, C, generator=g).to(dev)
    kabc = (torch.rand(3, C, generator=g) + 0.5).to(dev)
    w = torch.randn(C, 1, k, k, generator=g).to(dev)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    mi = torch.stack([torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5]).to(dev)
    slots = 8

    def run(parts_list):
        du_in = torch.full((N * H * H, C), float("nan"), device=dev)
        stats = torch.zeros(slots, 2, C, dtype=torch.float64, device=dev)
        dw = torch.zeros(C, 1, k, k, device=dev)
        for parts in parts_list:
            L.check(lib.mt_dwconv_bwd(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(w), L.ptr(zin), L.ptr(sc), L.ptr(sh), L.ptr(mi),
                                      L.ptr(du_in), L.ptr(stats), slots, L.ptr(dw), N, H, H, C, k, s, parts, 1, None, None,
                                      L.stream_ptr()), "mt_dwconv_bwd")
        torch.cuda.synchronize()
        return du_in, stats.sum(0), dw

    a_du, a_st, a_dw = run([2, 1])
    b_du, b_st, b_dw = run([3])
    assert torch.equal(a_du, b_du)
    assert_close(b_st, a_st, 1e-5, "BatchNorm-backward sums, fused vs separate")
    assert_close(b_dw, a_dw, 2e-5, "depthwise weight gradient, fused vs separate")
    # and the weight gradient against its definition in float64 (autograd of the same convolution)
    a_in = (zin.double() * sc.double() + sh.double())
    a_in = (a_in * torch.sigmoid(a_in)).view(N, H, H, C).permute(0, 3, 1, 2)
    pad_total = max((Ho - 1) * s + k - H, 0)
    a_pad = torch.nn.functional.pad(a_in, (pad_total // 2, pad_total - pad_total // 2, pad_total // 2, pad_total - pad_total // 2))
    wd = w.double().clone().requires_grad_(True)
    out = torch.nn.functional.conv2d(a_pad, wd, stride=s, groups=C)
    dz = (kabc[0].double() * du.double() + kabc[1].double() * z.double() + kabc[2].double()).view(N, Ho, Ho, C).permute(0, 3, 1, 2)
    out.backward(dz)
    assert_close(b_dw, wd.grad, 2e-4, "depthwise weight gradient vs float64 autograd")


def test_engine_with_fused_depthwise_backward(monkeypatch):
    """MT_DW_FUSED=1 (every depthwise layer's data + weight gradient from one pass) through the whole extractor: every parameter
    gradient against the default two-kernel path."""
    from mintime_amd import effnet_backward as EB
    x = _input(3, 6).cuda()
    wts = torch.randn(3, 1280, 7, 7, generator=torch.Generator().manual_seed(2)).cuda()
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setattr(EB, "FUSED_DW", mode)
        model, _ = _model(6, True)
        (model(x) * wts).sum().backward()
        out[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    for k, ref in out["0"].items():
        if k.endswith("_bn2.bias") and float(ref.norm()) < 1e-3 * float(out["0"][k.replace(".bias", ".weight")].norm()):
            continue
        assert_close(out["1"][k], ref, GRAD_TOL_UNIT, f"grad {k}, fused vs two kernels")


def test_nchw_contiguous_input_is_accepted():
    g = golden("ef_eval")
    model, _ = _model(int(g["seed"]), False)
    x = _input(int(g["n_img"]), int(g["seed"])).contiguous()
    with torch.no_grad():
        feats = model(x.cuda())
    assert_close(feats, g["features"], REL_TOL, "NCHW-contiguous input")


@pytest.mark.parametrize("n,H,u8", [(2, 224, False), (3, 224, True), (2, 225, False), (1, 96, True)])
def test_direct_stem_kernel_agrees_with_the_im2col_gemm_path(n, H, u8):
    """mt_stem_conv_fwd (streaming MFMA kernel, what the engine runs) and the im2col-prologue GEMM compute the same stem: values
    against fp64 torch (TF-SAME padding: (0,1) at 224, (1,1) at 225), BatchNorm sums against each other; fp32 and uint8 crops."""
    from mintime_amd import lib as L
    lib = L.get()
    Ho = (H + 1) // 2
    pad = max((Ho - 1) * 2 + 3 - H, 0)
    x8 = torch.randint(0, 256, (n, H, H, 3), dtype=torch.uint8)
    x = (x8 if u8 else x8.float()).cuda()
    w = (torch.randn(32, 3, 3, 3) * 0.01).cuda()
    z_direct = torch.full((n * Ho * Ho, 32), float("nan"), device="cuda")
    st_direct = torch.zeros(4, 2, 32, dtype=torch.float64, device="cuda")
    L.check(lib.mt_stem_conv_fwd(L.ptr(x), 1 if u8 else 0, L.ptr(w), L.ptr(z_direct), L.ptr(st_direct), 4, n, H, H, L.stream_ptr()), "stem")
    wp = torch.empty(32, 28, device="cuda")
    L.check(lib.mt_conv_weight_pack(L.ptr(w), L.ptr(wp), 32, 3, 3, 28, 0, L.stream_ptr()), "pack")
    z_gemm = torch.empty_like(z_direct)
    st_gemm = torch.zeros(4, 2, 32, dtype=torch.float64, device="cuda")
    L.gemm(L.OP_NT, x, wp, z_gemm, n * Ho * Ho, 32, 28, 28, 28, 32, prologue=L.PRO_IM2COL, epilogue=L.EPI_STATS, stats=st_gemm,
           stats_slots=4, conv=(H, H, 3, Ho, Ho, 3, 2, pad // 2, 0, 1 if u8 else 0))
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x8.permute(0, 3, 1, 2).double(), [pad // 2, pad - pad // 2] * 2),
                                     w.cpu().double(), None, 2).permute(0, 2, 3, 1).reshape(-1, 32)
    assert_close(z_direct, ref, 2e-5, "direct stem")
    assert_close(z_gemm, ref, 2e-5, "im2col GEMM stem")
    assert_close(st_direct.sum(0), st_gemm.sum(0), 1e-6, "BatchNorm sums")


def test_uint8_crops_are_ingested_directly():
    """Next-row f2: uint8 BGR crops (what the dataset holds before its .float(), deepfakes_dataset.py:257,339) go straight into the
    stem's gather.  Judged against the ORACLE run on x8.float() (features and every parameter gradient), eval and train mode."""
    x8 = torch.randint(0, 256, (3, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(9))
    gw = torch.randn(3, 1280, 7, 7, generator=torch.Generator().manual_seed(10)) * 0.1
    for training in (False, True):
        model, sd = _model(2, training)
        f = model(x8.permute(0, 3, 1, 2).cuda())
        (f * gw.cuda()).sum().backward()
        osd = {k: (v.double().requires_grad_("running_" not in k) if v.is_floating_point() else v) for k, v in sd.items()}
        fo = O.effnet_b0_forward(osd, x8.double().permute(0, 3, 1, 2), training=training)
        (fo * gw.double()).sum().backward()
        assert_close(f, fo, REL_TOL, f"uint8 features vs oracle (training={training})")
        _assert_all_grads(model, osd, training, f"uint8 (training={training})")


def _assert_all_grads(model, osd, training, what):
    """Every EfficientNet parameter gradient vs the oracle's.  In train mode d/d(_bn2.bias) is analytically zero (a per-channel
    shift of a block output is removed by the train-mode BatchNorm behind the next 1x1 conv): both sides hold rounding noise."""
    named = dict(model.named_parameters())
    for k, p in named.items():
        if k.startswith("_fc"):
            continue
        ref = osd[k].grad
        wn = float(named[k.replace(".bias", ".weight")].grad.norm()) if k.endswith("_bn2.bias") else 0.0
        if training and k.endswith("_bn2.bias") and float(ref.norm()) < 1e-3 * wn:      # (not zero behind a drop-connect gate)
            assert float(p.grad.norm()) < 1e-3 * wn, k
            continue
        assert_close(p.grad, ref, GRAD_TOL_UNIT, f"{what}: grad {k}")


def _dc_model(g):
    n, seed, rate = int(g["n_img"]), int(g["seed"]), float(g["rate"])
    model = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=rate)
    sd = synth.effnet_b0_state(seed)
    model.load_state_dict(sd, strict=True)
    model.train(True)
    u = O.drop_connect_uniforms(seed, n, rate)          # the reference's own draws (torch.manual_seed(seed), block order)
    model.drop_connect_uniform = lambda rows, N, dev: torch.stack([u[i].reshape(N) for i in sorted(u)]).to(dev)
    return model.cuda(), sd, u, n, seed, rate


def test_drop_connect_matches_reference_fixture():
    """a5: train mode WITH drop-connect 0.2 (what bench.py runs).  The per-sample gate floor(keep + U)/keep (utils.py:148-153)
    is fed the reference's own uniform draws; features, gated block outputs and sampled gradients vs the reference fixture,
    every parameter gradient vs the oracle in fp64."""
    import numpy as np
    g = golden("ef_train_dc")
    model, sd, u, n, seed, rate = _dc_model(g)
    x = _input(n, seed)
    gw = torch.from_numpy(np.random.Generator(np.random.Philox(key=[seed, 777])).standard_normal((n, 1280, 7, 7)) * 0.1).float()
    from mintime_amd import effnet_engine
    feats, ys = effnet_engine.effnet_apply(model, x.cuda(), want_blocks=True)
    (feats * gw.cuda()).sum().backward()
    assert_close(feats, g["features"], REL_TOL, "features vs reference (drop-connect 0.2)")
    for i in (2, 7, 10, 14):
        assert_close(ys[i][:, :, :3, :3], g[f"block{i}_slice"], REL_TOL, f"block {i} output")
    named = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("gnorm."):
            key = k[len("gnorm."):]
            assert_close(named[key].grad.norm(), g[k], GRAD_TOL_UNIT, k)
            assert_close(named[key].grad.reshape(-1)[:256], g["gslice." + key], GRAD_TOL_UNIT, "gslice." + key)
    # a dropped sample's block output is exactly its block input (the gate is exactly 0)
    keep14 = 1 - rate * 14 / 16
    dropped = (torch.floor(keep14 + u[14].reshape(-1)) == 0).nonzero().reshape(-1).tolist()
    for smp in dropped:
        assert torch.equal(ys[14][smp], ys[13][smp])
    osd = {k: (v.double().requires_grad_("running_" not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    fo = O.effnet_b0_forward(osd, x.double(), training=True, drop_connect_rate=rate, dc_uniform=u)
    (fo * gw.double()).sum().backward()
    _assert_all_grads(model, osd, True, "drop-connect 0.2")


def test_drop_connect_default_sampler_statistics():
    """Without an injected sampler the gate comes from torch's device RNG: values are exactly {0, 1/keep} and eval ignores it."""
    from mintime_amd import effnet_engine
    model = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=0.2)
    model.load_state_dict(synth.effnet_b0_state(0), strict=True)
    model = model.cuda().train(True)
    x = _input(6, 4).cuda()
    torch.manual_seed(0)
    feat, saved, _ = effnet_engine.effnet_forward(model, x.permute(0, 2, 3, 1).contiguous(), effnet_engine.param_list(model), True, True)
    for bi, rec in enumerate(saved["blocks"]):
        dc = rec["dc"]
        if bi in (2, 4, 6, 7, 9, 10, 12, 13, 14):
            keep = 1 - 0.2 * bi / 16
            vals = set(round(float(v), 5) for v in dc.cpu())
            assert vals <= {0.0, round(1 / keep, 5)}, (bi, vals)
        else:
            assert dc is None
    model.eval()
    with torch.no_grad():
        a, b = model(x), model(x)
    assert torch.equal(a, b)


@pytest.mark.parametrize("cout,cin,rows,gated", [(96, 16, 100003, False), (16, 32, 70001, True), (24, 96, 50000, True),
                                                 (144, 24, 33333, False), (40, 144, 20001, True), (240, 40, 9999, False),
                                                 (40, 240, 10000, True), (8, 8, 17, False)])
def test_skinny_conv1x1_wgrad_kernel(cout, cin, rows, gated):
    """mt_conv1x1_wgrad (accumulator-resident weight gradient for many-row 1x1 convs) against fp64 torch."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator(device="cuda").manual_seed(cout * 1000 + cin)
    hw = 49
    du = torch.randn(rows, cout, device="cuda", generator=g)
    z = torch.randn(rows, cout, device="cuda", generator=g)
    kabc = torch.randn(3, cout, device="cuda", generator=g)
    x = torch.randn(rows, cin, device="cuda", generator=g)
    sc, sh = torch.randn(cin, device="cuda", generator=g), torch.randn(cin, device="cuda", generator=g)
    n_img = (rows + hw - 1) // hw
    gate = torch.rand(n_img, cin, device="cuda", generator=g)
    dw = torch.full((cout, cin), 0.5, device="cuda")
    assert lib.mt_conv1x1_wgrad_supported(cout, cin) == 1
    L.check(lib.mt_conv1x1_wgrad(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x), L.ptr(sc) if gated else None, L.ptr(sh) if gated else None,
                                 L.ptr(gate) if gated else None, hw, L.ptr(dw), rows, cout, cin, L.stream_ptr()), "mt_conv1x1_wgrad")
    dz = (kabc[0] * du + kabc[1] * z + kabc[2]).double()
    a = x.double()
    if gated:
        a = torch.nn.functional.silu(sc.double() * a + sh.double()) * gate.double().repeat_interleave(hw, 0)[:rows]
    want = dz.t() @ a + 0.5
    assert_close(dw, want.float(), 1e-4, f"dW {cout}x{cin}")


@pytest.mark.parametrize("cout,cin,rows,hw", [(192, 1152, 12544, 49), (80, 240, 50176 // 4, 196), (112, 672, 9973, 196),
                                              (320, 1152, 4999, 49), (192, 672, 31, 49), (112, 480, 50176 // 2, 196)])
def test_wide_project_conv_wgrad_kernel(cout, cin, rows, hw):
    """mt_conv1x1_wgrad_wide (late-stage project-conv weight gradient: 128-column slabs, producer / consumer wavefronts) against
    fp64 torch; partial last slab (240, 672, 480 columns), ragged last chunk, rows straddling images, fewer chunks than row ranges."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator(device="cuda").manual_seed(cout * 1000 + cin)
    du = torch.randn(rows, cout, device="cuda", generator=g)
    z = torch.randn(rows, cout, device="cuda", generator=g)
    kabc = torch.randn(3, cout, device="cuda", generator=g)
    x = torch.randn(rows, cin, device="cuda", generator=g)
    sc, sh = torch.randn(cin, device="cuda", generator=g), torch.randn(cin, device="cuda", generator=g)
    n_img = (rows + hw - 1) // hw
    gate = torch.rand(n_img, cin, device="cuda", generator=g)
    dw = torch.full((cout, cin), 0.5, device="cuda")
    assert lib.mt_conv1x1_wgrad_wide_supported(cout, cin) == 1
    assert lib.mt_conv1x1_wgrad_wide_supported(40, 240) == 0 and lib.mt_conv1x1_wgrad_wide_supported(192, 96) == 0
    L.check(lib.mt_conv1x1_wgrad_wide(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(gate), hw, L.ptr(dw), rows,
                                      cout, cin, L.stream_ptr()), "mt_conv1x1_wgrad_wide")
    dz = (kabc[0] * du + kabc[1] * z + kabc[2]).double()
    a = torch.nn.functional.silu(sc.double() * x.double() + sh.double()) * gate.double().repeat_interleave(hw, 0)[:rows]
    want = dz.t() @ a + 0.5
    assert_close(dw, want.float(), 1e-4, f"dW {cout}x{cin}")


@pytest.mark.parametrize("cin,cout,rows,mode", [(16, 96, 100003, 0), (32, 16, 70001, 1), (96, 24, 50000, 1), (24, 144, 10007, 0),
                                                (96, 16, 40001, 2), (16, 32, 30000, 2), (24, 96, 30001, 2), (24, 144, 20000, 2)])
def test_streaming_conv1x1_kernel(cin, cout, rows, mode):
    """mt_conv1x1_rows (forward / data gradient of many-row 1x1 convs as a streaming kernel) against fp64 torch: every operand
    transform, the residual input, the transposed-weight form and the BatchNorm statistics of the output."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + cout + mode)
    hw = 49
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, x2 = r(rows, cin), r(rows, cin)
    c0, c1 = r(cin), r(cin)
    n_img = (rows + hw - 1) // hw
    c2 = torch.rand(n_img, cin, device="cuda", generator=g) if mode == 1 else r(cin)
    transposed = mode == 2
    w = r(cin, cout) if transposed else r(cout, cin)            # data gradient: the forward weight [k, out] is used transposed
    res = r(rows, cout) if mode == 2 else None
    out = torch.empty(rows, cout, device="cuda")
    stats = torch.zeros(32, 2, cout, dtype=torch.float64, device="cuda") if mode != 2 else None
    L.check(lib.mt_conv1x1_rows(L.ptr(x), L.ptr(x2) if mode == 2 else None, L.ptr(w), w.shape[1], 1 if transposed else 0,
                                L.ptr(c0) if mode else None, L.ptr(c1) if mode else None, L.ptr(c2) if mode else None, hw, mode,
                                L.ptr(res), L.ptr(out), L.ptr(stats), 32, rows, cin, cout, L.stream_ptr()), "mt_conv1x1_rows")
    a = x.double()
    if mode == 1:
        a = torch.nn.functional.silu(c0.double() * a + c1.double()) * c2.double().repeat_interleave(hw, 0)[:rows]
    elif mode == 2:
        a = c0.double() * a + c1.double() * x2.double() + c2.double()
    want = a @ (w.double() if transposed else w.double().t())
    if res is not None:
        want = want + res.double()
    assert_close(out, want.float(), 1e-5, f"out {cin}->{cout} mode {mode}")
    if stats is not None:
        s = stats.sum(0)
        assert_close(s[0].float(), want.sum(0).float(), 1e-4, "column sums")
        assert_close(s[1].float(), (want * want).sum(0).float(), 1e-4, "column sums of squares")


@pytest.mark.parametrize("rows,cout,cin,with_res", [(100000 + 37, 96, 16, False), (200704, 144, 24, True), (4096 + 5, 96, 24, True), (130, 96, 16, True),
                                                    (64 * 256 * 3, 144, 24, False)])
def test_fused_expand_conv_backward_matches_fp64(rows, cout, cin, with_res):
    """mt_conv1x1_bwd_fused: dx = (ka*du + kb*z + kc) . W (+ res) and dW += (ka*du + kb*z + kc)^T . x in one streaming pass
    (autograd of efficientnet_pytorch/model.py:96-99 through _bn0), against float64 on the host; ragged last chunk included.
    z = x . W^T is not an argument: the kernel folds it into W^T diag(kb) W and x^T x."""
    from mintime_amd import lib as L
    g = torch.Generator().manual_seed(5)
    du, x = torch.randn(rows, cout, generator=g), torch.randn(rows, cin, generator=g)
    W = torch.randn(cout, cin, generator=g) * 0.2
    z = x.double() @ W.double().T
    kabc = torch.stack([torch.randn(cout, generator=g), torch.randn(cout, generator=g) * 0.3, torch.randn(cout, generator=g) * 0.1])
    res = torch.randn(rows, cin, generator=g) if with_res else None
    d = {k: v.cuda() for k, v in dict(du=du, x=x, W=W, kabc=kabc).items()}
    res_d = res.cuda() if with_res else None
    dx = torch.full((rows, cin), float("nan"), device="cuda")
    dW = torch.zeros(cout, cin, device="cuda")
    assert L.get().mt_conv1x1_bwd_fused_supported(cout, cin) == 1
    L.check(L.get().mt_conv1x1_bwd_fused(L.ptr(d["du"]), L.ptr(d["kabc"]), L.ptr(d["x"]), L.ptr(d["W"]), L.ptr(res_d),
                                         L.ptr(dx), L.ptr(dW), rows, cout, cin, L.stream_ptr()), "mt_conv1x1_bwd_fused")
    dz = kabc[0].double() * du.double() + kabc[1].double() * z + kabc[2].double()
    ref_dx = dz @ W.double() + (res.double() if with_res else 0)
    assert_close(dx, ref_dx, 2e-5, "fused data gradient")
    assert_close(dW, dz.T @ x.double(), 5e-5, "fused weight gradient")
    assert L.get().mt_conv1x1_bwd_fused_supported(240, 40) == 0 and L.get().mt_conv1x1_bwd_fused_supported(96, 40) == 0


@pytest.mark.parametrize("n_img,hw,co,c", [(9, 3136, 24, 96), (3, 12544, 16, 32), (5, 3136, 24, 144)])
def test_streaming_se_stage_matches_fp64(n_img, hw, co, c):
    """mt_se_stage_fused: the project conv's data gradient rebuilt from the narrow gradient inside the squeeze-excite reduction
    (mode 0) and inside the activation / BatchNorm backward (mode 1), against float64 on the host."""
    from mintime_amd import lib as L
    g = torch.Generator().manual_seed(8)
    rows = n_img * hw
    du, zp = torch.randn(rows, co, generator=g), torch.randn(rows, co, generator=g)
    kabc = torch.stack([torch.randn(co, generator=g), torch.randn(co, generator=g) * 0.3, torch.randn(co, generator=g) * 0.1])
    W = torch.randn(co, c, generator=g) * 0.2
    zd = torch.randn(rows, c, generator=g)
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    gate, dpool = torch.rand(n_img, c, generator=g), torch.randn(n_img, c, generator=g)
    mi = torch.stack([torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5])
    d = {k: v.cuda() for k, v in dict(du=du, zp=zp, kabc=kabc, W=W, zd=zd, scale=scale, shift=shift, gate=gate, dpool=dpool, mi=mi).items()}
    lib = L.get()
    assert lib.mt_se_stage_fused_supported(co, c, hw) == 1 and lib.mt_se_stage_fused_supported(40, 240, 784) == 0
    dgate = torch.zeros(n_img, c, device="cuda")
    L.check(lib.mt_se_stage_fused(L.ptr(d["du"]), L.ptr(d["zp"]), L.ptr(d["kabc"]), L.ptr(d["W"]), L.ptr(d["zd"]), L.ptr(d["scale"]),
                                  L.ptr(d["shift"]), 0, L.ptr(dgate), None, None, None, None, None, 1, rows, co, c, hw, L.stream_ptr()), "red")
    slots = 8
    dud = torch.full((rows, c), float("nan"), device="cuda")
    stats = torch.zeros(slots, 2, c, dtype=torch.float64, device="cuda")
    L.check(lib.mt_se_stage_fused(L.ptr(d["du"]), L.ptr(d["zp"]), L.ptr(d["kabc"]), L.ptr(d["W"]), L.ptr(d["zd"]), L.ptr(d["scale"]),
                                  L.ptr(d["shift"]), 1, None, L.ptr(d["gate"]), L.ptr(d["dpool"]), L.ptr(d["mi"]), L.ptr(dud), L.ptr(stats),
                                  slots, rows, co, c, hw, L.stream_ptr()), "act")
    dz = kabc[0].double() * du.double() + kabc[1].double() * zp.double() + kabc[2].double()
    da = dz @ W.double()
    u = zd.double() * scale.double() + shift.double()
    sg = torch.sigmoid(u)
    ref_dgate = (da * u * sg).reshape(n_img, hw, c).sum(1)
    assert_close(dgate, ref_dgate, 5e-5, "d gate")
    ref_du = (da * gate.double().repeat_interleave(hw, 0) + dpool.double().repeat_interleave(hw, 0) / hw) * (sg * (1 + u * (1 - sg)))
    assert_close(dud, ref_du, 2e-5, "du_d")
    st = stats.sum(0).cpu()
    assert_close(st[0], ref_du.sum(0), 1e-4, "sum du")
    assert_close(st[1], (ref_du * (zd.double() - mi[0].double()) * mi[1].double()).sum(0), 1e-4, "sum du * xhat")


@pytest.mark.parametrize("rows,C,act,with_res,with_gate", [(1000, 80, 0, True, True), (3000, 112, 0, False, False), (777, 320, 1, True, False),
                                                            (100, 40, 0, True, True), (640, 1280, 1, False, False), (49 * 8, 192, 0, True, True)])
def test_block_output_as_fp32_and_planes(rows, C, act, with_res, with_gate):
    """mt_bn_act_fwd_planes (late-stage block output: model.py:117-127 + the operand of the next expand convolution): y is mt_bn_act_fwd's
    y to the bit, the planes are the exact three-piece split of y, padding rows / columns are zeros."""
    from mintime_amd import lib as L
    lib = L.get()
    g = torch.Generator().manual_seed(rows + C)
    z = (torch.randn(rows, C, generator=g) * 2).cuda()
    sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    res = torch.randn(rows, C, generator=g).cuda() if with_res else None
    hw = 49
    gate = (torch.floor(0.8 + torch.rand((rows + hw - 1) // hw, generator=g)) / 0.8).cuda() if with_gate else None
    y_ref = torch.empty(rows, C, device="cuda")
    L.check(lib.mt_bn_act_fwd(L.ptr(z), L.ptr(sc), L.ptr(sh), L.ptr(res), L.ptr(y_ref), rows, C, act, L.ptr(gate), hw, L.stream_ptr()), "bn_act")
    y = torch.full((rows, C), float("nan"), device="cuda")
    y_p = L.planes_empty(rows, C, "cuda")
    y_p.fill_(float("nan"))
    L.check(lib.mt_bn_act_fwd_planes(L.ptr(z), L.ptr(sc), L.ptr(sh), L.ptr(res), L.ptr(y), rows, C, act, L.ptr(gate), hw, L.ptr(y_p),
                                     L.stream_ptr()), "bn_act_planes")
    assert torch.equal(y, y_ref)
    assert not torch.isnan(y_p.float()).any()
    full = L.planes_to_float(y_p, y_p.shape[1] * 32, y_p.shape[2] * 16)
    assert torch.equal(full[:rows, :C], y_ref)
    assert float(full[rows:].abs().sum()) == 0.0 and float(full[:, C:].abs().sum()) == 0.0
    zd = z.double() * sc.double() + sh.double()
    want = zd * torch.sigmoid(zd) if act == 1 else zd
    if gate is not None:
        want = want * gate.double()[torch.arange(rows, device="cuda") // hw][:, None]
    if res is not None:
        want = want + res.double()
    assert_close(y, want, 1e-5, "block output vs float64")


@pytest.mark.parametrize("n_img,hw,C,Co", [(16, 49, 1152, 192), (8, 196, 672, 112), (3, 49, 480, 80)])
def test_project_operand_planes_and_the_gemm_on_them(n_img, hw, C, Co):
    """mt_bn_swish_gate_planes: swish(bn1(z_d)) * gate (model.py:104-116) as planes, and the project convolution on them
    (mt_gemm_planes with BatchNorm statistics) against mt_gemm's operand prologue and against float64."""
    from mintime_amd import lib as L
    from mintime_amd.effnet_engine import SLOTS
    lib = L.get()
    rows = n_img * hw
    g = torch.Generator().manual_seed(C + hw)
    z = (torch.randn(rows, C, generator=g) * 2).cuda()
    sc, sh = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    gate = torch.rand(n_img, C, generator=g).cuda()
    w = (torch.randn(Co, C, generator=g) * 0.05).cuda()
    a_p = L.planes_empty(rows, C, "cuda")
    a_p.fill_(float("nan"))
    L.check(lib.mt_bn_swish_gate_planes(L.ptr(z), L.ptr(sc), L.ptr(sh), L.ptr(gate), hw, L.ptr(a_p), rows, C, L.stream_ptr()), "a planes")
    zd = z.double() * sc.double() + sh.double()
    a64 = zd * torch.sigmoid(zd) * gate.double().repeat_interleave(hw, 0)
    a = L.planes_to_float(a_p, a_p.shape[1] * 32, C)
    assert_close(a[:rows], a64, 2e-6, "operand planes vs float64")
    assert float(a[rows:].abs().sum()) == 0.0
    out = torch.empty(rows, Co, device="cuda")
    st = torch.zeros(SLOTS * 2 * Co, dtype=torch.float64, device="cuda")
    L.gemm_planes(L.OP_NT, a_p, L.split_planes_blk(w), rows, Co, C, Cout=out, ldc=Co, epilogue=L.EPI_STATS, stats=st, stats_slots=SLOTS)
    want = a64 @ w.double().T
    assert_close(out, want, 2e-5, "project convolution on plane operands vs float64")
    sums = st.view(SLOTS, 2, Co).sum(0)
    assert_close(sums[0], want.sum(0), 1e-4, "BatchNorm sum")
    assert_close(sums[1], (want * want).sum(0), 1e-4, "BatchNorm sum of squares")
    ref = torch.empty(rows, Co, device="cuda")
    L.gemm(L.OP_NT, z, w, ref, rows, Co, C, C, C, Co, prologue=L.PRO_BN_SWISH_GATE, scale=sc, shift=sh, gate=gate, hw=hw)
    assert_close(out, ref, 2e-5, "plane operands vs the operand-prologue GEMM")
