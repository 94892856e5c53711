"""Deterministic mode (reference train.py:110 asks cuDNN for deterministic kernels; here MT_DETERMINISTIC=1 / lib.set_deterministic):
with the switch on no floating-point atomic is issued in the training step, so the same state and inputs give BIT-identical
gradients, running statistics and loss run after run -- and the same values as the default (atomic) path up to rounding."""
import pytest
import torch

import mintime_amd
from mintime_amd import arch, synth, EfficientNet, SizeInvariantTimeSformer
from mintime_amd import lib as L
from tests.util import REL_TOL, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture
def det_mode():
    prev = L.set_deterministic(True)
    yield
    L.set_deterministic(prev)


def _models(seed, frames, rate=0.2):
    cfg = arch.default_tsf_config(1280, frames)
    ef = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=rate)
    ef.load_state_dict(synth.effnet_b0_state(seed))
    ef.train(True).cuda()
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=False)
    tsf.load_state_dict(synth.tsf_state(cfg, seed))
    tsf.train(True).cuda()
    return cfg, ef, tsf


def _grads(ef, tsf, inp, dc_seed=11):
    """One forward + BCE + backward from the models' current state (train.py:332-378); returns loss, gradients and BN running stats."""
    for p in list(ef.parameters()) + list(tsf.parameters()):
        p.grad = None
    g = torch.Generator(device="cuda").manual_seed(dc_seed)
    ef.drop_connect_uniform = lambda rows, N, dev: torch.rand(rows, N, device=dev, generator=g)
    videos = inp["videos"]
    b, f, h, w, c = videos.shape
    x = videos.reshape(b * f, h, w, c).permute(0, 3, 1, 2).cuda()
    feats = ef(x)
    feats = feats.reshape(b, f, *feats.shape[1:])
    y = tsf(feats, mask=inp["mask"].cuda(), size_embedding=inp["size_embedding"], identities_mask=inp["identities_mask"].cuda(),
            positions=inp["positions"].cuda())
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y, inp["labels"].reshape(-1, 1).cuda())
    loss.backward()
    torch.cuda.synchronize()
    out = {"loss": loss.detach().clone(), "logits": y.detach().clone()}
    for tag, m in (("ef.", ef), ("tsf.", tsf)):
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[tag + k] = p.grad.detach().clone()
        for k, bbuf in m.named_buffers():
            if "running_" in k:
                out[tag + k] = bbuf.detach().clone()
    return out


def _snapshot(*models):
    return [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in models]


def _restore(snaps, *models):
    for s, m in zip(snaps, models):
        m.load_state_dict(s)


@pytest.mark.parametrize("B,ids,ragged", [(1, 1, False), (2, 2, True), (32, 2, False)])
def test_two_training_steps_from_the_same_state_are_bit_identical(det_mode, B, ids, ragged):
    """One clip (393 token rows: the in-kernel-split TimeSformer path), a small ragged batch (short grids, masked frames) and BASELINE config 3 at full size (B = 32: 256 crops, 12 576 token rows --
    every split-K slab count, log rank count and streaming-kernel replacement of the benchmarked step)."""
    Fr, seed = 8, 5
    cfg, ef, tsf = _models(seed, Fr)
    inp = synth.clip_inputs(B, Fr, ids, seed, ragged=ragged)
    snaps = _snapshot(ef, tsf)
    a = _grads(ef, tsf, inp)
    _restore(snaps, ef, tsf)
    b = _grads(ef, tsf, inp)
    assert set(a) == set(b) and len(a) > 300
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, f"{len(diff)} of {len(a)} tensors differ between two deterministic runs, e.g. {diff[:5]}"
    _restore(snaps, ef, tsf)
    # ... and they are the default path's values up to rounding
    L.set_deterministic(False)
    c = _grads(ef, tsf, inp)
    L.set_deterministic(True)
    assert_close(a["logits"], c["logits"], 1e-4, "logits, deterministic vs default")
    worst = 0.0
    for k in a:
        if k.endswith("_bn2.bias") and float(c[k].norm()) < 1e-3 * float(c[k.replace(".bias", ".weight")].norm()):
            continue                                      # analytically zero gradients (rounding noise either way)
        if float(c[k].abs().max()) == 0.0:
            assert float(a[k].abs().max()) == 0.0, k
            continue
        worst = max(worst, assert_close(a[k], c[k], REL_TOL, "deterministic vs default: " + k))
    print(f"B={B}: {len(a)} tensors bit-identical across runs; worst relative difference to the default path {worst:.2e}")


def test_deterministic_bn_sums_match_fp64(det_mode):
    lib = L.get()
    torch.manual_seed(0)
    for rows, C in ((5000, 16), (4097, 40), (3000, 1152)):
        x = torch.randn(rows, C, device="cuda") * 3 + 1
        z = torch.randn(rows, C, device="cuda")
        mi = torch.stack([z.mean(0), 1.0 / z.std(0)]).contiguous()
        for mode in (0, 1):
            stats = torch.zeros(2, C, dtype=torch.float64, device="cuda")
            L.check(lib.mt_det_bn_sums(L.ptr(x), L.ptr(z), L.ptr(mi), rows, C, mode, L.ptr(stats), L.stream_ptr()), "mt_det_bn_sums")
            xd = x.double()
            s1 = xd.sum(0)
            s2 = (xd * xd).sum(0) if mode == 0 else (x * ((z - mi[0]) * mi[1])).double().sum(0)
            assert_close(stats[0], s1, 1e-9, f"s1 rows={rows} C={C} mode={mode}")
            assert_close(stats[1], s2, 1e-6, f"s2 rows={rows} C={C} mode={mode}")
            again = torch.zeros_like(stats)
            L.check(lib.mt_det_bn_sums(L.ptr(x), L.ptr(z), L.ptr(mi), rows, C, mode, L.ptr(again), L.stream_ptr()), "mt_det_bn_sums")
            assert torch.equal(stats, again)


@pytest.mark.parametrize("M,N,K", [(512, 2048, 12576), (96, 16, 200704), (1536, 512, 6288)])
def test_split_k_weight_gradient_is_reproducible_and_right(det_mode, M, N, K):
    """dW = dY^T X through every GEMM family's split-K path: slabs in split order instead of fp32 atomics."""
    torch.manual_seed(1)
    a = torch.randn(K, M, device="cuda")
    b = torch.randn(K, N, device="cuda")
    ref = (a.double().t() @ b.double())
    outs = []
    for _ in range(2):
        c = torch.zeros(M, N, device="cuda")
        L.gemm(L.OP_TN, a, b, c, M, N, K, M, N, N, epilogue=L.EPI_ATOMIC, split_k=0)
        outs.append(c)
    assert torch.equal(outs[0], outs[1])
    assert_close(outs[0], ref, 1e-5, "split-K TN, deterministic")
    if M % 16 == 0 and N % 16 == 0:
        ap, bp = L.split_planes_blk(a, K, M), L.split_planes_blk(b, K, N)
        outs = []
        for _ in range(2):
            c = torch.zeros(M, N, device="cuda")
            L.gemm_planes(L.OP_TN, ap, bp, M, N, K, Cout=c, ldc=N, epilogue=L.EPI_ATOMIC)
            outs.append(c)
        assert torch.equal(outs[0], outs[1])
        assert_close(outs[0], ref, 1e-5, "plane-operand TN, deterministic")


def test_xception_training_step_is_bit_identical(det_mode):
    """BASELINE config 5's extractor (2 clips x 16 slots x 3 identities, train-mode BatchNorm): the max-pool adjoint runs as a gather,
    BatchNorm sums in fixed order, every weight gradient through split-K slabs."""
    from mintime_amd import xception
    B, F, seed = 2, 16, 6
    cfg = arch.default_tsf_config(2048, F)
    xc = xception(num_classes=1, pretrain_path=None)
    xc.load_state_dict(synth.xception_state(seed))
    xc.cuda().train()
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=False)
    tsf.load_state_dict(synth.tsf_state(cfg, seed))
    tsf.cuda().train()
    inp = synth.clip_inputs(B, F, 3, seed, ragged=False)
    snaps = _snapshot(xc, tsf)
    a = _grads(xc, tsf, inp)
    _restore(snaps, xc, tsf)
    b = _grads(xc, tsf, inp)
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert len(a) > 250 and not diff, f"{len(diff)} of {len(a)} tensors differ between two deterministic runs, e.g. {diff[:5]}"
    _restore(snaps, xc, tsf)
    L.set_deterministic(False)
    c = _grads(xc, tsf, inp)
    L.set_deterministic(True)
    assert_close(a["logits"], c["logits"], REL_TOL, "logits, deterministic vs default")
    # (ReLU / max-pool masks flip at rounding level between any two summation orders: relative L2, like tests/test_gpu_xception.py)
    for k in a:
        if float(c[k].norm()) == 0.0:
            continue
        e = float((a[k].double() - c[k].double()).norm() / c[k].double().norm())
        assert e <= 2e-2, f"{k}: deterministic vs default relative L2 {e:.2e}"


def test_training_trajectory_is_reproducible(det_mode):
    """Three optimizer steps (harness.train_step: forward, BCE, backward, fused SGD -- train.py:332-378) run twice from the same
    state and RNG seed end in bit-identical weights, BatchNorm statistics and losses."""
    from mintime_amd import harness

    def run():
        torch.manual_seed(123)                                  # drop-connect draws (torch.rand inside the extractor's forward)
        cfg, ef, tsf = harness.build_models(8, seed=2, device="cuda")
        opt = harness.make_optimizer(cfg, ef, tsf)
        losses = []
        for s in range(3):
            batch = harness.device_batch(4, 8, 2, seed=s, device="cuda", ragged=(s == 1))
            losses.append(harness.train_step(ef, tsf, opt, batch).detach().clone())
        torch.cuda.synchronize()
        state = {"ef." + k: v.detach().clone() for k, v in ef.state_dict().items()}
        state.update({"tsf." + k: v.detach().clone() for k, v in tsf.state_dict().items()})
        return losses, state

    l1, s1 = run()
    l2, s2 = run()
    assert all(torch.equal(a, b) for a, b in zip(l1, l2)), (l1, l2)
    diff = [k for k in s1 if not torch.equal(s1[k], s2[k])]
    assert not diff, f"{len(diff)} of {len(s1)} state tensors differ after 3 deterministic steps, e.g. {diff[:5]}"
    assert float(l1[0]) != float(l1[2])                         # (the steps did train)
