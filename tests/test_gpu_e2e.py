"""GPU parity of the whole hot path (EfficientNet-B0 -> SizeInvariantTimeSformer, forward + backward) driven exactly
like the reference call sites (train.py:332-378, test.py:235-247)."""
import os

import pytest
import torch

import mintime_amd
from mintime_amd import arch, synth, EfficientNet, SizeInvariantTimeSformer
from oracle import mintime_oracle as O
from tests.util import REL_TOL, assert_close, golden, rel_err

# Gradient gates.  north_star states 1e-3 for OUTPUTS (REL_TOL: logits, features, loss, attentions' per-tensor gate); gradients had
# been gated at 2-3x that.  Round 6 measured every comparison of this file on an MI355X (MT_TEST_VERBOSE=1, gpurun_out/r06_e2e_verbose.log)
# and set each gate to ~2x the worst case seen (DESIGN.md section 4 lists the measurements):
GRAD_TOL_EF = 6e-4          # EfficientNet parameter gradients, every size: worst 2.6e-4 (config 2, live fp64 oracle), 3.0e-4 (config 3 fixture)
GRAD_TOL_TSF = 4e-4         # TimeSformer gradients behind the EfficientNet: worst 1.9e-4
GRAD_TOL_TSF_XS = 2e-3      # ... behind the train-mode Xception (config 5, ReLU / max-pool decisions upstream): worst 1.0e-3
GRAD_TOL_SLICE = 8e-4       # 256-element gradient slices of the small fixtures: worst 3.5e-4
ATT_TOL = 3e-4              # cls attention maps: worst 1.1e-4

pytestmark = pytest.mark.gpu


def _models(seed, frames, training, require_attention=True):
    cfg = arch.default_tsf_config(1280, frames)
    ef = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=0.0)
    ef_sd = synth.effnet_b0_state(seed)
    ef.load_state_dict(ef_sd)
    ef.train(training)
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=require_attention)
    tsf_sd = synth.tsf_state(cfg, seed)
    tsf.load_state_dict(tsf_sd)
    return cfg, ef.cuda(), tsf.cuda(), ef_sd, tsf_sd


def _step(ef, tsf, inp, require_attention=True):
    """The reference's training-step body (train.py:332-368) on our modules."""
    videos = inp["videos"]
    b, f, h, w, c = videos.shape
    videos = videos.reshape(b * f, h, w, c).permute(0, 3, 1, 2).cuda()           # rearrange "b f h w c -> (b f) c h w"
    features = ef(videos)
    features = features.reshape(b, f, *features.shape[1:])                       # "(b f) c h w -> b f c h w"
    out = tsf(features, mask=inp["mask"].cuda(), size_embedding=inp["size_embedding"],
              identities_mask=inp["identities_mask"].cuda(), positions=inp["positions"].cuda())
    return features, out


@pytest.mark.parametrize("name,deterministic", [("e2e_cfg1_eval", False), ("e2e_2id_train", False), ("e2e_2id_train", True)])
def test_step_matches_reference_fixture(name, deterministic):
    """... and once in deterministic mode (train.py:110; MT_DETERMINISTIC=1): the fixed-order / integer-limb reductions are judged
    against the reference's float64 pass directly, not only against the default path."""
    from mintime_amd import lib as L
    prev = L.set_deterministic(deterministic)
    try:
        _check_step_against_fixture(name)
    finally:
        L.set_deterministic(prev)


def _check_step_against_fixture(name):
    g = golden(name)
    B, Fr, seed, training = int(g["batch"]), int(g["frames"]), int(g["seed"]), bool(g["training"])
    cfg, ef, tsf, _, _ = _models(seed, Fr, training)
    inp = synth.clip_inputs(B, Fr, int(g["identities"]), seed, ragged=bool(g["ragged"]))
    features, (y_pred, (s_att, t_att)) = _step(ef, tsf, inp)
    y_pred = y_pred.cpu()                                                         # train.py:367
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y_pred, inp["labels"].reshape(-1, 1))
    loss.backward()
    # Judge against the reference's float64 pass (its exact arithmetic).  The reference's own fp32 CPU run deviates from
    # that by up to 8e-3 on the train-mode case (tools/make_golden.py: e2e_case) -- more than this path does.
    assert_close(y_pred, g["logits64"], REL_TOL, "logits")
    assert_close(loss, g["loss64"], REL_TOL, "loss")
    assert_close(s_att, g["space_att64"], ATT_TOL, "space attention")
    assert_close(t_att, g["time_att64"], ATT_TOL, "time attention")
    assert_close(features.mean(dim=(0, 1, 3, 4)), g["feat_mean64"], REL_TOL, "feature mean")
    ours_vs_exact = rel_err(y_pred, g["logits64"])
    ref32_vs_exact = rel_err(g["logits"], g["logits64"])
    print(f"{name}: logits rel err  ours {ours_vs_exact:.2e}   reference-fp32 {ref32_vs_exact:.2e}")
    for model, tag in ((ef, "ef."), (tsf, "tsf.")):
        named = dict(model.named_parameters())
        for k in g.files:
            if k.startswith("gnorm64." + tag):
                key = k[len("gnorm64." + tag):]
                assert named[key].grad is not None, key
                assert_close(named[key].grad.norm(), g[k], GRAD_TOL_TSF, k)                      # (norms: worst 4.5e-5)
                assert_close(named[key].grad.reshape(-1)[:256], g["gslice64." + tag + key], GRAD_TOL_SLICE, "gslice " + k)


# The full-size steps are judged against committed float64 runs of the imported reference (test_full_size_training_step_vs_reference_
# fixture); ONE configuration is also re-run live on the fp64 oracle as a cross-check of fixture and oracle against each other: the one
# whose reference fixture is missing, else config 2 (60 s of host time; config 3 takes 140 s of the suite's 1200 s).
import os as _os
from tests.util import GOLDEN as _GOLDEN
FULL_SIZE_LIVE = ([(16, 1, "config 2")] if _os.path.exists(_os.path.join(_GOLDEN, "e2e_full_config3.npz"))
                  else [(32, 2, "config 3")])


@pytest.mark.parametrize("training", [False, True])
def test_effnet_backward_all_parameters_vs_oracle(training):
    seed, n = 7, 3
    ef = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=0.0)
    sd = synth.effnet_b0_state(seed)
    ef.load_state_dict(sd)
    ef.train(training).cuda()
    x = synth.clip_inputs(1, n, 1, seed)["videos"].reshape(n, 224, 224, 3).permute(0, 3, 1, 2)
    g = torch.Generator().manual_seed(1)
    wgt = torch.randn(n, 1280, 7, 7, generator=g) * 0.1
    feat = ef(x.cuda())
    (feat * wgt.cuda()).sum().backward()
    osd = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    ofeat = O.effnet_b0_forward(osd, x, training=training)
    (ofeat * wgt).sum().backward()
    assert_close(feat, ofeat, REL_TOL, "features")
    worst = 0.0
    for k, p in ef.named_parameters():
        if k.startswith("_fc"):
            assert p.grad is None          # unused by forward (model.py:206-208)
            continue
        ref = osd[k].grad.detach()
        if training and k.endswith("_bn2.bias"):
            # analytically zero: a per-channel shift of a block output is removed by the train-mode BatchNorm that
            # follows the next 1x1 conv; both implementations only hold rounding noise here
            wn = float(dict(ef.named_parameters())[k.replace(".bias", ".weight")].grad.norm())
            assert float(p.grad.norm()) < 1e-3 * wn and float(ref.norm()) < 1e-3 * wn, k
            continue
        worst = max(worst, assert_close(p.grad, ref, GRAD_TOL_EF, "grad " + k))          # (worst measured here: 2.3e-5)
    print("worst relative gradient error", worst)


@pytest.mark.parametrize("B,ids,tag", FULL_SIZE_LIVE)
def test_config3_full_size_training_step_vs_oracle(B, ids, tag):
    """BASELINE configs 3 and 2 exactly as bench.py times them: B = 32 clips x 8 slots (256 crops), 2 identities [4,4] / B = 16,
    1 identity (128 crops; 6288 token rows: not a multiple of 32, the operand planes' zero padding is live), train-mode BatchNorm,
    drop-connect 0.2 (gates fed from the reference's RNG draws), BCE loss, backward.  Logits, loss and every gradient of both
    networks against the CPU oracle run in float64 (train.py:332-378)."""
    import time
    Fr, seed, rate = 8, 4, 0.2
    cfg = arch.default_tsf_config(1280, Fr)
    ef = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=rate)
    ef_sd = synth.effnet_b0_state(seed)
    ef.load_state_dict(ef_sd)
    ef.train(True).cuda()
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=False)
    tsf_sd = synth.tsf_state(cfg, seed)
    tsf.load_state_dict(tsf_sd)
    tsf.cuda()
    u = O.drop_connect_uniforms(seed, B * Fr, rate)
    ef.drop_connect_uniform = lambda rows, N, dev: torch.stack([u[i].reshape(N) for i in sorted(u)]).to(dev)
    inp = synth.clip_inputs(B, Fr, ids, seed, ragged=False)
    _, y_pred = _step(ef, tsf, inp, require_attention=False)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y_pred.cpu(), inp["labels"].reshape(-1, 1))
    loss.backward()
    torch.cuda.synchronize()
    t0 = time.time()
    torch.set_num_threads(min(64, torch.get_num_threads()))
    eo = {k: (v.double().requires_grad_("running_" not in k) if v.is_floating_point() else v) for k, v in ef_sd.items()}
    to = {k: v.double().requires_grad_(True) for k, v in tsf_sd.items()}
    inp64 = dict(inp)
    inp64["videos"] = inp["videos"].double()
    yo = O.clip_forward(eo, to, cfg, inp64, training_extractor=True, drop_connect_rate=rate, dc_uniform=u)
    lo = O.bce_with_logits(yo, inp["labels"])
    lo.backward()
    print(f"oracle fp64 {tag} step on the host: {time.time() - t0:.1f} s")
    assert_close(y_pred, yo, REL_TOL, f"logits ({tag}, B={B})")
    assert bool(((y_pred.detach().cpu().double() - yo.detach()).abs() <= 1e-3 * yo.detach().abs() + 1e-5).all())
    assert_close(loss, lo, REL_TOL, "loss")
    worst = 0.0
    for k, p in tsf.named_parameters():
        ref = to[k].grad
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            worst = max(worst, assert_close(p.grad, ref, GRAD_TOL_TSF, "tsf grad " + k))
    for k, p in ef.named_parameters():
        if k.startswith("_fc"):
            continue
        ref = eo[k].grad
        if k.endswith("_bn2.bias") or k == "_bn1.bias" and False:
            wn = float(dict(ef.named_parameters())[k.replace(".bias", ".weight")].grad.norm())
            if float(ref.norm()) < 1e-3 * wn:       # analytically zero (see test_effnet_backward_all_parameters_vs_oracle)
                assert float(p.grad.norm()) < 1e-3 * wn, k
                continue
        worst = max(worst, assert_close(p.grad, ref, GRAD_TOL_EF, "ef grad " + k))
    print(tag, "full size: worst relative gradient error", worst)


def _full_size_step(B, ids, seed=4, rate=0.2, Fr=8):
    cfg = arch.default_tsf_config(1280, Fr)
    ef = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=rate)
    ef_sd = synth.effnet_b0_state(seed)
    ef.load_state_dict(ef_sd)
    ef.train(True).cuda()
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=False)
    tsf_sd = synth.tsf_state(cfg, seed)
    tsf.load_state_dict(tsf_sd)
    tsf.cuda()
    u = O.drop_connect_uniforms(seed, B * Fr, rate)
    ef.drop_connect_uniform = lambda rows, N, dev: torch.stack([u[i].reshape(N) for i in sorted(u)]).to(dev)
    inp = synth.clip_inputs(B, Fr, ids, seed, ragged=False)
    _, y_pred = _step(ef, tsf, inp, require_attention=False)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y_pred.cpu(), inp["labels"].reshape(-1, 1))
    loss.backward()
    torch.cuda.synchronize()
    return cfg, ef, tsf, ef_sd, tsf_sd, u, inp, y_pred, loss


@pytest.mark.parametrize("name,B,ids", [("e2e_full_config2", 16, 1), ("e2e_full_config3", 32, 2)])
def test_full_size_training_step_vs_reference_fixture(name, B, ids):
    """BASELINE configs 2 and 3 exactly as bench.py times them (see the live-oracle test below), against the IMPORTED REFERENCE run in
    float64 at full size (tools/make_golden.py e2e_full_case; minutes of host time and tens of GB there, nothing here): logits,
    loss, the extractor's updated running statistics and, for EVERY parameter of both networks, the gradient's norm and a
    256-element strided sample."""
    import os
    from tests.util import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip(f"{name}.npz is not committed (its float64 reference run did not fit the build container's memory)")
    g = golden(name)
    assert int(g["batch"]) == B and int(g["identities"]) == ids
    cfg, ef, tsf, _, _, _, inp, y_pred, loss = _full_size_step(B, ids, seed=int(g["seed"]), rate=float(g["rate"]))
    assert abs(float(inp["videos"].double().sum()) - float(g["input_sum"])) <= 1e-6 * abs(float(g["input_sum"]))
    yo = torch.from_numpy(g["logits64"])
    assert_close(y_pred, yo, REL_TOL, f"logits ({name})")
    assert bool(((y_pred.detach().cpu().double() - yo).abs() <= 1e-3 * yo.abs() + 1e-5).all())
    assert_close(loss, torch.from_numpy(g["loss64"]), REL_TOL, "loss")
    esd = ef.state_dict()
    for key in [k[len("stat64."):] for k in g.files if k.startswith("stat64.")]:
        assert_close(esd[key], torch.from_numpy(g["stat64." + key]), REL_TOL, "running statistic " + key)
    worst, n, worst_c, worst_s, n_c = 0.0, 0, 0.0, 0.0, 0
    for tag, model, tol in (("tsf.", tsf, GRAD_TOL_EF), ("ef.", ef, GRAD_TOL_EF)):      # (sample / norm / checksum errors: worst 3.0e-4 / 3.7e-4)
        named = dict(model.named_parameters())
        for key in [k[len("gnorm64." + tag):] for k in g.files if k.startswith("gnorm64." + tag)]:
            got = named[key].grad.detach().double().cpu().reshape(-1)
            ref_norm, ref_max = float(g["gnorm64." + tag + key]), float(g["gabsmax64." + tag + key])
            if ref_max == 0.0:
                assert float(got.abs().max()) == 0.0, key
                continue
            if key.endswith("_bn2.bias"):
                wn = float(named[key.replace(".bias", ".weight")].grad.norm())
                if ref_norm < 1e-3 * wn:              # analytically zero (see test_effnet_backward_all_parameters_vs_oracle)
                    assert float(got.norm()) < 1e-3 * wn, key
                    continue
            step = max(1, got.numel() // 256)
            sample = torch.from_numpy(g["gsample64." + tag + key])
            err = float((got[::step][:256] - sample).abs().max()) / ref_max
            nerr = abs(float(got.norm()) - ref_norm) / ref_norm
            assert err <= tol and nerr <= tol, f"{tag}{key}: sample error {err:.2e}, norm error {nerr:.2e} (tolerance {tol:.0e})"
            worst = max(worst, err, nerr)
            n += 1
            if "gdot64." + tag + key in g.files:
                # whole-tensor checksum pair of the reference's gradient: sum g and sum g*r for the fixed pseudo-random r of
                # tests/util.probe_vector -- EVERY element is weighted, so a tile-local defect that sits between the 256 samples and
                # moves the norm by less than the gate still shows here.  Normalised so that an error vector d with random signs
                # gives |d . r| / (rms(r) |g|) ~ |sum d| / |g| ~ |d| / |g| (the relative L2 error).
                from tests.util import probe_vector
                r = torch.from_numpy(probe_vector(tag + key, got.numel(), int(g["seed"])))
                cdot = abs(float((got * r).sum()) - float(g["gdot64." + tag + key])) / (0.57735 * ref_norm)
                csum = abs(float(got.sum()) - float(g["gsum64." + tag + key])) / ref_norm
                assert cdot <= 8e-4, f"{tag}{key}: checksum sum(g*r) off by {cdot:.2e} |g|"          # (worst measured 3.7e-4)
                # sum g adds a SYSTEMATIC error coherently (sqrt(numel) x the random-sign case): it is the bias detector
                assert csum <= 1e-2, f"{tag}{key}: checksum sum(g) off by {csum:.2e} |g|"             # (worst measured 4.8e-3)
                worst_c = max(worst_c, cdot)
                worst_s = max(worst_s, csum)
                n_c += 1
    assert n > 300
    print(f"{name}: {n} parameter gradients within {worst:.2e} of the reference's float64 step; whole-tensor checksums of {n_c}: "
          f"sum(g*r) within {worst_c:.2e} |g|, sum(g) within {worst_s:.2e} |g|")
    assert n_c == n or n_c == 0


def test_hip_graph_replay_matches_eager_eval():
    """The whole eval forward captured in a HIP graph: replays bit-identically (the eval forward has no atomics) and on new inputs."""
    from mintime_amd import harness
    cfg, ef, tsf, _, _ = _models(0, 8, False, require_attention=False)
    ef.eval(); tsf.eval()
    inp = synth.clip_inputs(2, 8, 2, 0, ragged=True)
    b0 = {k: (v.cuda() if k != "size_embedding" else v) for k, v in inp.items()}
    with torch.no_grad():
        eager0 = harness.forward(ef, tsf, b0).clone()
    graphed = harness.GraphedEval(ef, tsf, b0)
    assert torch.equal(graphed(b0), eager0)
    inp1 = synth.clip_inputs(2, 8, 2, 0, ragged=True)
    inp1["videos"] = synth.clip_inputs(2, 8, 2, 5, ragged=True)["videos"]
    b1 = {k: (v.cuda() if k != "size_embedding" else v) for k, v in inp1.items()}
    with torch.no_grad():
        eager1 = harness.forward(ef, tsf, b1).clone()
    assert not torch.equal(eager1, eager0)
    assert torch.equal(graphed(b1), eager1)


_RCCL_SCRIPT = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import mintime_amd
from mintime_amd import harness, ddp
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[1])
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
finals = []
for overlapped in (False, True):
    cfg, ef, tsf = harness.build_models(seed=3, device="cuda", drop_connect_rate=0.0)
    opt = harness.make_optimizer(cfg, ef, tsf)
    live = {tsf.pos_emb.weight: 8 * 49 + 1, tsf.size_emb.weight: 21}        # what bench.py declares: only these rows go on the wire
    red = ddp.OverlappedGradReducer([tsf, ef], force=True, live_rows=live) if overlapped else None
    for step in range(3):
        batch = harness.device_batch(2, seed=step)
        loss = harness.train_step(ef, tsf, opt, batch, red)
    torch.cuda.synchronize()
    finals.append([p.detach().clone() for p in list(tsf.parameters()) + list(ef.parameters())] + [loss.detach().reshape(1)])
    if red is not None: stats = dict(red.stats)
dist.destroy_process_group()
# whole-tensor relative L2 distance between the two runs (sums of parameters cancel and would amplify the atomics' run-to-run noise)
worst = max(float((a - b).norm() / a.norm().clamp_min(1e-12)) for a, b in zip(*finals))
print("RESULT " + json.dumps({"worst_rel_l2": worst, "stats": stats}))
"""


def test_overlapped_allreduce_runs_on_rccl_single_rank(tmp_path):
    """The bucketed, backward-overlapped gradient reducer on a real RCCL communicator (1 rank: averaging is the identity, so
    three optimisation steps must land where the un-reduced run lands) -- exercises the hooks, the async launch from the
    autograd thread and the in-place flat-buffer path on the GPU."""
    import json, subprocess, sys, os
    script = tmp_path / "rccl_overlap.py"
    script.write_text(_RCCL_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), str(23456 + os.getpid() % 1000)], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    st = res["stats"]
    assert st["synchronous"] == 0 and st["overlapped_launches"] == 6, st      # both buckets, all three steps, from the engines' hook
    assert st["in_place"] == 6 and st["staged"] == 0, st           # p.grad aliases the engines' flat gradient buffers: no copies
    assert res["worst_rel_l2"] <= 1e-3, res


def test_config5_xception_timesformer_step_vs_oracle():
    """BASELINE config 5 (the "XS" variant): Xception extractor -> TimeSformer with 16 face slots of 3 identities [7,5,4],
    one ragged clip, eval extractor (what train.py does when the extractor is frozen) and train-mode loss/backward through
    the TimeSformer and the extractor.  Logits, loss and a spread of gradients against the CPU oracle on the same inputs."""
    from mintime_amd import xception
    F = 16
    cfg = arch.default_tsf_config(2048, F)
    xc_sd, tsf_sd = synth.xception_state(2), synth.tsf_state(cfg, 2)
    xc = xception(num_classes=1, pretrain_path=None)
    xc.load_state_dict(xc_sd)
    xc.cuda().eval()
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=False)
    tsf.load_state_dict(tsf_sd)
    tsf.cuda()
    inp = synth.clip_inputs(1, F, 3, 4, ragged=True)
    labels = torch.tensor([[1.0]])

    feats, out = _step(xc, tsf, inp, require_attention=False)
    assert feats.shape == (1, F, 2048, 7, 7)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, labels.cuda())
    loss.backward()

    v = inp["videos"]
    vid = v.reshape(F, 224, 224, 3).permute(0, 3, 1, 2)
    o_xc = {k: t.clone().requires_grad_(t.is_floating_point() and "running_" not in k) for k, t in xc_sd.items()}
    o_tsf = {k: t.clone().requires_grad_(True) for k, t in tsf_sd.items()}
    ofeat = O.xception_forward(o_xc, vid, training=False)
    ologits = O.tsf_forward(o_tsf, cfg, ofeat.reshape(1, F, 2048, 7, 7), inp["mask"], inp["identities_mask"], inp["size_embedding"],
                            inp["positions"])
    oloss = O.bce_with_logits(ologits, labels)
    oloss.backward()

    assert_close(feats.reshape(F, 2048, 7, 7), ofeat.detach(), REL_TOL, "Xception features")
    if os.environ.get("MT_TEST_VERBOSE"):
        print("[config5] logit", float(out), "oracle", float(ologits))
    assert_close(out, ologits.detach(), REL_TOL, "logits")
    assert abs(float(loss.detach()) - float(oloss.detach())) <= 1e-4 * max(1.0, abs(float(oloss.detach())))
    tsf_named = dict(tsf.named_parameters())
    for k in ("to_patch_embedding.weight", "pos_emb.weight", "layers.0.0.fn.to_qkv.weight", "layers.4.1.fn.to_out.0.weight",
              "layers.8.2.fn.net.0.weight", "to_out.1.weight"):
        assert_close(tsf_named[k].grad, o_tsf[k].grad, REL_TOL, "grad " + k)               # (worst measured: 4.8e-4)
    xc_named = dict(xc.named_parameters())
    for k in ("conv4.pointwise.weight", "block12.rep.4.pointwise.weight", "block6.rep.1.conv1.weight", "block1.skip.weight",
              "conv1.weight"):
        got, ref = xc_named[k].grad, o_xc[k].grad
        e = float((got.cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
        assert e <= 2e-2, f"grad {k}: relative L2 error {e:.3e}"      # ReLU/max-pool mask flips: see test_gpu_xception.py


def test_config5_train_mode_step_vs_fp64_oracle():
    """BASELINE config 5 the way bench.py --config 5 times it, at a size the fp64 oracle finishes on the host: 4 clips x 16 slots
    x 3 identities [7,5,4] (64 crops), TRAIN-mode BatchNorm in the Xception extractor (batch statistics in all 40 BN layers,
    running stats updated), BCE loss, backward through both networks.  Judged against the oracle in float64."""
    import time
    from mintime_amd import xception
    B, F, seed = 4, 16, 6
    cfg = arch.default_tsf_config(2048, F)
    xc_sd, tsf_sd = synth.xception_state(seed), synth.tsf_state(cfg, seed)
    xc = xception(num_classes=1, pretrain_path=None)
    xc.load_state_dict(xc_sd)
    xc.cuda().train()
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=False)
    tsf.load_state_dict(tsf_sd)
    tsf.cuda().train()
    inp = synth.clip_inputs(B, F, 3, seed, ragged=False)
    feats, out = _step(xc, tsf, inp, require_attention=False)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, inp["labels"].reshape(-1, 1).cuda())
    loss.backward()
    torch.cuda.synchronize()

    t0 = time.time()
    torch.set_num_threads(min(64, torch.get_num_threads()))
    vid = inp["videos"].reshape(B * F, 224, 224, 3).permute(0, 3, 1, 2).double()
    o_xc = {k: (t.double().requires_grad_("running_" not in k) if t.is_floating_point() else t.clone()) for k, t in xc_sd.items()}
    o_tsf = {k: t.double().requires_grad_(True) for k, t in tsf_sd.items()}
    ofeat = O.xception_forward(o_xc, vid, training=True)
    ologits = O.tsf_forward(o_tsf, cfg, ofeat.reshape(B, F, 2048, 7, 7), inp["mask"], inp["identities_mask"], inp["size_embedding"],
                            inp["positions"])
    oloss = O.bce_with_logits(ologits, inp["labels"])
    oloss.backward()
    print(f"oracle fp64 config-5 step (64 crops) on the host: {time.time() - t0:.1f} s")
    # the same oracle in float32: how far fp32 arithmetic ITSELF lands from fp64 for each tensor (ReLU / max-pool decisions that
    # rounding flips) -- the per-tensor floor the HIP path is held to below
    t0 = time.time()
    f_xc = {k: (t.clone().requires_grad_("running_" not in k) if t.is_floating_point() else t.clone()) for k, t in xc_sd.items()}
    f_tsf = {k: t.clone().requires_grad_(True) for k, t in tsf_sd.items()}
    ffeat = O.xception_forward(f_xc, vid.float(), training=True)
    flogits = O.tsf_forward(f_tsf, cfg, ffeat.reshape(B, F, 2048, 7, 7), inp["mask"], inp["identities_mask"], inp["size_embedding"],
                            inp["positions"])
    O.bce_with_logits(flogits, inp["labels"]).backward()
    print(f"oracle fp32 config-5 step on the host: {time.time() - t0:.1f} s")

    assert_close(feats.reshape(B * F, 2048, 7, 7), ofeat.detach(), REL_TOL, "Xception features (train-mode BN)")
    assert_close(out, ologits.detach(), REL_TOL, "logits")          # per tensor: max|d| <= 1e-3 max|ref| (north_star)
    # (no per-element gate here: one logit of this batch is 0.048, and behind 36 ReLU / 5 max-pool layers in train mode a handful of
    # rounding-level mask flips move it by 1e-4 absolute -- the reference's own fp32 run is as far from its fp64 run)
    assert abs(float(loss.detach()) - float(oloss.detach())) <= 1e-4 * max(1.0, abs(float(oloss.detach())))
    worst = 0.0
    for k, p in tsf.named_parameters():
        ref = o_tsf[k].grad
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        worst = max(worst, assert_close(p.grad, ref, GRAD_TOL_TSF_XS, "grad " + k))
    print("config 5 train step: worst TimeSformer gradient error", worst)
    xc_named = dict(xc.named_parameters())
    errs = {}
    for k in ("conv4.pointwise.weight", "bn4.weight", "conv3.pointwise.weight", "block12.rep.4.pointwise.weight", "block12.skip.weight",
              "block6.rep.1.conv1.weight", "block6.rep.4.pointwise.weight", "block3.rep.1.pointwise.weight", "block1.skip.weight",
              "conv2.weight", "conv1.weight"):
        got, ref = xc_named[k].grad, o_xc[k].grad
        errs[k] = float((got.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-30))
    print("config 5 train step: Xception gradient rel-L2 errors", {k: f"{v:.2e}" for k, v in errs.items()})
    # the tail of the extractor sits behind no ReLU / max-pool decision that fp32 rounding could flip: tight; further up the flips of
    # test_oracle_golden.py::test_xception_fp32_rounding_flips_relu_and_maxpool_decisions set the floor (the reference's own
    # fp32 run is this far from its fp64 run)
    for k in ("conv4.pointwise.weight", "bn4.weight"):
        assert errs[k] <= 3 * REL_TOL, (k, errs[k])
    # EVERY Xception gradient, each against its own floor: 3e-3 + 2 x (the fp32 oracle's distance from the fp64 oracle for that
    # tensor).  (Rounds 2-5 gated eleven tensors at a blanket 2e-2.)
    worst_x, floors = (0.0, ""), {}
    for k, p in xc.named_parameters():
        ref = o_xc[k].grad
        if p.grad is None or ref is None:
            assert p.grad is None and ref is None, k
            continue
        den = float(ref.norm().clamp_min(1e-30))
        floors[k] = float((f_xc[k].grad.double() - ref).norm()) / den
        e = float((p.grad.cpu().double() - ref).norm()) / den
        assert e <= 3 * REL_TOL + 2 * floors[k], f"grad {k}: rel-L2 {e:.3e} > 3e-3 + 2 x fp32 floor {floors[k]:.3e}"
        if e > worst_x[0]:
            worst_x = (e, k)
    print("config 5 train step: worst Xception gradient", worst_x, "fp32-oracle floor there", floors.get(worst_x[1]),
          "largest floor", max(floors.values()))
    # running statistics moved (momentum 0.1, torch defaults of models/xception.py) and the counters were bumped
    assert int(dict(xc.named_buffers())["bn1.num_batches_tracked"]) == 1


def test_config5_full_size_equals_eight_quarter_steps():
    """BASELINE config 5 at FULL size (B = 32 clips x 16 slots = 512 crops, 3 identities [7,5,4]) through a size-independent
    property: with eval-mode BatchNorm no operation couples clips, so one step on the 32 clips must equal eight steps on 4 clips each
    -- same features and logits, and the gradients of the 32-clip mean loss = the mean of the eight 4-clip gradients.  The 512-crop
    launch geometry (1.5 GB tensors, other split-K / chunk choices than the 64-crop oracle test) is what bench.py --config 5 times.
    Then one TRAIN-mode step at full size: finite loss / gradients / running statistics, loss within reach of the eval one."""
    from mintime_amd import harness
    B, F, seed = 32, 16, 5
    cfg, xc, tsf = harness.build_models_xs(F, seed=seed, device="cuda")
    xc.eval()
    batch = harness.device_batch(B, F, 3, seed=seed, device="cuda")
    params = list(xc.parameters()) + list(tsf.parameters())

    def run(bt):
        for p_ in params:
            p_.grad = None
        y = harness.forward(xc, tsf, bt)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(y, bt["labels"].reshape(-1, 1))
        loss.backward()
        return y.detach().clone(), float(loss.detach()), [None if p_.grad is None else p_.grad.detach().clone() for p_ in params]

    y_full, loss_full, g_full = run(batch)
    ys, acc = [], None
    for q in range(8):
        sl = slice(4 * q, 4 * q + 4)
        y_q, _, g_q = run({k: v[sl] for k, v in batch.items()})
        ys.append(y_q)
        acc = g_q if acc is None else [a if b is None else (b if a is None else a + b) for a, b in zip(acc, g_q)]
    y_cat = torch.cat(ys)
    assert y_cat.shape == y_full.shape == (B, 1)
    e_y = float((y_cat - y_full).abs().max() / y_full.abs().max())
    print("config 5 full size: logits 32 clips vs 8 x 4 clips, max rel diff", e_y, "bit-equal" if torch.equal(y_cat, y_full) else "")
    assert e_y <= 2e-6
    worst = (0.0, "")
    names = [k for k, _ in xc.named_parameters()] + [k for k, _ in tsf.named_parameters()]
    for k, gf, ga in zip(names, g_full, acc):
        if gf is None or ga is None:
            assert gf is None and ga is None, k
            continue
        ga = ga / 8.0
        den = float(gf.norm())
        if den == 0.0:
            assert float(ga.abs().max()) == 0.0, k
            continue
        e = float((ga - gf).norm()) / den
        if e > worst[0]:
            worst = (e, k)
        # split-K ranges, atomic orders and chunk counts differ between the two geometries: rounding-level differences only
        assert e <= 2e-5, f"grad {k}: 32-clip step vs mean of eight 4-clip steps, rel-L2 {e:.3e}"
    print("config 5 full size: worst gradient difference", worst)
    # train-mode BatchNorm at full size (what bench.py --config 5 runs): everything finite
    xc.train()
    y_t, loss_t, g_t = run(batch)
    assert bool(torch.isfinite(y_t).all()) and loss_t == loss_t and abs(loss_t - loss_full) < 0.5
    assert all(bool(torch.isfinite(g_).all()) for g_ in g_t if g_ is not None)
    assert all(bool(torch.isfinite(b_).all()) for b_ in xc.buffers() if b_.is_floating_point())


def test_native_bce_and_fused_sgd_match_torch():
    """Next-row f4: mt_bce_logits (value + gradient) against torch's BCEWithLogitsLoss(pos_weight) and the one-launch
    multi-tensor SGD against torch.optim.SGD(lr, weight_decay) over tensors of awkward sizes (scalar, odd, unaligned views)."""
    from mintime_amd import optim
    g = torch.Generator().manual_seed(3)
    for n, pw in ((1, None), (7, 2.5), (32, 0.4), (300, None)):
        x = (torch.randn(n, 1, generator=g) * 4).requires_grad_(True)
        y = (torch.rand(n, 1, generator=g) > 0.5).float()
        ref = torch.nn.BCEWithLogitsLoss(pos_weight=None if pw is None else torch.tensor([pw]))(x, y)
        ref.backward()
        xd = x.detach().cuda().requires_grad_(True)
        got = optim.bce_with_logits(xd, y.cuda(), pw)
        (got * 1.5).backward()
        assert abs(float(got.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
        assert_close(xd.grad, 1.5 * x.grad, 1e-5, "d loss / d logits")
    sizes = [(1,), (3, 5), (4097,), (128, 64), (2, 3, 3, 3), (100003,)]
    flat = torch.randn(sum(torch.Size(s).numel() for s in sizes) + 8, generator=g)
    ps_ref, ps_gpu, off = [], [], 1                                  # offset 1: parameters that are NOT 16-byte aligned
    for s in sizes:
        n = torch.Size(s).numel()
        w = flat[off:off + n].reshape(s).clone()
        ps_ref.append(torch.nn.Parameter(w.clone()))
        ps_gpu.append(torch.nn.Parameter(w.clone().cuda()))
        off += n
    o_ref = torch.optim.SGD(ps_ref, lr=0.05, weight_decay=1e-2)
    o_gpu = optim.FusedSGD(ps_gpu, lr=0.05, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(o_gpu, step_size=1, gamma=0.5)       # schedulers drive param_groups as usual
    sched_ref = torch.optim.lr_scheduler.StepLR(o_ref, step_size=1, gamma=0.5)
    for step in range(3):
        for pr, pg in zip(ps_ref, ps_gpu):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad, pg.grad = gr.clone(), gr.clone().cuda()
        ps_gpu[1].grad = None if step == 1 else ps_gpu[1].grad               # a parameter without gradient is skipped
        ps_ref[1].grad = None if step == 1 else ps_ref[1].grad
        o_ref.step(); o_gpu.step(); sched.step(); sched_ref.step()
    for pr, pg in zip(ps_ref, ps_gpu):
        assert_close(pg.detach(), pr.detach(), 1e-6, "parameters after 3 SGD steps")


@pytest.mark.parametrize("kind", ["adam", "adamw"])
def test_fused_adam_and_adamw_match_torch(kind):
    """train.py:187-190: the AdamW / Adam branches of the YAML's `optimizer` key as one multi-tensor launch, against torch's own
    optimizers over awkward tensors (scalar, odd, unaligned), several steps, an lr schedule, and a state_dict round trip."""
    from mintime_amd import optim
    g = torch.Generator().manual_seed(11)
    sizes = [(1,), (3, 5), (4097,), (128, 64), (2, 3, 3, 3), (100003,)]
    ps_ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in sizes]
    ps_gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ps_ref]
    T, F_ = (torch.optim.AdamW, optim.FusedAdamW) if kind == "adamw" else (torch.optim.Adam, optim.FusedAdam)
    o_ref, o_gpu = T(ps_ref, lr=0.01, weight_decay=1e-2), F_(ps_gpu, lr=0.01, weight_decay=1e-2)
    s_ref = torch.optim.lr_scheduler.StepLR(o_ref, step_size=2, gamma=0.5)
    for step in range(5):
        for pr, pg in zip(ps_ref, ps_gpu):
            gr = torch.randn(pr.shape, generator=g) * (10.0 ** (step - 2))       # gradient scales over 5 decades
            pr.grad, pg.grad = gr.clone(), gr.clone().cuda()
        o_ref.step(); o_gpu.step(); s_ref.step()
        if step == 2:                                                           # checkpoint / resume keeps the moments
            o_new = F_(ps_gpu, lr=0.01, weight_decay=1e-2)
            o_new.load_state_dict(o_gpu.state_dict())
            o_gpu = o_new
        o_gpu.param_groups[0]["lr"] = o_ref.param_groups[0]["lr"]              # lr is read from the group at every step
    for pr, pg in zip(ps_ref, ps_gpu):
        assert_close(pg.detach(), pr.detach(), 2e-6, f"parameters after 5 {kind} steps")
        assert_close(o_gpu.state[pg]["exp_avg_sq"], o_ref.state[pr]["exp_avg_sq"], 2e-6, "second moment")
    # the default weight decay differs between the two, like torch's
    assert F_(ps_gpu).defaults["weight_decay"] == T(ps_ref).defaults["weight_decay"]


def test_fused_adam_keeps_a_step_count_per_parameter():
    """A parameter that gets its first gradient after others have stepped (unfreezing mid-run: train.py:153-170) has its own bias
    corrections in torch.optim.Adam; the fused optimizer buckets a group's parameters by step count instead of refusing."""
    from mintime_amd import optim
    g = torch.Generator().manual_seed(3)
    ps_ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in [(33,), (8, 9), (257,)]]
    ps_gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ps_ref]
    o_ref, o_gpu = torch.optim.Adam(ps_ref, lr=0.01), optim.FusedAdam(ps_gpu, lr=0.01)
    for step in range(4):
        for i, (pr, pg) in enumerate(zip(ps_ref, ps_gpu)):
            if i == 2 and step < 2:                                            # the third parameter is "frozen" for two steps
                pr.grad = pg.grad = None
                continue
            gr = torch.randn(pr.shape, generator=g)
            pr.grad, pg.grad = gr.clone(), gr.clone().cuda()
        o_ref.step(); o_gpu.step()
    assert int(o_gpu.state[ps_gpu[2]]["step"]) == 2 and int(o_gpu.state[ps_gpu[0]]["step"]) == 4
    for pr, pg in zip(ps_ref, ps_gpu):
        assert_close(pg.detach(), pr.detach(), 2e-6, "parameters with different step counts")


_DP2_SCRIPT = r"""
import os, sys, json, time, torch, torch.distributed as dist
T0 = time.time()
def mark(what):
    print(f"TIMING rank {sys.argv[1]} {what}: {time.time() - T0:.1f} s", file=sys.stderr, flush=True)
# two processes share the one GPU of the test box: with the package's default of 8 hardware queues per process the queues are
# oversubscribed and the driver time-slices between the processes in long quanta (every small copy of build_models waited for one:
# 50 s per model build in round 4's suite); 2 queues per process fit side by side
os.environ["GPU_MAX_HW_QUEUES"] = "2"
sys.path.insert(0, os.getcwd())
rank = int(sys.argv[1]); port = sys.argv[2]; outdir = sys.argv[3]
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = port
import mintime_amd
from mintime_amd import harness, ddp, optim
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=2)           # two ranks share the one GPU of the test box; gloo moves CUDA tensors
lo, hi = ddp.shard_range(4, rank, 2)                                # global batch of 4 clips, 2 per rank

def shard(step):
    full = harness.device_batch(4, seed=10 + step)
    return {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in full.items()}

def picked(ef, tsf):
    tp, ep = list(tsf.named_parameters()), [(k, p) for k, p in ef.named_parameters() if not k.startswith("_fc") and not k.endswith("_bn2.bias")]   # _bn2.bias: analytically zero gradient (rounding noise only)
    return tp[:6] + tp[-6:] + tp[40:44] + ep[:6] + ep[-6:] + ep[100:104]

# (a) what this rank computes ALONE on its shard (no reducer, no process group involved).  The ranks take turns: two processes
# that start their HIP context, load the library's code objects and build models on ONE GPU at the same time are time-sliced against
# each other in long quanta (111 s for this phase in round 5's first suite run; a few seconds each when they do it one after the other)
for turn in (0, 1):
    if rank == turn:
        cfg, ef0, tsf0 = harness.build_models(seed=3, device="cuda", drop_connect_rate=0.0)
        start = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (ef0, tsf0)]
        y = harness.forward(ef0, tsf0, shard(0))
        optim.bce_with_logits(y, shard(0)["labels"], None).backward()
        alone = {k: p.grad.detach().cpu().clone() for k, p in picked(ef0, tsf0)}
        torch.cuda.synchronize()
        mark("alone step done")
    dist.barrier()
# (b) the data-parallel step, once per reducer, each from the same starting state (one model pair: building it is the slow part)
ef, tsf = ef0, tsf0
for mode in ("overlap", "flat"):
    ef.load_state_dict(start[0]); tsf.load_state_dict(start[1])
    ef._grads_ready_hook = tsf._grads_ready_hook = None
    for p in list(ef.parameters()) + list(tsf.parameters()):
        p.grad = None
    opt = harness.make_optimizer(cfg, ef, tsf)
    if mode == "overlap":
        red = ddp.OverlappedGradReducer([tsf, ef], live_rows={tsf.pos_emb.weight: 8 * 49 + 1, tsf.size_emb.weight: 21})
    else:
        red = ddp.GradAllReducer(list(ef.parameters()) + list(tsf.parameters()))
    loss = harness.train_step(ef, tsf, opt, shard(0), red)
    torch.cuda.synchronize()
    mark(mode + ": first data-parallel step done")
    averaged = {k: p.grad.detach().cpu().clone() for k, p in picked(ef, tsf)}
    torch.save({"alone": alone, "averaged": averaged}, os.path.join(outdir, f"grads_{mode}_{rank}.pt"))
    loss = harness.train_step(ef, tsf, opt, shard(1), red)
    torch.cuda.synchronize()
    sig = [float(p.detach().double().norm()) for p in list(tsf.parameters())[:8] + list(ef.parameters())[:8]]
    gsig = [float(p.grad.detach().double().norm()) for p in list(tsf.parameters())[:8] + list(ef.parameters())[:8] if p.grad is not None]
    print("RESULT " + json.dumps({"mode": mode, "rank": rank, "params": sig, "grads": gsig, "stats": dict(getattr(red, "stats", {}))}), flush=True)
    mark(mode + ": second step done")
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_data_parallel_step_keeps_replicas_identical(tmp_path):
    """Row (e) end to end with the real engines: two processes (sharing this box's single GPU, gloo transport) each train on their
    shard of a 4-clip batch.  VALUE check: the gradient every rank ends up with equals the mean of what each rank computes alone on
    its shard (a separate, reducer-free run inside each process); the engine-level bucket hooks must fire on both ranks
    (mode "overlap") / the flat fallback reducer bench.py falls back to must give the same result (mode "flat"); after two steps
    parameters and gradients agree between the ranks.  Both reducers run in the same pair of processes, from the same state."""
    import json, subprocess, sys, os
    script = tmp_path / "dp2.py"
    script.write_text(_DP2_SCRIPT)
    port = str(24500 + os.getpid() % 1000)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, cwd=root) for r in (0, 1)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        print("\n".join(l for l in se.splitlines() if l.startswith("TIMING")))
    for mode in ("overlap", "flat"):
        res = [[json.loads(l[7:]) for l in so.splitlines() if l.startswith("RESULT ") and json.loads(l[7:])["mode"] == mode][-1]
               for so, _ in outs]
        if mode == "overlap":
            for r in res:
                assert r["stats"]["overlapped_launches"] == 4 and r["stats"]["synchronous"] == 0, r["stats"]
        for a, b in zip(res[0]["params"], res[1]["params"]):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (a, b)
        for a, b in zip(res[0]["grads"], res[1]["grads"]):
            assert abs(a - b) <= 1e-6 * max(1e-3, abs(a)), (a, b)
        g0, g1 = (torch.load(tmp_path / f"grads_{mode}_{r}.pt") for r in (0, 1))
        assert len(g0["alone"]) >= 30
        for k in g0["alone"]:
            want = (g0["alone"][k].double() + g1["alone"][k].double()) / 2
            for g in (g0, g1):
                got = g["averaged"][k].double()
                den = float(want.norm())
                if den == 0.0:
                    assert float(got.norm()) == 0.0, k
                else:
                    assert float((got - want).norm()) <= 2e-4 * den, (mode, k, float((got - want).norm()) / den)


def test_nn_dataparallel_wrap_on_one_gpu_is_transparent():
    """train.py:294-296 / test.py:129-133 / predict.py:378-379 wrap both modules in nn.DataParallel.  On one visible GPU that
    wrapper scatters the inputs (moving the host-resident size_embedding to the device) and calls the module: forward, backward
    and state_dict keys ('module.' prefix, as the reference's checkpoints have) must behave exactly like the bare modules."""
    cfg, ef, tsf, _, _ = _models(0, 8, True, require_attention=True)
    inp = synth.clip_inputs(2, 8, 2, 0, ragged=True)
    _, (y0, (s0, t0)) = _step(ef, tsf, inp)
    torch.nn.functional.binary_cross_entropy_with_logits(y0, inp["labels"].reshape(-1, 1).cuda()).backward()
    g0 = {k: p.grad.clone() for k, p in list(tsf.named_parameters()) + list(ef.named_parameters()) if p.grad is not None}
    cfg, ef1, tsf1, _, _ = _models(0, 8, True, require_attention=True)
    dp_ef, dp_tsf = torch.nn.DataParallel(ef1, device_ids=[0]), torch.nn.DataParallel(tsf1, device_ids=[0])
    assert all(k.startswith("module.") for k in dp_tsf.state_dict())
    _, (y1, (s1, t1)) = _step(dp_ef, dp_tsf, inp)
    torch.nn.functional.binary_cross_entropy_with_logits(y1, inp["labels"].reshape(-1, 1).cuda()).backward()
    assert_close(y1, y0, 1e-5, "logits through DataParallel")
    assert_close(s1, s0, 1e-5, "space attention through DataParallel")
    for k, p in list(tsf1.named_parameters()) + list(ef1.named_parameters()):
        if p.grad is not None and not k.endswith("_bn2.bias"):                      # _bn2.bias: analytically zero, noise only
            assert_close(p.grad, g0[k], 2e-4, "grad through DataParallel " + k)      # atomics order only


def _apply_unfreeze_rule(named_params, unfreeze_blocks):
    """train.py:157-170: with --extractor_unfreeze_blocks k > -1 only MBConv blocks >= 16 - k stay trainable; the stem, the head
    conv and every other block get requires_grad = False (the extractor itself stays in train() mode)."""
    for name, p in named_params:
        if "blocks" in name:
            p.requires_grad_(int(name.split(".")[1]) >= 16 - unfreeze_blocks)
        else:
            p.requires_grad_(False)


@pytest.mark.parametrize("unfreeze", [3, 0])
def test_partially_frozen_extractor_matches_oracle_and_skips_frozen_work(unfreeze):
    """Frozen parameters get NO gradient (None, like torch autograd), the trainable ones match the oracle, and the reverse
    walk neither launches a frozen weight gradient nor descends below the lowest trainable block."""
    from mintime_amd import effnet_backward as EB
    seed, B, Fr = 5, 1, 8
    cfg, ef, tsf, ef_sd, tsf_sd = _models(seed, Fr, True, require_attention=False)
    _apply_unfreeze_rule(ef.named_parameters(), unfreeze)
    inp = synth.clip_inputs(B, Fr, 2, seed)
    EB.LAST_RUN.update(blocks_run=-1, wgrad_launches=-1, stem_run=None)
    _, y = _step(ef, tsf, inp, require_attention=False)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y.cpu(), inp["labels"].reshape(-1, 1))
    loss.backward()
    eo = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in ef_sd.items()}
    for k, v in eo.items():
        if v.is_floating_point() and "running_" not in k and not k.startswith("_fc"):
            v.requires_grad_(True)
    _apply_unfreeze_rule([(k, v) for k, v in eo.items() if v.is_floating_point() and "running_" not in k and "num_batches" not in k], unfreeze)
    to = {k: v.double().requires_grad_(True) for k, v in tsf_sd.items()}
    inp64 = dict(inp)
    inp64["videos"] = inp["videos"].double()
    yo = O.clip_forward(eo, to, cfg, inp64, training_extractor=True)
    O.bce_with_logits(yo, inp["labels"]).backward()
    assert_close(y, yo, REL_TOL, "logits")
    n_train = 0
    for k, p in ef.named_parameters():
        if k.startswith("_fc"):
            continue
        if not p.requires_grad:
            assert p.grad is None and eo[k].grad is None, k
            continue
        n_train += 1
        if k.endswith("_bn2.bias"):
            continue                       # analytically zero under train-mode BN (see test_effnet_backward_all_parameters_vs_oracle)
        assert_close(p.grad, eo[k].grad, GRAD_TOL_EF, "grad " + k)
    for k, p in tsf.named_parameters():
        if float(to[k].grad.abs().max()) > 0:
            assert_close(p.grad, to[k].grad, GRAD_TOL_TSF, "tsf grad " + k)
    if unfreeze == 0:
        # nothing in the extractor is trainable: its backward never runs, the TimeSformer skips the feature gradient
        assert n_train == 0 and EB.LAST_RUN["blocks_run"] == -1
    else:
        assert EB.LAST_RUN["blocks_run"] == unfreeze and EB.LAST_RUN["stem_run"] is False
        # per trainable block: expand, depthwise, squeeze-excite (one launch for its four tensors), project; the frozen head conv: none
        assert EB.LAST_RUN["wgrad_launches"] == 4 * unfreeze


def test_frozen_backbone_step_like_train_py():
    """--freeze_backbone (train.py:153-155,344-346,180-181): extractor in eval() under no_grad, only the TimeSformer trains."""
    seed, B, Fr = 6, 2, 8
    cfg, ef, tsf, ef_sd, tsf_sd = _models(seed, Fr, False, require_attention=False)
    inp = synth.clip_inputs(B, Fr, 2, seed)
    videos = inp["videos"].reshape(B * Fr, 224, 224, 3).permute(0, 3, 1, 2).cuda()
    with torch.no_grad():
        features = ef(videos)
    assert not features.requires_grad
    features = features.reshape(B, Fr, *features.shape[1:])
    y = tsf(features, mask=inp["mask"].cuda(), size_embedding=inp["size_embedding"], identities_mask=inp["identities_mask"].cuda(),
            positions=inp["positions"].cuda())
    torch.nn.functional.binary_cross_entropy_with_logits(y.cpu(), inp["labels"].reshape(-1, 1)).backward()
    assert all(p.grad is None for p in ef.parameters())
    eo = {k: v.clone() for k, v in ef_sd.items()}
    to = {k: v.clone().requires_grad_(True) for k, v in tsf_sd.items()}
    with torch.no_grad():
        fo = O.effnet_b0_forward(eo, inp["videos"].reshape(B * Fr, 224, 224, 3).permute(0, 3, 1, 2), training=False)
    yo = O.tsf_forward(to, cfg, fo.reshape(B, Fr, *fo.shape[1:]), inp["mask"], inp["identities_mask"], inp["size_embedding"], inp["positions"])
    yo = yo[0] if isinstance(yo, tuple) else yo
    O.bce_with_logits(yo, inp["labels"]).backward()
    assert_close(y, yo, REL_TOL, "logits")
    for k, p in tsf.named_parameters():
        if float(to[k].grad.abs().max()) > 0:
            assert_close(p.grad, to[k].grad, GRAD_TOL_TSF, "tsf grad " + k)
