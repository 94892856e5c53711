"""GPU parity of the HIP SizeInvariantTimeSformer against the CPU oracle and the reference-generated fixtures."""
import os

import pytest
import torch

import mintime_amd
from mintime_amd import arch, synth, SizeInvariantTimeSformer
from oracle import mintime_oracle as O
from tests.util import REL_TOL, assert_close, golden, GRAD_TOL_UNIT

pytestmark = pytest.mark.gpu


def _build(cfg, seed, require_attention=True):
    model = SizeInvariantTimeSformer(config=cfg, require_attention=require_attention)
    sd = synth.tsf_state(cfg, seed)
    model.load_state_dict(sd, strict=True)
    return model.cuda(), sd


def _cfg(g, C, Fr):
    cfg = arch.default_tsf_config(C, Fr)
    if "pos_emb" in g.files:        # the embedding switches of size_invariant_timesformer.py:235-248
        cfg["model"]["enable-pos-emb"], cfg["model"]["enable-size-emb"] = bool(g["pos_emb"]), bool(g["size_emb"])
    return cfg


def _inputs(g):
    B, Fr, C = int(g["batch"]), int(g["frames"]), int(g["channels"])
    feats = synth.features(B, Fr, C, int(g["seed"]))
    aux = synth.clip_inputs(B, Fr, int(g["identities"]), int(g["seed"]), ragged=bool(g["ragged"]), with_video=False)
    return B, Fr, C, feats, aux


@pytest.mark.parametrize("name", ["tsf_cfg1", "tsf_2id_ragged", "tsf_xs_3id", "tsf_nopos", "tsf_nosize"])
def test_forward_matches_reference_fixture(name):
    g = golden(name)
    B, Fr, C, feats, aux = _inputs(g)
    cfg = _cfg(g, C, Fr)
    model, sd = _build(cfg, int(g["seed"]))
    with torch.no_grad():
        # size_embedding stays on the CPU like the reference call sites (train.py:355)
        logits, (s_att, t_att) = model(feats.cuda(), mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                                       size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
    assert logits.shape == (B, 1) and s_att.shape == (B * 8, 1, 1 + Fr * 49)
    assert_close(logits, g["logits"], REL_TOL, "logits vs reference")
    assert_close(s_att, g["space_att"], REL_TOL, "space cls attention vs reference")
    assert_close(t_att, g["time_att"], REL_TOL, "time cls attention vs reference")
    # elementwise gate on the logits: |d| <= 1e-3*|ref| + 1e-5
    ref = torch.as_tensor(g["logits"])
    assert bool(((logits.cpu() - ref).abs() <= 1e-3 * ref.abs() + 1e-5).all())
    # and against the oracle run in this process
    with torch.no_grad():
        o_logits, (o_s, o_t) = O.tsf_forward(sd, cfg, feats, aux["mask"], aux["identities_mask"], aux["size_embedding"],
                                             aux["positions"], require_attention=True)
    assert_close(logits, o_logits, REL_TOL, "logits vs oracle")
    assert_close(t_att, o_t, REL_TOL, "time att vs oracle")


def test_masked_keys_get_exactly_zero_attention():
    g = golden("tsf_2id_ragged")
    B, Fr, C, feats, aux = _inputs(g)
    cfg = arch.default_tsf_config(C, Fr)
    model, _ = _build(cfg, int(g["seed"]))
    with torch.no_grad():
        _, (s_att, t_att) = model(feats.cuda(), mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                                  size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
    m = aux["mask"]
    cm = torch.cat([torch.ones(B, 1, dtype=torch.bool), m.repeat_interleave(49, 1)], 1)
    cm = cm[:, None].expand(-1, 8, -1).reshape(-1, 1, cm.shape[-1])
    assert float(t_att.cpu()[~cm].abs().max()) == 0.0
    assert float(s_att.cpu()[~cm].abs().max()) == 0.0
    assert_close(t_att.sum(-1), torch.ones(B * 8, 1), 1e-5, "probabilities sum to one")


def test_nchw_contiguous_features_are_accepted():
    """The drop-in contract: any [B,F,C,7,7] tensor works, not only our NHWC-strided view."""
    g = golden("tsf_cfg1")
    B, Fr, C, feats, aux = _inputs(g)
    cfg = arch.default_tsf_config(C, Fr)
    model, _ = _build(cfg, int(g["seed"]), require_attention=False)
    with torch.no_grad():
        out = model(feats.contiguous().cuda(), mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                    size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
    assert_close(out, g["logits"], REL_TOL, "NCHW-contiguous input")


@pytest.mark.parametrize("name", ["tsf_cfg1", "tsf_2id_ragged", "tsf_nopos", "tsf_nosize"])
def test_backward_matches_reference_fixture(name):
    g = golden(name)
    B, Fr, C, feats, aux = _inputs(g)
    cfg = _cfg(g, C, Fr)
    model, sd = _build(cfg, int(g["seed"]), require_attention=False)
    x = feats.cuda().requires_grad_(True)
    out = model(x, mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
    # loss on the CPU like the reference (train.py:367-368)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out.cpu(), aux["labels"].reshape(-1, 1))
    loss.backward()
    assert_close(loss, g["loss"], REL_TOL, "loss")
    named = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("gnorm."):
            key = k[len("gnorm."):]
            assert named[key].grad is not None, key
            assert_close(named[key].grad.norm(), g[k], REL_TOL, k)
            if "gslice." + key in g.files:
                assert_close(named[key].grad.reshape(-1)[:256], g["gslice." + key], GRAD_TOL_UNIT, "gslice." + key)
    assert_close(named["pos_emb.weight"].grad[:8], g["gslice.pos_emb.rows"], REL_TOL, "pos_emb grad rows")
    if "gslice.size_emb.rows" in g.files:
        assert_close(named["size_emb.weight"].grad[:21], g["gslice.size_emb.rows"], REL_TOL, "size_emb grad rows")
    else:
        assert "size_emb.weight" not in named
    assert float(named["pos_emb.weight"].grad[Fr * 49 + 1:].abs().max()) == 0.0
    assert_close(x.grad.norm(), g["dfeats_norm"], REL_TOL, "dfeats norm")
    assert_close(x.grad.permute(0, 1, 3, 4, 2).reshape(-1)[:512], g["dfeats_slice"], GRAD_TOL_UNIT, "dfeats slice")


@pytest.mark.parametrize("B,Fr,C,ids", [(2, 8, 1280, 2), (1, 16, 2048, 3), (1, 32, 1280, 3)])
def test_backward_all_parameters_vs_oracle(B, Fr, C, ids):
    """Every parameter gradient (not just the fixture's sample) against torch autograd over the CPU oracle;
    the second case is the XS configuration (Xception features, 16 frames, 3 identities [7,5,4]), the third the largest
    frame count the reference accepts (train.py:101: 32 slots, identities [14,10,8], 1569 tokens)."""
    seed = 5
    cfg = arch.default_tsf_config(C, Fr)
    model, sd = _build(cfg, seed, require_attention=False)
    feats = synth.features(B, Fr, C, seed)
    aux = synth.clip_inputs(B, Fr, ids, seed, ragged=True, with_video=False)
    out = model(feats.cuda(), mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
    w = torch.tensor([[1.0], [-0.7]])[:B]
    (out.cpu() * w).sum().backward()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.tsf_forward(osd, cfg, feats, aux["mask"], aux["identities_mask"], aux["size_embedding"], aux["positions"])
    (oout * w).sum().backward()
    for k, p in model.named_parameters():
        ref = osd[k].grad
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            assert_close(p.grad, ref, REL_TOL, "grad " + k)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_attention_aggregation_matches_reference_fixture(tag):
    """Next-row f3: mt_attn_aggregate against utils.py:68-96 ITSELF (tests/golden/agg_att.npz was produced by importing the
    reference's utils.py in the build container), on the cls attentions of three fixtures (2 ids ragged, XS 3 ids, 1 id)."""
    from mintime_amd import harness
    g = golden("agg_att")
    fx = golden(str(g[tag + "_fixture"]))
    frames, fpi = int(g[tag + "_frames"]), [int(v) for v in g[tag + "_fpi"]]
    s_att, t_att = torch.as_tensor(fx["space_att"]), torch.as_tensor(fx["time_att"])
    got, got_id = harness.aggregate_attentions([s_att.cuda(), t_att.cuda()], 8, frames, fpi, scale_factor=50000)
    assert_close(torch.tensor(got), g[tag + "_agg"], 1e-4, "aggregated attentions vs reference")
    assert_close(torch.tensor(got_id), g[tag + "_ident"], 1e-4, "identity attentions vs reference")


def test_clip_with_a_fully_padded_identity_and_a_single_valid_slot():
    """Edge of the masking rules (:252-260): one identity has no valid face at all (all its slots padded) and the other has a
    single valid slot; every masked key must still get exactly zero probability and the logits must match the oracle."""
    Fr, C, seed = 8, 1280, 11
    cfg = arch.default_tsf_config(C, Fr)
    model, sd = _build(cfg, seed, require_attention=True)
    feats = synth.features(1, Fr, C, seed)
    aux = synth.clip_inputs(1, Fr, 2, seed, ragged=False, with_video=False)
    mask = torch.zeros(1, Fr, dtype=torch.bool)
    mask[0, 0] = True                                       # identity 0: slot 0 valid only; identity 1: nothing valid
    size = aux["size_embedding"].clone()
    size[~mask] = 0
    out, (space, time_) = model(feats.cuda(), mask=mask.cuda(), identities_mask=aux["identities_mask"].cuda(), size_embedding=size,
                                positions=aux["positions"].cuda())
    oout, (ospace, otime) = O.tsf_forward(sd, cfg, feats, mask, aux["identities_mask"], size, aux["positions"], require_attention=True)
    assert_close(out, oout, REL_TOL, "logits")
    assert_close(space, ospace, REL_TOL, "space cls attention")
    assert_close(time_, otime, REL_TOL, "time cls attention")
    padded_tokens = (~mask[0]).repeat_interleave(49)
    assert float(space[:, 0, 1:][:, padded_tokens.cuda()].abs().max()) == 0.0
    assert torch.isfinite(out).all()


def test_full_size_batch_logits_and_all_gradients_vs_oracle():
    """The configuration bench.py times (config 3: B = 32, 8 slots, 2 identities, M = 12 576 token rows): at this size the engine
    takes its large-M branches (FF2 and the N = 512 data gradients as split-K + atomics, 64x64 tiles, L2-blocked tile order).
    Logits, both cls attentions and EVERY parameter gradient against the CPU oracle (size_invariant_timesformer.py:263-268)."""
    B, Fr, C, seed = 32, 8, 1280, 7
    cfg = arch.default_tsf_config(C, Fr)
    model, sd = _build(cfg, seed, require_attention=True)
    feats = synth.features(B, Fr, C, seed)
    aux = synth.clip_inputs(B, Fr, 2, seed, ragged=True, with_video=False)
    x = feats.cuda().requires_grad_(True)
    out, (s_att, t_att) = model(x, mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                                size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
    w = torch.linspace(-1.0, 1.3, B).reshape(B, 1)          # (weights that do not sum to zero: d/d(head bias) = sum(w))
    (out.cpu() * w).sum().backward()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = feats.clone().requires_grad_(True)
    oout, (o_s, o_t) = O.tsf_forward(osd, cfg, xo, aux["mask"], aux["identities_mask"], aux["size_embedding"], aux["positions"],
                                     require_attention=True)
    (oout * w).sum().backward()
    assert_close(out, oout, REL_TOL, "logits at B=32")
    assert bool(((out.detach().cpu() - oout.detach()).abs() <= 1e-3 * oout.detach().abs() + 1e-5).all())
    assert_close(s_att, o_s, REL_TOL, "space cls attention at B=32")
    assert_close(t_att, o_t, REL_TOL, "time cls attention at B=32")
    assert_close(x.grad, xo.grad, REL_TOL, "feature gradient at B=32")
    for k, p in model.named_parameters():
        ref = osd[k].grad
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            assert_close(p.grad, ref, REL_TOL, "grad " + k)


def test_out_of_range_indices_are_caught_not_dereferenced():
    """nn.Embedding raises on a bad index (size_invariant_timesformer.py:236,248).  Host-resident indices are range-checked
    immediately; device-resident ones are clamped in the kernels (no out-of-bounds read or atomic) and reported when the next
    forward starts."""
    Fr, C = 8, 1280
    cfg = arch.default_tsf_config(C, Fr)
    model, _ = _build(cfg, 0, require_attention=False)
    feats = synth.features(1, Fr, C, 0).cuda()
    aux = synth.clip_inputs(1, Fr, 1, 0, with_video=False)
    kw = dict(mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda())
    bad_size = aux["size_embedding"].clone()
    bad_size[0, 3] = 10 ** 6
    with pytest.raises(IndexError):
        model(feats, size_embedding=bad_size, positions=aux["positions"].cuda(), **kw)
    neg = aux["size_embedding"].clone()
    neg[0, 0] = -1
    with pytest.raises(IndexError):
        model(feats, size_embedding=neg, positions=aux["positions"].cuda(), **kw)
    bad_pos = aux["positions"].clone()
    bad_pos[0, 5] = 10 ** 9
    with pytest.raises(IndexError):
        model(feats, size_embedding=aux["size_embedding"], positions=bad_pos, **kw)          # host tensor: immediate
    x = feats.clone().requires_grad_(True)
    out = model(x, size_embedding=aux["size_embedding"], positions=bad_pos.cuda(), **kw)     # device tensor: clamped + flagged
    out.sum().backward()                                                                     # the scatter-add is clamped too
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(model.pos_emb.weight.grad).all()
    with pytest.raises(IndexError, match="positions"):
        model(feats, size_embedding=aux["size_embedding"], positions=aux["positions"].cuda(), **kw)
    good = model(feats, size_embedding=aux["size_embedding"], positions=aux["positions"].cuda(), **kw)   # flag was cleared
    assert torch.isfinite(good).all()


def test_eval_forward_is_bit_reproducible_at_large_batch():
    """Inference never takes the split-K + atomics branch (tsf_engine: `M >= 4096 and save`): same bits every run at B = 16."""
    B, Fr, C = 16, 8, 1280
    cfg = arch.default_tsf_config(C, Fr)
    model, _ = _build(cfg, 0, require_attention=False)
    model.eval()
    feats = synth.features(B, Fr, C, 1).cuda()
    aux = synth.clip_inputs(B, Fr, 2, 1, ragged=True, with_video=False)
    outs = []
    with torch.no_grad():
        for _ in range(3):
            outs.append(model(feats, mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                              size_embedding=aux["size_embedding"], positions=aux["positions"].cuda()).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_second_backward_through_the_same_forward_raises_clearly():
    Fr, C = 8, 1280
    cfg = arch.default_tsf_config(C, Fr)
    model, _ = _build(cfg, 0, require_attention=False)
    feats = synth.features(1, Fr, C, 0).cuda()
    aux = synth.clip_inputs(1, Fr, 1, 0, with_video=False)
    out = model(feats, mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
    out.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        out.sum().backward()


def test_two_graphs_share_the_weight_planes_until_the_weights_change():
    """The Linear weights' operand planes live on the module (tsf_planes.weight_planes).  Two forwards followed by their two
    backwards (gradient accumulation) see the same planes; a backward that would run on planes of UPDATED weights is refused (torch
    raises its version-counter error in that situation: the fused optimizers update through raw pointers, so the engine checks)."""
    from mintime_amd import optim
    Fr, C = 8, 1280
    cfg = arch.default_tsf_config(C, Fr)
    model, _ = _build(cfg, 0, require_attention=False)
    feats = synth.features(2, Fr, C, 0).cuda()
    aux = synth.clip_inputs(2, Fr, 1, 0, with_video=False)
    kw = dict(mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(), size_embedding=aux["size_embedding"],
              positions=aux["positions"].cuda())
    model(feats, **kw).sum().backward()
    ref = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    for p in model.parameters():
        p.grad = None
    a, b = model(feats, **kw), model(feats, **kw)            # two live graphs, no update in between
    a.sum().backward()
    got = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    b.sum().backward()
    for g, r in zip(got, ref):
        assert_close(g, r, 1e-5, "first of two live graphs")
    opt = optim.FusedSGD(model.parameters(), lr=0.01)
    stale = model(feats, **kw)
    opt.step()                                                # weights change through raw pointers ...
    model(feats, **kw)                                        # ... and the next training forward re-splits them
    if os.environ.get("MT_TSF_PLANES", "1") != "0":
        with pytest.raises(RuntimeError, match="weights were updated"):
            stale.sum().backward()


@pytest.mark.parametrize("rows,skip", [(2 * 393, 0), (5 * 393 + 0, 393), (1000, 0), (32 * 393, 393)])
def test_layernorm_backward_emits_column_sums_of_updated_dx(rows, skip):
    """mt_layernorm_bwd's dx_colsum: the bias gradient of the Linear below (sum over rows of the updated residual-stream
    gradient), with the cls rows (row % skip == 0) left out for the patch embedding."""
    from mintime_amd import lib as L
    D = 512
    g = torch.Generator().manual_seed(rows)
    x, dy, dx0 = torch.randn(rows, D, generator=g), torch.randn(rows, D, generator=g), torch.randn(rows, D, generator=g)
    gamma = torch.rand(D, generator=g) + 0.5
    xd = x.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (D,), gamma.double(), torch.zeros(D, dtype=torch.float64), 1e-5)
    (y * dy.double()).sum().backward()
    dx_ref = dx0.double() + xd.grad
    keep = torch.ones(rows, dtype=torch.bool)
    if skip:
        keep[::skip] = False
    mean, var = x.double().mean(1), x.double().var(1, unbiased=False)
    stats = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], 1).float().cuda()
    dxd, dg, db, cs = dx0.cuda().clone(), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dy_d, x_d, gamma_d = dy.cuda(), x.cuda(), gamma.cuda()          # keep the device copies alive across the raw-pointer call
    L.check(L.get().mt_layernorm_bwd(L.ptr(dy_d), L.ptr(x_d), L.ptr(stats), L.ptr(gamma_d), L.ptr(dxd), L.ptr(dg), L.ptr(db),
                                     rows, D, 1, L.ptr(cs), skip, None, L.stream_ptr()), "ln bwd")
    assert_close(dxd, dx_ref, 1e-5, "dx")
    # out-of-place form (dx = LN'(dy) + dx_in): same numbers, the input gradient untouched
    dx_in, dx_out = dx0.cuda().clone(), torch.full((rows, D), float("nan"), device="cuda")
    dg2, db2, cs2 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    L.check(L.get().mt_layernorm_bwd(L.ptr(dy_d), L.ptr(x_d), L.ptr(stats), L.ptr(gamma_d), L.ptr(dx_out), L.ptr(dg2), L.ptr(db2),
                                     rows, D, 1, L.ptr(cs2), skip, L.ptr(dx_in), L.stream_ptr()), "ln bwd (out of place)")
    assert torch.equal(dx_out, dxd) and torch.equal(dx_in, dx0.cuda())
    assert_close(cs, dx_ref[keep].sum(0), 1e-4, "column sums of the updated dx")
    assert_close(db, dy.double().sum(0), 1e-4, "dbeta")
    # the same backward split by queue: rows kernel (dx only) + cols kernel (the three parameter-gradient sums)
    dx_rows = torch.full((rows, D), float("nan"), device="cuda")
    dg3, db3, cs3 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    L.check(L.get().mt_layernorm_bwd_rows(L.ptr(dy_d), L.ptr(x_d), L.ptr(stats), L.ptr(gamma_d), L.ptr(dx_rows), L.ptr(dx_in), rows, D,
                                          None, L.stream_ptr()), "ln bwd rows")
    L.check(L.get().mt_layernorm_bwd_cols(L.ptr(dy_d), L.ptr(x_d), L.ptr(stats), L.ptr(dx_rows), L.ptr(dg3), L.ptr(db3), L.ptr(cs3), skip,
                                          rows, D, L.stream_ptr()), "ln bwd cols")
    assert_close(dx_rows, dx_ref, 1e-5, "dx (rows kernel)")
    xh = (x.double() - mean[:, None]) * (1.0 / torch.sqrt(var + 1e-5))[:, None]
    assert_close(dg3, (dy.double() * xh).sum(0), 1e-4, "dgamma (cols kernel)")
    assert_close(dg, (dy.double() * xh).sum(0), 1e-4, "dgamma (fused kernel)")
    assert_close(db3, dy.double().sum(0), 1e-4, "dbeta (cols kernel)")
    assert_close(cs3, dx_ref[keep].sum(0), 1e-4, "column sums (cols kernel)")
    # ... and with the sums folded into the rows kernel as per-block partials + a block-order reduce (the default of the plane path):
    # dx bit-identical to the rows kernel, sums deterministic (two runs agree to the bit) and accumulated ONTO the targets
    lib = L.get()
    nb = lib.mt_layernorm_bwd_rows_blocks(rows)
    res = []
    for rep in range(2):
        dx_f = torch.full((rows, D), float("nan"), device="cuda")
        part = torch.full((nb, 3, D), float("nan"), device="cuda")
        dg4, db4, cs4 = torch.ones(D, device="cuda"), torch.ones(D, device="cuda"), torch.ones(D, device="cuda")
        L.check(lib.mt_layernorm_bwd_rows_sums(L.ptr(dy_d), L.ptr(x_d), L.ptr(stats), L.ptr(gamma_d), L.ptr(dx_f), L.ptr(dx_in), rows, D, None,
                                               L.ptr(part), skip, L.stream_ptr()), "ln bwd rows + sums")
        L.check(lib.mt_layernorm_bwd_cols_reduce(L.ptr(part), nb, D, L.ptr(dg4), L.ptr(db4), L.ptr(cs4), L.stream_ptr()), "ln cols reduce")
        res.append((dx_f, dg4, db4, cs4))
    dx_f, dg4, db4, cs4 = res[0]
    assert_close(dx_f, dx_ref, 1e-5, "dx (rows + sums kernel)")
    assert_close(dg4 - 1.0, (dy.double() * xh).sum(0), 1e-4, "dgamma (folded)")
    assert_close(db4 - 1.0, dy.double().sum(0), 1e-4, "dbeta (folded)")
    assert_close(cs4 - 1.0, dx_ref[keep].sum(0), 1e-4, "column sums (folded)")
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1])), "block-order reduce: run-to-run identical"
    # dx_colsum is optional
    dg5, db5 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    L.check(lib.mt_layernorm_bwd_cols_reduce(L.ptr(part), nb, D, L.ptr(dg5), L.ptr(db5), None, L.stream_ptr()), "ln cols reduce")
    assert torch.equal(dg5 + 1.0, dg4) or float((dg5 + 1.0 - dg4).abs().max()) < 1e-4


@pytest.mark.parametrize("B,Fr,ids,ragged", [(2, 8, 2, True), (3, 16, 3, False)])
def test_last_layer_dead_row_pruning_is_exact(B, Fr, ids, ragged, monkeypatch):
    """MT_TSF_PRUNE_LAST=1: the last layer's space-attention tail and feed-forward block run on the cls rows only (the head reads
    nothing else, size_invariant_timesformer.py:270-276).  Logits, both attention maps and EVERY parameter / feature gradient must
    be the numbers of the full computation (whose dead rows carry exact zeros) -- and match the oracle."""
    C, seed = 1280, 9
    cfg = arch.default_tsf_config(C, Fr)
    feats = synth.features(B, Fr, C, seed)
    aux = synth.clip_inputs(B, Fr, ids, seed, ragged=ragged, with_video=False)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("MT_TSF_PRUNE_LAST", flag)
        model, sd = _build(cfg, seed)
        x = feats.cuda().requires_grad_(True)
        logits, (s_att, t_att) = model(x, mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(),
                                       size_embedding=aux["size_embedding"], positions=aux["positions"].cuda())
        torch.nn.functional.binary_cross_entropy_with_logits(logits, aux["labels"].reshape(-1, 1).cuda()).backward()
        runs[flag] = (logits.detach(), s_att, t_att, x.grad, {k: p.grad for k, p in model.named_parameters()})
    full, pruned = runs["0"], runs["1"]
    assert_close(pruned[0], full[0], 1e-5, "logits")
    assert torch.equal(pruned[1], full[1]) and torch.equal(pruned[2], full[2])          # same cls kernels, same inputs
    assert_close(pruned[3], full[3], 2e-5, "feature gradient")
    for k, g in full[4].items():
        if float(g.abs().max()) == 0.0:
            assert float(pruned[4][k].abs().max()) == 0.0, k
        else:
            assert_close(pruned[4][k], g, 5e-5, "grad " + k)
    o_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o_logits = O.tsf_forward(o_sd, cfg, feats, aux["mask"], aux["identities_mask"], aux["size_embedding"], aux["positions"])
    O.bce_with_logits(o_logits, aux["labels"]).backward()
    assert_close(pruned[0], o_logits, REL_TOL, "logits vs oracle")
    for k in ("layers.8.2.fn.net.0.weight", "layers.8.2.fn.net.3.bias", "layers.8.1.fn.to_out.0.weight", "layers.8.1.fn.to_qkv.weight",
              "layers.8.0.fn.to_qkv.weight", "layers.0.0.fn.to_qkv.weight", "to_patch_embedding.weight"):
        assert_close(pruned[4][k], o_sd[k].grad, GRAD_TOL_UNIT, "grad vs oracle " + k)


@pytest.mark.gpu
def test_transpose_multi_is_exact():
    """mt_transpose_multi (the TimeSformer's transposed Linear weights, one launch per step): bit-exact against torch, ragged tiles."""
    from mintime_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(512, 1536), (37, 65), (2048, 512), (1, 33), (64, 64)]
    src = [torch.randn(r, c, device="cuda", generator=g) for r, c in shapes]
    dst = [torch.full((c, r), float("nan"), device="cuda") for r, c in shapes]
    rows, tiles = [], 0
    for s, d in zip(src, dst):
        rows.append((s.data_ptr(), d.data_ptr(), s.shape[0], s.shape[1], tiles))
        tiles += ((s.shape[0] + 31) // 32) * ((s.shape[1] + 31) // 32)
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    L.check(L.get().mt_transpose_multi(table.data_ptr(), len(rows), tiles, L.stream_ptr()), "mt_transpose_multi")
    for s, d in zip(src, dst):
        assert torch.equal(d, s.t().contiguous())


def _dropout_model(g):
    from tests.util import dropout_multipliers
    B, Fr, C, depth = int(g["batch"]), int(g["frames"]), int(g["channels"]), int(g["depth"])
    cfg = arch.default_tsf_config(C, Fr)
    cfg["model"]["depth"], cfg["model"]["attn-dropout"], cfg["model"]["ff-dropout"] = depth, float(g["attn_p"]), float(g["ff_p"])
    model, _ = _build(cfg, int(g["seed"]), require_attention=False)
    dm = dropout_multipliers(g, B, 1 + Fr * 49, cfg["model"]["dim"])
    order = [(li, kind) for li in range(depth) for kind in range(3)]          # forward order: time, space, feed-forward per layer
    calls = []

    def sampler(shape, dev):          # uniforms that reproduce the reference's keeps: 1 >= p keeps, 0 < p drops
        key = order[len(calls) % len(order)]
        calls.append(key)
        keep = (dm[key] != 0).reshape(shape)
        return keep.float().to(dev)

    model.dropout_uniform = sampler
    feats = synth.features(B, Fr, C, int(g["seed"]))
    aux = synth.clip_inputs(B, Fr, int(g["identities"]), int(g["seed"]), ragged=True, with_video=False)
    return model, cfg, feats, aux, calls


def test_dropout_train_step_matches_reference_fixture():
    """attn-dropout 0.1 / ff-dropout 0.2, train mode (size_invariant_timesformer.py:66-70, 98-101), fed the reference's own draws:
    logits, loss and the gradients of every layer parameter against the reference's fp32 run (tests/golden/tsf_dropout.npz)."""
    g = golden("tsf_dropout")
    model, cfg, feats, aux, calls = _dropout_model(g)
    model.train()
    x = feats.cuda().requires_grad_(True)
    kw = dict(mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(), size_embedding=aux["size_embedding"],
              positions=aux["positions"].cuda())
    out = model(x, **kw)
    assert len(calls) == 3 * int(g["depth"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out.cpu(), aux["labels"].reshape(-1, 1))
    loss.backward()
    assert_close(out, g["logits"], REL_TOL, "logits")
    assert_close(loss, g["loss"], REL_TOL, "loss")
    named = dict(model.named_parameters())
    n = 0
    for k in g.files:
        if k.startswith("gnorm."):
            key = k[len("gnorm."):]
            assert_close(named[key].grad.norm(), g[k], REL_TOL, k)
            assert_close(named[key].grad.reshape(-1)[:128], g["gslice." + key], GRAD_TOL_UNIT, "gslice." + key)
            n += 1
    assert n >= 16 * int(g["depth"])
    assert_close(x.grad.norm(), g["dfeats_norm"], REL_TOL, "dfeats norm")
    assert_close(x.grad.permute(0, 1, 3, 4, 2).reshape(-1)[:512], g["dfeats_slice"], GRAD_TOL_UNIT, "dfeats slice")
    # eval mode: nn.Dropout is the identity -- same logits as a model built without dropout, no draws taken
    model.eval()
    with torch.no_grad():
        ev = model(feats.cuda(), **kw)
    cfg0 = {**cfg, "model": {**cfg["model"], "attn-dropout": 0.0, "ff-dropout": 0.0}}
    plain, _ = _build(cfg0, int(g["seed"]), require_attention=False)
    plain.eval()
    with torch.no_grad():
        ref = plain(feats.cuda(), **kw)
    assert torch.equal(ev, ref) and len(calls) == 3 * int(g["depth"])
    assert float((ev.cpu() - torch.as_tensor(g["logits"])).abs().max()) > 1e-3       # ... and train mode did drop something


def test_dropout_default_draws_keep_the_expected_fraction():
    """Without a sampler the multipliers come from torch.rand: keep rate 1 - p, scale 1 / (1 - p); two forwards differ."""
    from mintime_amd import tsf_planes
    cfg = arch.default_tsf_config(1280, 8)
    cfg["model"]["depth"], cfg["model"]["attn-dropout"], cfg["model"]["ff-dropout"] = 2, 0.25, 0.5
    model, _ = _build(cfg, 3, require_attention=False)
    model.train()
    m = tsf_planes._dropout_mult(model, (4096, 512), 0.25, "cuda")
    assert abs(float((m != 0).float().mean()) - 0.75) < 0.01 and abs(float(m.max()) - 1.0 / 0.75) < 1e-6
    feats = synth.features(2, 8, 1280, 3).cuda()
    aux = synth.clip_inputs(2, 8, 2, 3, ragged=False, with_video=False)
    kw = dict(mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(), size_embedding=aux["size_embedding"],
              positions=aux["positions"].cuda())
    with torch.no_grad():
        a, b = model(feats, **kw), model(feats, **kw)
    assert not torch.equal(a, b)
