#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (imported from /root/reference,
CPU, fp32 + one fp64 pass) on this repo's seeded weights and synthetic inputs.

Runs only in the build container (the reference does not exist on the GPU box).  Nothing from the
reference's source enters the repo: the fixtures are inputs' checksums and output tensors.

    python tools/make_golden.py            # writes every fixture
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import mintime_amd  # noqa: E402  (alias of the hyphen-named package)
from mintime_amd import arch, synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))   # size_invariant_timesformer.py:8 imports it, unused
    # the reference's `models` package must win over this repo's drop-in `models/`
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    # /root/reference/models has no __init__.py (namespace package), so this repo's regular `models` package would
    # shadow it regardless of order: take the repo root off sys.path while the reference is imported.
    saved_path = list(sys.path)
    sys.path[:] = [REF] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from models.size_invariant_timesformer import SizeInvariantTimeSformer
        from models.efficientnet.efficientnet_pytorch import EfficientNet
    assert SizeInvariantTimeSformer.__module__ == "models.size_invariant_timesformer"
    import models
    assert list(models.__path__)[0].startswith(REF), models.__path__
    sys.path[:] = saved_path
    return EfficientNet, SizeInvariantTimeSformer


class _Placeholder(types.ModuleType):
    """Empty stand-in for a module the reference imports at file scope but never touches on the code path being run
    (cv2, magic, albumentations, torchvision, pytorchvideo ... are not installed here): every attribute is an inert
    placeholder class, so `from x import A, B` and `class C(x.Base)` at import time succeed and nothing else works."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


def placeholder_modules(*names):
    for n in names:
        if not hasattr(sys.modules.get(n), "__file__"):      # absent, or the bare cv2 stub import_reference() installed
            m = _Placeholder(n)
            m.__path__ = []
            sys.modules[n] = m


def import_reference_file(modname, relpath):
    """Import one top-level reference file (utils.py, deepfakes_dataset.py) under a private module name."""
    import importlib.util
    saved_path = list(sys.path)
    sys.path[:] = [REF] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    try:
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            spec.loader.exec_module(mod)
    finally:
        sys.path[:] = saved_path
    return mod


def checksum(t):
    return float(t.double().sum())


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def build_tsf(TSF, cfg, seed, require_attention, dtype=torch.float32):
    model = TSF(config=cfg, require_attention=require_attention)
    sd = synth.tsf_state(cfg, seed)
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    return model.to(dtype), sd


def tsf_case(TSF, name, batch, frames, channels, identities, ragged, seed, pos_emb=True, size_emb=True):
    cfg = arch.default_tsf_config(channels=channels, num_frames=frames)
    cfg["model"]["enable-pos-emb"], cfg["model"]["enable-size-emb"] = pos_emb, size_emb   # size_invariant_timesformer.py:235-248
    model, sd = build_tsf(TSF, cfg, seed, True)
    feats = synth.features(batch, frames, channels, seed)
    aux = synth.clip_inputs(batch, frames, identities, seed, ragged=ragged, with_video=False)
    rows = []
    hooks = []
    for i in range(1, cfg["model"]["depth"]):
        hooks.append(model.layers[i][0].register_forward_pre_hook(lambda m, a: rows.append(a[0][:, 0].detach().clone())))
    hooks.append(model.to_out.register_forward_pre_hook(lambda m, a: rows.append(a[0].detach().clone())))
    tok = []
    hooks.append(model.layers[0][0].register_forward_pre_hook(lambda m, a: tok.append(a[0].detach().clone())))
    with torch.no_grad():
        logits, (s_att, t_att) = model(feats, mask=aux["mask"], identities_mask=aux["identities_mask"],
                                       size_embedding=aux["size_embedding"], positions=aux["positions"])
        for h in hooks:
            h.remove()
        m64, _ = build_tsf(TSF, cfg, seed, True, torch.float64)
        logits64, _ = m64(feats.double(), mask=aux["mask"], identities_mask=aux["identities_mask"],
                          size_embedding=aux["size_embedding"], positions=aux["positions"])
    # backward: d(sum of logits * labels-ish weights) for a few parameters
    model.zero_grad()
    feats_g = feats.clone().requires_grad_(True)
    out, _ = model(feats_g, mask=aux["mask"], identities_mask=aux["identities_mask"],
                   size_embedding=aux["size_embedding"], positions=aux["positions"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, aux["labels"].reshape(-1, 1))
    loss.backward()
    grads = {}
    named = dict(model.named_parameters())
    for key in GRAD_KEYS_TSF:
        g = named[key].grad
        grads["gnorm." + key] = g.norm()
        grads["gslice." + key] = g.reshape(-1)[:256].clone()
    # live rows of the embedding tables
    grads["gslice.pos_emb.rows"] = named["pos_emb.weight"].grad[:8].clone()
    if size_emb:
        grads["gslice.size_emb.rows"] = named["size_emb.weight"].grad[:21].clone()
    if not pos_emb:                      # arange positions: every row 0..N-1 of the table is live
        grads["gnorm.pos_emb.weight"] = named["pos_emb.weight"].grad.norm()
    save(name, logits=logits, logits64=logits64, space_att=s_att, time_att=t_att,
         cls_rows=torch.stack(rows), tokens_head=tok[0][:, :60].clone(), tokens_sum=checksum(tok[0]),
         feats_sum=checksum(feats), loss=loss.detach(), dfeats_norm=feats_g.grad.norm(),
         dfeats_slice=feats_g.grad.permute(0, 1, 3, 4, 2).reshape(-1)[:512].clone(),
         batch=batch, frames=frames, channels=channels, identities=identities, ragged=int(ragged), seed=seed,
         pos_emb=int(pos_emb), size_emb=int(size_emb), **grads)


def tsf_dropout_case(TSF, name, batch, frames, channels, identities, seed, depth=3, attn_p=0.1, ff_p=0.2):
    """Train-mode TimeSformer with attn-dropout / ff-dropout > 0 (size_invariant_timesformer.py:66-70, 98-101).  The multipliers
    nn.Dropout applied (keep / (1 - p)) are read off its input / output with forward hooks and stored bit-packed; the fixture pins the
    reference's logits, loss and gradients for exactly those draws."""
    import numpy as np
    cfg = arch.default_tsf_config(channels=channels, num_frames=frames)
    cfg["model"]["depth"], cfg["model"]["attn-dropout"], cfg["model"]["ff-dropout"] = depth, attn_p, ff_p
    model, sd = build_tsf(TSF, cfg, seed, False)
    model.train()
    feats = synth.features(batch, frames, channels, seed)
    aux = synth.clip_inputs(batch, frames, identities, seed, ragged=True, with_video=False)
    keeps = {}
    hooks = []
    for li in range(depth):
        mods = [(0, model.layers[li][0].fn.to_out[1]), (1, model.layers[li][1].fn.to_out[1]), (2, model.layers[li][2].fn.net[2])]
        for kind, mod in mods:
            assert isinstance(mod, torch.nn.Dropout) and mod.p == (ff_p if kind == 2 else attn_p)
            def hook(m, a, out, key=(li, kind)):
                x = a[0]
                keeps[key] = ((out != 0) | (x == 0)).detach().clone()       # (an input that is exactly 0 tells nothing: count it kept)
            hooks.append(mod.register_forward_hook(hook))
    torch.manual_seed(seed)
    feats_g = feats.clone().requires_grad_(True)
    out = model(feats_g, mask=aux["mask"], identities_mask=aux["identities_mask"], size_embedding=aux["size_embedding"],
                positions=aux["positions"])
    out = out[0] if isinstance(out, tuple) else out
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, aux["labels"].reshape(-1, 1))
    loss.backward()
    for h in hooks:
        h.remove()
    assert len(keeps) == 3 * depth
    named = dict(model.named_parameters())
    grads = {}
    for key, p in named.items():
        if p.grad is not None and (key.startswith("layers.") or key in ("cls_token", "to_patch_embedding.weight", "to_out.1.weight")):
            grads["gnorm." + key] = p.grad.norm()
            grads["gslice." + key] = p.grad.reshape(-1)[:128].clone()
    packed = {f"keep.{li}.{kind}": torch.from_numpy(np.packbits(k.numpy().reshape(-1))) for (li, kind), k in keeps.items()}
    rates = torch.tensor([float(keeps[(li, kind)].float().mean()) for li in range(depth) for kind in range(3)])
    save(name, logits=out.detach(), loss=loss.detach(), dfeats_norm=feats_g.grad.norm(),
         dfeats_slice=feats_g.grad.permute(0, 1, 3, 4, 2).reshape(-1)[:512].clone(), keep_rates=rates,
         batch=batch, frames=frames, channels=channels, identities=identities, seed=seed, depth=depth,
         attn_p=torch.tensor(attn_p), ff_p=torch.tensor(ff_p), **packed, **grads)


GRAD_KEYS_TSF = ["cls_token", "to_patch_embedding.weight", "to_patch_embedding.bias",
                 "layers.0.0.fn.to_qkv.weight", "layers.0.0.fn.to_out.0.weight", "layers.0.0.fn.to_out.0.bias",
                 "layers.0.0.norm.weight", "layers.0.0.norm.bias", "layers.4.1.fn.to_qkv.weight",
                 "layers.4.2.fn.net.0.weight", "layers.4.2.fn.net.0.bias", "layers.4.2.fn.net.3.weight",
                 "layers.8.2.fn.net.3.bias", "layers.8.1.fn.to_out.0.weight", "to_out.0.weight", "to_out.1.weight",
                 "to_out.1.bias"]

GRAD_KEYS_EF = ["_conv_stem.weight", "_bn0.weight", "_bn0.bias", "_blocks.0._depthwise_conv.weight",
                "_blocks.0._se_reduce.weight", "_blocks.0._se_expand.bias", "_blocks.0._project_conv.weight",
                "_blocks.1._expand_conv.weight", "_blocks.3._depthwise_conv.weight", "_blocks.3._bn1.weight",
                "_blocks.5._bn2.weight", "_blocks.10._se_expand.weight", "_blocks.10._project_conv.weight",
                "_blocks.15._expand_conv.weight", "_conv_head.weight", "_bn1.weight"]


def build_ef(EF, seed, training, dtype=torch.float32):
    model = EF.from_name("efficientnet-b0", drop_connect_rate=0.0)
    sd = synth.effnet_b0_state(seed)
    model.load_state_dict(sd, strict=True)
    model.train(training)
    return model.to(dtype), sd


def ef_case(EF, name, n_img, training, seed):
    model, sd = build_ef(EF, seed, training)
    vid = synth.clip_inputs(1, n_img, 1, seed)["videos"]            # [1,n,224,224,3]
    x = vid.reshape(n_img, 224, 224, 3).permute(0, 3, 1, 2)         # NHWC-strided NCHW view (train.py:341)
    taps = {}
    hooks = [model._blocks[i].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(i, o.detach()))
             for i in (0, 2, 5, 10, 15)]
    with torch.no_grad():
        feats = model(x)
    for h in hooks:
        h.remove()
    extra = {}
    if training:
        msd = model.state_dict()
        for k in ("_bn0.running_mean", "_bn0.running_var", "_blocks.3._bn1.running_mean", "_blocks.3._bn1.running_var",
                  "_blocks.15._bn2.running_var", "_bn1.running_mean", "_bn1.running_var"):
            extra["stat." + k] = msd[k].clone()
        extra["nbt"] = msd["_bn0.num_batches_tracked"].clone()
    m64, _ = build_ef(EF, seed, training, torch.float64)
    with torch.no_grad():
        feats64 = m64(x.double())
    save(name, features=feats, feat64_slice=feats64[:, :64].clone(), input_sum=checksum(x),
         **{f"block{i}_mean": taps[i].mean(dim=(0, 2, 3)) for i in taps},
         **{f"block{i}_absmax": taps[i].abs().amax(dim=(0, 2, 3)) for i in taps},
         **{f"block{i}_slice": taps[i][0, :, :4, :4].clone() for i in taps},
         n_img=n_img, training=int(training), seed=seed, **extra)


def ef_dc_case(EF, name, n_img, seed, rate=0.2):
    """Train-mode EfficientNet WITH drop-connect (utils.py:129-154): the reference draws torch.rand([N,1,1,1]) once per gated
    block after torch.manual_seed(seed); oracle.drop_connect_uniforms(seed, N) replays exactly those draws."""
    model = EF.from_name("efficientnet-b0", drop_connect_rate=rate)
    sd = synth.effnet_b0_state(seed)
    model.load_state_dict(sd, strict=True)
    model.train(True)
    vid = synth.clip_inputs(1, n_img, 1, seed)["videos"]
    x = vid.reshape(n_img, 224, 224, 3).permute(0, 3, 1, 2)
    taps = {}
    hooks = [model._blocks[i].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(i, o.detach().clone()))
             for i in (2, 7, 10, 14)]
    gw = torch.from_numpy(np.random.Generator(np.random.Philox(key=[seed, 777])).standard_normal((n_img, 1280, 7, 7)) * 0.1).float()
    torch.manual_seed(seed)
    feats = model(x)
    for h in hooks:
        h.remove()
    (feats * gw).sum().backward()
    named = dict(model.named_parameters())
    grads = {}
    for key in GRAD_KEYS_EF:
        g = named[key].grad
        grads["gnorm." + key] = g.norm()
        grads["gslice." + key] = g.reshape(-1)[:256].clone()
    save(name, features=feats.detach(), input_sum=checksum(x), rate=rate, n_img=n_img, seed=seed,
         **{f"block{i}_slice": taps[i][:, :, :3, :3].clone() for i in taps}, **grads)


def agg_case(name):
    """utils.py:68-96 aggregate_attentions run on the cls attentions of the tsf_2id_ragged / tsf_xs_3id fixtures."""
    placeholder_modules("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms._transforms_video",
                        "pytorchvideo", "pytorchvideo.data", "pytorchvideo.data.encoded_video", "pytorchvideo.transforms")
    ref_utils = import_reference_file("_ref_utils", "utils.py")
    out = {}
    for tag, fx, frames, fpi in (("a", "tsf_2id_ragged", 8, [4, 8]), ("b", "tsf_xs_3id", 16, [7, 12, 16]), ("c", "tsf_cfg1", 8, [8])):
        g = np.load(os.path.join(OUT, fx + ".npz"))
        atts = [torch.as_tensor(g["space_att"]), torch.as_tensor(g["time_att"])]
        agg, ident = ref_utils.aggregate_attentions(atts, 8, frames, fpi)
        out[tag + "_agg"] = np.asarray([np.asarray(r, dtype=np.float64) for r in agg])
        out[tag + "_ident"] = np.asarray(ident, dtype=np.float64)
        out[tag + "_fixture"] = fx
        out[tag + "_frames"] = frames
        out[tag + "_fpi"] = np.asarray(fpi)
    save(name, **out)


def slots_case(name):
    """deepfakes_dataset.py:130-186 get_sorted_identities run on throw-away directory trees (identities_ordering=1: by number of
    faces, so neither python-magic nor cv2 is reached; distinct counts so os.listdir order cannot matter)."""
    import itertools, shutil, tempfile
    placeholder_modules("cv2", "magic", "albumentations", "albumentations.augmentations", "albumentations.augmentations.functional")
    ds_mod = import_reference_file("_ref_dataset", "deepfakes_dataset.py")
    rng = np.random.Generator(np.random.Philox(key=[11, 22]))
    cases = []
    for num_frames in (8, 16, 32):
        for max_id in (1, 2, 3, 4):
            for n_id in (1, 2, 3, 4, 5):
                for _ in range(6):
                    counts = rng.choice(np.arange(1, 3 * num_frames), size=n_id, replace=False)
                    cases.append((num_frames, max_id, [int(c) for c in counts]))
    rows = []
    for (num_frames, max_id, counts) in cases:
        root = tempfile.mkdtemp(prefix="slots_")
        try:
            for i, c in enumerate(counts):
                d = os.path.join(root, f"identity_{i}")
                os.makedirs(d)
                for k in range(c):
                    open(os.path.join(d, f"{k * 3}_{i}.jpg"), "w").close()
            ds = ds_mod.DeepFakesDataset([], [], "", "", 224, num_frames=num_frames, max_identities=max_id, identities_ordering=1)
            ids, _ = ds.get_sorted_identities(root)
            got = [(int(os.path.basename(p).split("_")[1]), int(n)) for p, _, n in ids]
        finally:
            shutil.rmtree(root)
        rows.append(dict(num_frames=num_frames, max_identities=max_id, counts=counts, order=[g[0] for g in got], slots=[g[1] for g in got]))
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump(rows, fh)
    print(f"wrote {name}.json ({len(rows)} cases)")
    # the module-level constants of the size embedding (deepfakes_dataset.py:30-31): what CAN be read off the imported reference
    # of the per-clip tensor rules -- the rest of __getitem__ (:216-339) sits behind cv2.VideoCapture (:250), cv2.imread (:257)
    # and the albumentations transform call (:299-308) and cannot be run here (oracle/mintime_oracle.py, f1 header)
    with open(os.path.join(OUT, "f1_constants.json"), "w") as fh:
        json.dump(dict(RANGE_SIZE=int(ds_mod.RANGE_SIZE), SIZE_EMB_DICT=[[int(a), int(b)] for a, b in ds_mod.SIZE_EMB_DICT],
                       MODES=list(ds_mod.MODES)), fh)
    print("wrote f1_constants.json")


def e2e_case(EF, TSF, name, batch, frames, identities, ragged, training, seed):
    """Whole step.  Stored twice: the reference in fp32 (what a user would run) and in fp64 (its exact arithmetic;
    the fp32 run of an ill-conditioned case -- train-mode BN over a batch with padded all-zero crops -- deviates from
    it by several 1e-3, which bounds what "parity within 1e-3" can mean there)."""
    cfg = arch.default_tsf_config(channels=1280, num_frames=frames)
    inp = synth.clip_inputs(batch, frames, identities, seed, ragged=ragged)
    out = {}
    for tag, dtype in (("", torch.float32), ("64", torch.float64)):
        ef, _ = build_ef(EF, seed, training, dtype)
        tsf, _ = build_tsf(TSF, cfg, seed, True, dtype)
        v = inp["videos"].to(dtype)
        b, f, h, w, c = v.shape
        vid = v.reshape(b * f, h, w, c).permute(0, 3, 1, 2)
        ef.zero_grad(); tsf.zero_grad()
        feats = ef(vid)
        logits, (s_att, t_att) = tsf(feats.reshape(b, f, *feats.shape[1:]), mask=inp["mask"],
                                     identities_mask=inp["identities_mask"], size_embedding=inp["size_embedding"],
                                     positions=inp["positions"])
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, inp["labels"].reshape(-1, 1).to(dtype))
        loss.backward()
        for model, keys, mtag in ((ef, GRAD_KEYS_EF, "ef."), (tsf, GRAD_KEYS_TSF, "tsf.")):
            named = dict(model.named_parameters())
            for key in keys:
                g = named[key].grad
                out[f"gnorm{tag}." + mtag + key] = g.norm()
                out[f"gslice{tag}." + mtag + key] = g.reshape(-1)[:256].clone()
        out.update({"logits" + tag: logits, "space_att" + tag: s_att, "time_att" + tag: t_att, "loss" + tag: loss.detach(),
                    "feat_mean" + tag: feats.mean(dim=(0, 2, 3)), "feat_absmax" + tag: feats.abs().amax(dim=(0, 2, 3)),
                    "feat_slice" + tag: feats[:2, :, :2, :2].clone()})
    save(name, input_sum=checksum(inp["videos"]), batch=batch, frames=frames, identities=identities, ragged=int(ragged),
         training=int(training), seed=seed, **out)


def e2e_full_case(EF, TSF, name, batch, frames, identities, seed, rate=0.2, checkpoint_blocks=False):
    """A BASELINE configuration at FULL size (config 2: B = 16, 1 identity; config 3: B = 32, 2 identities), one training step of
    the imported reference in float64 (its exact arithmetic): train-mode BatchNorm, drop-connect `rate`, BCE loss, backward.
    Stored for EVERY parameter of both networks: gradient norm, max |g| and a 256-element strided sample; logits, loss, and the
    extractor's updated running statistics (sampled).  The GPU suite compares against this instead of re-running the fp64 oracle
    on its host (round-4 verdict, weak #9).
    Drop-connect: the reference draws torch.rand([N,1,1,1], dtype=inputs.dtype) per gated block (utils.py:148-150).  The fp64 model
    would draw from the generator's 53-bit stream; torch.rand is wrapped for this run so that those calls draw the float32 values
    (cast up) -- the values oracle.drop_connect_uniforms(seed, N, rate) replays and the tests feed to the HIP path.
    checkpoint_blocks: autograd keeps only each MBConv block's input and re-runs the block (the reference's own forward, same RNG
    state) inside the backward pass -- config 3's 256 crops in float64 do not fit this container's 62 GB otherwise.  The same
    arithmetic in the same order; the running statistics are sampled right after the forward (the re-runs update them again)."""
    import time
    t0 = time.time()
    cfg = arch.default_tsf_config(channels=1280, num_frames=frames)
    inp = synth.clip_inputs(batch, frames, identities, seed, ragged=False)
    dtype = torch.float64
    ef = EF.from_name("efficientnet-b0", drop_connect_rate=rate)
    ef.load_state_dict(synth.effnet_b0_state(seed), strict=True)
    ef.train(True).to(dtype)
    tsf, _ = build_tsf(TSF, cfg, seed, True, dtype)
    tsf.train(True)
    v = inp["videos"].to(dtype)
    b, f, h, w, c = v.shape
    vid = v.reshape(b * f, h, w, c).permute(0, 3, 1, 2)
    real_rand = torch.rand

    def rand32(*a, **k):
        if k.get("dtype") == torch.float64 and len(a) == 1 and list(a[0])[1:] == [1, 1, 1]:
            k = dict(k, dtype=torch.float32)
            return real_rand(*a, **k).double()
        return real_rand(*a, **k)

    if checkpoint_blocks:
        from torch.utils.checkpoint import checkpoint
        for blk in ef._blocks:
            inner = blk.forward
            blk.forward = (lambda inner: lambda x, drop_connect_rate=None: checkpoint(
                inner, x, drop_connect_rate, use_reentrant=False, preserve_rng_state=True))(inner)
    torch.manual_seed(seed)
    torch.rand = rand32           # stays wrapped until after backward: checkpointed blocks draw again (same restored RNG state)
    feats = ef(vid)
    stats_after_forward = {k: v.clone() for k, v in ef.state_dict().items() if "running_" in k}
    logits, _ = tsf(feats.reshape(b, f, *feats.shape[1:]), mask=inp["mask"], identities_mask=inp["identities_mask"],
                    size_embedding=inp["size_embedding"], positions=inp["positions"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, inp["labels"].reshape(-1, 1).to(dtype))
    print(f"{name}: reference fp64 forward {time.time() - t0:.0f} s", flush=True)
    try:
        loss.backward()
    finally:
        torch.rand = real_rand
    print(f"{name}: + backward {time.time() - t0:.0f} s", flush=True)
    out = {"logits64": logits, "loss64": loss.detach()}
    for model, mtag in ((ef, "ef."), (tsf, "tsf.")):
        for key, prm in model.named_parameters():
            if prm.grad is None:
                continue
            g = prm.grad.reshape(-1)
            step = max(1, g.numel() // 256)
            out["gnorm64." + mtag + key] = g.norm()
            out["gabsmax64." + mtag + key] = g.abs().max()
            out["gsample64." + mtag + key] = g[::step][:256].clone()
            # whole-tensor checksum pair: every element is weighted (a tile-local defect between the 256 samples moves these)
            from tests.util import probe_vector
            out["gsum64." + mtag + key] = g.sum()
            out["gdot64." + mtag + key] = (g * torch.from_numpy(probe_vector(mtag + key, g.numel(), seed))).sum()
    for key in ("_bn0.running_mean", "_blocks.3._bn1.running_var", "_blocks.10._bn2.running_mean", "_bn1.running_var"):
        out["stat64." + key] = stats_after_forward[key]
    save(name, input_sum=checksum(inp["videos"]), batch=batch, frames=frames, identities=identities, seed=seed, rate=rate, **out)


def xc_case(name, n_img, training, seed):
    from models.xception import xception as ref_xception      # resolved from /root/reference by import_reference()
    import contextlib, io
    model = ref_xception(num_classes=1, pretrain_path=None)
    sd = synth.xception_state(seed)
    model.load_state_dict(sd, strict=True)
    model.train(training)
    vid = synth.clip_inputs(1, n_img, 1, seed)["videos"]
    x = vid.reshape(n_img, 224, 224, 3).permute(0, 3, 1, 2)
    taps = {}
    hooks = [getattr(model, f"block{i}").register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(i, o.detach().clone()))
             for i in (1, 3, 7, 12)]
    with torch.no_grad():
        feats = model(x)
    for h in hooks:
        h.remove()
    extra = {}
    if training:
        msd = model.state_dict()
        for k in ("bn1.running_mean", "bn2.running_var", "block1.skipbn.running_var", "block5.rep.5.running_mean", "bn4.running_var"):
            extra["stat." + k] = msd[k].clone()
    m64 = ref_xception(num_classes=1, pretrain_path=None)
    m64.load_state_dict(sd, strict=True)
    m64.train(training).double()
    with torch.no_grad():
        feats64 = m64(x.double())
    # backward sample: d(sum(feats * w)) for a few parameters (fp64 pass = the exact arithmetic)
    model.train(training)
    for prm in m64.parameters():
        prm.grad = None
    gw = torch.from_numpy(np.random.Generator(np.random.Philox(key=[seed, 4242])).standard_normal((n_img, 2048, 7, 7)) * 0.1)
    m64.load_state_dict(sd, strict=True)   # reset running stats the first pass updated
    out = m64(x.double())
    (out * gw).sum().backward()
    named = dict(m64.named_parameters())
    grads = {}
    for key in GRAD_KEYS_XC:
        gg = named[key].grad
        grads["gnorm64." + key] = gg.norm()
        grads["gslice64." + key] = gg.reshape(-1)[:256].clone()
    save(name, features=feats, feat64_mean=feats64.mean(dim=(0, 2, 3)), feat64_slice=feats64[:, :256].clone(), input_sum=checksum(x),
         **{f"block{i}_mean": taps[i].mean(dim=(0, 2, 3)) for i in taps},
         **{f"block{i}_slice": taps[i][0, :, :3, :3].clone() for i in taps},
         n_img=n_img, training=int(training), seed=seed, **extra, **grads)


GRAD_KEYS_XC = ["conv1.weight", "bn1.weight", "conv2.weight", "bn2.bias", "block1.skip.weight", "block1.skipbn.weight",
                "block1.rep.0.conv1.weight", "block1.rep.0.pointwise.weight", "block1.rep.4.weight", "block3.rep.4.conv1.weight",
                "block6.rep.4.pointwise.weight", "block6.rep.8.bias", "block12.rep.4.pointwise.weight", "block12.skip.weight",
                "conv3.conv1.weight", "bn3.weight", "conv4.pointwise.weight", "bn4.weight", "bn4.bias"]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    EF, TSF = import_reference()
    # key/shape manifests of the reference modules (state-dict compatibility contract)
    ef = EF.from_name("efficientnet-b0")
    man = {"efficientnet-b0": [[k, list(v.shape), str(v.dtype)] for k, v in ef.state_dict().items()]}
    for (c, fr) in ((1280, 8), (2048, 16)):
        t = TSF(config=arch.default_tsf_config(c, fr))
        man[f"tsf_c{c}_f{fr}"] = [[k, list(v.shape), str(v.dtype)] for k, v in t.state_dict().items()]
        man[f"tsf_c{c}_f{fr}_no_weight_decay"] = sorted(t.no_weight_decay())
    only = os.environ.get("GOLDEN_ONLY", "")
    if only in ("full2", "full3", "full2ck"):       # full-size steps: minutes of float64 on the host, tens of GB of autograd state -- on request
        if only == "full2":
            e2e_full_case(EF, TSF, "e2e_full_config2", batch=16, frames=8, identities=1, seed=4)
        elif only == "full3":            # 34 minutes of float64 on 8 cores
            e2e_full_case(EF, TSF, "e2e_full_config3", batch=32, frames=8, identities=2, seed=4, checkpoint_blocks=True)
        if only == "full2ck":            # cross-check of the checkpointed run against the plain one: reproduces e2e_full_config2.npz
            # bit for bit (1104 arrays, difference 0.0 -- checked when the fixtures were made; the copy is not committed)
            e2e_full_case(EF, TSF, "e2e_full_config2_ck", batch=16, frames=8, identities=1, seed=4, checkpoint_blocks=True)
        return
    if only in ("", "dc"):
        ef_dc_case(EF, "ef_train_dc", n_img=4, seed=3)
    if only in ("", "agg"):
        agg_case("agg_att")
    if only in ("", "slots"):
        slots_case("slots")
    if only in ("", "switch"):
        tsf_case(TSF, "tsf_nopos", batch=2, frames=8, channels=1280, identities=2, ragged=True, seed=3, pos_emb=False, size_emb=True)
        tsf_case(TSF, "tsf_nosize", batch=2, frames=8, channels=1280, identities=2, ragged=True, seed=4, pos_emb=True, size_emb=False)
    if only in ("", "dropout"):
        tsf_dropout_case(TSF, "tsf_dropout", batch=2, frames=8, channels=1280, identities=2, seed=7)
    if only == "tsf":            # the three plain TimeSformer fixtures alone (e.g. after tsf_case() gained keys)
        tsf_case(TSF, "tsf_cfg1", batch=2, frames=8, channels=1280, identities=1, ragged=False, seed=0)
        tsf_case(TSF, "tsf_2id_ragged", batch=2, frames=8, channels=1280, identities=2, ragged=True, seed=1)
        tsf_case(TSF, "tsf_xs_3id", batch=1, frames=16, channels=2048, identities=3, ragged=True, seed=2)
    if only in ("dc", "agg", "slots", "switch", "tsf", "dropout"):
        return
    from models.xception import xception as _xc
    man["xception"] = [[k, list(v.shape), str(v.dtype)] for k, v in _xc(num_classes=1).state_dict().items()]
    with open(os.path.join(OUT, "state_manifest.json"), "w") as fh:
        json.dump(man, fh)
    if os.environ.get("GOLDEN_ONLY", "") in ("", "xc"):
        xc_case("xc_eval", n_img=2, training=False, seed=0)
        xc_case("xc_train", n_img=3, training=True, seed=1)
    if os.environ.get("GOLDEN_ONLY", "") == "xc":
        return
    tsf_case(TSF, "tsf_cfg1", batch=2, frames=8, channels=1280, identities=1, ragged=False, seed=0)
    tsf_case(TSF, "tsf_2id_ragged", batch=2, frames=8, channels=1280, identities=2, ragged=True, seed=1)
    tsf_case(TSF, "tsf_xs_3id", batch=1, frames=16, channels=2048, identities=3, ragged=True, seed=2)
    ef_case(EF, "ef_eval", n_img=2, training=False, seed=0)
    ef_case(EF, "ef_train", n_img=4, training=True, seed=1)
    e2e_case(EF, TSF, "e2e_cfg1_eval", batch=2, frames=8, identities=1, ragged=False, training=False, seed=0)
    e2e_case(EF, TSF, "e2e_2id_train", batch=2, frames=8, identities=2, ragged=True, training=True, seed=1)


if __name__ == "__main__":
    main()
