#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import mintime_amd
from mintime_amd import lib as L, arch, synth, tsf_engine, SizeInvariantTimeSformer

cfg = arch.default_tsf_config(1280, 8)
model = SizeInvariantTimeSformer(config=cfg)
model.load_state_dict(synth.tsf_state(cfg, 0))
model = model.cuda().eval()
orig_new = tsf_engine._new
for fill in ("empty", "zero", "nan", "big"):
    if fill == "empty":
        tsf_engine._new = orig_new
    elif fill == "zero":
        tsf_engine._new = lambda dev, *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
    elif fill == "nan":
        tsf_engine._new = lambda dev, *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
    else:
        tsf_engine._new = lambda dev, *shape: torch.full(shape, 1e30, dtype=torch.float32, device=dev)
    for B in (8, 11, 16):
        feats = synth.features(B, 8, 1280, 1).cuda()
        aux = synth.clip_inputs(B, 8, 2, 1, ragged=True, with_video=False)
        with torch.no_grad():
            o = [model(feats, mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(), size_embedding=aux["size_embedding"],
                       positions=aux["positions"].cuda()).clone() for _ in range(3)]
        print(f"fill={fill:5s} B={B}: identical {torch.equal(o[0], o[1]) and torch.equal(o[0], o[2])} finite {bool(torch.isfinite(o[0]).all())} "
              f"maxdiff {float((o[0]-o[1]).abs().max()):.3e} out0 {float(o[0][0,0]):.6f}", flush=True)
