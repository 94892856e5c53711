#!/usr/bin/env python3
"""Quick timing of the TimeSformer forward (and backward when available) on one GPU."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mintime_amd
from mintime_amd import arch, synth, SizeInvariantTimeSformer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--bwd", action="store_true")
a = ap.parse_args()
cfg = arch.default_tsf_config(1280, 8)
model = SizeInvariantTimeSformer(config=cfg)
model.load_state_dict(synth.tsf_state(cfg, 0))
model.cuda()
B = a.batch
feats = synth.features(2, 8, 1280, 0).repeat(B // 2, 1, 1, 1, 1).cuda()
feats = feats.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
aux = synth.clip_inputs(B, 8, 2, 0, with_video=False)
kw = dict(mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(), size_embedding=aux["size_embedding"],
          positions=aux["positions"].cuda())


def step():
    if a.bwd:
        for p in model.parameters():
            p.grad = None       # like optimizer.zero_grad(set_to_none=True): the engines' gradient views are adopted, not accumulated
        out = model(feats, **kw)
        out.sum().backward()
    else:
        with torch.no_grad():
            model(feats, **kw)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
flops = 2 * 19021359104 * B * (3 if a.bwd else 1)
print(f"B={B} {'fwd+bwd' if a.bwd else 'fwd'}: {dt*1e3:.2f} ms/iter  {B/dt:.1f} clips/s  {flops/dt/1e12:.1f} TFLOP/s (TSF only)")
