#!/usr/bin/env python3
"""Quick timing of the EfficientNet-B0 forward (and backward when available) on one GPU."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mintime_amd
from mintime_amd import synth, EfficientNet

ap = argparse.ArgumentParser()
ap.add_argument("--crops", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--eval", action="store_true")
a = ap.parse_args()
m = EfficientNet.from_name("efficientnet-b0")
m.load_state_dict(synth.effnet_b0_state(0))
m.cuda().train(not a.eval)
x = torch.randint(0, 256, (a.crops, 224, 224, 3), device="cuda").float().permute(0, 3, 1, 2)


def step():
    if a.bwd:
        for p in m.parameters():
            p.grad = None       # like optimizer.zero_grad(set_to_none=True): the engines' gradient views are adopted, not accumulated
        m(x).sum().backward()
    else:
        with torch.no_grad():
            m(x)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print(f"crops={a.crops} {'fwd+bwd' if a.bwd else 'fwd'} {'eval' if a.eval else 'train'}: {dt*1e3:.2f} ms/iter  {a.crops/dt:.0f} crops/s")
