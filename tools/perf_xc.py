#!/usr/bin/env python3
"""Quick timing of the Xception extractor (config 5) forward / forward+backward on one GPU."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mintime_amd
from mintime_amd import synth, xception

ap = argparse.ArgumentParser()
ap.add_argument("--crops", type=int, default=128)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--bwd", action="store_true")
a = ap.parse_args()
m = xception(num_classes=1, pretrain_path=None)
m.load_state_dict(synth.xception_state(0))
m.cuda().train(True)
x = torch.randint(0, 256, (a.crops, 224, 224, 3), device="cuda").float().permute(0, 3, 1, 2)


def step():
    if a.bwd:
        for p in m.parameters():
            p.grad = None
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
gflop = 8.4 * a.crops * (3 if a.bwd else 1)       # ~4.2 GMAC per 224^2 crop up to bn4
print(f"crops={a.crops} {'fwd+bwd' if a.bwd else 'fwd'}: {dt*1e3:.2f} ms/iter  {a.crops/dt:.0f} crops/s  {gflop/dt/1e3:.1f} TFLOP/s")
