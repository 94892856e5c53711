#!/usr/bin/env python3
"""Is the host ahead of the device in the steady-state training loop?  Host time at which each step's enqueue RETURNS vs the device
time at which that step ends (events).  If step k's enqueue returns after step k-1 has finished on the device, something in the
step blocks the host (a synchronising runtime call) and the device idles at that point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import harness

dev = "cuda:0"
cfg, ef, tsf = harness.build_models(8, seed=0, device=dev)
opt = harness.make_optimizer(cfg, ef, tsf)
batch = harness.device_batch(32, 8, 2, seed=0, device=dev)
if os.environ.get("SE_ON_DEVICE"):
    batch["size_embedding"] = batch["size_embedding"].to(dev)
for _ in range(6):
    harness.train_step(ef, tsf, opt, batch)
torch.cuda.synchronize()
K = 12
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
evs[0].record()
t0 = time.perf_counter()
for k in range(K):
    harness.train_step(ef, tsf, opt, batch)
    evs[k + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
dev_end = [evs[0].elapsed_time(evs[k + 1]) for k in range(K)]
for k in range(K):
    print(f"step {k:2d}: host enqueue returned at {host[k] * 1e3:8.2f} ms, device finished it at {dev_end[k]:8.2f} ms, lead {dev_end[k] - host[k] * 1e3:8.2f} ms")
