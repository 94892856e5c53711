#!/usr/bin/env python3
"""Host-side enqueue time of one training step (no device sync inside the timed region) vs the device time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import harness

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg, ef, tsf = harness.build_models(seed=0, device="cuda")
opt = harness.make_optimizer(cfg, ef, tsf)
batch = harness.device_batch(B, seed=0)
for _ in range(3):
    harness.train_step(ef, tsf, opt, batch)
torch.cuda.synchronize()
host = []
t_all0 = time.perf_counter()
for _ in range(10):
    t0 = time.perf_counter()
    harness.train_step(ef, tsf, opt, batch)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
total = (time.perf_counter() - t_all0) / 10
print(f"B={B}: device-bound step {total*1e3:.1f} ms; host enqueue per step: first {host[0]*1e3:.1f} ms, median {sorted(host)[5]*1e3:.1f} ms "
      f"(once the launch queue is full the host blocks, so only the FIRST step's figure is pure host cost)")
# phases of host time in the first step after a sync
torch.cuda.synchronize()
t0 = time.perf_counter(); y = harness.forward(ef, tsf, batch); t1 = time.perf_counter()
loss = mintime_amd.optim.bce_with_logits(y, batch["labels"], None); opt.zero_grad(set_to_none=True); loss.backward(); t2 = time.perf_counter()
opt.step(); t3 = time.perf_counter()
torch.cuda.synchronize(); t4 = time.perf_counter()
print(f"host: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, sgd {1e3*(t3-t2):.1f} ms, drain {1e3*(t4-t3):.1f} ms")
