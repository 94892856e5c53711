#!/usr/bin/env python3
"""Stability soak: N training steps at B=32 with fresh inputs every step; reports loss trend, step-time drift and allocator growth."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import harness
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg, ef, tsf = harness.build_models(seed=0)
opt = harness.make_optimizer(cfg, ef, tsf)
batches = [harness.device_batch(32, seed=s, ragged=(s % 2 == 1)) for s in range(4)]
losses, times, mem = [], [], []
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = harness.train_step(ef, tsf, opt, batches[i % 4])
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    losses.append(float(loss)); mem.append(torch.cuda.memory_allocated() / 2**30)
    if not (losses[-1] == losses[-1]) or abs(losses[-1]) > 1e4:
        print("DIVERGED at step", i, losses[-5:]); break
q = max(1, steps // 4)
print(f"steps {len(losses)}  loss first/last quarter {sum(losses[:q]) / q:.4f} -> {sum(losses[-q:]) / q:.4f}")
print(f"ms/step first/last quarter {1e3 * sum(times[5:q]) / max(1, q - 5):.2f} -> {1e3 * sum(times[-q:]) / q:.2f}")
print(f"allocated GiB after step 5 / last: {mem[min(5, len(mem) - 1)]:.2f} / {mem[-1]:.2f}; peak {torch.cuda.max_memory_allocated() / 2**30:.2f}; reserved {torch.cuda.memory_reserved() / 2**30:.2f}")
