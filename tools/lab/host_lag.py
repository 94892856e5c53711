#!/usr/bin/env python3
"""How far the host runs ahead of the device at the phase boundaries of a train step (config 3): host wall-clock and a HIP
event are taken at entry / exit of the four engine functions; after the run, device time - host time at each mark = the lead
the host had when it enqueued that point.  A lead near zero means the device was waiting for the host there."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import harness, lib, tsf_backward, effnet_backward, tsf_engine, effnet_engine

dev = torch.device("cuda:0")
lib.get()
cfg, ef, tsf = harness.build_models(8, seed=0, device=dev)
opt = harness.make_optimizer(cfg, ef, tsf)
batch = harness.device_batch(32, 8, 2, seed=0, device=dev)
marks = []

def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, time.perf_counter(), e))

def wrap(mod, fn, tag):
    orig = getattr(mod, fn)
    def f(*a, **k):
        mark(tag + ":in")
        r = orig(*a, **k)
        mark(tag + ":out")
        return r
    setattr(mod, fn, f)

wrap(effnet_engine, "effnet_forward", "ef_fwd")
wrap(tsf_engine, "tsf_forward", "tsf_fwd")
wrap(tsf_engine, "tsf_backward", "tsf_bwd") if hasattr(tsf_engine, "tsf_backward") else wrap(tsf_backward, "tsf_backward", "tsf_bwd")
wrap(effnet_engine, "effnet_backward", "ef_bwd") if hasattr(effnet_engine, "effnet_backward") else wrap(effnet_backward, "effnet_backward", "ef_bwd")

_pool_init = effnet_engine._StatsPool.__init__
def _pi(self, dev, total):
    mark("pool:pre")
    _pool_init(self, dev, total)
    mark("pool:post")
effnet_engine._StatsPool.__init__ = _pi
_fin = effnet_engine._finalize
_cnt = [0]
def _f(*a, **k):
    r = _fin(*a, **k)
    if _cnt[0] == 0:
        mark("stem:done")
    _cnt[0] += 1
    return r
effnet_engine._finalize = _f
for _ in range(5):
    harness.train_step(ef, tsf, opt, batch)
torch.cuda.synchronize()
marks.clear()
steps = 6
from mintime_amd import optim
for s in range(steps):
    mark("step%d" % s)
    _cnt[0] = 0
    y = harness.forward(ef, tsf, batch)
    y = y[0] if isinstance(y, tuple) else y
    mark("fwd_done")
    loss = optim.bce_with_logits(y, batch["labels"], None)
    opt.zero_grad(set_to_none=True)
    mark("pre_bwd")
    loss.backward()
    mark("bwd_done")
    opt.step()
    mark("opt_done")
mark("end")
torch.cuda.synchronize()
h0, e0 = marks[0][1], marks[0][2]
print(f"{'mark':>14} {'host ms':>9} {'device ms':>10} {'host lead ms':>12}")
for name, h, e in marks:
    hd, dd = (h - h0) * 1e3, e0.elapsed_time(e)
    print(f"{name:>14} {hd:9.2f} {dd:10.2f} {dd - hd:12.2f}")
