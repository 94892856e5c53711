#!/usr/bin/env python3
"""Replays every distinct mt_gemm call of one B=32 training step under each tile configuration (MT_FORCE_CFG) and prints
the time per configuration next to the dispatcher's own choice -- the evidence behind pick_cfg's rules in csrc/gemm.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import harness, lib as L

B = int(os.environ.get("TUNE_BATCH", "32"))
cfg, ef, tsf = harness.build_models(seed=0)
opt = harness.make_optimizer(cfg, ef, tsf)
batch = harness.device_batch(B, seed=0)
os.environ["MT_SIDE_STREAM"] = "0"
harness.train_step(ef, tsf, opt, batch)
torch.cuda.synchronize()

calls, count = {}, {}
orig = L.gemm


def rec(op, A, Bm, Cout, M, N, K, lda, ldb, ldc, **kw):
    key = (op, M, N, K, kw.get("prologue", 0), kw.get("epilogue", 0), kw.get("b_prologue", 0), kw.get("split_k", 1),
           kw.get("conv") is not None)
    count[key] = count.get(key, 0) + 1
    if key not in calls:
        calls[key] = ((op, A, Bm, Cout, M, N, K, lda, ldb, ldc), dict(kw))
    return orig(op, A, Bm, Cout, M, N, K, lda, ldb, ldc, **kw)


L.gemm = rec
for m in (mintime_amd.tsf_engine, mintime_amd.tsf_backward, mintime_amd.effnet_engine, mintime_amd.effnet_backward):
    pass   # the engines call L.gemm through the module attribute, so patching lib is enough
harness.train_step(ef, tsf, opt, batch)
torch.cuda.synchronize()
L.gemm = orig


def timed(args, kw, iters=6):
    for _ in range(2):
        orig(*args, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        orig(*args, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


names = {None: "auto", 0: "128x128", 1: "128x64", 2: "256x32", 3: "64x64"}
total_auto = total_best = 0.0
rows = []
for key, (args, kw) in calls.items():
    res = {}
    for c in (None, 0, 1, 2, 3):
        if c is None:
            os.environ.pop("MT_FORCE_CFG", None)
        else:
            os.environ["MT_FORCE_CFG"] = str(c)
        try:
            res[c] = timed(args, kw)
        except Exception:
            res[c] = float("inf")
    os.environ.pop("MT_FORCE_CFG", None)
    best = min((c for c in res if c is not None), key=lambda c: res[c])
    n = count[key]
    total_auto += n * res[None]
    total_best += n * min(res[None], res[best])
    rows.append((n * (res[None] - min(res[None], res[best])), key, n, res, best))
rows.sort(key=lambda r: -r[0])
print(f"{len(calls)} distinct GEMMs; per step: auto {total_auto / 1e3:.2f} ms, best-of-configs {total_best / 1e3:.2f} ms")
print("gain/step  calls  op  M N K pro epi bpro splitk conv | auto " + " ".join(names[c] for c in (0, 1, 2, 3)))
for gain, key, n, res, best in rows:
    print(f"{gain:8.1f}us {n:3d}  {key} | {res[None]:7.1f} " + " ".join(f"{res[c]:7.1f}" for c in (0, 1, 2, 3)) + f"  best {names[best]}")
