#!/usr/bin/env python3
"""Small-batch eval latency: eager launch sequence vs HIP-graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import harness
for B in (1, 2, 8):
    cfg, ef, tsf = harness.build_models(8, 0, "cuda", train_extractor=False)
    ef.eval(); tsf.eval()
    batch = harness.device_batch(B, 8, 2, 0, "cuda")
    def timeit(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        t_eager = timeit(lambda: harness.forward(ef, tsf, batch))
    g = harness.GraphedEval(ef, tsf, batch)
    t_graph = timeit(lambda: g(batch))
    print(f"B={B}: eager {t_eager:.2f} ms ({B/t_eager*1e3:.0f} clips/s)   HIP graph {t_graph:.2f} ms ({B/t_graph*1e3:.0f} clips/s)")
