"""Lab: per-step wall time of the first steps of a process at BASELINE config 5 (Xception, 512 crops), with the caching allocator's
segment traffic per step (hipMalloc / hipFree counts, reserved bytes): why are the first ~10 steps of a process 3-12x slow?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mintime_amd
from mintime_amd import harness
import bench

cfgn = int(os.environ.get("CONFIG", 5))
wl = bench.WORKLOADS[cfgn]
dev = "cuda:0"
B, frames = wl["B"], wl["frames"]
cfg, ef, tsf = (harness.build_models_xs if wl["extractor"] == "xception" else harness.build_models)(frames, seed=0, device=dev)
opt = harness.make_optimizer(cfg, ef, tsf)
batch = harness.device_batch(B, frames, wl["ids"], seed=0, device=dev)
SYNC = os.environ.get("SYNC", "1") != "0"      # 0: the host runs ahead of the device, as in bench.py's loops
prev = torch.cuda.memory_stats()
for i in range(int(os.environ.get("STEPS", 16))):
    if SYNC:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    harness.train_step(ef, tsf, opt, batch)
    t1 = time.perf_counter()
    if SYNC:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    d = lambda k: st.get(k, 0) - prev.get(k, 0)
    print(f"step {i:2d}: {1e3 * (t2 - t0):8.1f} ms (host {1e3 * (t1 - t0):8.1f})  segments +{d('segment.all.allocated'):3d} -{d('segment.all.freed'):3d}  "
          f"retries {d('num_alloc_retries')}  reserved {st['reserved_bytes.all.current'] / 2**30:6.1f} GiB  peak alloc {st['allocated_bytes.all.peak'] / 2**30:6.1f} GiB",
          flush=True)
    prev = st
torch.cuda.synchronize()
print("end: retries", torch.cuda.memory_stats().get("num_alloc_retries"), "reserved peak", torch.cuda.memory_stats()["reserved_bytes.all.peak"] / 2**30, flush=True)
