#!/usr/bin/env python3
"""Headline benchmark: clips/s of one MINTIME training step (EfficientNet-B0 -> SizeInvariantTimeSformer,
forward + backward + gradient all-reduce + SGD step) on 8-frame, 2-identity, 224x224 synthetic clips.

    python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel (FF1 GEMM with fused GEGLU, fp32 MFMA) timed live with HIP events on its launch stream
  cpu_baseline  the CPU oracle (restatement pinned to the reference) timed on this host's cores, rank 0, N == 1 only
"""
import argparse
import json
import os
import sys
import time

# Before the HIP runtime starts: with the default 4 hardware queues, merely creating an RCCL communicator re-maps HIP streams so
# that the weight-gradient side stream stops overlapping the main stream (+6.5 ms/step measured, backward phases only); 8 queues
# keep them apart (and cost nothing without RCCL).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FLOP_PER_CLIP_FWD = 2 * 22097637120          # BASELINE.md §2 (EF 3.076 G MAC + TSF 19.021 G MAC per 8-frame clip)
FLOP_PER_CLIP_STEP = 3 * FLOP_PER_CLIP_FWD   # backward = 2x forward MACs
PEAK_FP32_MFMA = 157.3e12                    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline_subprocess(num_frames, timeout_s=150):
    """Run the CPU leg in its own process under a hard timeout so a slow host can never stall the benchmark."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--frames", str(num_frames)],
                           capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "clips/s", "cores": usable_cores(), "kind": "port", "sample": "failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "clips/s", "cores": usable_cores(), "kind": "port", "sample": f"timed out after {timeout_s}s"}


def cpu_baseline(num_frames, seed, budget_s=20.0):
    """Reference CPU path (the oracle: plain torch fp32 ops + torch autograd) on a bounded sample of the workload."""
    import mintime_amd
    from mintime_amd import arch, synth
    from oracle import mintime_oracle as O
    # torch's intra-op pool stops scaling long before a 256-core host is full on these small tensors (and 256 threads
    # on B=4 is slower than 16: measured 4.0 / 3.0 / 1.4 clips/s at 16 / 32 / 64 threads); MT_CPU_THREADS overrides.  `cores` in the result is what was actually used.
    torch.set_num_threads(int(os.environ.get("MT_CPU_THREADS", min(usable_cores(), 16))))
    cfg = arch.default_tsf_config(1280, num_frames)
    B = 4
    ef = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k and not k.startswith("_fc"))
          for k, v in synth.effnet_b0_state(seed).items()}
    ts = {k: v.clone().requires_grad_(True) for k, v in synth.tsf_state(cfg, seed).items()}
    inp = synth.clip_inputs(B, num_frames, 2, seed)

    def step():
        for d in (ef, ts):
            for v in d.values():
                v.grad = None
        out = O.clip_forward(ef, ts, cfg, inp, training_extractor=True)
        O.bce_with_logits(out, inp["labels"]).backward()

    step()                                     # warm-up (thread pools, allocator)
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 8:
            break
    return {"value": round(B * n / dt, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} fwd+bwd steps of {B} clips (8-frame, 2-identity, 224x224), train-mode BN, fp32 torch CPU ops"}


def pmc_traffic(B):
    """HBM/fabric bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes (separate runs, gfx950
    correction applied; profiles/r01_ff1_geglu_gemm_pmc.json).  Only valid for the shape they were taken on (B = 32)."""
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_ff1_geglu_gemm_pmc.json")
    if B != 32 or not os.path.exists(f):
        return {"traffic": None}
    d = json.load(open(f))
    return {"traffic": d["traffic_bytes_per_launch"], "traffic_unit": "bytes/launch",
            "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (profiles/r01_ff1_geglu_gemm_pmc.json)",
            "algorithmic_bytes": sum(d["algorithmic_bytes_per_launch"].values())}


def attention_modules_leg(dev, B, F=8, reps=3):
    """north_star sub-metric: forward of the divided space-time attention MODULES (QKV GEMM + attention core + out-proj GEMM,
    time and space, 9 layers = 18 modules) at batch B, as a fraction of the fp32 MFMA peak.  Timed with HIP events."""
    from mintime_amd import lib
    h = lib.get()
    D, H, n = 512, 8, 49
    N = 1 + F * n
    M = B * N
    g = torch.Generator(device=dev).manual_seed(7)
    xn = torch.randn(M, D, device=dev, generator=g)
    wqkv = torch.randn(3 * D, D, device=dev, generator=g) * 0.08
    wo, bo = torch.randn(D, D, device=dev, generator=g) * 0.03, torch.randn(D, device=dev, generator=g) * 0.02
    qkv, o, x = torch.empty(M, 3 * D, device=dev), torch.empty(M, D, device=dev), torch.randn(M, D, device=dev, generator=g)
    mask = torch.ones(B, F, dtype=torch.uint8, device=dev)
    ident = torch.block_diag(torch.ones(F // 2, F // 2), torch.ones(F - F // 2, F - F // 2)).to(torch.uint8).to(dev).repeat(B, 1, 1).contiguous()

    def modules():
        for layer in range(9):
            for mode in (0, 1):
                lib.gemm(lib.OP_NT, xn, wqkv, qkv, M, 3 * D, D, D, D, 3 * D)
                lib.check(h.mt_attn_fwd(lib.ptr(qkv), lib.ptr(o), None, lib.ptr(mask), lib.ptr(ident), B, H, F, n, mode, 0.125,
                                        lib.stream_ptr()), "mt_attn_fwd")
                lib.gemm(lib.OP_NT, o, wo, x, M, D, D, D, D, D, epilogue=lib.EPI_BIAS_RES, bias=bo, R=x, ldr=D)

    modules()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        modules()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = B * 2 * 7638018048            # attention-module MACs per clip (QKV + core + out-proj, x18)
    return {"batch": B, "fwd_ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 2),
            "mfma_frac": round(flops / (ms * 1e-3) / PEAK_FP32_MFMA, 4), "target_frac": 0.40}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-reducer", action="store_true",
                    help="single-GPU check of the multi-GPU step: 1-rank RCCL group + the overlapped gradient reducer")
    a = ap.parse_args()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(a.frames, 0)))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1 or a.force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    import mintime_amd
    from mintime_amd import harness, lib, ddp
    lib.get()
    B = a.batch_per_gpu
    cfg, ef, tsf = harness.build_models(a.frames, seed=0, device=dev)            # train-mode BN + drop-connect 0.2 (train.py:157)
    opt = harness.make_optimizer(cfg, ef, tsf)
    batch = harness.device_batch(B, a.frames, 2, seed=rank, device=dev)          # config 3 masks: 2 identities [4,4]
    # buckets in the order backward finishes them: the TimeSformer's 48 M gradients all-reduce under the EfficientNet backward
    reducer = (ddp.OverlappedGradReducer([tsf, ef], force=a.force_reducer)
               if world > 1 or a.force_reducer else None)

    def step():
        return harness.train_step(ef, tsf, opt, batch, reducer)

    if reducer is not None:
        try:                                 # never lose a scaling run to the overlapped path: fall back to the plain flat all-reduce
            step()
            torch.cuda.synchronize()
        except Exception as e:               # noqa: BLE001
            print(f"[bench] overlapped gradient reducer failed ({type(e).__name__}: {e}); using the plain flat all-reduce",
                  file=sys.stderr)
            ef._grads_ready_hook = tsf._grads_ready_hook = None
            opt.zero_grad(set_to_none=True)
            reducer = ddp.GradAllReducer(list(ef.parameters()) + list(tsf.parameters()))
    for _ in range(a.warmup):
        step()
    # live timing of the dominant kernel: FF1 GEMM + GEGLU epilogue (9 launches per step), HIP events on its stream
    prof = {"match": lambda d: d.epilogue == lib.EPI_GEGLU, "events": []}
    lib.PROFILE = prof
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.PROFILE = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = None
    if rank == 0:
        ms = 1e3 * dt / a.steps
        clips_s = world * B * a.steps / dt
        durs = [e0.elapsed_time(e1) * 1e-3 for e0, e1, _ in prof["events"]]
        flops = prof["events"][0][2] if prof["events"] else 0.0
        avg = sum(durs) / max(len(durs), 1)
        achieved = flops / avg / 1e12 if avg > 0 else 0.0
        out = {
            "metric": "clips/sec (8-frame, 2-identity, 224^2 crops) fwd+bwd", "value": round(clips_s, 2), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config 3: EfficientNet-B0 + SizeInvariantTimeSformer(dim 512, depth 9, heads 8), "
                                   f"B={B}/GPU, {a.frames} frames, 2 identities [4,4] identity-masked, 224x224 crops, random-init "
                                   "seeded weights, train-mode BN + drop-connect, step = fwd + BCE loss + bwd"
                                   + (" + RCCL grad all-reduce" if world > 1 else "") + " + SGD(lr .01, wd 1e-4)",
                       "global_batch": world * B, "frames": a.frames, "parallelism": f"dp{world}",
                       "model_tflops": round(clips_s * FLOP_PER_CLIP_STEP / 1e12, 2),
                       "model_mfma_frac": round(clips_s * FLOP_PER_CLIP_STEP / (world * PEAK_FP32_MFMA), 4),
                       "loss": round(float(loss.item()), 5)},
            "roofline": {"bound": "mfma", "kernel": "mt::gemm_kernel<2,2,2,2,NT,EPI_GEGLU> (FF1 512->4096 + GEGLU, M=B*393)",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved / (PEAK_FP32_MFMA / 1e12), 4), **pmc_traffic(B),
                         "launches_timed": len(durs), "avg_launch_us": round(avg * 1e6, 1),
                         "flops_per_launch": flops},
        }
        if world == 1:
            out["attention_modules"] = attention_modules_leg(dev, B, a.frames)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(a.frames)
    line = json.dumps(out) if rank == 0 else None
    if world > 1 or a.force_reducer:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        import ctypes
        ctypes.CDLL(None).fflush(None)   # RCCL's version banner sits in C stdio's buffer and would otherwise land after the line
        sys.stdout.flush()
        print(line, flush=True)          # the last thing on stdout


if __name__ == "__main__":
    main()
