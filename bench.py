#!/usr/bin/env python3
"""Headline benchmark: clips/s of one MINTIME training step (EfficientNet-B0 -> SizeInvariantTimeSformer,
forward + backward + gradient all-reduce + SGD step) on 8-frame, 2-identity, 224x224 synthetic clips.

    python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline       the TIME-DOMINANT kernel family of the step (the TimeSformer's weight-gradient GEMMs, fp32 MFMA), timed live
                 with HIP events on the stream it is launched on, over the timed region
  roofline_ff1   the largest single GEMM (FF1 + GEGLU epilogue), same method (round 1's roofline object, kept for continuity)
  roofline_hbm   the dominant HBM-bound kernel family (EfficientNet depthwise data gradient): algorithmic GB/s vs 8 TB/s
  phases         event-timed phases of the step + the fraction of the fp32-MFMA peak each network reaches
  forward_only   inference clips/s (eval-mode and train-mode BatchNorm), seeds: ms/step on synthetic seeds 0/1/2 + median
  cpu_baseline   the CPU oracle (restatement pinned to the reference) timed on this host's cores, rank 0, N == 1 only
--config 2 | 5 time BASELINE.json's other single-GPU configurations (B=16 1 identity; Xception "XS" B=32 x 16 frames x 3 ids).
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
PEAK_BF16_MFMA = 2500e12                     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
SPLIT_PIPE = ("split-operand fp32 (csrc/gemm_planes.hpp, gemm_split.hpp): each fp32 operand = 3 exact bf16 pieces (TimeSformer Linear layers: "
              "written once by the producing kernel as plane tensors, both GEMM operands by LDS-DMA), 6 piece products per fp32 product on "
              "v_mfma_f32_32x32x16_bf16, fp32 accumulators; error vs fp64 equal to the fp32 MFMA pipe's (tests/test_gpu_gemm.py, test_gpu_planes.py)")


PEAK_SPLIT_PIPE = PEAK_BF16_MFMA / 6          # fp32-equivalent ceiling of the split-operand loop: 6 bf16 MFMA flops per fp32 flop


def mfma_roofline(achieved_flops_s, split_on):
    """`achieved` / `peak` / `frac` of a GEMM family against the ceiling of the matrix pipe it ACTUALLY ran on (SURVEY section 7:
    the denominator must match the dtype used).  Split-operand loop: every fp32 product costs six bf16 MFMA products, so the
    ceiling is 2500 / 6 = 416.7 fp32-equivalent TFLOP/s and `frac` is how busy the bf16 matrix cores are; the figure against the
    fp32 MFMA peak (157.3, which this pipe can exceed) is kept as `frac_vs_fp32_mfma` and is NOT a roofline fraction."""
    a = achieved_flops_s
    peak = PEAK_SPLIT_PIPE if split_on else PEAK_FP32_MFMA
    out = {"pipe": SPLIT_PIPE if split_on else "v_mfma_f32_32x32x2_f32",
           "achieved": round(a / 1e12, 2) if a else None, "peak": round(peak / 1e12, 1), "unit": "TFLOP/s",
           "frac": round(a / peak, 4) if a else None}
    if split_on:
        out.update(peak_note="2500 TFLOP/s dense bf16 MFMA / 6 piece products per fp32 product", pipe_flops_per_flop=6,
                   frac_vs_fp32_mfma=round(a / PEAK_FP32_MFMA, 4) if a else None)
    return out


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
    res = {"value": round(B * n / dt, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n} fwd+bwd steps of {B} clips (8-frame, 2-identity, 224x224), train-mode BN, fp32 torch CPU ops"}
    # BASELINE config 1: the reference's own CPU-runnable case (2 clips, 1 identity, eval forward only)
    inp1 = synth.clip_inputs(2, num_frames, 1, seed)
    with torch.no_grad():
        O.clip_forward(ef, ts, cfg, inp1, training_extractor=False)
        t1, m = time.perf_counter(), 0
        while time.perf_counter() - t1 < 4.0 and m < 10:
            O.clip_forward(ef, ts, cfg, inp1, training_extractor=False)
            m += 1
    res["config1_forward_clips_s"] = round(2 * m / (time.perf_counter() - t1), 3)
    return res


PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_families.json")
_PMC_WARNED = []


def csrc_hash():
    """sha256 (16 hex digits) over the kernel sources the PMC passes were collected on (tools/pmc_report.py stores the same value)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(pkg, "*.hip")) + glob.glob(os.path.join(pkg, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def committed_counters(key):
    """Per-launch HBM-side bytes of a kernel family from the committed rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE
    runs, gfx950 correction applied; written by tools/pmc_report.py --json-out).  Valid for the default workload only, and only
    for the kernel sources they were measured on: the file carries a hash of csrc/, and when that differs from the tree being
    timed the counters are dropped (traffic = null) with a warning instead of being reported as current."""
    if not os.path.exists(PMC_FILE):
        return None
    d = json.load(open(PMC_FILE))
    if d.get("csrc_sha16") != csrc_hash():
        if not _PMC_WARNED:
            _PMC_WARNED.append(1)
            print(f"[bench] {os.path.basename(PMC_FILE)} was measured on csrc {d.get('csrc_sha16')} but this tree is {csrc_hash()}: "
                  "stale PMC counters dropped (traffic = null); re-run tools/final_profiles.sh", file=sys.stderr)
        return None
    return d.get(key)


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

    xn_p, wqkv_p, wo_p = lib.split_planes_blk(xn), lib.split_planes_blk(wqkv), lib.split_planes_blk(wo)
    o_p = lib.planes_empty(M, D, dev)
    planes = [False]

    def modules():
        for layer in range(9):
            for mode in (0, 1):
                if planes[0]:       # the engine's default on the split pipe (tsf_planes.py): plane operands, o leaves the attention kernels as planes
                    lib.gemm_planes(lib.OP_NT, xn_p, wqkv_p, M, 3 * D, D, Cout=qkv, ldc=3 * D)
                    lib.check(h.mt_attn_fwd(lib.ptr(qkv), None, None, lib.ptr(mask), lib.ptr(ident), B, H, F, n, mode, 0.125,
                                            lib.ptr(o_p), lib.stream_ptr()), "mt_attn_fwd")
                    lib.gemm_planes(lib.OP_NT, o_p, wo_p, M, D, D, Cout=x, ldc=D, epilogue=lib.EPI_BIAS_RES, bias=bo, R=x, ldr=D)
                    continue
                lib.gemm(lib.OP_NT, xn, wqkv, qkv, M, 3 * D, D, D, D, 3 * D)
                lib.check(h.mt_attn_fwd(lib.ptr(qkv), lib.ptr(o), None, lib.ptr(mask), lib.ptr(ident), B, H, F, n, mode, 0.125,
                                        None, lib.stream_ptr()), "mt_attn_fwd")
                lib.gemm(lib.OP_NT, o, wo, x, M, D, D, D, D, D, epilogue=lib.EPI_BIAS_RES, bias=bo, R=x, ldr=D)

    flops = B * 2 * 7638018048            # attention-module MACs per clip (QKV + core + out-proj, x18)

    def timed_leg():
        modules()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            modules()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # north_star's figure is MFMA UTILISATION: measured on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32 everywhere) against its
    # 157.3 TFLOP/s peak.  The default (split-operand) pipe is timed too and priced against ITS ceiling (2500/6 TFLOP/s).
    was = lib.gemm_split_enabled()
    lib.set_gemm_split(False)
    ms32 = timed_leg()
    out = {"batch": B, "target_frac": 0.40,
           "fp32_mfma_pipe": {"fwd_ms": round(ms32, 3), "tflops": round(flops / ms32 / 1e9, 2), "peak": PEAK_FP32_MFMA / 1e12,
                              "mfma_frac": round(flops / (ms32 * 1e-3) / PEAK_FP32_MFMA, 4)}}
    if was:
        lib.set_gemm_split(True)
        planes[0] = os.environ.get("MT_TSF_PLANES", "1") != "0"
        ms = timed_leg()
        out["split_pipe"] = {"fwd_ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 2), "peak": round(PEAK_SPLIT_PIPE / 1e12, 1),
                             "frac": round(flops / (ms * 1e-3) / PEAK_SPLIT_PIPE, 4),
                             "frac_vs_fp32_mfma": round(flops / (ms * 1e-3) / PEAK_FP32_MFMA, 4)}
    return out


WGRAD_KERNEL = {True: "TimeSformer weight-gradient GEMMs: mt::gemm_planes_kernel<128x128, k-major x k-major, EPI_ATOMIC> (dW = dY^T X over B*393 rows, both "
                      "operands as stored plane tensors by LDS-DMA + ds_read_b64_tr_b16, K-range-major split-K over the XCDs; side stream, "
                      "next to the main stream's data-gradient GEMMs; 54 launches + the row-mapped patch-embedding gradient on gemm_split_kernel)",
                False: "TimeSformer weight-gradient GEMMs: mt::gemm_dma_kernel<..., TN> (dW = dY^T X over B*393 rows; side stream)"}

WORKLOADS = {   # BASELINE.json configs that fit one GPU: (clips/GPU, frames, identities, extractor, fwd GFLOP/clip)
    2: dict(B=16, frames=8, ids=1, extractor="efficientnet-b0", flop_fwd=FLOP_PER_CLIP_FWD, name="config 2"),
    3: dict(B=32, frames=8, ids=2, extractor="efficientnet-b0", flop_fwd=FLOP_PER_CLIP_FWD, name="config 3"),
    5: dict(B=32, frames=16, ids=3, extractor="xception", flop_fwd=2 * 111174895104, name="config 5 (XS)"),
}


def _probe_summary(pr, tag=None):
    """(launches, seconds, work) of a probed kernel family over the timed region: launches issued eagerly carry torch events
    (lib.PROFILE), launches re-issued from a recorded plan are bracketed with HIP events inside mt_plan_run (their tag)."""
    durs = [e0.elapsed_time(e1) * 1e-3 for e0, e1, _ in pr["events"]]
    work = [w for _, _, w in pr["events"]]
    n, tot_t, tot_w = len(durs), sum(durs), sum(work)
    if tag is not None:
        from mintime_amd import plans
        pn, pt, pw = plans.probe_totals(tag)
        n, tot_t, tot_w = n + pn, tot_t + pt, tot_w + pw
    return n, tot_t, tot_w


def phases_leg(ef, tsf, opt, batch, reps=3):
    """Event-timed phases of the step on the main stream (side-stream work is inside the phase that waits for it)."""
    from mintime_amd import harness, optim
    acc = {}

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    for it in range(reps + 1):
        t0 = ev()
        v = batch["videos"]
        b, f, h, w, c = v.shape
        feats = ef(v.reshape(b * f, h, w, c).permute(0, 3, 1, 2))
        t1 = ev()
        y = tsf(feats.reshape(b, f, *feats.shape[1:]), mask=batch["mask"], size_embedding=batch["size_embedding"],
                identities_mask=batch["identities_mask"], positions=batch["positions"])
        loss = optim.bce_with_logits(y, batch["labels"], None)
        t2 = ev()
        opt.zero_grad(set_to_none=True)
        mark = []
        feats.register_hook(lambda g: mark.append(ev()))          # fires when the TimeSformer backward has produced dfeat
        loss.backward()
        t4 = ev()
        opt.step()
        t5 = ev()
        torch.cuda.synchronize()
        if it >= 1:
            for k, (a, b_) in dict(ef_fwd=(t0, t1), tsf_fwd=(t1, t2), tsf_bwd=(t2, mark[0]), ef_bwd=(mark[0], t4), sgd=(t4, t5)).items():
                acc.setdefault(k, []).append(a.elapsed_time(b_))
    return {k: round(sum(v) / len(v), 3) for k, v in acc.items()}


def forward_only_leg(ef, tsf, batch, iters=5):
    """Inference throughput of the same path (no_grad; nothing is kept for a backward pass), eval-mode and train-mode BatchNorm."""
    from mintime_amd import harness
    out = {}
    was = ef.training
    for name, mode in (("eval_bn", False), ("train_bn", True)):
        ef.train(mode)
        with torch.no_grad():
            for _ in range(2):
                harness.forward(ef, tsf, batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                harness.forward(ef, tsf, batch)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        out[name + "_ms"] = round(dt * 1e3, 3)
        out[name + "_clips_s"] = round(batch["videos"].shape[0] / dt, 1)
    ef.train(was)
    return out


def pin_launch_thread(local_rank, world):
    """One launch thread per rank: give each rank its own slice of the host's cores (rank r of N takes the r-th N-th of the cores
    this process may use -- on a two-socket node that is also the socket next to GPUs r of each half), so that 8 Python launch loops
    and their RCCL proxy threads do not migrate over each other.  MT_BENCH_NO_PIN=1 leaves the affinity alone."""
    if world <= 1 or os.environ.get("MT_BENCH_NO_PIN"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per < 8:                  # too few cores to fence off (launch thread, autograd thread, RCCL proxies per rank): leave it to the OS
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per] or cores
        os.sched_setaffinity(0, mine)
        return [mine[0], mine[-1]]
    except (AttributeError, OSError):
        return None


def other_config_leg(config, dev, steps=6, warm=10):
    """Another single-GPU BASELINE configuration timed in the same process (the driver's line carries config 3 as `value`; configs 2
    and 5 ride along so that they are driver-timed too): fresh models, `warm` untimed steps (they also grow the caching allocator's
    pools: the first steps of a large configuration on a warm box pay fresh hipMallocs), then `steps` timed steps."""
    import gc
    from mintime_amd import harness
    wl = WORKLOADS[config]
    B, frames = wl["B"], wl["frames"]
    try:
        if wl["extractor"] == "xception":
            cfg, ef, tsf = harness.build_models_xs(frames, seed=0, device=dev)
        else:
            cfg, ef, tsf = harness.build_models(frames, seed=0, device=dev)
        opt = harness.make_optimizer(cfg, ef, tsf)
        batch = harness.device_batch(B, frames, wl["ids"], seed=0, device=dev)
        for _ in range(warm):
            harness.train_step(ef, tsf, opt, batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = harness.train_step(ef, tsf, opt, batch)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        t_enq = 0.0
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            harness.train_step(ef, tsf, opt, batch)
            t_enq += time.perf_counter() - t1
        torch.cuda.synchronize()
        out = {"workload": f"{wl['name']}: B={B}, {frames} frames, {wl['ids']} identit{'y' if wl['ids'] == 1 else 'ies'}, {wl['extractor']}",
               "ms_per_step": round(ms, 3), "clips_s": round(B / (ms * 1e-3), 2), "steps": steps, "warmup": warm,
               "host_enqueue_ms_per_step": round(1e3 * t_enq / 2, 3), "loss": round(float(loss.item()), 5)}
    except Exception as e:       # noqa: BLE001  (an extra leg must never cost the line)
        out = {"error": f"{type(e).__name__}: {e}"}
    ef = tsf = opt = batch = None
    gc.collect()
    torch.cuda.empty_cache()
    return out


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run, one rank per GPU
    (--standalone style rendezvous on 127.0.0.1, a free port).  The ranks' chatter goes to stderr; the ONE JSON line of rank 0
    is the last thing on stdout."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    sys.stderr.flush()
    if line is not None:
        print(line, flush=True)
    return r.returncode if r.returncode else (0 if line is not None else 1)


def stub_main(a, rank, world):
    """The launcher / barrier / max-over-ranks / one-JSON-line plumbing of this file on CPU ranks over gloo with a stand-in step
    (tests/test_host_logic.py runs `bench.py --gpus 2 --stub` here; no kernel of the product is involved and the line says so)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.ones(1024) * (rank + 1)

    def step():
        y = x * 2.0
        if world > 1:
            dist.all_reduce(y)
        return y

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    mine = torch.tensor([1e3 * dt / a.steps, float(rank)], dtype=torch.float64)       # the per-rank gather of main(), on CPU ranks
    per_rank = [mine]
    if world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "stub steps/sec (launcher plumbing test, no product kernel)", "value": round(world * a.steps / dt, 2),
                          "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "stub", "config": {"workload": "stub", "parallelism": f"dp{world}",
                                                                        "checksum": float(y.sum().item()),
                                                                        "per_rank": {"ms_per_step": [round(float(r[0]), 3) for r in per_rank],
                                                                                     "rank": [int(r[1]) for r in per_rank]}}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=sorted(WORKLOADS), help="BASELINE.json configuration (3 = the headline)")
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--ragged", action="store_true", help="config-3 variant with the last slot of each identity padded")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the phase / forward-only / seed legs")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-reducer", action="store_true",
                    help="single-GPU check of the multi-GPU step: 1-rank RCCL group + the overlapped gradient reducer")
    a = ap.parse_args()
    wl = dict(WORKLOADS[a.config])
    if a.batch_per_gpu:
        wl["B"] = a.batch_per_gpu
    if a.frames:
        wl["frames"] = a.frames
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(wl["frames"], 0)))
        return

    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(a.gpus))        # `python bench.py --gpus N`: become the launcher of N ranks (one per GPU)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} is running with WORLD_SIZE={world}")
    if a.stub:
        return stub_main(a, rank, world)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    pinned = pin_launch_thread(local_rank, world)
    if world > 1 or a.force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    import mintime_amd
    from mintime_amd import harness, lib, ddp
    lib.get()
    if os.environ.get("MT_BENCH_STREAM"):        # experiment: run the step on a non-default (non-blocking) stream
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    B, frames = wl["B"], wl["frames"]
    if wl["extractor"] == "xception":
        cfg, ef, tsf = harness.build_models_xs(frames, seed=0, device=dev)
    else:
        cfg, ef, tsf = harness.build_models(frames, seed=0, device=dev)          # train-mode BN + drop-connect 0.2 (train.py:157)
    opt = harness.make_optimizer(cfg, ef, tsf)
    batch = harness.device_batch(B, frames, wl["ids"], seed=rank, device=dev, ragged=a.ragged)
    if os.environ.get("MT_BENCH_SE_DEVICE"):        # lab: size_embedding already on the device (the reference's callers keep it on the host)
        batch["size_embedding"] = batch["size_embedding"].to(dev)
    # buckets in the order backward finishes them: the TimeSformer's 48 M gradients all-reduce under the EfficientNet backward
    # ... and only the rows of the TimeSformer's two 21 MB embedding-gradient tables that the data can index go on the wire: positions
    # reach F * 49, size buckets 20 (synth.clip_inputs / sequence.py keep the reference's ranges: deepfakes_dataset.py:259-263,324-329)
    live_rows = {tsf.pos_emb.weight: frames * 49 + 1}
    if getattr(tsf, "enable_size_emb", False):
        live_rows[tsf.size_emb.weight] = 21
    reducer = (ddp.OverlappedGradReducer([tsf, ef], force=a.force_reducer, live_rows=live_rows)
               if world > 1 or a.force_reducer else None)
    reducer_path = "none" if reducer is None else "overlapped"

    def step(bt=None):
        return harness.train_step(ef, tsf, opt, batch if bt is None else bt, reducer)

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
            reducer_path = "flat"
    for _ in range(a.warmup):
        step()
    # live kernel timing over the timed region, HIP events on the stream each kernel is launched on:
    #   the TimeSformer's weight-gradient GEMMs (TN, plain operands; side stream) = the time-dominant kernel family of the step,
    #   FF1 + GEGLU (the largest single GEMM), the EfficientNet depthwise data gradient (the largest HBM-bound family)
    # (mt_gemm launches: the in-kernel split / fp32 pipe; mt_gemm_planes launches: the plane path -- whichever the engine runs)
    p_wgrad = {"match": lambda d: d.op == lib.OP_TN and d.prologue == lib.PRO_NONE and d.b_prologue == lib.BPRO_NONE,
               "match_planes": lambda d: d.op == lib.OP_TN, "events": []}
    p_ff1 = {"match": lambda d: d.epilogue == lib.EPI_GEGLU, "match_planes": lambda d: d.epilogue == lib.EPI_GEGLU, "events": []}
    p_dw = {"name": "dwconv_dgrad", "events": []}
    if reducer is not None and hasattr(reducer, "time_exposed"):
        reducer.time_exposed = True
    lib.PROFILE = [p_ff1, p_wgrad, p_dw]
    from mintime_amd import plans
    plans.PROBE_MASK[0] = (1 << lib.TAG_WGRAD) | (1 << lib.TAG_FF1) | (1 << lib.TAG_DWCONV_DGRAD)
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
    plans.PROBE_MASK[0] = 0
    # launch-side cost: wall time the host needs to ENQUEUE one step (~790 launches through ctypes) onto an idle device, i.e. without
    # back-pressure from a full queue.  The step is device-bound as long as this stays below ms_per_step.
    t_enq = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        t_enq += time.perf_counter() - t1
    torch.cuda.synchronize()
    t_enq /= 3
    exposed = reducer.exposed_ms() if reducer is not None and hasattr(reducer, "exposed_ms") else None
    if reducer is not None and hasattr(reducer, "time_exposed"):
        reducer.time_exposed = False
    # per-rank readings (first contact with an 8-GPU node should be diagnostic, not one number): step time, host enqueue time and
    # the exposed part of the gradient all-reduce of every rank
    mine = torch.tensor([1e3 * dt / a.steps, 1e3 * t_enq, -1.0 if exposed is None else exposed], device=dev, dtype=torch.float64)
    per_rank = [mine]
    if world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    per_rank = [[round(float(v), 3) for v in r.tolist()] for r in per_rank]
    out = None
    if rank == 0:
        ms = 1e3 * dt / a.steps
        clips_s = world * B * a.steps / dt
        flop_step = 3 * wl["flop_fwd"]
        headline = a.config == 3 and B == 32 and frames == 8 and not a.ragged
        n_w, t_w, f_w = _probe_summary(p_wgrad, lib.TAG_WGRAD)
        n_f, t_f, f_f = _probe_summary(p_ff1, lib.TAG_FF1)
        n_d, t_d, b_d = _probe_summary(p_dw, lib.TAG_DWCONV_DGRAD)
        pmc = lambda key: (committed_counters(key) or {}) if headline else {}
        split_on = lib.gemm_split_enabled()
        out = {
            "metric": "clips/sec (8-frame, 2-identity, 224^2 crops) fwd+bwd" if a.config != 5 else
                      "clips/sec (16-frame, 3-identity, 224^2 crops, Xception extractor) fwd+bwd",
            "value": round(clips_s, 2), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl['name']}: {'Xception' if a.config == 5 else 'EfficientNet-B0'} + SizeInvariantTimeSformer(dim 512, "
                                   f"depth 9, heads 8), B={B}/GPU, {frames} frames, {wl['ids']} identit{'y' if wl['ids'] == 1 else 'ies'} "
                                   f"{'(identity-masked' + (', last slot of each identity padded' if a.ragged else '') + ')'}, 224x224 crops, "
                                   "random-init seeded weights, train-mode BN" + (" + drop-connect 0.2" if a.config != 5 else "")
                                   + ", step = fwd + BCE loss + bwd"
                                   + (" + RCCL grad all-reduce" if world > 1 else "") + " + SGD(lr .01, wd 1e-4)",
                       "global_batch": world * B, "frames": frames, "parallelism": f"dp{world}", "reducer_path": reducer_path,
                       "per_rank": {"ms_per_step": [r[0] for r in per_rank], "host_enqueue_ms_per_step": [r[1] for r in per_rank],
                                    "allreduce_exposed_ms": [None if r[2] < 0 else r[2] for r in per_rank],
                                    "launch_thread_cores": pinned},
                       "model_tflops": round(clips_s * flop_step / 1e12, 2),
                       "model_frac_vs_fp32_mfma": round(clips_s * flop_step / (world * PEAK_FP32_MFMA), 4),
                       "matrix_pipe": SPLIT_PIPE + "; convolution GEMMs with operand prologues and K < 512: v_mfma_f32_32x32x2_f32"
                       if split_on else "v_mfma_f32_32x32x2_f32 (MT_GEMM_SPLIT=0)",
                       "loss": round(float(loss.item()), 5),
                       "host_enqueue_ms_per_step": round(1e3 * t_enq, 3),
                       "launch_plans": {"enabled": plans.ENABLED, **{k: plans.STATS[k] for k in ("recorded", "replayed")},
                                        "calls_per_phase": sorted(pl.ops for np_ in plans.ALL for pl in (np_.fwd, np_.bwd)
                                                                  if pl is not None)}},
            # the time-dominant kernel family: in-step duration (next to the main stream's data-gradient GEMMs), all launches summed
            "roofline": {"bound": "mfma", "kernel": WGRAD_KERNEL[split_on],
                         **mfma_roofline(f_w / t_w if t_w else None, split_on),
                         "traffic": pmc("tsf_wgrad").get("bytes_per_launch"), "traffic_unit": "bytes/launch (family mean)",
                         "traffic_source": pmc("tsf_wgrad").get("source"),
                         # the counter fields of this object and of roofline_ff1 / roofline_hbm are NOT measured in this run: they are read
                         # from profiles/*_pmc_families.json (hash-tied to csrc/), collected with the weight-gradient stream off
                         "pmc_mode": "serialised (side stream off), committed file, hash-tied to csrc/" if pmc("tsf_wgrad") else None,
                         "algorithmic_bytes": pmc("tsf_wgrad").get("algorithmic_bytes_per_launch"),
                         # matrix-pipe counters of the same family from the hash-tied PMC passes (serialised: side stream off):
                         # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), clock = GRBM_GUI_ACTIVE / 8 / wall
                         "mfma_busy_frac": pmc("tsf_wgrad").get("mfma_busy_frac"),
                         "effective_clock_ghz": pmc("tsf_wgrad").get("effective_clock_ghz"),
                         "mfma_busy_vs_algorithmic": pmc("tsf_wgrad").get("busy_vs_algorithmic"),
                         "launches_timed": n_w, "launches_per_step": n_w // max(a.steps, 1),
                         "avg_launch_us": round(t_w / max(n_w, 1) * 1e6, 1), "flops_per_launch": f_w / max(n_w, 1),
                         "ms_per_step": round(t_w / max(a.steps, 1) * 1e3, 3)},
            "roofline_ff1": {"bound": "mfma", "kernel": "mt::gemm_" + ("planes" if split_on else "dma")
                                                        + "_kernel<128x128, NT, EPI_GEGLU> (FF1 512->4096 + GEGLU, M=B*393"
                                                        + ("; emits h as operand planes)" if split_on else ")"),
                             **mfma_roofline(f_f / t_f if t_f else None, split_on),
                             "traffic": pmc("tsf_ff1").get("bytes_per_launch"), "traffic_unit": "bytes/launch",
                             "algorithmic_bytes": pmc("tsf_ff1").get("algorithmic_bytes_per_launch"),
                             "mfma_busy_frac": pmc("tsf_ff1").get("mfma_busy_frac"),
                             "effective_clock_ghz": pmc("tsf_ff1").get("effective_clock_ghz"),
                             "mfma_busy_vs_algorithmic": pmc("tsf_ff1").get("busy_vs_algorithmic"),
                             "launches_timed": n_f, "avg_launch_us": round(t_f / max(n_f, 1) * 1e6, 1),
                             "flops_per_launch": f_f / max(n_f, 1)},
        }
        if n_d:
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "EfficientNet depthwise-conv data gradient (dwconv_dgrad_tiled_kernel, 16 "
                                                             "launches/step, main stream, next to the side stream's weight gradients)",
                                   "achieved": round(b_d / t_d / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(b_d / t_d / 8e12, 4),
                                   "traffic": pmc("ef_dwconv_dgrad").get("bytes_per_step"), "traffic_unit": "bytes/step (16 launches)",
                                   "algorithmic_bytes": round(b_d / max(a.steps, 1)), "launches_timed": n_d,
                                   "ms_per_step": round(t_d / max(a.steps, 1) * 1e3, 3)}
        if world == 1 and not a.no_extras:
            ph = phases_leg(ef, tsf, opt, batch)
            tsf_flop = 3 * B * (wl["flop_fwd"] - (2 * 3076278016 if a.config != 5 else 2 * 72813297152))
            ef_flop = 3 * wl["flop_fwd"] * B - tsf_flop
            tsf_rate = tsf_flop / ((ph["tsf_fwd"] + ph["tsf_bwd"]) * 1e-3)
            ph["tsf_tflops"] = round(tsf_rate / 1e12, 1)
            ph["tsf_pipe_frac"] = round(tsf_rate / (PEAK_SPLIT_PIPE if split_on else PEAK_FP32_MFMA), 4)     # vs the pipe its GEMMs run on
            ph["tsf_frac_vs_fp32_mfma"] = round(tsf_rate / PEAK_FP32_MFMA, 4)
            ph["extractor_frac_vs_fp32_mfma"] = round(ef_flop / ((ph["ef_fwd"] + ph["ef_bwd"]) * 1e-3) / PEAK_FP32_MFMA, 4)
            step_bytes = pmc("ef_step").get("bytes_per_step")
            if step_bytes:
                ph["extractor_hbm_GBps"] = round(step_bytes / ((ph["ef_fwd"] + ph["ef_bwd"]) * 1e-3) / 1e9, 1)
                ph["extractor_hbm_frac"] = round(step_bytes / ((ph["ef_fwd"] + ph["ef_bwd"]) * 1e-3) / 8e12, 4)
            out["phases"] = ph
            out["forward_only"] = forward_only_leg(ef, tsf, batch)
            seeds = {"0": round(ms, 3)}
            for sd in (1, 2):
                bt = harness.device_batch(B, frames, wl["ids"], seed=sd, device=dev, ragged=a.ragged)
                for _ in range(2):
                    step(bt)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n_sd = max(3, a.steps // 2)
                for _ in range(n_sd):
                    step(bt)
                torch.cuda.synchronize()
                seeds[str(sd)] = round(1e3 * (time.perf_counter() - t1) / n_sd, 3)
            out["seeds"] = {"ms_per_step": seeds, "median_ms": sorted(seeds.values())[1],
                            "median_clips_s": round(B / (sorted(seeds.values())[1] * 1e-3), 2)}
            if split_on:                     # the same step with every contraction on the fp32 MFMA pipe
                lib.set_gemm_split(False)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n_f32 = max(3, a.steps // 2)
                for _ in range(n_f32):
                    step()
                torch.cuda.synchronize()
                ms_f32 = 1e3 * (time.perf_counter() - t1) / n_f32
                lib.set_gemm_split(True)
                out["fp32_mfma_pipe"] = {"ms_per_step": round(ms_f32, 3), "clips_s": round(B / (ms_f32 * 1e-3), 2),
                                         "note": "mt_gemm_set_split(0): v_mfma_f32_32x32x2_f32 everywhere"}
            if a.config != 5:
                out["attention_modules"] = attention_modules_leg(dev, B, frames)
            # optional exact optimisation, NOT part of `value`: the last layer's dead rows pruned (tsf_engine.py, MT_TSF_PRUNE_LAST)
            prev = os.environ.get("MT_TSF_PRUNE_LAST")
            os.environ["MT_TSF_PRUNE_LAST"] = "1"
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_pr = max(3, a.steps // 2)
            for _ in range(n_pr):
                step()
            torch.cuda.synchronize()
            ms_pr = 1e3 * (time.perf_counter() - t1) / n_pr
            if prev is None:
                os.environ.pop("MT_TSF_PRUNE_LAST")
            else:
                os.environ["MT_TSF_PRUNE_LAST"] = prev
            out["last_layer_dead_rows_pruned"] = {"ms_per_step": round(ms_pr, 3), "clips_s": round(B / (ms_pr * 1e-3), 2),
                                                  "note": "MT_TSF_PRUNE_LAST=1 (off in `value`): the head reads the cls token only, so the last layer's "
                                                          "space-attention tail and feed-forward run on the cls rows; same logits and gradients "
                                                          "(tests/test_gpu_tsf.py::test_last_layer_dead_row_pruning_is_exact)"}
            if a.config != 5:
                # deterministic mode (reference train.py:110; NOT part of `value`): the same step with every atomic reduction replaced by
                # logs / split-K slabs summed in a fixed order, and the streaming fusions that need atomics swapped for their GEMM forms
                try:                         # (an extra leg must never cost the line)
                    lib.set_deterministic(True)
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    n_dt = max(3, a.steps // 2)
                    for _ in range(n_dt):
                        step()
                    torch.cuda.synchronize()
                    ms_dt = 1e3 * (time.perf_counter() - t1) / n_dt
                    out["deterministic"] = {"ms_per_step": round(ms_dt, 3), "clips_s": round(B / (ms_dt * 1e-3), 2),
                                            "slowdown": round(ms_dt / ms, 3),
                                            "note": "MT_DETERMINISTIC=1 / mt_set_deterministic(1): bit-identical gradients run to run "
                                                    "(tests/test_gpu_determinism.py); off in `value`"}
                except Exception as e:       # noqa: BLE001
                    out["deterministic"] = {"error": f"{type(e).__name__}: {e}"}
                finally:
                    lib.set_deterministic(False)
        if world == 1 and not a.no_extras and headline:
            out["other_configs"] = {"config2": other_config_leg(2, dev), "config5": other_config_leg(5, dev)}
        if world == 1 and not a.no_cpu_baseline and a.config != 5:
            out["cpu_baseline"] = cpu_baseline_subprocess(frames)
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
