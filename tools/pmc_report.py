#!/usr/bin/env python3
"""Joins the passes of tools/ef_pmc.sh (FETCH_SIZE, WRITE_SIZE, SQ counters, kernel trace) by dispatch order and prints, for the
LAST step in the trace, per kernel family: launches, time, HBM-side bytes (gfx950 correction: FETCH_SIZE doubled, see
MI355X_MICROARCH.md), GB/s, MFMA-busy fraction, effective clock and VALU instructions -- plus every launch >= --min us with the
same columns.

Normalisation of the matrix-pipe counter (round-4 verdict, weak #6): rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (each
XCD's GRBM counts its own active cycles) and SQ_VALU_MFMA_BUSY_CYCLES summed over all 1024 SIMDs, in shader cycles (32 per
v_mfma_f32_32x32x16_bf16).  So   effective clock = GRBM_GUI_ACTIVE / 8 / wall   and   MFMA busy = BUSY / (1024 x GRBM_GUI_ACTIVE / 8).
Rounds 2-4 divided by 1024 x GRBM_GUI_ACTIVE and printed values 8x too low.  The tool checks the assumption per launch (the
effective clock of every launch >= 100 us must come out between 0.8 and 2.6 GHz) and, for the two TimeSformer families whose flop counts it knows, that
busy cycles x 1024 FLOP / wall agrees with 6 x 2MNK / wall."""
import argparse
import csv
import glob
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--min", type=float, default=1e9)
ap.add_argument("--start", default="stem_mfma_kernel", help="substring of the kernel that starts a step (EfficientNet: the stem kernel)")
ap.add_argument("--fetch-scale", type=float, default=2.0)
ap.add_argument("--json-out", default=None, help="merge family totals into this JSON file (profiles/r02_pmc_families.json)")
ap.add_argument("--kind", default="ef", choices=["ef", "tsf"])
a = ap.parse_args()


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mt::", "")
    return re.sub(r"\(.*", "", n)[:60]


def load_counters(sub):
    f = glob.glob(f"{a.dir}/{sub}/**/*counter_collection.csv", recursive=True)
    if not f:
        return None
    per = defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(f[0])):
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    order = sorted(per)
    return [(names[d], per[d]) for d in order]


trace = glob.glob(f"{a.dir}/trace/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
kt = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else "") for r in rows]
passes = {k: load_counters(k) for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES")}


def last_step(seq, name_of):
    starts = [i for i, x in enumerate(seq) if a.start in name_of(x)]
    return seq[starts[-1]:] if starts else seq


kt = last_step(kt, lambda x: x[0])
for k in passes:
    if passes[k] is not None:
        passes[k] = last_step(passes[k], lambda x: x[0])
        if len(passes[k]) != len(kt):
            print(f"# warning: pass {k} has {len(passes[k])} dispatches in its last step, the trace {len(kt)}")
fam = defaultdict(lambda: defaultdict(float))
per_launch = []
N_XCD = 8
bad_clock = []
print(f"{'idx':>4} {'us':>8} {'rd MB':>8} {'wr MB':>8} {'GB/s':>7} {'mfma%':>6} {'GHz':>5} {'valu/wave':>9}  kernel")
for i, (name, us, grid) in enumerate(kt):
    c = {}
    for k in passes:
        if passes[k] is not None and i < len(passes[k]) and short(passes[k][i][0]) == short(name):
            c.update(passes[k][i][1])
    rd = c.get("FETCH_SIZE", 0.0) * 1024 * a.fetch_scale        # FETCH_SIZE / WRITE_SIZE are reported in KiB
    wr = c.get("WRITE_SIZE", 0.0) * 1024
    per_launch.append({"rd": rd, "wr": wr, "busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "gui": c.get("GRBM_GUI_ACTIVE", 0.0)})
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    clk = gui / N_XCD / us / 1e3 if gui and us else 0.0          # GHz
    if gui and us >= 100 and not 0.8 <= clk <= 2.6:         # (short launches: the counter pass's active cycles include dispatch, the trace's wall does not)
        bad_clock.append((short(name), us, clk))
    mfma = busy / (gui / N_XCD * 256 * 4) if gui else 0.0        # busy cycles of 1024 SIMDs over the kernel's active cycles
    waves = c.get("SQ_WAVES", 0.0)
    f = fam[re.sub(r"<.*", "", short(name))]
    f["n"] += 1; f["us"] += us; f["rd"] += rd; f["wr"] += wr; f["busy"] += busy; f["gui"] += gui
    f["valu"] += c.get("SQ_INSTS_VALU", 0.0); f["wait"] += c.get("SQ_WAIT_ANY", 0.0); f["wcyc"] += c.get("SQ_WAVE_CYCLES", 0.0)
    if us >= a.min:
        print(f"{i:4d} {us:8.1f} {rd / 1e6:8.1f} {wr / 1e6:8.1f} {(rd + wr) / us / 1e3:7.0f} {100 * mfma:6.1f} {clk:5.2f} {c.get('SQ_INSTS_VALU', 0.0):9.0f}  {short(name)}")
tot = sum(f["us"] for f in fam.values())
print(f"\nstep: {len(kt)} launches, kernel time {tot / 1e3:.2f} ms, HBM-side traffic {sum(f['rd'] + f['wr'] for f in fam.values()) / 1e9:.2f} GB")
print(f"{'ms':>7} {'n':>4} {'rd GB':>7} {'wr GB':>7} {'TB/s':>6} {'mfma%':>6} {'GHz':>5} {'wait%':>6}  family")
for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
    mf = f["busy"] / (f["gui"] / N_XCD * 1024) if f["gui"] else 0.0
    ck = f["gui"] / N_XCD / f["us"] / 1e3 if f["gui"] else 0.0       # the counter pass's clock (a profiled pass runs a few % lower)
    wt = f["wait"] / f["wcyc"] if f["wcyc"] else 0.0
    print(f"{f['us'] / 1e3:7.2f} {int(f['n']):4d} {f['rd'] / 1e9:7.2f} {f['wr'] / 1e9:7.2f} {(f['rd'] + f['wr']) / f['us'] / 1e6:6.2f} {100 * mf:6.1f} {ck:5.2f} {100 * wt:6.1f}  {k}")
if bad_clock:
    print(f"# WARNING: {len(bad_clock)} launches give an effective clock outside 0.8-2.6 GHz under the 8-XCD normalisation, e.g. {bad_clock[:3]}")

if a.json_out:
    import json, os
    doc = json.load(open(a.json_out)) if os.path.exists(a.json_out) else {}
    src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, side stream off (tools/ef_pmc.sh); FETCH_SIZE doubled "
           "(gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported")

    def family(pred):
        sel = [(us, c) for (name, us, _), c in zip(kt, per_launch) if pred(short(name))]
        return sel

    if a.kind == "ef":
        tot = sum(f["rd"] + f["wr"] for f in fam.values())
        dw = fam.get("dwconv_dgrad_tiled_kernel")
        doc["ef_step"] = {"bytes_per_step": tot, "kernel_ms": round(sum(f["us"] for f in fam.values()) / 1e3, 3), "launches": len(kt),
                          "source": src, "workload": "EfficientNet-B0 forward+backward, 256 crops (config 3), train-mode BN"}
        if dw:
            doc["ef_dwconv_dgrad"] = {"bytes_per_step": dw["rd"] + dw["wr"], "read_bytes": dw["rd"], "write_bytes": dw["wr"],
                                      "launches": int(dw["n"]), "kernel_ms": round(dw["us"] / 1e3, 3), "source": src}
        doc["ef_families"] = {k: {"ms": round(f["us"] / 1e3, 3), "n": int(f["n"]), "read_GB": round(f["rd"] / 1e9, 3),
                                  "write_GB": round(f["wr"] / 1e9, 3)} for k, f in fam.items() if f["us"] > 100}
    else:
        M = 32 * 393
        # TN weight gradients / FF1 + GEGLU on whichever main loop ran (gemm_split_kernel by default, gemm_dma_kernel with MT_GEMM_SPLIT=0)
        # ... or the plane-operand loop (gemm_planes_kernel<..., AKM, BKM, EPI, ...>): both operands are 6 B / element plane tensors
        tn_pl = family(lambda n: re.match(r"gemm_planes_kernel<\d+, \d+, \d+, \d+, true, true, 4,", n) is not None)
        ff1_pl = family(lambda n: re.match(r"gemm_planes_kernel<2, 2, 2, 2, false, false, 2,", n) is not None)
        tn = tn_pl or family(lambda n: re.match(r"gemm_(dma|split)_kernel<\d+, \d+, \d+, \d+, 1, 1, 4,", n) is not None)
        ff1 = ff1_pl or family(lambda n: re.match(r"gemm_(dma|split)_kernel<2, 2, 2, 2, 0, 0, 2,", n) is not None)
        shapes = [(512, 2048, 9), (4096, 512, 9), (1536, 512, 18), (512, 512, 18)]
        opb = 6.0 if tn_pl else 4.0          # operand bytes per element
        alg = sum((opb * M * (n1 + n2) + 4.0 * n1 * n2) * c for n1, n2, c in shapes) / sum(c for _, _, c in shapes)
        def mfma_fields(sel, flops_total, products):
            """Matrix-pipe evidence of a family: busy fraction and clock from the counters, and the cross-check that the counted
            busy cycles (1024 FLOP each on the bf16 pipe, 64 on the fp32 pipe) deliver the family's products."""
            us = sum(u for u, _ in sel)
            busy = sum(c["busy"] for _, c in sel)
            gui = sum(c["gui"] for _, c in sel)
            if not (busy and gui):
                return {}
            per_cycle = 1024.0 if products == 6 else 64.0
            out = {"mfma_busy_frac": round(busy / (gui / N_XCD * 1024), 4), "effective_clock_ghz": round(gui / N_XCD / us / 1e3, 3),
                   "mfma_tflops_from_busy_cycles": round(busy * per_cycle / us / 1e6, 1),
                   "mfma_tflops_algorithmic": round(products * flops_total / us / 1e6, 1)}
            ratio = out["mfma_tflops_from_busy_cycles"] / out["mfma_tflops_algorithmic"]
            out["busy_vs_algorithmic"] = round(ratio, 3)
            if not 0.9 <= ratio <= 1.1:
                print(f"# WARNING: busy-cycle flops / algorithmic flops = {ratio:.3f} (expected within 10 % of 1)")
            return out

        if tn:
            doc["tsf_wgrad"] = {**mfma_fields(tn, sum(2.0 * M * n1 * n2 * c for n1, n2, c in shapes), 6 if tn_pl else 1),
                                "bytes_per_launch": sum(c["rd"] + c["wr"] for _, c in tn) / len(tn), "launches": len(tn),
                                "kernel_ms": round(sum(us for us, _ in tn) / 1e3, 3), "algorithmic_bytes_per_launch": alg,
                                "source": src}
        if ff1:
            doc["tsf_ff1"] = {**mfma_fields(ff1, 2.0 * M * 4096 * 512 * len(ff1), 6 if ff1_pl else 1),
                              "bytes_per_launch": sum(c["rd"] + c["wr"] for _, c in ff1) / len(ff1), "launches": len(ff1),
                              "read_bytes": sum(c["rd"] for _, c in ff1) / len(ff1), "write_bytes": sum(c["wr"] for _, c in ff1) / len(ff1),
                              "algorithmic_bytes_per_launch": (6.0 * (M * 512 + 4096 * 512) + 4.0 * 4096 + 6.0 * M * 2048 + 4.0 * M * 4096) if ff1_pl
                              else 4.0 * (M * 512 + 4096 * 512 + 4096 + M * 2048 + M * 4096), "source": src}
        doc["tsf_families"] = {k: {"ms": round(f["us"] / 1e3, 3), "n": int(f["n"]), "read_GB": round(f["rd"] / 1e9, 3),
                                   "write_GB": round(f["wr"] / 1e9, 3)} for k, f in fam.items() if f["us"] > 100}
    # which kernel sources these counters belong to: bench.py refuses them for any other tree (same hash as bench.csrc_hash)
    import glob as _g, hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd", "csrc")
    for f in sorted(_g.glob(os.path.join(pkg, "*.hip")) + _g.glob(os.path.join(pkg, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    doc["csrc_sha16"] = h.hexdigest()[:16]
    json.dump(doc, open(a.json_out, "w"), indent=1)
    print("wrote", a.json_out)
