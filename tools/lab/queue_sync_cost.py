#!/usr/bin/env python3
"""What cross-queue synchronisation costs the MAIN queue on this device: a chain of 100 kernels of ~110 us on the main stream with
(a) nothing between them, (b) an event record, (c) event record + side stream waiting for it + a short side kernel,
(d) as (c) and the main stream waits for the side kernel's event before its next kernel, (e) as (c) with the main-stream wait
on the side event of TWO iterations ago (already complete)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mintime_amd import lib as L

dev = torch.device("cuda:0")
main = torch.cuda.current_stream(dev)
side = L.SideStream(dev).stream
x = torch.zeros(1 << 26, device=dev)      # 256 MB: ~110 us per add_ (the device, not the host, sets the pace)
y = torch.zeros(1 << 24, device=dev)      # side kernel ~30 us
N = 100

def run(mode):
    evs = []
    for i in range(N):
        x.add_(1.0)
        if mode >= 1:
            e = torch.cuda.Event(); e.record(main)
        if mode >= 2:
            side.wait_event(e)
            with torch.cuda.stream(side):
                y.add_(1.0)
            d = torch.cuda.Event(); d.record(side); evs.append(d)
        if mode == 3:
            main.wait_event(evs[-1])
        if mode == 4 and len(evs) > 2:
            main.wait_event(evs[-3])
    main.wait_stream(side)

for mode, name in enumerate(["plain chain", "+ event record on main", "+ side stream waits, side kernel", "+ main waits for the side kernel",
                             "+ main waits for the side kernel of two iterations ago"]):
    run(mode); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(); run(mode); e1.record(); torch.cuda.synchronize()
    print(f"{name:58s} {e0.elapsed_time(e1) * 1e3 / N:7.1f} us per iteration (host {1e6 * (time.perf_counter() - t0) / N:6.1f} us)")
