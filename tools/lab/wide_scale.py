#!/usr/bin/env python3
"""mt_conv1x1_wgrad_wide: time against the row count (fixed cost vs per-chunk cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mintime_amd import lib as L
lib = L.get()
for cout, cin, hw in [(192, 1152, 49), (112, 672, 196)]:
    for rows in (hw * 64, hw * 256, hw * 1024, hw * 4096):
        g = torch.Generator(device="cuda").manual_seed(1)
        r = lambda *s: torch.randn(*s, device="cuda", generator=g)
        du, z, kabc, x, sc, sh = r(rows, cout), r(rows, cout), r(3, cout), r(rows, cin), r(cin), r(cin)
        gate = torch.rand(rows // hw, cin, device="cuda", generator=g)
        dw = torch.zeros(cout, cin, device="cuda")
        def wide():
            L.check(lib.mt_conv1x1_wgrad_wide(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(gate), hw, L.ptr(dw),
                                              rows, cout, cin, L.stream_ptr()), "wide")
        for _ in range(3): wide()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): wide()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        print(f"{cout}x{cin} rows={rows}: {us:.0f} us  {2.0 * rows * cout * cin / us / 1e6:.0f} TF  chunks/block={(rows + 31) // 32 / (8 * max(1, 256 // (8 * ((cin + 127) // 128)))):.1f}")
