#!/usr/bin/env python3
"""Micro-benchmark: late-stage project-conv weight gradients, mt_conv1x1_wgrad_wide against the generic TN GEMM (256-crop batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mintime_amd import lib as L

lib = L.get()
for rows, cout, cin, hw in [(50176, 80, 240, 196), (50176, 80, 480, 196), (50176, 112, 480, 196), (50176, 112, 672, 196),
                            (12544, 192, 672, 49), (12544, 192, 1152, 49), (12544, 320, 1152, 49)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    du, z, kabc, x, sc, sh = r(rows, cout), r(rows, cout), r(3, cout), r(rows, cin), r(cin), r(cin)
    gate = torch.rand(rows // hw, cin, device="cuda", generator=g)
    dw = torch.zeros(cout, cin, device="cuda")
    def wide():
        L.check(lib.mt_conv1x1_wgrad_wide(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(gate), hw, L.ptr(dw),
                                          rows, cout, cin, L.stream_ptr()), "wide")
    def gemm():
        L.gemm(L.OP_TN, du, x, dw, cout, cin, rows, cout, cin, cin, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0, A2=z,
               scale=kabc[0], shift=kabc[1], gate=kabc[2], b_prologue=L.BPRO_BN_SWISH_GATE, b_scale=sc, b_shift=sh, b_gate=gate, b_hw=hw)
    out = []
    for fn in (wide, gemm):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 20)
    fl = 2.0 * rows * cout * cin
    print(f"rows={rows} {cout}x{cin}: wide {out[0]:.0f} us ({fl / out[0] / 1e6:.0f} TF)  generic {out[1]:.0f} us ({fl / out[1] / 1e6:.0f} TF)")
