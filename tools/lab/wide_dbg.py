import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mintime_amd import lib as L
lib = L.get()
cout, cin, hw, rows = 192, 1152, 49, 49 * 4096
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
du, z, kabc, x, sc, sh = r(rows, cout), r(rows, cout), r(3, cout), r(rows, cin), r(cin), r(cin)
gate = torch.rand(rows // hw, cin, device="cuda", generator=g)
dw = torch.zeros(cout, cin, device="cuda")
def wide():
    L.check(lib.mt_conv1x1_wgrad_wide(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(gate), hw, L.ptr(dw),
                                      rows, cout, cin, L.stream_ptr()), "wide")
for _ in range(2): wide()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): wide()
e1.record(); torch.cuda.synchronize()
print("dbg", os.environ.get("MT_WIDE_DBG"), f"{e0.elapsed_time(e1) * 1e3 / 5:.0f} us")
