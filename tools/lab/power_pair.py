"""Lab: the plane-operand GEMM loop at 4096^3 on RANDOM operands, then on ZERO operands (same binary, same launch geometry): 30 launches
each, printed as wall time per launch.  Run under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` and under `--kernel-trace`
(tools/lab/power_pair.sh) the pair shows what bounds the loop: equal busy cycles, different clocks."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mintime_amd
from mintime_amd import lib as L

n = 4096
torch.manual_seed(0)
for tag, mk in (("random", lambda: torch.randn(n, n, device="cuda")), ("zero", lambda: torch.zeros(n, n, device="cuda"))):
    a, b = L.split_planes_blk(mk()), L.split_planes_blk(mk())
    c = torch.empty(n, n, device="cuda")
    for _ in range(10):
        L.gemm_planes(L.OP_NT, a, b, n, n, n, Cout=c, ldc=n, streamk=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        L.gemm_planes(L.OP_NT, a, b, n, n, n, Cout=c, ldc=n, streamk=False)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e3
    print(f"{tag}: {t:.1f} us / launch = {2.0 * n ** 3 / t / 1e6:.1f} TF fp32-equivalent ({6 * 2.0 * n ** 3 / t / 1e6:.0f} TF bf16)")
