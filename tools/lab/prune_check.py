"""Config-3 size check of MT_TSF_PRUNE_LAST: one forward+backward each way on identical weights / inputs, gradients compared."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import mintime_amd
from mintime_amd import harness, optim

out = {}
for flag in ("0", "1"):
    os.environ["MT_TSF_PRUNE_LAST"] = flag
    torch.manual_seed(0); torch.cuda.manual_seed_all(0)
    cfg, ef, tsf = harness.build_models(8, seed=0, device="cuda", drop_connect_rate=0.0)
    batch = harness.device_batch(32, 8, 2, seed=0, device="cuda")
    y = harness.forward(ef, tsf, batch)
    loss = optim.bce_with_logits(y, batch["labels"], None)
    loss.backward()
    torch.cuda.synchronize()
    out[flag] = (y.detach().clone(), float(loss), {k: p.grad.clone() for k, p in list(tsf.named_parameters()) + list(ef.named_parameters()) if p.grad is not None})
a, b = out["0"], out["1"]
print("loss", a[1], b[1], "max logit diff", float((a[0] - b[0]).abs().max()))
worst = []
for k in a[2]:
    d = float((a[2][k] - b[2][k]).norm() / a[2][k].norm().clamp_min(1e-30))
    worst.append((d, k))
for d, k in sorted(worst, reverse=True)[:12]:
    print(f"{d:.3e} {k}")
