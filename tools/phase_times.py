#!/usr/bin/env python3
"""Wall time of the phases of one training step at B=32 (events on the main stream; side-stream work is included in the
phase that waits for it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mintime_amd
from mintime_amd import harness
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if os.environ.get("PHASE_INIT_RCCL"):          # what does merely creating a 1-rank RCCL communicator cost the step?
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
cfg, ef, tsf = harness.build_models(8, 0, "cuda")
opt = harness.make_optimizer(cfg, ef, tsf)
batch = harness.device_batch(B, 8, 2, 0, "cuda")
import torch.nn.functional as F
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
acc = {}
for it in range(6):
    t0 = ev()
    v = batch["videos"]; b, f, h, w, c = v.shape
    feats = ef(v.reshape(b * f, h, w, c).permute(0, 3, 1, 2))
    t1 = ev()
    feats5 = feats.reshape(b, f, *feats.shape[1:])
    y = tsf(feats5, mask=batch["mask"], size_embedding=batch["size_embedding"], identities_mask=batch["identities_mask"], positions=batch["positions"])
    loss = F.binary_cross_entropy_with_logits(y, batch["labels"].reshape(-1, 1))
    t2 = ev()
    opt.zero_grad(set_to_none=True)
    mark = []
    feats.register_hook(lambda g: mark.append(ev()))     # fires when the TimeSformer backward has produced dfeat
    loss.backward()
    t3 = mark[0]
    t4 = ev()
    opt.step()
    t5 = ev()
    torch.cuda.synchronize()
    if it >= 2:
        for k, (a, b_) in dict(ef_fwd=(t0, t1), tsf_fwd=(t1, t2), tsf_bwd=(t2, t3), ef_bwd=(t3, t4), sgd=(t4, t5), total=(t0, t5)).items():
            acc.setdefault(k, []).append(a.elapsed_time(b_))
for k, v in acc.items():
    print(f"{k:12s} {sum(v)/len(v):8.2f} ms")
