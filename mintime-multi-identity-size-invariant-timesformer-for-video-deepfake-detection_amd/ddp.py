"""Clip-level data parallelism: one process per GPU, weights replicated, gradients summed with one flat RCCL
all-reduce per step over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

The reference only has single-process nn.DataParallel (train.py:294-296): scatter / replicate / gather every step.
Here every rank owns its shard of the clips end-to-end (BatchNorm statistics stay per replica, like DataParallel);
the only exchange is the gradient sum -- 62.4 M floats ~ 250 MB for EfficientNet-B0 + TimeSformer.
Parameters that get no gradient (`_fc`, reference model.py:206-208) are skipped deterministically on every rank.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) share of `total` clips for `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradAllReducer:
    """Average gradients over the process group through one persistent flat fp32 buffer."""

    def __init__(self, params, group=None, skip_unused=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.flat = None
        self.live = None
        self.skip_unused = skip_unused

    def _layout(self):
        # fixed after the first step: parameters that received a gradient (identical on every rank by construction)
        self.live = [p for p in self.params if p.grad is not None or not self.skip_unused]
        n = sum(p.numel() for p in self.live)
        self.flat = torch.empty(n, dtype=torch.float32, device=self.live[0].device)
        self.views = []
        off = 0
        for p in self.live:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def allreduce(self):
        if self.world == 1:
            return 0
        if self.flat is None:
            self._layout()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.live]
        torch._foreach_copy_(self.views, grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, self.views)
        for p, g in zip(self.live, grads):
            if p.grad is None:
                p.grad = g
        return self.flat.numel()
