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


class OverlappedGradReducer:
    """Gradient averaging overlapped with the backward pass, one bucket per network.

    The hot path runs backwards through the TimeSformer first and the EfficientNet second, and 77 % of the gradient bytes
    (48 of 62.4 M floats) are the TimeSformer's: its all-reduce is launched (async, on RCCL's own stream) the moment its last
    parameter has accumulated and rides under the ~20 ms EfficientNet backward; only the EfficientNet bucket (16 MB) is exposed.
    xGMI is point-to-point, so the buckets are as large as the dependency structure allows (two), not 25 MB slices.

    Both engines hand autograd views of ONE flat gradient buffer per network (lib.zero_grads); when `p.grad` still aliases that
    buffer the bucket is reduced in place with no copies, otherwise it goes through a persistent flat staging buffer.
    The first step runs synchronously and records which parameters receive gradients (`_fc` never does, model.py:206-208).

        reducer = OverlappedGradReducer([tsf.parameters(), ef.parameters()])   # order = order in which backward finishes them
        loss.backward(); reducer.allreduce(); optimizer.step()
    """

    def __init__(self, buckets, group=None, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or force
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.live = [None] * len(self.buckets)        # parameters that get gradients (known after the first step)
        self.seen = [0] * len(self.buckets)
        self.pending = []
        self.staging = [None] * len(self.buckets)
        self.stats = {"overlapped_launches": 0, "in_place": 0, "staged": 0, "synchronous": 0}
        if self.active:
            for bi, b in enumerate(self.buckets):
                for p in b:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _make_hook(self, bi):
        def hook(p):
            if self.live[bi] is None:
                return
            self.seen[bi] += 1
            if self.seen[bi] == len(self.live[bi]):
                self._launch(bi, async_op=True)
                self.stats["overlapped_launches"] += 1
        return hook

    def _launch(self, bi, async_op):
        params = self.live[bi]
        grads = [p.grad for p in params]
        base = grads[0].untyped_storage()
        aliased = all(g.untyped_storage().data_ptr() == base.data_ptr() for g in grads)
        if aliased:
            flat = torch.empty(0, dtype=grads[0].dtype, device=grads[0].device).set_(base)
            back = None
            self.stats["in_place"] += 1
        else:
            n = sum(g.numel() for g in grads)
            if self.staging[bi] is None or self.staging[bi][0].numel() != n:
                flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
                views, off = [], 0
                for g in grads:
                    views.append(flat[off:off + g.numel()].view_as(g))
                    off += g.numel()
                self.staging[bi] = (flat, views)
            flat, views = self.staging[bi]
            torch._foreach_copy_(views, grads)
            back = (grads, views)
            self.stats["staged"] += 1
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        self.pending.append((work, flat, back))

    def allreduce(self):
        """Call after backward(): waits for the launched buckets (launching any that could not be overlapped) and averages."""
        if not self.active:
            return 0
        for bi, b in enumerate(self.buckets):
            if self.live[bi] is None:                                   # first step: learn the layout, reduce synchronously
                self.live[bi] = [p for p in b if p.grad is not None]
                if self.live[bi]:
                    self._launch(bi, async_op=False)
                    self.stats["synchronous"] += 1
            elif self.seen[bi] != len(self.live[bi]):
                raise RuntimeError(f"bucket {bi}: {self.seen[bi]} of {len(self.live[bi])} gradients arrived; the set of "
                                   "parameters receiving gradients must not change between steps")
        n = 0
        for work, flat, back in self.pending:
            if work is not None:
                work.wait()
            flat.mul_(1.0 / self.world)
            if back is not None:
                torch._foreach_copy_(back[0], back[1])
            n += flat.numel()
        self.pending = []
        self.seen = [0] * len(self.buckets)
        return n
