"""Clip-level data parallelism: one process per GPU, weights replicated, gradients summed with one flat RCCL
all-reduce per step over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

The reference only has single-process nn.DataParallel (train.py:294-296): scatter / replicate / gather every step.
Here every rank owns its shard of the clips end-to-end (BatchNorm statistics stay per replica, like DataParallel);
the only exchange is the gradient sum -- 62.4 M floats ~ 250 MB for EfficientNet-B0 + TimeSformer.
Parameters that get no gradient (`_fc`, reference model.py:206-208) are skipped deterministically on every rank.
"""
import contextlib

import torch
import torch.distributed as dist


def broadcast_module_state(modules, group=None, src=0):
    """Make every rank start from rank `src`'s parameters and buffers (what torch's DistributedDataParallel does at
    construction); the reference's nn.DataParallel gets the same effect by replicating module 0 every step (train.py:294-296)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    n = 0
    with torch.no_grad():
        for m in modules:
            tensors = list(m.parameters()) + list(m.buffers()) if isinstance(m, torch.nn.Module) else list(m)
            for t in tensors:
                dist.broadcast(t.data, src=src, group=group)
                n += t.numel()
    return n


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
    (48 of 62.4 M floats) are the TimeSformer's: its all-reduce is launched (async, on RCCL's own stream) the moment its
    backward has finished and rides under the ~19 ms EfficientNet backward; only the EfficientNet bucket (16 MB) is exposed.
    xGMI is point-to-point, so the buckets are as large as the dependency structure allows (two), not 25 MB slices.

    A bucket is either
      * an engine-backed module (EfficientNet / SizeInvariantTimeSformer / Xception): its backward produces every parameter
        gradient as a view of ONE flat buffer and announces it through `lib.grads_ready`; the buffer is reduced in place, with
        no per-parameter Python hooks (366 hook calls cost ~5 ms of launch-thread time per step) and no copies; or
      * a plain list of parameters (any torch module): post-accumulate hooks count arrivals, the bucket is launched when the
        last one has arrived (in place when the gradients alias one storage, else through a persistent staging buffer); the
        first step runs synchronously and records which parameters receive gradients (`_fc` never does, model.py:206-208).

        reducer = OverlappedGradReducer([tsf, ef])          # order = order in which backward finishes them
        loss.backward(); reducer.allreduce(); optimizer.step()
    """

    def __init__(self, buckets, group=None, force=False, broadcast_init=True, live_rows=None):
        """live_rows: {parameter: n} -- only the first n rows of that (embedding-table) parameter ever receive gradient: the rest
        of its gradient is exactly zero on every rank and is left out of the collective.  The TimeSformer's two tables are
        [num_frames * channels + 1, dim] = 21 MB each (the reference's (sic) sizing, size_invariant_timesformer.py:172-180) while
        positions only reach F * 49 and size buckets 20 (deepfakes_dataset.py:259-263,324-329): 42 of the bucket's 234 MB.  The
        bound is the caller's data contract; the first `LIVE_ROW_CHECKS` engine launches verify that the skipped rows are zero."""
        self.group = group
        self.sync = True
        self.live_rows = {id(p): int(n) for p, n in (live_rows or {}).items()}
        self._segments = {}
        self._live_checks_left = self.LIVE_ROW_CHECKS
        if broadcast_init:
            broadcast_module_state(buckets, group)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or force
        self.modules = [b if isinstance(b, torch.nn.Module) else None for b in buckets]
        self.buckets = [[p for p in (b.parameters() if isinstance(b, torch.nn.Module) else b) if p.requires_grad] for b in buckets]
        self.live = [None] * len(self.buckets)        # hook buckets: parameters that get gradients (known after the first step)
        self.seen = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.pending = []
        self.staging = [None] * len(self.buckets)
        self.stats = {"overlapped_launches": 0, "in_place": 0, "staged": 0, "synchronous": 0}
        # bench.py: event pairs around the wait in allreduce() = the part of the collectives the backward pass did not hide
        self.time_exposed = False
        self._exposed = []
        if self.active:
            for bi, b in enumerate(self.buckets):
                if self.modules[bi] is not None:
                    self.modules[bi]._grads_ready_hook = self._make_engine_hook(bi)
                else:
                    for p in b:
                        p.register_post_accumulate_grad_hook(self._make_hook(bi))

    LIVE_ROW_CHECKS = 2

    def _flat_segments(self, bi, params, flat):
        """Contiguous [start, end) ranges of an engine's flat gradient buffer that go on the wire: everything but the dead rows of
        the `live_rows` parameters (carving rule of lib.zero_grads: every view starts on a multiple of 4 floats)."""
        key = (bi, flat.numel(), tuple(id(p) for p in params))
        if key not in self._segments:
            segs, dead, off = [], [], 0

            def add(a, b):
                if b <= a:
                    return
                if segs and segs[-1][1] >= a:
                    segs[-1][1] = max(segs[-1][1], b)
                else:
                    segs.append([a, b])
            for p in params:
                size = p.numel()
                padded = (size + 3) // 4 * 4
                n = self.live_rows.get(id(p))
                if n is not None and p.dim() >= 1 and 0 <= n < p.shape[0]:
                    live_end = off + n * (size // p.shape[0])
                    add(off, live_end)
                    dead.append((live_end, off + size))
                else:
                    add(off, min(off + padded, flat.numel()))          # (the alignment padding -- zeros -- rides along)
                off += padded
            self._segments[key] = ([tuple(s_) for s_ in segs], dead)
        return self._segments[key]

    # ---- engine-backed buckets ------------------------------------------------------------------------------------
    def _make_engine_hook(self, bi):
        def hook(params, flat):
            if not self.sync:                      # inside no_sync(): gradients only accumulate locally
                return
            params = [p for p in params if p is not None]      # optional parameters (size_emb off) hold no space in `flat`
            if self.launched[bi]:
                # a second backward in the same step: autograd is about to ADD into the buffer the in-flight collective owns,
                # and that contribution would never be reduced.  Drain the collective so memory stays sane, then refuse.
                for work, _, back in self.pending:
                    if work is not None and back is not None and back[0] == "engine" and back[1] == bi:
                        for w_ in (work if isinstance(work, list) else [work]):
                            w_.wait()
                raise RuntimeError("OverlappedGradReducer: backward ran twice before allreduce(); wrap all but the last "
                                   "micro-batch in `with reducer.no_sync():` to accumulate gradients")
            # gradients that already exist (accumulated under no_sync) are ADDED to by autograd after this hook returns: leave
            # such a bucket to the synchronous path in allreduce()
            if any(p.grad is not None for p in params):
                return
            segs, dead = self._flat_segments(bi, params, flat) if self.live_rows else ([(0, flat.numel())], [])
            if dead and self._live_checks_left > 0:
                self._live_checks_left -= 1
                if any(bool(flat[a:b].any()) for a, b in dead):
                    raise RuntimeError("OverlappedGradReducer: a parameter's gradient is non-zero beyond its `live_rows` bound; "
                                       "the bound does not hold for this data")
            if len(segs) == 1 and segs[0] == (0, flat.numel()):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                work = [dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for a, b in segs]
            self.pending.append((work, flat, ("engine", bi, params, segs)))
            self.launched[bi] = True
            self.stats["overlapped_launches"] += 1
            self.stats["in_place"] += 1
        return hook

    @staticmethod
    def _views_like_zero_grads(params, flat):
        """Same carving rule as lib.zero_grads (None entries, already filtered by the hook, take no space there either)."""
        views, off = [], 0
        for p in params:
            if p is None:
                continue
            views.append(flat[off:off + p.numel()].view(p.shape))
            off += (p.numel() + 3) // 4 * 4
        return views

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only accumulate locally; the first backward outside it
        (followed by allreduce()) reduces the accumulated sum -- synchronously, because autograd adds into existing gradients."""
        prev, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = prev

    # ---- plain parameter buckets ----------------------------------------------------------------------------------
    def _make_hook(self, bi):
        def hook(p):
            if self.live[bi] is None or not self.sync:
                return
            self.seen[bi] += 1
            if self.seen[bi] == len(self.live[bi]):
                self._launch(bi, async_op=True)
                self.launched[bi] = True
                self.stats["overlapped_launches"] += 1
        return hook

    def _launch(self, bi, async_op, params=None):
        params = self.live[bi] if params is None else params
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
            back = ("copy", grads, views)
            self.stats["staged"] += 1
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        self.pending.append((work, flat, back))

    # ---- after backward() -----------------------------------------------------------------------------------------
    def allreduce(self):
        """Call after backward(): waits for the launched buckets (launching any that could not be overlapped) and averages."""
        if not self.active:
            return 0
        if not self.sync:
            raise RuntimeError("allreduce() called inside no_sync()")
        for bi, b in enumerate(self.buckets):
            if self.launched[bi]:
                continue
            if self.modules[bi] is not None:                             # engine bucket that could not be launched early
                ps = [p for p in b if p.grad is not None]
                if ps:
                    self._launch(bi, async_op=False, params=ps)
                    self.stats["synchronous"] += 1
            elif self.live[bi] is None:                                   # first step: learn the layout, reduce synchronously
                self.live[bi] = [p for p in b if p.grad is not None]
                if self.live[bi]:
                    self._launch(bi, async_op=False)
                    self.stats["synchronous"] += 1
            elif self.seen[bi] == 0 and all(p.grad is not None for p in self.live[bi]):
                self._launch(bi, async_op=False)                          # accumulated under no_sync(): nothing was launched
                self.stats["synchronous"] += 1
            elif self.seen[bi] != len(self.live[bi]):
                raise RuntimeError(f"bucket {bi}: {self.seen[bi]} of {len(self.live[bi])} gradients arrived; the set of "
                                   "parameters receiving gradients must not change between steps")
        n = 0
        ev = None
        if self.time_exposed and self.pending and self.pending[0][1].is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for work, flat, back in self.pending:
            if work is not None:
                for w_ in (work if isinstance(work, list) else [work]):
                    w_.wait()
            if back is not None and back[0] == "engine" and isinstance(work, list):
                for a, b in back[3]:                                      # (the dead rows are zeros: nothing to scale)
                    flat[a:b].mul_(1.0 / self.world)
            else:
                flat.mul_(1.0 / self.world)
            if back is not None and back[0] == "copy":
                torch._foreach_copy_(back[1], back[2])
            elif back is not None:                                        # engine bucket: p.grad must still alias the buffer
                _, bi, params = back[:3]
                base = flat.untyped_storage().data_ptr()
                stale = [(p, v) for p, v in zip(params, self._views_like_zero_grads(params, flat))
                         if p.grad is not None and p.grad.untyped_storage().data_ptr() != base]
                if stale:                                                 # autograd cloned instead of adopting the views
                    torch._foreach_copy_([p.grad for p, _ in stale], [v for _, v in stale])
                    self.stats["staged"] += 1
            n += flat.numel()
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
        self.pending = []
        self.seen = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        return n

    def exposed_ms(self):
        """Mean stream time per step between entering allreduce() and the averaged gradients being ready (collective wait + scaling):
        what the overlap with the backward pass left exposed.  Clears the readings; call after a synchronize."""
        if not self._exposed:
            return None
        t = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return sum(t) / len(t)
