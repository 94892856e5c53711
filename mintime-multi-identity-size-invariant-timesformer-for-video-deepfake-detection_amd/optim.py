"""Training-step ends on libmintime_hip (next-row f4): a drop-in for the reference's `torch.optim.SGD(parameters, lr=...,
weight_decay=...)` (train.py:186) whose step() is ONE multi-tensor launch, and BCE-with-logits (train.py:261,367-368) computed on
the device together with its gradient."""
import torch

from . import lib as L

_CHUNK = 4096


def _weights_changed():
    """Parameters were updated through raw pointers (torch's version counters did not move): cached operand planes are stale."""
    from . import tsf_planes
    tsf_planes.WEIGHT_EPOCH[0] += 1


class FusedSGD(torch.optim.Optimizer):
    """`p -= lr * (grad + weight_decay * p)` (torch.optim.SGD without momentum) for every parameter of a group in one launch.
    param_groups / lr schedulers work as with torch.optim.SGD (lr and weight_decay are read from the group at every step)."""

    def __init__(self, params, lr=1e-3, weight_decay=0.0):
        if lr < 0.0 or weight_decay < 0.0:
            raise ValueError("lr and weight_decay must be non-negative")
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay))
        self._tables = {}
        self.table_uploads = 0

    def _table(self, gi, ps):
        """Device table of (param ptr, grad ptr, numel, first block).  Gradient buffers come from the caching allocator, so their
        addresses repeat after a step or two: tables are kept per address set (a few KB each) and uploaded through pinned memory
        without blocking the host."""
        key = (gi,) + tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in ps)
        hit = self._tables.get(key)
        if hit is not None:
            return hit[0], hit[1]
        rows, block = [], 0
        for p in ps:
            rows.append((p.data_ptr(), p.grad.data_ptr(), p.numel(), block))
            block += (p.numel() + _CHUNK - 1) // _CHUNK
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        table = host.to(ps[0].device, non_blocking=True)
        if len(self._tables) >= 8:
            self._tables.pop(next(iter(self._tables)))
        # the pinned source stays alive with the entry: under HIP-graph capture the upload is a memcpy node that re-reads it at replay
        self._tables[key] = (table, block, host)
        self.table_uploads += 1
        return table, block

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.get()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise L.MintimeHipError("FusedSGD needs contiguous fp32 parameters and gradients")
                L.ptr(p)                                   # refuses CPU tensors: there is no CPU path
            table, blocks = self._table(gi, ps)
            L.check(lib.mt_sgd_multi(table.data_ptr(), len(ps), blocks, float(group["lr"]), float(group["weight_decay"]),
                                     L.stream_ptr()), "mt_sgd_multi")
        _weights_changed()
        return loss


class _FusedAdamBase(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW (train.py:187-190 pass only lr and weight_decay: betas (0.9, 0.999), eps 1e-8, amsgrad off) with
    step() as ONE multi-tensor launch per parameter group.  State ('step', 'exp_avg', 'exp_avg_sq') uses torch's keys, so
    state_dict() round-trips with the torch optimizers.  The moments of a group live in two flat buffers."""

    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=None):
        if weight_decay is None:
            weight_decay = 1e-2 if self._decoupled else 0.0        # the torch defaults of AdamW / Adam
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}

    def _table(self, gi, ps):
        key = (gi,) + tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), p.numel()) for p in ps)   # gi: (group, step bucket)
        hit = self._tables.get(key)
        if hit is not None:
            return hit[0], hit[1]
        rows, block = [], 0
        for p in ps:
            st = self.state[p]
            rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), block))
            block += (p.numel() + _CHUNK - 1) // _CHUNK
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        table = host.to(ps[0].device, non_blocking=True)
        if len(self._tables) >= 8:
            self._tables.pop(next(iter(self._tables)))
        self._tables[key] = (table, block, host)
        return table, block

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.get()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            fresh = [p for p in ps if "exp_avg" not in self.state[p]]
            if fresh:
                flat = torch.zeros(2, sum(p.numel() for p in fresh), dtype=torch.float32, device=fresh[0].device)
                o = 0
                for p in fresh:
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise L.MintimeHipError("FusedAdam needs contiguous fp32 parameters")
                    L.ptr(p)
                    self.state[p].update(step=0, exp_avg=flat[0, o:o + p.numel()].view(p.shape),
                                         exp_avg_sq=flat[1, o:o + p.numel()].view(p.shape))
                    o += p.numel()
            for p in ps:
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise L.MintimeHipError("FusedAdam needs contiguous fp32 gradients")
            # torch.optim.Adam keeps a step count PER PARAMETER (a parameter that gets its first gradient late, or a loaded state
            # dict with mixed steps, has its own bias corrections): one launch per distinct step value -- one launch in the usual case
            by_step = {}
            for p in ps:
                by_step.setdefault(int(self.state[p]["step"]), []).append(p)
            b1, b2 = group["betas"]
            for t0, members in sorted(by_step.items()):
                t = t0 + 1
                for p in members:
                    self.state[p]["step"] = t
                step_size = group["lr"] / (1.0 - b1 ** t)
                bc2_sqrt = (1.0 - b2 ** t) ** 0.5
                table, blocks = self._table((gi, t0 if len(by_step) > 1 else -1), members)
                L.check(lib.mt_adam_multi(table.data_ptr(), len(members), blocks, float(group["lr"]), float(group["weight_decay"]),
                                          float(b1), float(b2), float(group["eps"]), float(step_size), float(bc2_sqrt),
                                          1 if self._decoupled else 0, L.stream_ptr()), "mt_adam_multi")
        _weights_changed()
        return loss


class FusedAdam(_FusedAdamBase):
    """Drop-in for torch.optim.Adam(parameters, lr=..., weight_decay=...) (train.py:189-190): L2 term added to the gradient."""
    _decoupled = False


class FusedAdamW(_FusedAdamBase):
    """Drop-in for torch.optim.AdamW(parameters, lr=..., weight_decay=...) (train.py:187-188): decoupled weight decay."""
    _decoupled = True


class _BCEWithLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, pos_weight):
        x = logits.reshape(-1).contiguous().float()
        y = labels.reshape(-1).contiguous().float()
        if x.numel() != y.numel():
            raise ValueError("logits and labels must have the same number of elements")
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        L.check(L.get().mt_bce_logits(L.ptr(x), L.ptr(y), float(pos_weight), L.ptr(loss), L.ptr(dx), x.numel(), L.stream_ptr()),
                "mt_bce_logits")
        ctx.save_for_backward(dx)
        ctx.shape = logits.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return (dx * g).reshape(ctx.shape), None, None


def bce_with_logits(logits, labels, pos_weight=None):
    """torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([w]))(logits, labels) on the device (mean reduction)."""
    return _BCEWithLogits.apply(logits, labels, 1.0 if pos_weight is None else float(pos_weight))
