"""Launch plans per network: a phase (EfficientNet forward / backward, TimeSformer forward / backward) is recorded ONCE into the
library (csrc/plan.hip, include/mintime_hip.h "Launch plans") and re-issued from C with one ctypes call per phase afterwards.

The reference's step is Python issuing every op (train.py:332-378); so was this package through round 4: ~790 ctypes launches,
24 ms of host time per 47 ms step.  What a recorded phase needs from the host side, and what this module provides:

  * fixed addresses -- every tensor a recorded call touches is pinned by the plan (lib.ptr does that while recording), so nothing
    is freed or re-used inside a step; the phase's inputs live in static buffers (an input that is another plan's output is used
    in place, anything else is copied in), its outputs are static buffers handed to autograd as fresh aliases each step;
  * the dependency structure -- the side stream's joins are entry points of the library (mt_plan_fork) and part of the recording;
  * the few per-step values that are not launches of the library: drop-connect gates (drawn eagerly, copied into the plan's gate
    buffer), the BatchNorm `num_batches_tracked` counters (one foreach add), the deferred index check of the embeddings;
  * the way out -- anything the recording did not see (other shapes, other parameter storage, a second forward before the backward,
    gradients that already exist and would be accumulated into, stream capture, MT_PLAN=0) runs the eager launch sequence.

A key is recorded the second time it is seen (one-off shapes never pin memory).  Memory: a plan pins every buffer of its phase, ~25 GB
for BASELINE config 3 (activations + every backward temporary), against 288 GB of HBM.
"""
import os
import weakref

import torch

from . import lib as L

ENABLED = os.environ.get("MT_PLAN", "1") != "0"
RECORD_AFTER = int(os.environ.get("MT_PLAN_AFTER", "1"))     # eager occurrences of a key before it is recorded
MAX_PLANS = int(os.environ.get("MT_PLAN_MAX", "2"))          # per module

_REG = weakref.WeakKeyDictionary()   # module -> {key: NetPlan}; kept OUT of the module's __dict__ (plans hold ctypes handles: a
                                     # copy.deepcopy / pickle of the module must not meet them)
OWNED = set()                 # storage addresses of plan-owned output buffers (feat, dfeat, logits): consumed in place by the next plan
ALL = weakref.WeakSet()       # live NetPlans (bench.py reads their probes)
PROBE_MASK = [0]              # bit t: time the calls tagged t (lib.TAG_*) inside mt_plan_run
STATS = {"recorded": 0, "replayed": 0, "eager_in_flight": 0, "eager_accumulate": 0, "dropped": 0}


class _Token:
    """Lives in the autograd node of a planned forward; when the node dies (backward done, or the graph dropped) the plan's static
    buffers are free for the next forward."""
    __slots__ = ("__weakref__",)


class NetPlan:
    def __init__(self, key):
        self.key = key
        self.seen = 0
        self.fwd = None            # L.Plan
        self.bwd = None
        self.broken = False
        self.in_flight = False
        self.live = None           # weak reference to the token of the forward that currently owns the static buffers
        self.inputs = {}           # name -> static tensor the recorded calls read
        self.state_ptrs = None
        self.stream = None
        self.extra = {}            # engine-specific: saved activations, outputs, gate buffer, tracked counters, gradient views ...
        ALL.add(self)

    def begin(self):
        tok = _Token()
        self.in_flight = True
        self.live = weakref.ref(tok)
        weakref.finalize(tok, _release, weakref.ref(self), id(tok))
        return tok

    def release(self, token=None):
        """The forward that holds `token` is done with the static buffers.  Only the LIVE token may clear `in_flight`: the token of
        step N sits on its autograd node and can die during step N+1, after begin() has handed the buffers to the new forward
        (a stale release there would let a second same-key forward replay over activations a pending backward still needs)."""
        if token is None or (self.live is not None and self.live() is token):
            self.in_flight = False
            self.live = None

    def __del__(self):
        for p in self.extra.get("owned", ()):
            OWNED.discard(p)


def _release(ref, tok_id):
    """Finalizer of a token: clears `in_flight` only when no newer forward has taken the plan since (the dead token's weak
    reference is the one stored in `live`: it now reads None; a newer token's reads the newer token)."""
    np_ = ref()
    if np_ is not None and np_.live is not None and np_.live() is None:
        np_.in_flight = False
        np_.live = None


def own(np_, *tensors):
    for t in tensors:
        if t is not None:
            p = t.untyped_storage().data_ptr()
            OWNED.add(p)
            np_.extra.setdefault("owned", []).append(p)


def state_ptrs(tensors):
    return tuple(0 if t is None else t.data_ptr() for t in tensors)


def lookup(model, key):
    """-> (NetPlan | None, 'eager' | 'record' | 'replay')."""
    if not ENABLED or torch.cuda.is_current_stream_capturing():
        return None, "eager"
    reg = _REG.setdefault(model, {})
    ent = reg.get(key)
    if ent is not None:
        reg[key] = reg.pop(key)           # most recently used last: eviction below takes the least recently used idle plan
    if ent is None:
        if len(reg) >= MAX_PLANS:
            victim = next((k for k, v in reg.items() if not v.in_flight), None)
            if victim is None:
                return None, "eager"
            if reg.pop(victim).fwd is not None:
                STATS["rerecord_risk"] = STATS.get("rerecord_risk", 0) + 1
                if STATS["rerecord_risk"] == 8:       # a workload cycling through more keys than MT_PLAN_MAX re-records (and re-pins) forever
                    import warnings
                    warnings.warn("mintime_amd.plans: recorded launch plans keep being evicted (more distinct shapes / modes per module "
                                  f"than MT_PLAN_MAX={MAX_PLANS}); raise MT_PLAN_MAX or set MT_PLAN=0")
            STATS["dropped"] += 1
        ent = reg[key] = NetPlan(key)
    if ent.broken:
        return None, "eager"
    if ent.fwd is None:
        ent.seen += 1
        if ent.seen <= RECORD_AFTER:
            return None, "eager"
        return ent, "record"
    if ent.in_flight:
        STATS["eager_in_flight"] += 1
        return None, "eager"
    return ent, "replay"


def drop(model, np_):
    _REG.get(model, {}).pop(np_.key, None)
    STATS["dropped"] += 1


def static_input(np_, name, t):
    """Recording: the tensor the phase reads input `name` from -- `t` itself when it is (a contiguous alias of) another plan's
    static output, otherwise a buffer of the plan's own holding a copy."""
    if t is None:
        np_.inputs[name] = None
        return None
    if t.is_contiguous() and t.untyped_storage().data_ptr() in OWNED:
        buf = t
    else:
        buf = torch.empty(t.shape, dtype=t.dtype, device=t.device)
        buf.copy_(t)
    np_.inputs[name] = buf
    return buf


def refresh_input(np_, name, t):
    """Replay: make the static buffer hold this step's input (nothing to do when the input IS the buffer)."""
    buf = np_.inputs[name]
    if buf is None:
        if t is not None:
            raise L.MintimeHipError(f"launch plan: input {name} was absent when the phase was recorded")
        return
    if t is None or t.shape != buf.shape:
        raise L.MintimeHipError(f"launch plan: input {name} changed shape")
    if t.data_ptr() != buf.data_ptr() or t.stride() != buf.stride():
        buf.copy_(t, non_blocking=True)


def run(plan):
    plan.run(PROBE_MASK[0])
    STATS["replayed"] += 1


def grads_exist(params):
    return any(p is not None and p.grad is not None for p in params)


def fresh_aliases(views):
    """New tensor objects over the plan's gradient buffer: autograd adopts a gradient it holds the only reference to (no copy)."""
    return [None if v is None else v.detach() for v in views]


def probe_totals(tag):
    """(launches, seconds, work) of the calls tagged `tag`, summed over every live plan; clears the readings."""
    n, ms, w = 0, 0.0, 0.0
    for np_ in list(ALL):
        for pl in (np_.fwd, np_.bwd):
            if pl is not None:
                a, b, c = pl.probe_read(tag)
                n, ms, w = n + a, ms + b, w + c
    return n, ms * 1e-3, w
