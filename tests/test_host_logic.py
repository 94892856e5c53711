"""CPU-only tests: host logic, state-dict compatibility, C-ABI surface (no compute calls without a GPU)."""
import ctypes
import json
import os
import re

import pytest
import torch

import mintime_amd
from mintime_amd import arch, synth, lib, ddp, EfficientNet, SizeInvariantTimeSformer
from tests.util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function include/mintime_hip.h declares is exported by the built .so and bound in lib.PROTOTYPES."""
    hdr = open(os.path.join(ROOT, "include", "mintime_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(mt_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    handle = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert declared == set(lib.PROTOTYPES), declared ^ set(lib.PROTOTYPES)
    assert lib.get().mt_version() == lib.header_version() == lib.ABI_VERSION


def test_every_launching_entry_point_has_a_recording_thunk():
    """Launch plans (header section "Launch plans"): csrc/gen_plan.py turns every declaration whose last parameter is `void* stream`
    into an exported thunk `mt_xxx` around the kernels' own entry point `mti_xxx`; both must be in the library, and the plan API itself
    must not be wrapped."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_plan", os.path.join(lib.CSRC, "gen_plan.py"))
    gp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gp)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mintime_hip.h")).read()
    launching = [name for ret, name, ps in gp.prototypes(hdr) if ps[-1] == ("void*", "stream") and ret == "int"]
    assert len(launching) >= 50 and "mt_gemm" in launching and "mt_memset_async" in launching
    handle = ctypes.CDLL(lib.LIB_PATH)
    for name in launching:
        assert hasattr(handle, name) and hasattr(handle, "mti_" + name[3:]), name
    for name in ("mt_plan_run", "mt_plan_fork", "mt_plan_create", "mt_version"):
        assert hasattr(handle, name) and not hasattr(handle, "mti_" + name[3:]), name
    # descriptors are captured by value: the generated thunk copies *d before the closure is made
    thunks = open(os.path.join(lib.CSRC, "plan_thunks.inc")).read()
    assert "const auto d_v = *d;" in thunks and "mti_gemm_planes(&d_v, stream)" in thunks


def test_launch_plan_registry_state_machine(monkeypatch):
    """plans.lookup: a key runs eagerly the first time it is seen, is recorded the second time, replays afterwards; while a planned
    forward's autograd node is alive the plan's static buffers are taken (a second forward goes eager); at most MT_PLAN_MAX plans per
    module, the oldest idle one makes room; the registry lives outside the module (deepcopy / pickle never meet a ctypes handle)."""
    import copy
    import gc
    import pickle
    from mintime_amd import plans
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(plans, "ENABLED", True)
    monkeypatch.setattr(plans, "MAX_PLANS", 2)
    monkeypatch.setattr(plans, "RECORD_AFTER", 1)
    m = torch.nn.Linear(2, 2)
    assert plans.lookup(m, "a") == (None, "eager")
    np_a, mode = plans.lookup(m, "a")
    assert mode == "record" and np_a is not None
    np_a.fwd = object()                                   # (the engine stores the recorded L.Plan here)
    assert plans.lookup(m, "a") == (np_a, "replay")
    tok = np_a.begin()                                    # a planned forward whose backward has not run yet
    before = plans.STATS["eager_in_flight"]
    assert plans.lookup(m, "a") == (None, "eager") and plans.STATS["eager_in_flight"] == before + 1
    del tok
    gc.collect()                                          # the autograd node died: the buffers are free again
    assert plans.lookup(m, "a") == (np_a, "replay")
    np_a.begin()
    np_a.release()                                        # ... or the backward ran
    assert plans.lookup(m, "a")[1] == "replay"
    # a stale token must not free buffers a NEWER forward has taken: the token of step N lives on its autograd node and can die
    # during step N+1 (reference-style loop: `loss` is rebound after the next forward has begun)
    tok_old = np_a.begin()
    np_a.release(tok_old)                                 # backward of step N
    tok_new = np_a.begin()                                # forward of step N+1
    del tok_old
    gc.collect()                                          # step N's graph dies now
    assert np_a.in_flight and plans.lookup(m, "a") == (None, "eager")
    np_a.release(object())                                # a release carrying a token that is not the live one: ignored
    assert np_a.in_flight
    np_a.release(tok_new)
    assert not np_a.in_flight and plans.lookup(m, "a") == (np_a, "replay")
    plans.lookup(m, "b")
    np_b, mode = plans.lookup(m, "b")
    assert mode == "record"
    np_b.fwd = object()
    tok_b = np_b.begin()
    assert plans.lookup(m, "c") == (None, "eager")        # third key: evicts the oldest IDLE plan ("a"), then counts its first sighting
    assert plans.lookup(m, "a") == (None, "eager")        # ... "a" starts over (and "b", in flight, stayed)
    assert plans.lookup(m, "b") == (None, "eager") and np_b.in_flight
    del tok_b
    np_b.broken = True                                    # a recording that raised: that key stays eager
    gc.collect()
    assert plans.lookup(m, "b") == (None, "eager")
    assert "_mt_plans" not in m.__dict__
    pickle.loads(pickle.dumps(copy.deepcopy(m)))          # nothing plan-related travels with the module
    monkeypatch.setattr(plans, "ENABLED", False)
    assert plans.lookup(m, "a") == (None, "eager")


def test_errors_are_reported_not_swallowed():
    h = lib.get()
    d = lib.GemmDesc()          # all-null descriptor
    rc = h.mt_gemm(ctypes.byref(d), None)
    assert rc < 0 and b"null pointer" in h.mt_last_error()
    with pytest.raises(lib.MintimeHipError):
        lib.ptr(torch.zeros(4))                       # CPU tensors are refused: no CPU fallback exists


def test_modules_refuse_cpu_tensors():
    cfg = arch.default_tsf_config(1280, 8)
    tsf = SizeInvariantTimeSformer(config=cfg)
    with pytest.raises(lib.MintimeHipError):
        tsf(torch.zeros(1, 8, 1280, 7, 7), mask=torch.ones(1, 8, dtype=torch.bool),
            identities_mask=torch.ones(1, 8, 8, dtype=torch.bool), size_embedding=torch.ones(1, 8, dtype=torch.int32),
            positions=torch.zeros(1, 393, dtype=torch.long))
    ef = EfficientNet.from_name("efficientnet-b0")
    with pytest.raises(lib.MintimeHipError):
        ef(torch.zeros(1, 3, 224, 224))


def test_state_dicts_match_reference_manifest():
    with open(os.path.join(GOLDEN, "state_manifest.json")) as fh:
        man = json.load(fh)
    ef = EfficientNet.from_name("efficientnet-b0")
    assert [[k, list(v.shape), str(v.dtype)] for k, v in ef.state_dict().items()] == man["efficientnet-b0"]
    for (c, f) in ((1280, 8), (2048, 16)):
        m = SizeInvariantTimeSformer(config=arch.default_tsf_config(c, f))
        ours = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
        ref = {k: [s, d] for k, s, d in man[f"tsf_c{c}_f{f}"]}
        assert ours == ref
        assert sorted(m.no_weight_decay()) == man[f"tsf_c{c}_f{f}_no_weight_decay"]
    # train.py:159-167 parses parameter names as "_blocks.<i>.<sub>"
    for name, _ in ef.named_parameters():
        if "blocks" in name:
            assert 0 <= int(name.split(".")[1]) < 16


def test_drop_in_import_paths():
    """The reference's import lines (train.py:27-28) resolve to the HIP-backed classes."""
    from models.efficientnet.efficientnet_pytorch import EfficientNet as E2
    from models.size_invariant_timesformer import SizeInvariantTimeSformer as T2
    assert E2 is EfficientNet and T2 is SizeInvariantTimeSformer


def test_reference_config_yaml_schema_is_accepted():
    import yaml
    text = """
model: {image-size: 224, patch-size: 1, num-classes: 1, num-patches: 49, num-frames: 16, max-identities: 2, dim: 512,
        depth: 9, dim-head: 64, channels: 2048, heads: 8, attn-dropout: 0., ff-dropout: 0., shift-tokens: False,
        enable-size-emb: True, enable-pos-emb: True, enable-identity-attention: True}
"""
    m = SizeInvariantTimeSformer(config=yaml.safe_load(text))
    assert m.pos_emb.weight.shape == (16 * 2048 + 1, 512)
    bad = yaml.safe_load(text)
    bad["model"]["shift-tokens"] = True
    with pytest.raises(NotImplementedError):
        SizeInvariantTimeSformer(config=bad)
    # dropout > 0 is accepted like the reference's constructor (size_invariant_timesformer.py:89-106); out of range raises
    drop = yaml.safe_load(text)
    drop["model"]["attn-dropout"], drop["model"]["ff-dropout"] = 0.1, 0.2
    md = SizeInvariantTimeSformer(config=drop)
    assert (md.attn_dropout, md.ff_dropout) == (0.1, 0.2) and md.dropout_uniform is None
    drop["model"]["ff-dropout"] = 1.0
    with pytest.raises(ValueError):
        SizeInvariantTimeSformer(config=drop)


def test_load_matching_state_dict_semantics():
    ef = EfficientNet.from_name("efficientnet-b0")
    sd = synth.effnet_b0_state(3)
    prefixed = {"efficient_net." + k: v for k, v in sd.items()}
    prefixed["not.a.key"] = torch.zeros(3)
    ef.load_matching_state_dict(prefixed)
    assert torch.equal(ef.state_dict()["_blocks.3._bn1.running_var"], sd["_blocks.3._bn1.running_var"])


def test_b0_block_table():
    b = arch.effnet_b0_blocks()
    assert len(b) == 16
    assert [(x.k, x.s, x.cin, x.cout, x.hin, x.hout, x.pad0, x.pad1, x.skip) for x in (b[0], b[1], b[3], b[11], b[15])] == [
        (3, 1, 32, 16, 112, 112, 1, 1, False), (3, 2, 16, 24, 112, 56, 0, 1, False), (5, 2, 24, 40, 56, 28, 1, 2, False),
        (5, 2, 112, 192, 14, 7, 1, 2, False), (3, 1, 192, 320, 7, 7, 1, 1, False)]
    assert [x.idx for x in b if x.skip] == [2, 4, 6, 7, 9, 10, 12, 13, 14]
    assert [x.cse for x in b[:4]] == [8, 4, 6, 6]
    assert arch.same_pad(224, 3, 2) == (0, 1)


def test_synthetic_inputs_follow_the_dataset_contract():
    inp = synth.clip_inputs(2, 8, 2, seed=0, ragged=True, with_video=False)
    assert inp["identities_mask"][0].int().tolist() == [[1] * 4 + [0] * 4] * 4 + [[0] * 4 + [1] * 4] * 4
    assert inp["mask"][0].tolist() == [True, True, True, False] * 2
    pos = inp["positions"][0]
    assert pos[0] == 0 and pos[1] == 1 and pos[49] == 49 and pos[50] == 50
    assert pos[1 + 3 * 49] == 2 * 49 + 1            # padded slot re-uses the previous temporal rank
    assert pos[1 + 4 * 49] == 1                     # second identity restarts at rank 1
    assert synth.identity_split(16, 3) == [7, 5, 4]
    a, b = synth.effnet_b0_state(5), synth.effnet_b0_state(5)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_shard_range():
    assert [ddp.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert ddp.shard_range(256, 7, 8) == (224, 256)


def _gloo_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    ps[0].grad = torch.full((5, 3), float(rank + 1))
    ps[1].grad = torch.arange(7.0) * (rank + 1)
    # ps[2] gets no gradient on any rank (like EfficientNet._fc): skipped deterministically
    red = ddp.GradAllReducer(ps)
    n = red.allreduce()
    red.allreduce()                                   # second step re-uses the flat buffer
    out[rank] = (n, ps[0].grad.clone(), ps[1].grad.clone(), ps[2].grad)
    dist.destroy_process_group()


def test_gradient_allreduce_world_size_2_gloo():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in (0, 1):
        n, g0, g1, g2 = out[rank]
        assert n == 22
        assert torch.allclose(g0, torch.full((5, 3), 1.5))
        assert torch.allclose(g1, torch.arange(7.0) * 1.5)
        assert g2 is None


class _FlatGradFn(torch.autograd.Function):
    """Mimics the engines: gradients of all parameters are views of one flat buffer."""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(x)
        return x.sum() * sum(p.sum() for p in params)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        total = sum((torch.Size(s).numel() + 3) // 4 * 4 for s in ctx.shapes)
        flat = torch.full((total,), float(x.sum()) * float(g))
        out, off = [], 0
        for s in ctx.shapes:
            n = torch.Size(s).numel()
            out.append(flat[off:off + n].view(s))
            off += (n + 3) // 4 * 4
        return (None,) + tuple(out)


def _overlap_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.ones(5, 3)), torch.nn.Parameter(torch.ones(7))]          # "TimeSformer": flat-view gradients
    lin = torch.nn.Linear(4, 2)                                                              # "EfficientNet": ordinary gradients
    unused = torch.nn.Parameter(torch.zeros(3))                                              # like _fc: never gets a gradient
    red = ddp.OverlappedGradReducer([a, list(lin.parameters()) + [unused]])
    res = []
    for step in range(3):
        for p in a + list(lin.parameters()):
            p.grad = None
        x = torch.full((2, 4), float(rank + 1 + step))
        loss = _FlatGradFn.apply(lin(x), *a) + lin(x).sum()
        loss.backward()
        n = red.allreduce()
        res.append((n, a[0].grad.clone(), a[1].grad.clone(), lin.weight.grad.clone(), unused.grad))
    out[rank] = (res, dict(red.stats))
    dist.destroy_process_group()


def test_overlapped_bucketed_allreduce_world_size_2_gloo():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_overlap_worker, args=(2, port, out), nprocs=2, join=True)
    (r0, s0), (r1, s1) = out[0], out[1]
    assert s0 == s1
    assert s0["synchronous"] == 2 and s0["overlapped_launches"] == 4          # step 0 learns the layout, steps 1-2 overlap
    assert s0["in_place"] >= 3 and s0["staged"] >= 3                          # flat-view bucket in place, Linear bucket staged
    for step in range(3):
        for k in range(1, 4):
            assert torch.equal(r0[step][k], r1[step][k])                      # every rank ends with the same averaged gradient
        assert r0[step][4] is None
    # the averaged gradient is the mean of what each rank computed alone
    lin = torch.nn.Linear(4, 2)
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 2)
    want = 0
    for rank in (0, 1):
        x = torch.full((2, 4), float(rank + 1 + 2))
        want = want + float(lin(x).sum().detach()) / 2
    assert torch.allclose(r0[2][1], torch.full((5, 3), want), rtol=1e-5)


class _EngineLike(torch.nn.Module):
    """Stands in for an engine-backed network on the CPU: its backward builds every gradient as a view of one flat buffer with
    lib.zero_grads and announces it with lib.grads_ready, exactly like tsf_backward / effnet_backward do on the GPU."""

    def __init__(self, shapes, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes])

    def forward(self, x):
        model = self

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, *params):
                ctx.save_for_backward(x)
                return x * sum(p.sum() for p in params)

            @staticmethod
            def backward(ctx, gout):
                (x,) = ctx.saved_tensors
                params = list(model.ps)
                grads, flat = lib.zero_grads(params, with_flat=True)
                for gr in grads:
                    gr += float((gout * x).sum())
                lib.grads_ready(model, params, flat)
                return (gout * sum(p.sum() for p in params).detach(),) + tuple(grads)

        return Fn.apply(x, *self.ps)


def _engine_hook_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    second, first = _EngineLike([(3, 5), (7,), (2, 2, 2)], 1), _EngineLike([(4,), (6, 2)], 2)     # "tsf" consumes "ef"'s output
    red = ddp.OverlappedGradReducer([second, first])                # order in which backward finishes them
    res = []
    for step in range(3):
        for p in list(first.parameters()) + list(second.parameters()):
            p.grad = None
        x = torch.full((3,), float(rank + 1 + step))
        second(first(x)).sum().backward()
        n = red.allreduce()
        res.append((n, [p.grad.clone() for p in second.parameters()], [p.grad.clone() for p in first.parameters()]))
    # a step that starts from existing gradients must not be reduced under autograd's feet: it falls back to the synchronous path
    x = torch.full((3,), float(rank + 7))
    second(first(x)).sum().backward()
    red.allreduce()
    out[rank] = (res, dict(red.stats), [p.grad.clone() for p in second.parameters()])
    dist.destroy_process_group()


def test_engine_level_bucket_hooks_world_size_2_gloo():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_engine_hook_worker, args=(2, port, out), nprocs=2, join=True)
    (r0, s0, acc0), (r1, s1, acc1) = out[0], out[1]
    assert s0 == s1
    assert s0["overlapped_launches"] == 6 and s0["synchronous"] == 2, s0      # 3 clean steps x 2 buckets early; the 4th step late
    for step in range(3):
        assert r0[step][0] == r1[step][0] > 0
        for a, b in zip(r0[step][1] + r0[step][2], r1[step][1] + r1[step][2]):
            assert torch.equal(a, b)                                           # same averaged gradient on both ranks
    # value check: gradient of every element of `second`'s parameters is sum(first(x)) on each rank; the average of ranks 0 and 1
    first = _EngineLike([(4,), (6, 2)], 2)
    want = sum(float(first(torch.full((3,), float(rank + 1 + 2))).sum().detach()) for rank in (0, 1)) / 2
    assert torch.allclose(r0[2][1][0], torch.full((3, 5), want), rtol=1e-5)
    for a, b in zip(acc0, acc1):
        assert torch.equal(a, b)


class _EngineEmbed(_EngineLike):
    """An engine with an embedding-like table whose gradient only touches its first rows (the TimeSformer's pos_emb / size_emb:
    [num_frames * channels + 1, dim] tables of which F * 49 + 1 / 21 rows are ever indexed)."""

    def __init__(self, live):
        super().__init__([(6,), (40, 4), (5,), (30, 4), (3,)], 3)
        self.live = live

    def forward(self, x):
        model = self

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, *params):
                ctx.save_for_backward(x)
                return x * sum(p.sum() for p in params)

            @staticmethod
            def backward(ctx, gout):
                (x,) = ctx.saved_tensors
                params = list(model.ps)
                grads, flat = lib.zero_grads(params, with_flat=True)
                for p, gr in zip(params, grads):
                    rows = model.live.get(id(p))
                    (gr if rows is None else gr[:rows]).add_(float((gout * x).sum()))
                lib.grads_ready(model, params, flat)
                return (gout * sum(p.sum() for p in params).detach(),) + tuple(grads)

        return Fn.apply(x, *self.ps)


def _live_rows_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for tag, declare in (("full", False), ("live", True), ("wrong", True)):
        net = _EngineEmbed({})
        ps = list(net.parameters())
        net.live = {id(ps[1]): 3, id(ps[3]): 7} if tag != "wrong" else {id(ps[1]): 5, id(ps[3]): 7}    # rows the backward really touches
        red = ddp.OverlappedGradReducer([net], live_rows={ps[1]: 3, ps[3]: 7} if declare else None)
        try:
            net(torch.full((3,), float(rank + 1))).sum().backward()
            red.allreduce()
            res[tag] = ([p.grad.clone() for p in ps], dict(red.stats), list(red._segments.values()))
        except RuntimeError as e:
            res[tag] = str(e)
            for w_, _, _ in red.pending:
                for x_ in (w_ if isinstance(w_, list) else [w_]):
                    x_.wait()
    out[rank] = res
    dist.destroy_process_group()


def test_live_rows_of_embedding_tables_world_size_2_gloo():
    """SURVEY 8(e) / round-5 verdict item 8: only the live rows of the TimeSformer's embedding-gradient tables go on the wire.  Same
    averaged gradients as the full all-reduce; the wire ranges skip the dead rows; a bound the data violates is refused loudly."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 36500 + (os.getpid() % 2000)
    mp.spawn(_live_rows_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    for a, b, c, d in zip(r0["full"][0], r0["live"][0], r1["full"][0], r1["live"][0]):
        assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    assert torch.allclose(r0["live"][0][1][:3], torch.full((3, 4), 4.5)) and not r0["live"][0][1][3:].any()     # mean of 3 and 6; dead rows zero
    (segs, dead), = r0["live"][2]
    # flat layout: 6 -> [0, 8); table 40 x 4 at 8: live [8, 20), dead [20, 168); 5 -> [168, 176); table 30 x 4 at 176: live [176, 204), dead [204, 296); 3 -> [296, 300)
    assert segs == [(0, 20), (168, 204), (296, 300)] and dead == [(20, 168), (204, 296)]
    assert sum(b - a for a, b in segs) == 60 and r0["live"][1]["overlapped_launches"] == 1
    assert "live_rows" in r0["wrong"] and "live_rows" in r1["wrong"]


class _EngineLikeOptional(_EngineLike):
    """An engine whose parameter list has an empty slot (SizeInvariantTimeSformer._param_list() puts None where size_emb would
    be when enable-size-emb is False): the None is handed to lib.zero_grads / lib.grads_ready exactly like tsf_backward does."""

    def forward(self, x):
        model = self

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, *params):
                ctx.save_for_backward(x)
                return x * sum(p.sum() for p in params)

            @staticmethod
            def backward(ctx, gout):
                (x,) = ctx.saved_tensors
                real = list(model.ps)
                params = real[:1] + [None] + real[1:]
                grads, flat = lib.zero_grads(params, with_flat=True)
                for gr in grads:
                    if gr is not None:
                        gr += float((gout * x).sum())
                lib.grads_ready(model, params, flat)
                return (gout * sum(p.sum() for p in real).detach(),) + tuple(g for g in grads if g is not None)

        return Fn.apply(x, *self.ps)


def _advice_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    # (1) replicas that start from different weights are made identical at construction
    net = _EngineLikeOptional([(3, 5), (7,), (2, 3)], seed=10 + rank)
    before = [p.detach().clone() for p in net.parameters()]
    red = ddp.OverlappedGradReducer([net])
    res["init"] = ([p.detach().clone() for p in net.parameters()], before)
    # (2) a None slot in the engine's parameter list (no space in the flat buffer) is reduced correctly, in place
    x = torch.full((3,), float(rank + 1))
    net(x).sum().backward()
    red.allreduce()
    res["none_slot"] = ([p.grad.clone() for p in net.parameters()], dict(red.stats))
    # (3) gradient accumulation: two micro-batches, the first under no_sync()
    for p in net.parameters():
        p.grad = None
    with red.no_sync():
        net(torch.full((3,), float(rank + 1))).sum().backward()
    net(torch.full((3,), float(10 * (rank + 1)))).sum().backward()
    red.allreduce()
    res["accum"] = [p.grad.clone() for p in net.parameters()]
    # (4) two backward passes without no_sync(): refused loudly instead of silently dropping the second micro-batch
    for p in net.parameters():
        p.grad = None
    net(x).sum().backward()
    try:
        net(x).sum().backward()
        res["double"] = "no error"
    except RuntimeError as e:
        res["double"] = str(e)
    red.allreduce()
    out[rank] = res
    dist.destroy_process_group()


def test_reducer_broadcast_none_slot_accumulation_world_size_2_gloo():
    """initial broadcast, None parameter slots, gradient accumulation (no_sync) and the double-backward guard."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_advice_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    for a, b in zip(r0["init"][0], r1["init"][0]):
        assert torch.equal(a, b)                                   # rank 1 adopted rank 0's weights
    assert any(not torch.equal(a, b) for a, b in zip(r1["init"][0], r1["init"][1]))
    # after the broadcast both ranks hold rank 0's weights: d/dp = sum(x) = 3*(rank+1); mean over ranks = 4.5
    for g0, g1 in zip(r0["none_slot"][0], r1["none_slot"][0]):
        assert torch.equal(g0, g1) and torch.allclose(g0, torch.full_like(g0, 4.5))
    assert r0["none_slot"][1]["overlapped_launches"] == 1
    # accumulation: per rank 3*(r+1) + 30*(r+1) = 33*(r+1); mean of 33 and 66 = 49.5
    for g0, g1 in zip(r0["accum"], r1["accum"]):
        assert torch.equal(g0, g1) and torch.allclose(g0, torch.full_like(g0, 49.5))
    assert "no_sync" in r0["double"] and "no_sync" in r1["double"]


def test_bench_self_launches_multi_rank_and_prints_one_json_line_last():
    """`python bench.py --gpus 2` (what the driver's scaling run invokes, no launcher around it) must start its own ranks under
    torch.distributed.run and leave exactly ONE JSON line, last, on stdout.  The launcher / barrier / max-over-ranks plumbing is
    exercised here on gloo with a stand-in step (--stub); the product step itself needs GPUs."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["checksum"] == 1024 * 2.0 * (1 + 2)          # both ranks took part in the all-reduce
    pr = d["config"]["per_rank"]                                     # every rank's own readings reach the line (all_gather)
    assert pr["rank"] == [0, 1] and len(pr["ms_per_step"]) == 2 and all(v > 0 for v in pr["ms_per_step"])


def test_launch_threads_get_disjoint_core_slices_only_when_there_are_enough_cores(monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pinned = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: pinned.setdefault("cores", sorted(cores)), raising=False)
    assert bench.pin_launch_thread(0, 1) is None and not pinned                  # one rank: nothing to fence off
    assert bench.pin_launch_thread(3, 8) == [24, 31] and pinned["cores"] == list(range(24, 32))
    pinned.clear()
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)), raising=False)
    assert bench.pin_launch_thread(3, 8) is None and not pinned                  # 2 cores per rank: left to the OS


def test_bench_drops_pmc_counters_measured_on_other_sources(tmp_path, monkeypatch):
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "PMC_FILE", str(f))
    f.write_text(json.dumps({"csrc_sha16": bench.csrc_hash(), "tsf_wgrad": {"bytes_per_launch": 1.0}}))
    assert bench.committed_counters("tsf_wgrad") == {"bytes_per_launch": 1.0}
    f.write_text(json.dumps({"csrc_sha16": "0" * 16, "tsf_wgrad": {"bytes_per_launch": 1.0}}))
    assert bench.committed_counters("tsf_wgrad") is None             # stale counters are not reported as current
    # no roofline object may print a fraction above 1: the split pipe is priced against its own ceiling
    r = bench.mfma_roofline(170e12, True)
    assert r["peak"] == 416.7 and r["frac"] < 1 and r["frac_vs_fp32_mfma"] > 1
    assert bench.mfma_roofline(100e12, False)["peak"] == 157.3


def test_plane_tensor_layout_spec():
    """The operand format of mt_gemm_planes, restated with plain torch on the CPU: planes[3][Rp/32][Cp/16][32][16] bf16 holding the
    exact three-piece split x = p0 + p1 + p2 (round-to-nearest bf16 at each level) with zero padding -- what every producer kernel
    writes and what lib.planes_to_float() undoes.  mt_planes_elems (a host function of the library) gives the plane stride."""
    h = lib.get()
    for rows, cols in [(77, 40), (64, 32), (1, 8), (393 * 16, 512)]:
        rp, cp = (rows + 31) // 32 * 32, (cols + 15) // 16 * 16
        assert h.mt_planes_elems(rows, cols) == rp * cp
        assert lib.planes_shape(rows, cols) == (3, rp // 32, cp // 16, 32, 16)
        g = torch.Generator().manual_seed(rows)
        x = torch.randn(rows, cols, generator=g) * torch.logspace(-6, 3, cols)[None, :]
        p0 = x.bfloat16()
        r1 = x - p0.float()
        p1 = r1.bfloat16()
        p2 = (r1 - p1.float()).bfloat16()
        assert torch.equal(p0.float() + p1.float() + p2.float(), x), "three bf16 pieces carry all 24 mantissa bits"
        full = torch.zeros(3, rp, cp, dtype=torch.bfloat16)
        full[0, :rows, :cols], full[1, :rows, :cols], full[2, :rows, :cols] = p0, p1, p2
        planes = full.reshape(3, rp // 32, 32, cp // 16, 16).permute(0, 1, 3, 2, 4).contiguous()      # 1 KB blocks of 32 rows x 16 columns
        assert planes.shape == lib.planes_shape(rows, cols)
        assert torch.equal(lib.planes_to_float(planes, rows, cols), x)
        # element (r, c) of plane k sits at ((r // 32) * (cp // 16) + c // 16) * 512 + (r % 32) * 16 + c % 16
        r, c = rows - 1, cols - 1
        off = ((r // 32) * (cp // 16) + c // 16) * 512 + (r % 32) * 16 + c % 16
        assert planes[1].reshape(-1)[off] == p1[r, c]


def test_deterministic_switch_round_trips_without_a_gpu():
    """mt_set_deterministic / mt_get_deterministic (train.py:110's switch) are plain host state: callable on a CPU-only box."""
    prev = lib.set_deterministic(True)
    try:
        assert lib.deterministic() is True and lib.get().mt_get_deterministic() == 1
        assert lib.set_deterministic(False) is True
        assert lib.deterministic() is False
    finally:
        lib.set_deterministic(prev)


def test_planes_fit_is_the_four_gigabyte_rule_of_the_plane_loop():
    """lib.planes_fit mirrors csrc/gemm_planes.hip: three bf16 planes of the padded operand (rows to 32, columns to 16) must stay within
    32-bit byte offsets.  The cases are the operands of BASELINE configs 3 / 5 and the one that wrapped before the guard existed."""
    from mintime_amd import lib as L
    assert L.planes_fit(32 * 785, 4096)                   # TimeSformer feed-forward operand, B = 32
    assert L.planes_fit(222 * 785, 4096) and not L.planes_fit(223 * 785, 4096)
    assert L.planes_fit(512 * 55 * 55, 256)               # Xception block 2 at 512 crops
    assert not L.planes_fit(512 * 109 * 109, 128)         # block 1: 4.67 GB, stays on mt_gemm
    assert L.planes_fit(1, 1) and L.planes_fit(31, 15) == L.planes_fit(32, 16)
    rows = (0xFFFFFFFF // (6 * 16)) // 32 * 32
    assert L.planes_fit(rows, 16) and not L.planes_fit(rows + 1, 16)
