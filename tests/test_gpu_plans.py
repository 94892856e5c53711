"""Launch plans (include/mintime_hip.h "Launch plans", csrc/plan.hip, plans.py): a phase recorded once and re-issued from C must be
the eager launch sequence -- same kernels, same order, same values.  In deterministic mode that is checkable bit for bit: N training
steps (train.py:332-378) with plans give exactly the parameters, running statistics and losses of N eager steps."""
import ctypes

import pytest
import torch

import mintime_amd
from mintime_amd import harness, plans, synth
from mintime_amd import lib as L
from tests.util import REL_TOL, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture
def det_mode():
    prev = L.set_deterministic(True)
    yield
    L.set_deterministic(prev)


@pytest.fixture
def plan_switch():
    prev = plans.ENABLED
    yield
    plans.ENABLED = prev


def _state(*models):
    out = {}
    for tag, m in zip(("ef.", "tsf."), models):
        for k, v in m.state_dict().items():
            out[tag + k] = v.detach().clone()
    return out


def _train(n_steps, plan_on, B=2, ids=2, ragged=True, optimizer="sgd", uint8=False):
    plans.ENABLED = plan_on
    cfg, ef, tsf = harness.build_models(8, seed=4, device="cuda")
    cfg["training"]["optimizer"] = optimizer
    opt = harness.make_optimizer(cfg, ef, tsf)
    torch.manual_seed(123)                       # the drop-connect draws (torch.rand on the device) repeat across the two runs
    losses = []
    for i in range(n_steps):
        batch = harness.device_batch(B, 8, ids, seed=i, device="cuda", ragged=ragged and i % 2 == 1, as_uint8=uint8)
        losses.append(harness.train_step(ef, tsf, opt, batch).detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses), _state(ef, tsf), (ef, tsf, opt)


def test_plan_records_and_replays_calls_in_order():
    """The C layer on its own: recorded fills and copies re-run in order with the recorded arguments; tags are timed on request."""
    lib = L.get()
    a = torch.full((1024,), 7.0, device="cuda")
    b = torch.zeros(1024, device="cuda")
    pl = L.Plan()
    with pl:
        L.tag_next(5, 4096.0)
        L.check(lib.mt_copy_async(L.ptr(b), L.ptr(a), 4096, L.stream_ptr()), "copy")
        L.zero_(a)
    assert pl.ops == 2 and float(b.sum()) == 7.0 * 1024 and float(a.sum()) == 0.0
    a.fill_(3.0)
    b.zero_()
    pl.run(1 << 5)
    torch.cuda.synchronize()
    assert float(b.sum()) == 3.0 * 1024 and float(a.sum()) == 0.0
    n, ms, work = pl.probe_read(5)
    assert n == 1 and ms >= 0.0 and work == 4096.0
    assert pl.probe_read(5)[0] == 0                  # readings are cleared
    with pytest.raises(L.MintimeHipError):           # a plan holds one recording
        with pl:
            pass


def test_fork_orders_two_streams_inside_a_plan():
    lib = L.get()
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    src = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    mid = torch.zeros_like(src)
    dst = torch.zeros_like(src)
    pl = L.Plan()
    with pl:
        L.check(lib.mt_copy_async(L.ptr(mid), L.ptr(src), src.numel() * 4, ctypes.c_void_p(main.cuda_stream)), "copy")
        L.check(lib.mt_plan_fork(ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(side.cuda_stream)), "fork")
        L.check(lib.mt_copy_async(L.ptr(dst), L.ptr(mid), src.numel() * 4, ctypes.c_void_p(side.cuda_stream)), "copy")
        L.check(lib.mt_plan_fork(ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(main.cuda_stream)), "fork")
    for rep in range(3):
        src.add_(1.0)
        pl.run()
        torch.cuda.synchronize()
        assert torch.equal(dst, src)


def test_planned_training_is_bit_identical_to_eager(det_mode, plan_switch):
    """6 steps, alternating full / ragged batches with fresh inputs each step: step 1 eager, step 2 recorded, steps 3-6 replayed."""
    before = dict(plans.STATS)
    l_p, s_p, _ = _train(6, True)
    assert plans.STATS["recorded"] - before["recorded"] == 2              # EfficientNet + TimeSformer
    assert plans.STATS["replayed"] - before["replayed"] == 4 * 4            # 4 replayed steps x 4 phases
    l_e, s_e, _ = _train(6, False)
    assert torch.equal(l_p, l_e), (l_p, l_e)
    diff = [k for k in s_e if not torch.equal(s_e[k], s_p[k])]
    assert not diff, f"{len(diff)} of {len(s_e)} state tensors differ between planned and eager training, e.g. {diff[:5]}"


def test_replayed_step_matches_eager_step_default_mode(plan_switch):
    """Default (atomic) mode, uint8 crops: ONE replayed step against the eager step from the same state and batch (several steps of a
    random-init network at B = 2 amplify the atomics' run-to-run rounding too far to compare trajectories)."""
    plans.ENABLED = True
    cfg, ef, tsf = harness.build_models(8, seed=4, device="cuda")
    opt = harness.make_optimizer(cfg, ef, tsf)
    batches = [harness.device_batch(2, 8, 2, seed=i, device="cuda", as_uint8=True) for i in range(4)]
    torch.manual_seed(5)
    for i in range(3):
        harness.train_step(ef, tsf, opt, batches[i])            # eager, recorded, replayed
    snap = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (ef, tsf)]
    rng = torch.cuda.get_rng_state()
    replayed = plans.STATS["replayed"]
    loss_p = harness.train_step(ef, tsf, opt, batches[3]).detach().clone()
    assert plans.STATS["replayed"] == replayed + 4
    s_p = _state(ef, tsf)
    ef.load_state_dict(snap[0]), tsf.load_state_dict(snap[1])
    torch.cuda.set_rng_state(rng)                               # the same drop-connect draws
    plans.ENABLED = False
    loss_e = harness.train_step(ef, tsf, opt, batches[3]).detach().clone()
    s_e = _state(ef, tsf)
    assert_close(loss_p, loss_e, 1e-5, "loss, replayed vs eager step")
    for k in s_e:
        if s_e[k].dtype.is_floating_point and float(s_e[k].abs().max()) > 0:
            assert_close(s_p[k], s_e[k], 1e-4, "replayed vs eager step: " + k)


def test_plan_steps_aside_when_it_must(det_mode, plan_switch):
    """Gradient accumulation (gradients that already exist), a second forward before the backward, an eval forward and another
    batch size in between: each takes the eager sequence and the values stay those of eager training."""
    def run(plan_on):
        plans.ENABLED = plan_on
        cfg, ef, tsf = harness.build_models(8, seed=2, device="cuda")
        opt = harness.make_optimizer(cfg, ef, tsf)
        torch.manual_seed(9)
        bt = [harness.device_batch(2, 8, 2, seed=i, device="cuda") for i in range(4)]
        small = harness.device_batch(1, 8, 1, seed=7, device="cuda")
        for i in range(3):
            harness.train_step(ef, tsf, opt, bt[i])                         # eager, recorded, replayed
        # accumulation: two backward passes into the same gradients
        opt.zero_grad(set_to_none=True)
        for i in (0, 1):
            y = harness.forward(ef, tsf, bt[i])
            mintime_amd.optim.bce_with_logits(y, bt[i]["labels"]).backward()
        opt.step()
        # two forwards in flight, backward through both
        opt.zero_grad(set_to_none=True)
        y0 = harness.forward(ef, tsf, bt[2])
        y1 = harness.forward(ef, tsf, bt[3])
        (mintime_amd.optim.bce_with_logits(y0, bt[2]["labels"]) + mintime_amd.optim.bce_with_logits(y1, bt[3]["labels"])).backward()
        opt.step()
        ef.eval(), tsf.eval()
        with torch.no_grad():
            ev = harness.forward(ef, tsf, bt[0]).clone()
        ef.train(), tsf.train()
        harness.train_step(ef, tsf, opt, small)
        last = harness.train_step(ef, tsf, opt, bt[1]).detach().clone()     # back on the plan
        torch.cuda.synchronize()
        return ev, last, _state(ef, tsf)
    before = dict(plans.STATS)
    ev_p, last_p, s_p = run(True)
    assert plans.STATS["eager_accumulate"] > before["eager_accumulate"] and plans.STATS["eager_in_flight"] > before["eager_in_flight"]
    ev_e, last_e, s_e = run(False)
    assert torch.equal(ev_p, ev_e) and torch.equal(last_p, last_e)
    diff = [k for k in s_e if not torch.equal(s_e[k], s_p[k])]
    assert not diff, f"{len(diff)} state tensors differ, e.g. {diff[:5]}"


def test_plan_is_dropped_when_parameters_move(plan_switch):
    plans.ENABLED = True
    cfg, ef, tsf = harness.build_models(8, seed=1, device="cuda")
    opt = harness.make_optimizer(cfg, ef, tsf)
    batch = harness.device_batch(2, 8, 2, seed=0, device="cuda")
    for _ in range(3):
        harness.train_step(ef, tsf, opt, batch)
    dropped = plans.STATS["dropped"]
    with torch.no_grad():
        w = tsf.to_out[1].weight if hasattr(tsf.to_out, "__getitem__") else getattr(tsf.to_out, "1").weight
        w.data = w.data.clone()                                          # new storage for one parameter
        blk = ef._blocks[3]._bn1
        blk.running_mean = blk.running_mean.clone()                      # ... and for one BatchNorm buffer
    opt2 = harness.make_optimizer(cfg, ef, tsf)
    loss = harness.train_step(ef, tsf, opt2, batch)
    torch.cuda.synchronize()
    assert plans.STATS["dropped"] == dropped + 2 and bool(torch.isfinite(loss))


def _train_xs(n_steps, plan_on, B=1, F=16):
    plans.ENABLED = plan_on
    cfg, xc, tsf = harness.build_models_xs(F, seed=3, device="cuda")
    opt = harness.make_optimizer(cfg, xc, tsf)
    losses = []
    for i in range(n_steps):
        batch = harness.device_batch(B, F, 3, seed=i, device="cuda", ragged=i % 2 == 1)
        losses.append(harness.train_step(xc, tsf, opt, batch).detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses), _state(xc, tsf)


def test_planned_xception_training_is_bit_identical_to_eager(det_mode, plan_switch):
    """BASELINE config 5's extractor under launch plans (round 6): 5 training steps of Xception + TimeSformer (1 clip x 16 slots, fresh
    inputs each step, train-mode BatchNorm) -- step 1 eager, step 2 recorded, steps 3-5 replayed -- give exactly the losses, parameters
    and running statistics of 5 eager steps in deterministic mode."""
    before = dict(plans.STATS)
    l_p, s_p = _train_xs(5, True)
    assert plans.STATS["recorded"] - before["recorded"] == 2              # Xception + TimeSformer
    assert plans.STATS["replayed"] - before["replayed"] == 3 * 4            # 3 replayed steps x 4 phases
    l_e, s_e = _train_xs(5, False)
    assert torch.equal(l_p, l_e), (l_p, l_e)
    diff = [k for k in s_e if not torch.equal(s_e[k], s_p[k])]
    assert not diff, f"{len(diff)} of {len(s_e)} state tensors differ between planned and eager training, e.g. {diff[:5]}"
