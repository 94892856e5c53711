"""Launch sequences (forward, backward) of EfficientNet-B0 on libmintime_hip.  Plumbing only: device buffers via
torch, raw pointers + current stream into the C ABI, one torch.autograd.Function for the whole extractor."""
import threading

import torch

from . import arch
from . import lib as L

SLOTS = 32   # replicated fp64 BatchNorm accumulators (spreads atomic traffic)
STEM_DIRECT = __import__("os").environ.get("MT_STEM_DIRECT", "1") != "0"    # 0 = the im2col-prologue GEMM
STREAM_ROWS = int(__import__("os").environ.get("MT_STREAM_ROWS", "100000"))   # 1x1 convs with at least this many rows use the streaming kernels
# Late stages (14 x 14 and 7 x 7 grids, the head) on plane operands (csrc/effnet_planes.hip, gemm_planes.hpp): expand convolutions
# whose input grid is at most EF_PLANES_EXPAND_HW wide, project convolutions whose output grid is at most EF_PLANES_PROJECT_HW wide
# (tools/lab/ef_planes_lab.py: at 14 x 14 the project convolution's producer pass costs what its GEMM saves).  MT_EF_PLANES=0: off.
# Early stages (blocks 1-3: 112^2 / 56^2 grids): expand convolution recomputed inside the depthwise kernels from the 16- / 24-channel
# block input (csrc/rc.hpp) -- the 6x wider expanded tensor is never written (forward) nor read (depthwise forward, data and weight
# gradient); in train mode its BatchNorm statistics come from a statistics-only pass over the block input.  Measured (round 6,
# profiles/r06_expand_recompute_ab.txt): HBM-side traffic of the extractor step 76.7 -> 70.1 GB, kernel time 22.3 -> 23.3 ms, step
# +0.4 ms -- the depthwise kernels are bound by instruction issue at 2-3 wavefronts per SIMD (VALU 50-60 % busy), not by the bytes
# the recompute removes, and it adds MFMA + staging work to them.  OFF by default; MT_EF_RC=1 switches it on (parity-tested).
EF_RC = __import__("os").environ.get("MT_EF_RC", "0") != "0"
EF_RC_MIN_ROWS = int(__import__("os").environ.get("MT_EF_RC_ROWS", "100000"))    # pixel rows from which the recompute form is used
EF_PLANES = __import__("os").environ.get("MT_EF_PLANES", "1") != "0"
EF_PLANES_EXPAND_HW = int(__import__("os").environ.get("MT_EF_PLANES_EXPAND_HW", "14"))
EF_PLANES_PROJECT_HW = int(__import__("os").environ.get("MT_EF_PLANES_PROJECT_HW", "7"))


def _new(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


def param_list(model):
    ps = [model._conv_stem.weight, model._bn0.weight, model._bn0.bias]
    for blk in model._blocks:
        if blk.spec.has_expand:
            ps += [blk._expand_conv.weight, blk._bn0.weight, blk._bn0.bias]
        ps += [blk._depthwise_conv.weight, blk._bn1.weight, blk._bn1.bias, blk._se_reduce.weight, blk._se_reduce.bias,
               blk._se_expand.weight, blk._se_expand.bias, blk._project_conv.weight, blk._bn2.weight, blk._bn2.bias]
    ps += [model._conv_head.weight, model._bn1.weight, model._bn1.bias]
    return ps


class _BNCtx:
    """Per-BatchNorm scratch: accumulators + folded affine + saved statistics."""

    def __init__(self, dev, C, training, stats_pool):
        self.C = C
        self.scale, self.shift = _new(dev, C), _new(dev, C)
        self.mean_invstd = _new(dev, 2, C)
        self.stats = stats_pool.take(C) if training else None
        self.count = 0.0


class _StatsPool:
    """One zero-filled fp64 buffer per forward, carved into [SLOTS][2][C] accumulators.  Deterministic mode (det=True): each
    accumulator is two 64-bit integer limbs (csrc/common.hpp stat_add: integer atomics add up the same whatever the order the
    blocks arrive in), [2 limbs][SLOTS][2][C], and the kernels are handed `-SLOTS`."""

    def __init__(self, dev, total_channels, det=False):
        self.limbs = 2 if det else 1
        self.buf = L.zeros(total_channels * 2 * SLOTS * self.limbs, torch.float64, dev)
        self.off = 0

    def take(self, C):
        n = 2 * SLOTS * C * self.limbs
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v


def _finalize(lib, st, bn_mod, ctx, count, training, gamma, beta, slots=SLOTS):
    ctx.count = float(count)
    L.check(lib.mt_bn_finalize(L.ptr(ctx.stats), slots, float(count), L.ptr(gamma), L.ptr(beta), L.ptr(bn_mod.running_mean),
                               L.ptr(bn_mod.running_var), L.ptr(ctx.scale), L.ptr(ctx.shift), L.ptr(ctx.mean_invstd), ctx.C,
                               bn_mod.eps, bn_mod.momentum, 1 if training else 0, st), "mt_bn_finalize")
    if training:
        _track(bn_mod.num_batches_tracked)


# num_batches_tracked counters touched by the running forward: bumped by ONE multi-tensor add at its end.  Per thread: forwards of
# different replicas may run concurrently from different Python threads (the reference's nn.DataParallel does that).
_TLS = threading.local()


def _track(counter):
    if not hasattr(_TLS, "tracked"):
        _TLS.tracked = []
    _TLS.tracked.append(counter)


def planes_scope(model, N):
    """(expand-on-planes?, project-on-planes?) per block + head, for a batch of N crops: the plane GEMM wants >= 512 pixel rows."""
    on = EF_PLANES and L.gemm_split_enabled()
    blocks = model._blocks
    exp = [on and b.spec.has_expand and b.spec.hin <= EF_PLANES_EXPAND_HW and N * b.spec.hin * b.spec.hin >= 512 and b.spec.cin % 4 == 0
           for b in blocks]
    proj = [on and b.spec.hout <= EF_PLANES_PROJECT_HW and N * b.spec.hout * b.spec.hout >= 512 for b in blocks]
    last = blocks[-1].spec
    head = on and last.hout <= EF_PLANES_EXPAND_HW and N * last.hout * last.hout >= 512
    return exp, proj, head


def weight_planes(model, params, exp, proj, head):
    """Plane tensors of the 1x1-conv weights that run on plane operands, re-split from the fp32 weights by ONE launch per forward
    (mt_split_planes_blk_multi; cf. tsf_planes.weight_planes).  Returns {("e", block) | ("p", block) | "head": planes}."""
    lib = L.get()
    blocks = model._blocks
    sel, pos = [], 3
    for bi, blk in enumerate(blocks):
        if blk.spec.has_expand:
            if exp[bi]:
                sel.append((("e", bi), params[pos], blk.spec.cexp, blk.spec.cin))
            pos += 3
        if proj[bi]:
            sel.append((("p", bi), params[pos + 7], blk.spec.cout, blk.spec.cexp))
        pos += 10
    if head:
        sel.append(("head", params[pos], arch.HEAD_COUT, arch.HEAD_CIN))
    if not sel:
        return {}
    # one cache entry per (weight storage, selection): another batch size selects other convolutions, and a recorded launch plan
    # (plans.py) keeps reading the table and the plane tensors of the entry it was recorded with
    ident = tuple((k, w.data_ptr()) for k, w, _, _ in sel)
    caches = model.__dict__.setdefault("_ef_wplanes_cache", {})
    if len(caches) > 4 and ident not in caches:
        caches.pop(next(iter(caches)))
    cache = caches.get(ident)
    if cache is None:
        holder, rows, first = {}, [], 0
        dev = sel[0][1].device
        for key, w, r, c in sel:
            if w.dtype != torch.float32 or not w.is_contiguous() or w.numel() != r * c:
                raise L.MintimeHipError("plane path needs contiguous fp32 1x1-conv weights")
            t = L.planes_empty(r, c, dev)
            holder[key] = t
            rows.append((w.data_ptr(), t.data_ptr(), r, c, first))
            first += t.shape[1] * t.shape[2]
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        cache = caches[ident] = dict(holder=holder, host=host, table=host.to(dev, non_blocking=True), blocks=first, count=len(rows))
    L.check(lib.mt_split_planes_blk_multi(L.ptr(cache["table"]), cache["count"], cache["blocks"], L.stream_ptr()),
            "mt_split_planes_blk_multi")       # (L.ptr: a plan being recorded pins the table)
    weight_planes_touch(model, cache, [w for _, w, _, _ in sel])
    return cache["holder"]


def weight_planes_touch(model, cache=None, ws=None):
    """A new serial when the weights changed since the planes were last written (cf. tsf_planes.weight_planes): graphs that saved
    an older serial must not run their backward on the re-split planes.  A replayed forward (the split launch is in the plan)
    calls this with no arguments for the bookkeeping alone.  Returns the serial the forward's planes carry, None without planes."""
    from .tsf_planes import WEIGHT_EPOCH
    st = model.__dict__.setdefault("_ef_wplanes_state", dict(serial=0, stamp=None, ws=None))
    if ws is not None:
        st["ws"], st["ident"] = ws, id(cache)
    if st["ws"] is None:
        return None
    stamp = (tuple(w._version for w in st["ws"]), tuple(w.data_ptr() for w in st["ws"]), st["ident"], WEIGHT_EPOCH[0])
    if stamp != st["stamp"]:
        st["serial"] += 1
        st["stamp"] = stamp
    return st["serial"]


def check_weight_serial(model, saved):
    ser = saved.get("w_serial")
    if ser is not None and model.__dict__.get("_ef_wplanes_state", {}).get("serial") != ser:
        raise RuntimeError("EfficientNet: the 1x1-conv weights were updated between this graph's forward and its backward (their "
                           "operand planes were rewritten by a later forward): run backward before the optimizer step")


def _bump_tracked(plan=None):
    tracked = getattr(_TLS, "tracked", None)
    if tracked:
        if plan is not None:
            plan.extra["tracked"] = list(tracked)       # a replayed forward bumps the same counters (plans.py)
        torch._foreach_add_(tracked, 1)
        tracked.clear()


def _dc_gates(model, N, dev):
    """Per-sample drop-connect gates (utils.py:129-154), train mode only: floor(keep + U[0,1)) / keep, keep = 1 - rate*idx/16
    (model.py:280-282).  ONE draw for all gated blocks (rows in block order); `model.drop_connect_uniform`, when set, supplies
    the uniforms instead of torch.rand (callable (n_rows, N, device) -> [n_rows, N]; parity tests feed the oracle's draws).
    Returns (gated block indices, gates [n_gated, N]) or ([], None)."""
    blocks = model._blocks
    if not (model.training and model.drop_connect_rate > 0):
        return [], None
    gated = [bi for bi, blk in enumerate(blocks) if blk.spec.skip and model.drop_connect_rate * float(bi) / len(blocks) > 0]
    if not gated:
        return [], None
    sampler = getattr(model, "drop_connect_uniform", None)
    u = sampler(len(gated), N, dev) if sampler is not None else torch.rand(len(gated), N, device=dev, dtype=torch.float32)
    cache = model.__dict__.setdefault("_dc_keep_cache", {})
    key = (str(dev), model.drop_connect_rate)
    if key not in cache:    # uploaded once, not per step
        cache[key] = torch.tensor([1.0 - model.drop_connect_rate * float(bi) / len(blocks) for bi in gated],
                                  dtype=torch.float32, device=dev).unsqueeze(1)
    keep = cache[key]
    return gated, torch.floor(keep + u.to(device=dev, dtype=torch.float32)) / keep


def effnet_forward(model, x_nhwc, params, training, save, want_blocks=False, plan=None):
    """x_nhwc [N,H,W,3] contiguous fp32.  Returns (feat [N*Ho*Wo, 1280], saved or None, block outputs or None).
    plan: the plans.NetPlan being recorded (it keeps the gate buffer and the tracked counters for its replays)."""
    lib = L.get()
    st = L.stream_ptr()
    dev = x_nhwc.device
    N, H, W, _ = x_nhwc.shape
    blocks = model._blocks
    total_c = arch.STEM_COUT + arch.HEAD_COUT + sum((b.spec.cexp if b.spec.has_expand else 0) + b.spec.cexp + b.spec.cout
                                                    for b in blocks)
    det = training and L.deterministic()          # MT_DETERMINISTIC: the producers' statistics as integer limbs (order-independent)
    pool = _StatsPool(dev, total_c, det) if training else None
    it = iter(params)
    slots = -SLOTS if det else SLOTS
    epi = L.EPI_STATS if training else L.EPI_STORE
    sptr = (lambda b_: L.ptr(b_.stats)) if training else (lambda b_: None)
    saved = {"blocks": []} if save else None
    pl_exp, pl_proj, pl_head = planes_scope(model, N)
    wpl = weight_planes(model, params, pl_exp, pl_proj, pl_head)
    y_p = None                   # planes of the current block input y (written by the producing block's bn_act when the next conv wants them)

    # ---- stem
    w_stem, g0, b0 = next(it), next(it), next(it)
    Hc, Wc = (H + 1) // 2, (W + 1) // 2
    bn = _BNCtx(dev, arch.STEM_COUT, training, pool)
    z = _new(dev, N * Hc * Wc, arch.STEM_COUT)
    # stem: streaming MFMA kernel (stem_fwd.hip; uint8 crops converted on the fly, BatchNorm statistics in the same pass).  The
    # im2col-prologue GEMM it replaced (K = 27 taps padded to 28) stays the path for crops wider than the kernel's LDS row tile.
    if W <= 512 and H == W and STEM_DIRECT:
        L.check(lib.mt_stem_conv_fwd(L.ptr(x_nhwc), 1 if x_nhwc.dtype == torch.uint8 else 0, L.ptr(w_stem), L.ptr(z),
                                     sptr(bn), slots, N, H, W, st), "mt_stem_conv_fwd")
    else:
        wp = _new(dev, arch.STEM_COUT, 28)
        L.check(lib.mt_conv_weight_pack(L.ptr(w_stem), L.ptr(wp), arch.STEM_COUT, 3, 3, 28, 0, st), "mt_conv_weight_pack")
        L.gemm(L.OP_NT, x_nhwc, wp, z, N * Hc * Wc, arch.STEM_COUT, 28, 28, 28, arch.STEM_COUT, prologue=L.PRO_IM2COL, epilogue=epi,
               stats=bn.stats, stats_slots=slots, conv=(H, W, 3, Hc, Wc, 3, 2, 0, 0, 1 if x_nhwc.dtype == torch.uint8 else 0))
    _finalize(lib, st, model._bn0, bn, N * Hc * Wc, training, g0, b0, slots)
    if save:
        saved["stem"] = dict(x=x_nhwc, z=z, bn=bn)
    cur_z, cur_bn = z, bn        # "virtual" activated tensor: swish(bn(z))
    y = None                     # materialised block output (narrow tensor)
    ys = [] if want_blocks else None

    gated, gates = _dc_gates(model, N, dev) if training else ([], None)
    dc_gates = {bi: gates[j] for j, bi in enumerate(gated)}
    if plan is not None:
        plan.extra["gates"] = gates
    for bi, blk in enumerate(blocks):
        s = blk.spec
        M_in = N * s.hin * s.hin
        M_out = N * s.hout * s.hout
        rec = {"spec": s}
        if s.has_expand:
            w_e, g, b = next(it), next(it), next(it)
            bn_e = _BNCtx(dev, s.cexp, training, pool)
            rc_on = (EF_RC and not L.deterministic() and M_in >= EF_RC_MIN_ROWS and w_e.data_ptr() % 16 == 0
                     and lib.mt_dwconv_rc_supported(s.cin, s.cexp, s.k, s.s, s.hin) and lib.mt_conv1x1_rows_instance(s.cin, s.cexp))
            z_e = None if rc_on else _new(dev, M_in, s.cexp)
            if rc_on:
                if training:      # BatchNorm statistics of the expanded tensor without the tensor: the product is formed and summed, not stored
                    L.check(lib.mt_conv1x1_rows(L.ptr(y), None, L.ptr(w_e), s.cin, 0, None, None, None, 1, 0, None, None,
                                                sptr(bn_e), slots, M_in, s.cin, s.cexp, st), "mt_conv1x1_rows")
            elif pl_exp[bi]:
                # late stages: y arrives as planes (written by the block above), the weight planes were split at the top
                L.gemm_planes(L.OP_NT, y_p, wpl[("e", bi)], M_in, s.cexp, s.cin, Cout=z_e, ldc=s.cexp, epilogue=epi, stats=bn_e.stats,
                              stats_slots=slots)
                rec.update(y_p=y_p, we_p=wpl[("e", bi)])
            elif M_in >= STREAM_ROWS and lib.mt_conv1x1_rows_supported(s.cin, s.cexp, 0):   # few channels, very many rows: streaming kernel
                L.check(lib.mt_conv1x1_rows(L.ptr(y), None, L.ptr(w_e), s.cin, 0, None, None, None, 1, 0, None, L.ptr(z_e),
                                            sptr(bn_e), slots, M_in, s.cin, s.cexp, st), "mt_conv1x1_rows")
            else:
                L.gemm(L.OP_NT, y, w_e, z_e, M_in, s.cexp, s.cin, s.cin, s.cin, s.cexp, epilogue=epi, stats=bn_e.stats, stats_slots=slots)
            _finalize(lib, st, blk._bn0, bn_e, M_in, training, g, b, slots)
            rec.update(z_e=z_e, bn_e=bn_e, rc=rc_on, w_e=w_e)
            dw_in, dw_bn = z_e, bn_e
        else:
            rc_on = False
            dw_in, dw_bn = cur_z, cur_bn
        w_d, g, b = next(it), next(it), next(it)
        bn_d = _BNCtx(dev, s.cexp, training, pool)
        z_d = _new(dev, M_out, s.cexp)
        if rc_on:
            L.check(lib.mt_dwconv_fwd_rc(L.ptr(y), L.ptr(w_e), s.cin, L.ptr(dw_bn.scale), L.ptr(dw_bn.shift), L.ptr(w_d), L.ptr(z_d),
                                         sptr(bn_d), slots, N, s.hin, s.hin, s.cexp, s.k, s.s, st), "mt_dwconv_fwd_rc")
        else:
            L.check(lib.mt_dwconv_fwd(L.ptr(dw_in), L.ptr(dw_bn.scale), L.ptr(dw_bn.shift), L.ptr(w_d), L.ptr(z_d), sptr(bn_d),
                                      slots, N, s.hin, s.hin, s.cexp, s.k, s.s, 1, st), "mt_dwconv_fwd")
        _finalize(lib, st, blk._bn1, bn_d, M_out, training, g, b, slots)
        w_r, b_r, w_x, b_x = next(it), next(it), next(it), next(it)
        pooled, gate = _new(dev, N, s.cexp), _new(dev, N, s.cexp)
        hidden = _new(dev, N, s.cse)          # squeeze pre-activations: the gate kernel reads them (and backward keeps them)
        hw = s.hout * s.hout
        parts = lib.mt_se_pool_parts(N, hw, s.cexp)
        partial = _new(dev, N, parts, s.cexp)
        L.check(lib.mt_se_pool_fwd(L.ptr(z_d), L.ptr(bn_d.scale), L.ptr(bn_d.shift), L.ptr(partial), N, hw, s.cexp, parts, st),
                "mt_se_pool_fwd")
        L.check(lib.mt_se_gate_fwd(L.ptr(partial), parts, L.ptr(w_r), L.ptr(b_r), L.ptr(w_x), L.ptr(b_x), L.ptr(pooled), L.ptr(gate),
                                   L.ptr(hidden), N, s.cexp, s.cse, st), "mt_se_gate_fwd")
        w_p, g, b = next(it), next(it), next(it)
        bn_p = _BNCtx(dev, s.cout, training, pool)
        z_p = _new(dev, M_out, s.cout)
        if pl_proj[bi]:
            # 7 x 7 stage: the operand swish(bn1(z_d)) * gate written once as planes (kept for the weight gradient), the GEMM only DMAs
            a_p = L.planes_empty(M_out, s.cexp, dev)
            L.check(lib.mt_bn_swish_gate_planes(L.ptr(z_d), L.ptr(bn_d.scale), L.ptr(bn_d.shift), L.ptr(gate), hw, L.ptr(a_p), M_out,
                                                s.cexp, st), "mt_bn_swish_gate_planes")
            L.gemm_planes(L.OP_NT, a_p, wpl[("p", bi)], M_out, s.cout, s.cexp, Cout=z_p, ldc=s.cout, epilogue=epi, stats=bn_p.stats,
                          stats_slots=slots)
            rec.update(a_p=a_p, wp_p=wpl[("p", bi)])
        elif M_out >= STREAM_ROWS and lib.mt_conv1x1_rows_supported(s.cexp, s.cout, 1):
            L.check(lib.mt_conv1x1_rows(L.ptr(z_d), None, L.ptr(w_p), s.cexp, 0, L.ptr(bn_d.scale), L.ptr(bn_d.shift), L.ptr(gate), hw, 1,
                                        None, L.ptr(z_p), sptr(bn_p), slots, M_out, s.cexp, s.cout, st), "mt_conv1x1_rows")
        else:
            L.gemm(L.OP_NT, z_d, w_p, z_p, M_out, s.cout, s.cexp, s.cexp, s.cexp, s.cout, prologue=L.PRO_BN_SWISH_GATE, epilogue=epi,
                   scale=bn_d.scale, shift=bn_d.shift, gate=gate, hw=hw, stats=bn_p.stats, stats_slots=slots)
        _finalize(lib, st, blk._bn2, bn_p, M_out, training, g, b, slots)
        # block output: bn2(z_p) [* drop-connect gate] [+ block input]
        dc = dc_gates.get(bi)
        y_new = _new(dev, M_out, s.cout)
        next_planes = pl_exp[bi + 1] if bi + 1 < len(blocks) else pl_head
        if next_planes:
            y_p = L.planes_empty(M_out, s.cout, dev)
            L.check(lib.mt_bn_act_fwd_planes(L.ptr(z_p), L.ptr(bn_p.scale), L.ptr(bn_p.shift), L.ptr(y if s.skip else None), L.ptr(y_new),
                                             M_out, s.cout, 0, L.ptr(dc), hw, L.ptr(y_p), st), "mt_bn_act_fwd_planes")
        else:
            y_p = None
            L.check(lib.mt_bn_act_fwd(L.ptr(z_p), L.ptr(bn_p.scale), L.ptr(bn_p.shift), L.ptr(y if s.skip else None), L.ptr(y_new),
                                      M_out, s.cout, 0, L.ptr(dc), hw, st), "mt_bn_act_fwd")
        if save:
            rec.update(y_in=y, dw_in=dw_in, dw_bn=dw_bn, z_d=z_d, bn_d=bn_d, pooled=pooled, hidden=hidden, gate=gate, z_p=z_p,
                       bn_p=bn_p, dc=dc)
            saved["blocks"].append(rec)
        y = y_new
        if want_blocks:
            ys.append(y.view(N, s.hout, s.hout, s.cout).permute(0, 3, 1, 2))

    # ---- head
    w_h, g, b = next(it), next(it), next(it)
    s = blocks[-1].spec
    M = N * s.hout * s.hout
    bn_h = _BNCtx(dev, arch.HEAD_COUT, training, pool)
    z_h = _new(dev, M, arch.HEAD_COUT)
    if pl_head:
        L.gemm_planes(L.OP_NT, y_p, wpl["head"], M, arch.HEAD_COUT, arch.HEAD_CIN, Cout=z_h, ldc=arch.HEAD_COUT, epilogue=epi,
                      stats=bn_h.stats, stats_slots=slots)
    else:
        L.gemm(L.OP_NT, y, w_h, z_h, M, arch.HEAD_COUT, arch.HEAD_CIN, arch.HEAD_CIN, arch.HEAD_CIN, arch.HEAD_COUT, epilogue=epi,
               stats=bn_h.stats, stats_slots=slots)
    _finalize(lib, st, model._bn1, bn_h, M, training, g, b, slots)
    feat = _new(dev, M, arch.HEAD_COUT)
    L.check(lib.mt_bn_act_fwd(L.ptr(z_h), L.ptr(bn_h.scale), L.ptr(bn_h.shift), None, L.ptr(feat), M, arch.HEAD_COUT, 1, None, 1,
                              st), "mt_bn_act_fwd")
    if save:
        saved["head"] = dict(y_in=y, z=z_h, bn=bn_h, y_p=y_p if pl_head else None, w_p=wpl.get("head"))
        saved["w_serial"] = model.__dict__.get("_ef_wplanes_state", {}).get("serial") if wpl else None
    _bump_tracked(plan)
    return feat, saved, ys


def _state_tensors(model, params):
    """Everything a recorded phase holds an address of besides its own buffers: parameters and BatchNorm buffers."""
    return list(params) + [b for b in model.buffers()]


class _EffNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, want_blocks, x_nhwc, *params):
        from . import plans
        want_blocks, grad_on = want_blocks
        save = grad_on and any(ctx.needs_input_grad)      # see tsf_engine._TSFFunction.forward
        N, H, W, _ = x_nhwc.shape
        ctx.shape = (N, H, W)
        ctx.model, ctx.params, ctx.training = model, params, model.training
        ctx.plan = ctx.token = None
        np_, mode = None, "eager"
        if save and not want_blocks:
            stream = torch.cuda.current_stream(x_nhwc.device).cuda_stream
            key = ("ef", tuple(x_nhwc.shape), x_nhwc.dtype, model.training, L.deterministic(), L.gemm_split_enabled(),
                   float(model.drop_connect_rate), tuple(ctx.needs_input_grad[3:]), stream,
                   float(model._bn0.momentum), float(model._bn0.eps))       # (scalars a recorded call holds by value)
            np_, mode = plans.lookup(model, key)
            if mode == "replay" and np_.state_ptrs != plans.state_ptrs(_state_tensors(model, params)):
                plans.drop(model, np_)                   # parameters / buffers moved (load_state_dict, .to()): record afresh later
                np_, mode = None, "eager"
        if mode == "eager":
            feat, saved, ys = effnet_forward(model, x_nhwc, params, model.training, save, want_blocks)
            ctx.saved = saved
            outs = [feat]
            if want_blocks:
                for t in ys:
                    ctx.mark_non_differentiable(t)
                outs += ys
            return tuple(outs)
        if mode == "record":
            np_.stream = stream
            x_s = plans.static_input(np_, "x", x_nhwc)
            pl = L.Plan()
            try:
                with pl:
                    feat, saved, _ = effnet_forward(model, x_s, params, model.training, True, False, plan=np_)
            except Exception:
                np_.broken = True
                raise
            np_.fwd = pl
            np_.extra.update(saved=saved, feat=feat)
            np_.state_ptrs = plans.state_ptrs(_state_tensors(model, params))
            plans.own(np_, feat)
            plans.STATS["recorded"] += 1
        else:
            plans.refresh_input(np_, "x", x_nhwc)
            if np_.extra.get("gates") is not None:
                np_.extra["gates"].copy_(_dc_gates(model, N, x_nhwc.device)[1])
            plans.run(np_.fwd)
            if np_.extra["saved"].get("w_serial") is not None:
                np_.extra["saved"]["w_serial"] = weight_planes_touch(model)       # the split launch is in the plan
            if np_.extra.get("tracked"):
                torch._foreach_add_(np_.extra["tracked"], 1)
        ctx.plan, ctx.token = np_, np_.begin()
        ctx.saved = np_.extra["saved"]
        return (np_.extra["feat"].detach(),)

    @staticmethod
    def backward(ctx, dfeat, *unused):
        if ctx.saved is None:
            raise RuntimeError("EfficientNet: backward ran a second time through the same forward; the activation buffers are "
                               "released after the first pass (retain_graph is not supported by the HIP engine)")
        from . import plans
        from .effnet_backward import effnet_backward, LAST_RUN
        np_ = ctx.plan
        check_weight_serial(ctx.model, ctx.saved)
        dfeat = dfeat.contiguous()
        need_dx, need_dp = ctx.needs_input_grad[2], ctx.needs_input_grad[3:]
        if np_ is None:
            dx, dparams = effnet_backward(ctx.model, ctx.params, ctx.saved, ctx.shape, ctx.training, dfeat, need_dx, need_dp)
        elif (plans.grads_exist(ctx.params) or torch.cuda.current_stream(dfeat.device).cuda_stream != np_.stream
              or torch.cuda.is_current_stream_capturing()):
            # gradients that already exist are ADDED to by autograd: they alias the plan's gradient buffer, so this pass needs
            # fresh ones -- the eager launch sequence over the plan's saved activations (which stay for the next replay)
            plans.STATS["eager_accumulate"] += 1
            dx, dparams = effnet_backward(ctx.model, ctx.params, ctx.saved, ctx.shape, ctx.training, dfeat, need_dx, need_dp,
                                          keep_saved=True)
        elif np_.bwd is None:
            d_s = plans.static_input(np_, "dfeat", dfeat)
            pl = L.Plan()
            try:
                with pl:
                    dx, dparams = effnet_backward(ctx.model, ctx.params, ctx.saved, ctx.shape, ctx.training, d_s, need_dx, need_dp,
                                                  keep_saved=True, plan=np_)
            except Exception:
                np_.broken = True
                raise
            np_.bwd = pl
            np_.extra.update(grads=list(dparams), last_run=dict(LAST_RUN))
            dparams = plans.fresh_aliases(dparams)
        else:
            plans.refresh_input(np_, "dfeat", dfeat)
            plans.run(np_.bwd)
            LAST_RUN.update(np_.extra["last_run"])
            L.grads_ready(ctx.model, ctx.params, np_.extra["flat_grads"])
            dx, dparams = None, plans.fresh_aliases(np_.extra["grads"])
        ctx.saved = None
        if np_ is not None:
            np_.release(ctx.token)
            ctx.token = None
        return (None, None, dx) + tuple(dparams)


def effnet_apply(model, inputs, want_blocks=False):
    if not inputs.is_cuda:
        raise L.MintimeHipError("EfficientNet (MI355X build) needs device tensors; there is no CPU path")
    if inputs.dim() != 4 or inputs.shape[1] != 3:
        raise ValueError(f"expected [N,3,H,W] input, got {tuple(inputs.shape)}")
    if inputs.shape[2] != model.image_size or inputs.shape[3] != model.image_size:
        raise ValueError(f"static TF-SAME padding was built for {model.image_size}x{model.image_size} inputs "
                         f"(utils.py:248-276); got {tuple(inputs.shape[2:])}")
    # logical NCHW, physical NHWC: a view when the caller did `rearrange(videos, 'b f h w c -> (b f) c h w')` (train.py:341).
    # uint8 crops (what cv2 produces before deepfakes_dataset.py:339 casts them to float) are ingested as they are: the stem's
    # gather converts on the fly, which cuts the H2D copy and the stem's input traffic 4x (next-row f2).
    x_nhwc = (inputs if inputs.dtype == torch.uint8 else inputs.float()).permute(0, 2, 3, 1)
    if not x_nhwc.is_contiguous():
        x_nhwc = x_nhwc.contiguous()
    outs = _EffNetFunction.apply(model, (want_blocks, torch.is_grad_enabled()), x_nhwc, *param_list(model))
    feat = outs[0]
    n = inputs.shape[0]
    ho = model._blocks[-1].spec.hout
    feat_nchw = feat.view(n, ho, ho, arch.HEAD_COUT).permute(0, 3, 1, 2)     # NHWC-strided [N,1280,7,7]
    return feat_nchw, list(outs[1:])
