"""Launch sequences (forward and backward) of Xception on libmintime_hip (reference models/xception.py:161-217).

Same rules as the EfficientNet engine: NHWC rows, raw conv outputs in HBM with BatchNorm(+ReLU) applied by the consumer on
load, BatchNorm statistics accumulated by the producer, BatchNorm backward folded into GEMM operand loads.
A `_Src` describes how a consumer obtains a tensor's value: stored data + pending per-channel affine + pending activation.
"""
import torch

from . import arch
from . import lib as L
from .effnet_engine import SLOTS, _StatsPool, _BNCtx, _track, _bump_tracked

RELU, NONE = 2, 0


# pointwise convolutions with at least this many input channels run on plane operands (MT_XC_PLANES=0: the in-kernel-split loop)
XC_PLANES = __import__("os").environ.get("MT_XC_PLANES", "1") != "0"
PLANES_MIN_C = int(__import__("os").environ.get("MT_XC_PLANES_MIN_C", "256"))   # (128: +0.4 ms in config 5 once the 4 GB operand limit is respected, lib.planes_fit)
DW_PLANES = __import__("os").environ.get("MT_XC_DW_PLANES", "1") != "0"   # 0: depthwise output as fp32 + mt_split_planes_blk (round 4)
CONV2_WGRAD_SIDE = float(__import__("os").environ.get("MT_XC_CONV2_WGRAD_SIDE", "0.7"))   # share of conv2's weight gradient on the side stream
XC_STEM = __import__("os").environ.get("MT_XC_STEM", "1") != "0"             # 0: conv1 as an im2col GEMM (round 4)
SKIP_HALF = __import__("os").environ.get("MT_XC_SKIP_HALF", "1") != "0"    # 0: the skip path's data gradient scattered into a zeroed full-size tensor
POOL_ARG = __import__("os").environ.get("MT_XC_POOL_ARG", "1") != "0"      # 0: the adjoint of the max-pool as an arg-max scatter (round 4)


def _new(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


class _Src:
    def __init__(self, t, C, H, scale=None, shift=None, act=NONE, bn=None):
        self.t, self.C, self.H, self.scale, self.shift, self.act, self.bn = t, C, H, scale, shift, act, bn


class _Consts:
    """ones / zeros vectors for tensors that carry no pending affine (the depthwise kernels always take scale/shift)."""

    def __init__(self, dev):
        self.dev, self.cache = dev, {}

    def ident(self, C):
        if C not in self.cache:
            self.cache[C] = (torch.ones(C, device=self.dev), torch.zeros(C, device=self.dev))
        return self.cache[C]

    def kabc_ident(self, C):
        key = ("k", C)
        if key not in self.cache:
            k = torch.zeros(3, C, device=self.dev)
            k[0] = 1.0
            self.cache[key] = k
        return self.cache[key]


def param_list(model):
    ps = [model.conv1.weight, model.bn1.weight, model.bn1.bias, model.conv2.weight, model.bn2.weight, model.bn2.bias]
    for blk in model.blocks():
        if blk.skip is not None:
            ps += [blk.skip.weight, blk.skipbn.weight, blk.skipbn.bias]
        for (i_sep, i_bn, ci, co) in blk.unit_index:
            sep, bn = getattr(blk.rep, str(i_sep)), getattr(blk.rep, str(i_bn))
            ps += [sep.conv1.weight, sep.pointwise.weight, bn.weight, bn.bias]
    for sep, bn in ((model.conv3, model.bn3), (model.conv4, model.bn4)):
        ps += [sep.conv1.weight, sep.pointwise.weight, bn.weight, bn.bias]
    return ps


def _total_channels(model):
    c = 32 + 64 + 1536 + 2048
    for blk in model.blocks():
        if blk.skip is not None:
            c += blk.cfg[2]
        c += sum(co for (_, _, _, co) in blk.unit_index)
    return c


def _finalize(lib, bn_mod, ctx, count, training, gamma, beta, slots=SLOTS):
    """slots < 0 (deterministic mode): the producers' statistics are integer limbs (csrc/common.hpp stat_add, effnet_engine._StatsPool)."""
    ctx.count = float(count)
    L.check(lib.mt_bn_finalize(L.ptr(ctx.stats), slots, float(count), L.ptr(gamma), L.ptr(beta), L.ptr(bn_mod.running_mean),
                               L.ptr(bn_mod.running_var), L.ptr(ctx.scale), L.ptr(ctx.shift), L.ptr(ctx.mean_invstd), ctx.C,
                               bn_mod.eps, bn_mod.momentum, 1 if training else 0, L.stream_ptr()), "mt_bn_finalize")
    if training:
        _track(bn_mod.num_batches_tracked)


class _WeightPlanes:
    """Plane tensors of the pointwise-conv weights that run on plane operands.  The first forward splits them one by one and registers
    them; every later forward re-splits all of them with ONE mt_split_planes_blk_multi launch (33 launches per step before)."""

    def __init__(self):
        self.items = {}          # weight data_ptr -> (weight, planes, co, ci)
        self.table = None
        self.fresh = False
        self.serial, self.stamp = 0, None     # see tsf_planes.weight_planes: graphs that saved an older serial must not run their
                                              # backward on planes a later forward re-split from updated weights

    def begin(self, lib):
        self.fresh = False
        if self.table is None:
            return
        if any(w.data_ptr() != k for k, (w, _, _, _) in self.items.items()):       # a weight moved: start over
            self.items, self.table = {}, None
            return
        L.check(lib.mt_split_planes_blk_multi(L.ptr(self.table), len(self.items), self.blocks, L.stream_ptr()),
                "mt_split_planes_blk_multi")
        self.fresh = True

    def get(self, w_pw, co, ci):
        ent = self.items.get(w_pw.data_ptr())
        if ent is not None and self.fresh and ent[0] is w_pw:
            return ent[1]
        planes = L.split_planes_blk(w_pw.view(co, ci), co, ci)
        self.items[w_pw.data_ptr()] = (w_pw, planes, co, ci)
        self.table = None
        return planes

    def touch(self):
        """A new serial when the weights changed since the planes were last written (version counters; the fused optimizers that
        write through raw pointers bump tsf_planes.WEIGHT_EPOCH)."""
        from .tsf_planes import WEIGHT_EPOCH
        stamp = (tuple(w._version for w, _, _, _ in self.items.values()), tuple(self.items), WEIGHT_EPOCH[0])
        if stamp != self.stamp:
            self.serial += 1
            self.stamp = stamp
        return self.serial

    def end(self, dev):
        if self.table is not None or not self.items:
            return
        rows, first = [], 0
        for k, (w, pl, co, ci) in self.items.items():
            rows.append((k, pl.data_ptr(), co, ci, first))
            first += pl.shape[1] * pl.shape[2]
        self.host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        self.table = self.host.to(dev, non_blocking=True)
        self.blocks = first


def xception_forward(model, x, params, training, save, plan=None):
    """plan: the plans.NetPlan being recorded (it keeps the tracked counters for its replays)."""
    lib = L.get()
    dev = x.device
    N, H, W, _ = x.shape
    det = training and L.deterministic()
    pool = _StatsPool(dev, _total_channels(model), det) if training else None
    consts = _Consts(dev)
    slots = -SLOTS if det else SLOTS          # deterministic mode: BatchNorm sums as integer limbs -- order-independent, no second pass
    epi = L.EPI_STATS if training else L.EPI_STORE
    planes_on = XC_PLANES and L.gemm_split_enabled()
    wplanes = model.__dict__.setdefault("_xc_wplanes", _WeightPlanes())
    if planes_on:
        wplanes.begin(lib)
    it = iter(params)
    saved = {"x": x, "blocks": [], "consts": consts} if save else None

    def pack(w, Co, Ci, k, ld, transpose=0):
        out = _new(dev, Ci if transpose else Co, ld)
        L.check(lib.mt_conv_weight_pack(L.ptr(w), L.ptr(out), Co, Ci, k, ld, transpose, L.stream_ptr()), "mt_conv_weight_pack")
        return out

    def conv_im2col(src, wp, Cout, K, geom, bn_mod, gamma, beta, u8=False):
        """z = im2col(act(affine(src))) . wp^T  (+ BatchNorm statistics)."""
        Hh, Ww, Cc, Ho, Wo, k, s_, p_ = geom
        M = N * Ho * Wo
        ctx = _BNCtx(dev, Cout, training, pool)
        z = _new(dev, M, Cout)
        L.gemm(L.OP_NT, src.t, wp, z, M, Cout, K, K, K, Cout, prologue=L.PRO_IM2COL, epilogue=epi, scale=src.scale, shift=src.shift,
               stats=ctx.stats, stats_slots=slots, conv=(Hh, Ww, Cc, Ho, Wo, k, s_, p_, src.act, 1 if u8 else 0))
        _finalize(lib, bn_mod, ctx, M, training, gamma, beta, slots)
        return z, ctx

    def sep_unit(src, act, w_dw, w_pw, co, bn_mod, gamma, beta):
        """[act] -> depthwise 3x3 pad 1 -> pointwise 1x1 -> BatchNorm (xception.py:17-27, 44-58)."""
        Hh, ci = src.H, src.C
        M = N * Hh * Hh
        sc, sh = (src.scale, src.shift) if src.scale is not None else consts.ident(ci)
        eff = RELU if (act == RELU or src.act == RELU) else NONE
        ctx = _BNCtx(dev, co, training, pool)
        z = _new(dev, M, co)
        d = d_p = w_p = None
        if planes_on and ci >= PLANES_MIN_C and M >= 512 and L.planes_fit(M, max(ci, co)):
            # wide pointwise convolutions on plane operands (csrc/gemm_planes.hpp): forward, data gradient and weight gradient read
            # the same plane tensors by LDS-DMA (728 -> 728 over 100 352 rows: 0.81 -> 0.58 ms).  The weight is split once per
            # forward; the depthwise kernel writes its output as planes and nowhere else (DW_PLANES; round 4: fp32 + a split pass)
            d_p = L.planes_empty(M, ci, dev)
            if DW_PLANES:
                L.check(lib.mt_dwconv_fwd_planes(L.ptr(src.t), L.ptr(sc), L.ptr(sh), L.ptr(w_dw), L.ptr(d_p), N, Hh, Hh, ci, 3, 1, eff,
                                                 L.stream_ptr()), "mt_dwconv_fwd_planes")
            else:
                d = _new(dev, M, ci)
                L.check(lib.mt_dwconv_fwd(L.ptr(src.t), L.ptr(sc), L.ptr(sh), L.ptr(w_dw), L.ptr(d), None, SLOTS, N, Hh, Hh, ci, 3, 1, eff,
                                          L.stream_ptr()), "mt_dwconv_fwd")
                L.split_planes_blk(d, M, ci, out=d_p)
                d = None                                   # backward reads d through its planes only
            w_p = wplanes.get(w_pw, co, ci)
            L.gemm_planes(L.OP_NT, d_p, w_p, M, co, ci, Cout=z, ldc=co, epilogue=epi, stats=ctx.stats, stats_slots=slots)
        else:
            d = _new(dev, M, ci)
            L.check(lib.mt_dwconv_fwd(L.ptr(src.t), L.ptr(sc), L.ptr(sh), L.ptr(w_dw), L.ptr(d), None, SLOTS, N, Hh, Hh, ci, 3, 1, eff,
                                      L.stream_ptr()), "mt_dwconv_fwd")
            L.gemm(L.OP_NT, d, w_pw, z, M, co, ci, ci, ci, co, epilogue=epi, stats=ctx.stats, stats_slots=slots)
        _finalize(lib, bn_mod, ctx, M, training, gamma, beta, slots)
        rec = dict(src=src, eff=eff, sc=sc, sh=sh, d=d, d_p=d_p, w_p=w_p, z=z, bn=ctx, ci=ci, co=co, H=Hh) if save else None
        return _Src(z, co, Hh, ctx.scale, ctx.shift, NONE, ctx), rec

    # ---- conv1 (3x3 s2 p0, 3 -> 32) and conv2 (3x3 s1 p0, 32 -> 64) as im2col GEMMs
    w1, g1, b1, w2, g2, b2 = next(it), next(it), next(it), next(it), next(it), next(it)
    H1 = (H - 3) // 2 + 1
    if XC_STEM and W <= 512:
        # conv1 = the EfficientNet stem's streaming MFMA kernel without padding (K = 27 is far too short for the im2col GEMM: 4.3 ms
        # against 0.5 at 512 crops); reads uint8 crops as well
        bn1 = _BNCtx(dev, 32, training, pool)
        z1 = _new(dev, N * H1 * H1, 32)
        L.check(lib.mt_stem_conv_fwd_valid(L.ptr(x), 1 if x.dtype == torch.uint8 else 0, L.ptr(w1), L.ptr(z1), L.ptr(bn1.stats) if training
                                           else None, slots, N, H, W, L.stream_ptr()), "mt_stem_conv_fwd_valid")
        _finalize(lib, model.bn1, bn1, N * H1 * H1, training, g1, b1, slots)
    else:
        wp1 = pack(w1, 32, 3, 3, 28)
        z1, bn1 = conv_im2col(_Src(x, 3, H), wp1, 32, 28, (H, W, 3, H1, H1, 3, 2, 0), model.bn1, g1, b1, u8=x.dtype == torch.uint8)
    s1 = _Src(z1, 32, H1, bn1.scale, bn1.shift, RELU, bn1)
    H2 = H1 - 2
    wp2 = pack(w2, 64, 32, 3, 288)
    z2, bn2 = conv_im2col(s1, wp2, 64, 288, (H1, H1, 32, H2, H2, 3, 1, 0), model.bn2, g2, b2)
    cur = _Src(z2, 64, H2, bn2.scale, bn2.shift, RELU, bn2)
    if save:
        saved.update(z1=z1, bn1=bn1, s1=s1, z2=z2, bn2=bn2, H1=H1, H2=H2)

    # ---- blocks
    for blk in model.blocks():
        name, cin, cout, reps, stride, srelu, grow = blk.cfg
        inp = cur
        brec = {"inp": inp, "units": [], "stride": stride}
        if blk.skip is not None:
            w_s, g_s, b_s = next(it), next(it), next(it)
        src = cur
        for u, (i_sep, i_bn, ci, co) in enumerate(blk.unit_index):
            w_dw, w_pw, g, b = next(it), next(it), next(it), next(it)
            act = RELU if (u > 0 or srelu) else NONE
            src, rec = sep_unit(src, act, w_dw, w_pw, co, getattr(blk.rep, str(i_bn)), g, b)
            brec["units"].append(rec)
        Hin = inp.H
        if stride != 1:
            Ho = (Hin - 1) // 2 + 1
            z_s, bn_s = conv_im2col(inp, w_s.view(cout, cin), cout, cin, (Hin, Hin, cin, Ho, Ho, 1, 2, 0), blk.skipbn, g_s, b_s)
            y = _new(dev, N * Ho * Ho, cout)
            if save and POOL_ARG:
                # the adjoint's routing is recorded here (1 byte + the raw arg-max value per pooled element): backward never builds the
                # full-resolution routed gradient by atomics, and takes the pooled unit's BatchNorm sums from the pooled tensors
                arg = torch.empty(N * Ho * Ho, cout, dtype=torch.uint8, device=dev)
                zmax = _new(dev, N * Ho * Ho, cout)
                L.check(lib.mt_maxpool_add_fwd_arg(L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift), L.ptr(z_s), L.ptr(bn_s.scale),
                                                   L.ptr(bn_s.shift), L.ptr(y), L.ptr(arg), L.ptr(zmax), N, Hin, Hin, cout, L.stream_ptr()),
                        "mt_maxpool_add_fwd_arg")
                brec.update(arg=arg, zmax=zmax)
            else:
                L.check(lib.mt_maxpool_add_fwd(L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift), L.ptr(z_s), L.ptr(bn_s.scale),
                                               L.ptr(bn_s.shift), L.ptr(y), N, Hin, Hin, cout, L.stream_ptr()), "mt_maxpool_add_fwd")
            brec.update(z_s=z_s, bn_s=bn_s, Ho=Ho)
            cur = _Src(y, cout, Ho)
        else:
            y = _new(dev, N * Hin * Hin, cout)
            L.check(lib.mt_bn_act_fwd(L.ptr(src.t), L.ptr(src.scale), L.ptr(src.shift), L.ptr(inp.t), L.ptr(y), N * Hin * Hin, cout, 0,
                                      None, 1, L.stream_ptr()), "mt_bn_act_fwd")
            cur = _Src(y, cout, Hin)
        if save:
            saved["blocks"].append(brec)

    # ---- conv3 / conv4 tail
    tail = []
    for k_, (sep_mod, bn_mod, co, act) in enumerate(((model.conv3, model.bn3, 1536, NONE), (model.conv4, model.bn4, 2048, RELU))):
        w_dw, w_pw, g, b = next(it), next(it), next(it), next(it)
        cur, rec = sep_unit(cur, act, w_dw, w_pw, co, bn_mod, g, b)
        tail.append(rec)
    M = N * cur.H * cur.H
    feat = _new(dev, M, 2048)
    L.check(lib.mt_bn_act_fwd(L.ptr(cur.t), L.ptr(cur.scale), L.ptr(cur.shift), None, L.ptr(feat), M, 2048, 0, None, 1, L.stream_ptr()),
            "mt_bn_act_fwd")
    if save:
        saved["tail"] = tail
    if planes_on:
        wplanes.end(dev)
        if save:
            saved["w_serial"] = wplanes.touch()
    _bump_tracked(plan)
    return feat, saved, cur.H


def xception_backward(model, params, saved, shape, training, dfeat, need_dparams, keep_saved=False, plan=None):
    """keep_saved: the activation records belong to a launch plan (plans.py) and stay for its next replay; plan: the NetPlan whose
    backward phase is being recorded (it keeps the flat gradient buffer)."""
    lib = L.get()
    if "w_serial" in saved and model.__dict__["_xc_wplanes"].serial != saved["w_serial"]:
        raise RuntimeError("Xception: the pointwise weights were updated between this graph's forward and its backward (their operand "
                           "planes were rewritten by a later forward): run backward before the optimizer step")
    dev = dfeat.device
    N, H, W = shape
    P = list(params)
    grads, flat_grads = L.zero_grads(P, with_flat=True)
    consts = saved["consts"]
    det = L.deterministic()
    slots = -SLOTS if det else SLOTS          # (see xception_forward)
    pool = _StatsPool(dev, 2 * _total_channels(model) + 4096, det)
    tr = 1 if training else 0
    side = L.SideStream(dev, hold=True)
    # parameter index map
    pos = 6
    bmap = []
    for blk in model.blocks():
        d = {}
        if blk.skip is not None:
            d["skip"] = pos
            pos += 3
        d["units"] = []
        for _ in blk.unit_index:
            d["units"].append(pos)
            pos += 4
        bmap.append(d)
    tail_idx = [pos, pos + 4]

    def bn_sums(g, z, ctx, rows, act=0, dout=None):
        sums = pool.take(ctx.C)
        L.check(lib.mt_bn_act_bwd(L.ptr(g), L.ptr(z), L.ptr(ctx.scale), L.ptr(ctx.shift), L.ptr(ctx.mean_invstd), None, None, None,
                                  L.ptr(dout), L.ptr(sums), slots, rows, ctx.C, 1, act, L.stream_ptr()), "mt_bn_act_bwd")
        return sums

    def bn_kabc(ctx, sums, gi, d, z, rows):
        """(d, z, rows): the gradient w.r.t. the BatchNorm's output and its input tensor (round 4's deterministic mode retook the sums
        from them; they are integer limbs now)."""
        kabc = _new(dev, 3, ctx.C)
        L.check(lib.mt_bn_bwd_finalize(L.ptr(sums), slots, ctx.count, L.ptr(P[gi]), L.ptr(ctx.mean_invstd), L.ptr(kabc), L.ptr(grads[gi]),
                                       L.ptr(grads[gi + 1]), ctx.C, tr, L.stream_ptr()), "mt_bn_bwd_finalize")
        return kabc

    def unit_backward(rec, pi, g, sums, res_pre=None, res_post=None, pooled=None, res_half=False):
        """g = gradient w.r.t. this unit's BatchNorm output (sums already accumulated when `sums` is given).
        pooled = (dy, arg, zmax, Ho) instead of g: the unit feeds the block's max-pool and the gradient is dy routed by `arg`.
        res_half: res_pre / res_post are at half resolution (the stride-2 skip convolution's data gradient, even positions only).
        Returns (gradient w.r.t. the source's affine output [activation derivative applied], its BN sums or None)."""
        ci, co, Hh = rec["ci"], rec["co"], rec["H"]
        M = N * Hh * Hh
        if pooled is not None:
            p_dy, p_arg, p_zmax, p_Ho = pooled
            sums = bn_sums(p_dy, p_zmax, rec["bn"], N * p_Ho * p_Ho)      # sum du = sum dy, sum du xhat = sum dy xhat(arg-max)
            if rec["d_p"] is None:
                g = _new(dev, M, co)
                L.check(lib.mt_maxpool_bwd_arg(L.ptr(p_dy), L.ptr(p_arg), L.ptr(g), N, Hh, Hh, co, L.stream_ptr()), "mt_maxpool_bwd_arg")
        elif sums is None:
            sums = bn_sums(g, rec["z"], rec["bn"], M)
        kabc = bn_kabc(rec["bn"], sums, pi + 2, g, rec["z"], M)
        w_dw, w_pw = P[pi], P[pi + 1]
        # pointwise: z = d . Wpw^T
        dd = _new(dev, M, ci)
        if rec["d_p"] is not None:
            # plane operands: dz = ka g + kb z + kc is evaluated ONCE, as planes, and feeds both gradient GEMMs
            dz_p = L.planes_empty(M, co, dev)
            if pooled is not None:
                L.check(lib.mt_maxpool_bn_bwd_apply_planes(L.ptr(p_dy), L.ptr(p_arg), L.ptr(rec["z"]), L.ptr(kabc), L.ptr(dz_p), N, Hh, Hh, co,
                                                           L.stream_ptr()), "mt_maxpool_bn_bwd_apply_planes")
            else:
                L.check(lib.mt_bn_bwd_apply_planes(L.ptr(g), L.ptr(rec["z"]), L.ptr(kabc), L.ptr(dz_p), M, co, L.stream_ptr()),
                        "mt_bn_bwd_apply_planes")
            side.launch(lambda: L.gemm_planes(L.OP_TN, dz_p, rec["d_p"], co, ci, M, Cout=grads[pi + 1].view(co, ci), ldc=ci,
                                              epilogue=L.EPI_ATOMIC), reads=(dz_p, rec["d_p"]))
            L.gemm_planes(L.OP_NN, dz_p, rec["w_p"], M, ci, co, Cout=dd, ldc=ci)
        else:
            side.launch(lambda: L.gemm(L.OP_TN, g, rec["d"], grads[pi + 1], co, ci, M, co, ci, ci, prologue=L.PRO_BN_BWD,
                                       epilogue=L.EPI_ATOMIC, split_k=0, A2=rec["z"], scale=kabc[0], shift=kabc[1], gate=kabc[2]),
                        reads=(g, rec["d"], rec["z"], kabc))
            L.gemm(L.OP_NN, g, w_pw, dd, M, ci, co, co, ci, ci, prologue=L.PRO_BN_BWD, A2=rec["z"], scale=kabc[0], shift=kabc[1],
                   gate=kabc[2])
        # depthwise: d = dw(act(affine(src)))
        src = rec["src"]
        kid = consts.kabc_ident(ci)
        has_bn = src.bn is not None
        sums_in = pool.take(ci) if has_bn else None
        du_in = _new(dev, M, ci)

        def dw_part(parts):
            fn = lib.mt_dwconv_bwd_res2 if res_half else lib.mt_dwconv_bwd
            L.check(fn(L.ptr(dd), L.ptr(dd), L.ptr(kid), L.ptr(w_dw), L.ptr(src.t), L.ptr(rec["sc"]), L.ptr(rec["sh"]),
                       L.ptr(src.bn.mean_invstd) if has_bn else None, L.ptr(du_in), L.ptr(sums_in), slots, L.ptr(grads[pi]), N, Hh, Hh, ci,
                       3, 1, parts, rec["eff"], L.ptr(res_pre), L.ptr(res_post), L.stream_ptr()), "mt_dwconv_bwd")
        if XC_DW_FUSED and not det:
            dw_part(3)            # data + weight gradient from one pass (effnet_backward.FUSED_DW; deterministic mode keeps the logged weight-gradient kernel)
        else:
            side.launch(lambda: dw_part(1), reads=(dd, kid, src.t, rec["sc"], rec["sh"]))
            dw_part(2)
        return du_in, sums_in

    # ---- tail: feat = bn4(z4); conv4 consumes relu(bn3(z3)); conv3 consumes y12 (no relu)
    t3, t4 = saved["tail"]
    g, sums = unit_backward(t4, tail_idx[1], dfeat, None)            # -> grad wrt bn3 output (+ bn3 sums)
    dy, _ = unit_backward(t3, tail_idx[0], g, sums)                  # -> grad wrt y12

    # ---- blocks, last to first
    blocks = model.blocks()
    for bi in reversed(range(len(blocks))):
        blk, brec, bm = blocks[bi], saved["blocks"][bi], bmap[bi]
        name, cin, cout, reps, stride, srelu, grow = blk.cfg
        inp = brec["inp"]
        units = brec["units"]
        last = units[-1]
        Hin = inp.H
        res_pre = res_post = None
        if stride != 1:
            Ho = brec["Ho"]
            Mo = N * Ho * Ho
            # skip path: y += skipbn(z_s), z_s = conv1x1_s2(value of inp)
            ks = bn_kabc(brec["bn_s"], bn_sums(dy, brec["z_s"], brec["bn_s"], Mo), bm["skip"] + 1, dy, brec["z_s"], Mo)
            geom = (Hin, Hin, cin, Ho, Ho, 1, 2, 0, inp.act)
            side.launch(lambda dy=dy, ks=ks, brec=brec, geom=geom, inp=inp, gi=bm["skip"]:
                        L.gemm(L.OP_TN, dy, inp.t, grads[gi].view(cout, cin), cout, cin, Mo, cout, cin, cin, prologue=L.PRO_BN_BWD,
                               epilogue=L.EPI_ATOMIC, split_k=0, A2=brec["z_s"], scale=ks[0], shift=ks[1], gate=ks[2],
                               b_prologue=L.BPRO_IM2COL, b_scale=inp.scale, b_shift=inp.shift, conv=geom),
                        reads=(dy, inp.t, brec["z_s"], ks))
            d_small = _new(dev, Mo, cin)
            L.gemm(L.OP_NN, dy, P[bm["skip"]].view(cout, cin), d_small, Mo, cin, cout, cout, cin, cin, prologue=L.PRO_BN_BWD,
                   A2=brec["z_s"], scale=ks[0], shift=ks[1], gate=ks[2])
            # it lands on the even positions (2oh, 2ow) of the block input: the first unit's depthwise adjoint adds it from the
            # half-resolution tensor (SKIP_HALF; round 4 scattered it into an input-sized zero tensor first)
            if SKIP_HALF:
                d_skip = d_small
            else:
                d_skip = torch.zeros(N, Hin, Hin, cin, dtype=torch.float32, device=dev)
                d_skip[:, ::2, ::2, :] = d_small.view(N, Ho, Ho, cin)
                d_skip = d_skip.view(N * Hin * Hin, cin)
            if inp.bn is not None or inp.act == RELU:
                res_pre = d_skip          # the skip conv consumed the same activated tensor as the first unit
            else:
                res_post = d_skip         # the skip conv consumed the raw block input, the first unit relu(input)
            # main path: y += maxpool(bn(z_last))
            pooled = None
            if "arg" in brec:
                g, sums, pooled = None, None, (dy, brec["arg"], brec["zmax"], Ho)
            else:
                du_last = torch.zeros(N * Hin * Hin, cout, dtype=torch.float32, device=dev)
                L.check(lib.mt_maxpool_bwd(L.ptr(dy), L.ptr(last["z"]), L.ptr(last["bn"].scale), L.ptr(last["bn"].shift), L.ptr(du_last), N,
                                           Hin, Hin, cout, L.stream_ptr()), "mt_maxpool_bwd")
                g, sums = du_last, None
        else:
            g, sums, pooled = dy, None, None
            res_post = dy                 # identity skip: y = bn(z_last) + inp
        for u in reversed(range(len(units))):
            first = u == 0
            g, sums = unit_backward(units[u], bm["units"][u], g, sums, res_pre if first else None, res_post if first else None,
                                    pooled if u == len(units) - 1 else None, res_half=first and stride != 1 and SKIP_HALF)
        # for block1 the result is the gradient w.r.t. bn2's output (sums accumulated); otherwise w.r.t. the previous block's y
        dy = g
        bn2_sums = sums
        if not keep_saved:
            brec.clear()
        side.release_point()

    # ---- conv2 / conv1
    z1, z2, bn1, bn2, s1 = saved["z1"], saved["z2"], saved["bn1"], saved["bn2"], saved["s1"]
    H1, H2 = saved["H1"], saved["H2"]
    M1, M2 = N * H1 * H1, N * H2 * H2
    k2 = bn_kabc(bn2, bn2_sums, 4, dy, z2, M2)
    dwp2 = L.zeros((64, 288), torch.float32, dev)          # (a fill of the library: a recorded phase re-issues it)
    geom2 = (H1, H1, 32, H2, H2, 3, 1, 0, RELU)

    def conv2_wgrad(n0, n1):      # images [n0, n1): 6 M rows x (64 x 288) in all, 9 ms
        r0, r1 = n0 * H2 * H2, n1 * H2 * H2
        L.gemm(L.OP_TN, dy[r0:r1], z1[n0 * H1 * H1:n1 * H1 * H1], dwp2, 64, 288, r1 - r0, 64, 288, 288, prologue=L.PRO_BN_BWD,
               epilogue=L.EPI_ATOMIC, split_k=0, A2=z2[r0:r1], scale=k2[0], shift=k2[1], gate=k2[2], b_prologue=L.BPRO_IM2COL,
               b_scale=bn1.scale, b_shift=bn1.shift, conv=geom2)
    # most of it on the side stream, beside conv2's data gradient and conv1's weight gradient; the rest after those on the main stream,
    # so that both queues end together (the whole of it on either queue leaves the other idle for 3-9 ms at the end of the step)
    # (deterministic mode: every split-K GEMM of this backward stays on the side stream -- they share the slab arena in stream order)
    n_side = N if (not side.enabled or det) else max(1, min(N, int(round(N * CONV2_WGRAD_SIDE))))
    side.launch(lambda: conv2_wgrad(0, n_side), reads=(dy, z1, z2, k2, dwp2, bn1.scale, bn1.shift))
    dz2 = _new(dev, M2, 64)
    L.check(lib.mt_bn_bwd_apply(L.ptr(dy), L.ptr(z2), L.ptr(k2), L.ptr(dz2), M2, 64, L.stream_ptr()), "mt_bn_bwd_apply")
    # data gradient of conv2 = "full" correlation of dz2 with the flipped kernel: im2col(dz2, pad 2) . W2flip^T
    wp2t = _new(dev, 32, 576)
    L.check(lib.mt_conv_weight_pack(L.ptr(P[3]), L.ptr(wp2t), 64, 32, 3, 576, 1, L.stream_ptr()), "mt_conv_weight_pack")
    da1 = _new(dev, M1, 32)
    L.gemm(L.OP_NT, dz2, wp2t, da1, M1, 32, 576, 576, 576, 32, prologue=L.PRO_IM2COL, conv=(H2, H2, 64, H1, H1, 3, 1, 2, NONE))
    du1 = _new(dev, M1, 32)
    k1 = bn_kabc(bn1, bn_sums(da1, z1, bn1, M1, act=RELU, dout=du1), 1, du1, z1, M1)
    if XC_STEM and W <= 512:
        L.check(lib.mt_stem_conv_wgrad_valid(L.ptr(du1), L.ptr(z1), L.ptr(k1), L.ptr(saved["x"]), 1 if saved["x"].dtype == torch.uint8 else 0,
                                             L.ptr(grads[0]), N, H, W, L.stream_ptr()), "mt_stem_conv_wgrad_valid")
    else:
        dwp1 = torch.zeros(32, 28, dtype=torch.float32, device=dev)
        L.gemm(L.OP_TN, du1, saved["x"], dwp1, 32, 28, M1, 32, 28, 28, prologue=L.PRO_BN_BWD, epilogue=L.EPI_ATOMIC, split_k=0, A2=z1,
               scale=k1[0], shift=k1[1], gate=k1[2], b_prologue=L.BPRO_IM2COL,
               conv=(H, W, 3, H1, H1, 3, 2, 0, NONE, 1 if saved["x"].dtype == torch.uint8 else 0))
        L.check(lib.mt_conv_weight_unpack_grad(L.ptr(dwp1), L.ptr(grads[0]), 32, 3, 3, 28, L.stream_ptr()), "mt_conv_weight_unpack_grad")
    if n_side < N:
        conv2_wgrad(n_side, N)
    side.wait()
    L.check(lib.mt_conv_weight_unpack_grad(L.ptr(dwp2), L.ptr(grads[3]), 64, 32, 3, 288, L.stream_ptr()), "mt_conv_weight_unpack_grad")
    if plan is not None:
        plan.extra["flat_grads"] = flat_grads
    L.grads_ready(model, P, flat_grads)
    return [g_ if need else None for need, g_ in zip(need_dparams, grads)]


def _plannable():
    """The recorded form covers the default kernels only: the lab alternatives issue torch fills / scatters a recording cannot see."""
    return SKIP_HALF and POOL_ARG and XC_STEM and XC_PLANES and DW_PLANES and XC_PLAN


XC_DW_FUSED = __import__("os").environ.get("MT_XC_DW_FUSED", "1") != "0"   # depthwise data + weight gradient in one kernel (all of Xception's are 3x3 stride 1): config 5 200.0 -> 193.5 ms; 0 = two kernels on two streams
XC_PLAN = __import__("os").environ.get("MT_XC_PLAN", "1") != "0"      # 0: the Xception phases are never recorded (launch plans, plans.py)


class _XceptionFunction(torch.autograd.Function):
    """Forward / backward of the extractor; like effnet_engine._EffNetFunction the two launch sequences are recorded into the library
    the second time a (shape, mode) key is seen and re-issued from C afterwards (plans.py): ~750 ctypes launches per step become 2."""

    @staticmethod
    def forward(ctx, model, x_nhwc, *params):
        from . import plans
        model, grad_on = model
        save = grad_on and any(ctx.needs_input_grad)      # see tsf_engine._TSFFunction.forward
        N, H, W, _ = x_nhwc.shape
        ctx.shape = (N, H, W)
        ctx.model, ctx.params, ctx.training = model, params, model.training
        ctx.plan = ctx.token = None
        np_, mode = None, "eager"
        if save and _plannable() and W <= 512 and L.gemm_split_enabled():
            stream = torch.cuda.current_stream(x_nhwc.device).cuda_stream
            key = ("xc", tuple(x_nhwc.shape), x_nhwc.dtype, model.training, L.deterministic(), tuple(ctx.needs_input_grad[2:]), stream,
                   float(model.bn1.momentum), float(model.bn1.eps))
            np_, mode = plans.lookup(model, key)
            if mode == "replay" and np_.state_ptrs != plans.state_ptrs(list(params) + list(model.buffers())):
                plans.drop(model, np_)                   # parameters / buffers moved: record afresh later
                np_, mode = None, "eager"
        if mode == "eager":
            feat, saved, ho = xception_forward(model, x_nhwc, params, model.training, save)
            ctx.saved = saved
            return feat
        if mode == "record":
            np_.stream = stream
            x_s = plans.static_input(np_, "x", x_nhwc)
            pl = L.Plan()
            try:
                with pl:
                    feat, saved, _ = xception_forward(model, x_s, params, model.training, True, plan=np_)
            except Exception:
                np_.broken = True
                raise
            np_.fwd = pl
            np_.extra.update(saved=saved, feat=feat)
            np_.state_ptrs = plans.state_ptrs(list(params) + list(model.buffers()))
            plans.own(np_, feat)
            plans.STATS["recorded"] += 1
        else:
            plans.refresh_input(np_, "x", x_nhwc)
            plans.run(np_.fwd)
            if "w_serial" in np_.extra["saved"]:
                np_.extra["saved"]["w_serial"] = model.__dict__["_xc_wplanes"].touch()      # the split launch is in the plan
            if np_.extra.get("tracked"):
                torch._foreach_add_(np_.extra["tracked"], 1)
        ctx.plan, ctx.token = np_, np_.begin()
        ctx.saved = np_.extra["saved"]
        return np_.extra["feat"].detach()

    @staticmethod
    def backward(ctx, dfeat):
        if ctx.saved is None:
            raise RuntimeError("Xception: backward ran a second time through the same forward; the activation buffers are "
                               "released after the first pass (retain_graph is not supported by the HIP engine)")
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. the input crops is not part of the MINTIME training path")
        from . import plans
        np_ = ctx.plan
        if "w_serial" in ctx.saved and ctx.model.__dict__["_xc_wplanes"].serial != ctx.saved["w_serial"]:
            raise RuntimeError("Xception: the pointwise weights were updated between this graph's forward and its backward (their operand "
                               "planes were rewritten by a later forward): run backward before the optimizer step")
        dfeat = dfeat.contiguous()
        need = ctx.needs_input_grad[2:]
        if np_ is None:
            dparams = xception_backward(ctx.model, ctx.params, ctx.saved, ctx.shape, ctx.training, dfeat, need)
        elif (plans.grads_exist(ctx.params) or torch.cuda.current_stream(dfeat.device).cuda_stream != np_.stream
              or torch.cuda.is_current_stream_capturing()):
            # existing gradients are ADDED to by autograd (they alias the plan's gradient buffer): eager sequence over the plan's
            # saved activations, which stay for the next replay
            plans.STATS["eager_accumulate"] += 1
            dparams = xception_backward(ctx.model, ctx.params, ctx.saved, ctx.shape, ctx.training, dfeat, need, keep_saved=True)
        elif np_.bwd is None:
            d_s = plans.static_input(np_, "dfeat", dfeat)
            pl = L.Plan()
            try:
                with pl:
                    dparams = xception_backward(ctx.model, ctx.params, ctx.saved, ctx.shape, ctx.training, d_s, need, keep_saved=True,
                                                plan=np_)
            except Exception:
                np_.broken = True
                raise
            np_.bwd = pl
            np_.extra.update(grads=list(dparams))
            dparams = plans.fresh_aliases(dparams)
        else:
            plans.refresh_input(np_, "dfeat", dfeat)
            plans.run(np_.bwd)
            L.grads_ready(ctx.model, list(ctx.params), np_.extra["flat_grads"])
            dparams = plans.fresh_aliases(np_.extra["grads"])
        ctx.saved = None
        if np_ is not None:
            np_.release(ctx.token)
            ctx.token = None
        return (None, None) + tuple(dparams)


def xception_apply(model, inputs):
    if not inputs.is_cuda:
        raise L.MintimeHipError("Xception (MI355X build) needs device tensors; there is no CPU path")
    if inputs.dim() != 4 or inputs.shape[1] != 3:
        raise ValueError(f"expected [N,3,H,W] input, got {tuple(inputs.shape)}")
    x = (inputs if inputs.dtype == torch.uint8 else inputs.float()).permute(0, 2, 3, 1)      # uint8 crops are ingested as they are
    if not x.is_contiguous():
        x = x.contiguous()
    feat = _XceptionFunction.apply((model, torch.is_grad_enabled()), x, *param_list(model))
    n = inputs.shape[0]
    ho = feat.shape[0] // n
    side = int(round(ho ** 0.5))
    return feat.view(n, side, side, 2048).permute(0, 3, 1, 2)
