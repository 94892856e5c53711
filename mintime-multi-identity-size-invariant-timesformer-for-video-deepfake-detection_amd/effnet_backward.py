"""Reverse launch sequence of EfficientNet-B0 (analytic backward on libmintime_hip; see csrc/effnet_bwd.hip)."""
import torch

from . import arch
from . import lib as L
from .effnet_engine import SLOTS, STREAM_ROWS, _StatsPool


# Depthwise data AND weight gradient from one pass over du / z / the depthwise input (csrc/effnet_bwd.hip, WG = true): the gather leaves
# the activated input tile in LDS, the threads change role and walk it against the dz tile.  Rounds 2-5 measured every fused form
# slower than the two kernels on two streams; with the branch-free tap reads of round 6 it wins for the 3x3 layers (one box, serialised
# extractor 21.82 -> 21.35 ms, step 44.99 -> 44.43 ms; all layers: 21.34 / 44.60): "3" = the 3x3 layers (default), "1" = every layer, "0" = off.
FUSED_DW = __import__("os").environ.get("MT_DW_FUSED", "3")
# Squeeze-excite stage of the reverse walk without the project conv's data gradient `da` in memory: the GEMM  dz_p . W_project  is
# run twice and consumed in its accumulators (MT_EPI_SE_RED: d gate; MT_EPI_ACT_BWD: du of the depthwise BatchNorm + its sums).
# 3 passes over a block's expanded tensor (read z_d, read z_d, write du_d) instead of 6 (write da | read da, z_d | read da, z_d,
# write du_d), two launches fewer -- and SLOWER on every block (profiles/r03_se_fused_epilogue_vs_streaming.txt: block 1
# 406 -> 744 us, the 7x7 blocks 126 -> 193 us, step +1.3 ms): a one-tile-per-block GEMM with a load-z / swish / store epilogue
# per lane-column runs at 1.3-3.2 TB/s where the float4 streaming kernels it replaces run at 4.5-5.5.  Parity-tested, opt-in.
SE_FUSED = __import__("os").environ.get("MT_SE_FUSED", "0") != "0"
EXPAND_FUSED = __import__("os").environ.get("MT_EXPAND_FUSED", "1") != "0"
WIDE_WGRAD = __import__("os").environ.get("MT_WIDE_WGRAD", "1") != "0"      # late-stage project-conv weight gradients: wide_wgrad.hip
# squeeze-excite reverse stage of the early blocks (stages 0-2) as two STREAMING passes that rebuild the project conv's data gradient
# from the narrow gradient (skinny_se.hip): 3 instead of 6 passes over the expanded tensor, no da tensor
SE_STREAM = __import__("os").environ.get("MT_SE_STREAM", "1") != "0"     # expand-conv data + weight gradient in one pass (stages 1-3)


def _new(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


def _splits(rows):
    return max(1, min(64, rows // 1024))


# what the last backward pass actually launched (tests check that frozen parameters cost nothing: train.py:153-170)
LAST_RUN = {"blocks_run": 0, "wgrad_launches": 0, "stem_run": False}


def effnet_backward(model, params, saved, shape, training, dfeat, need_dx, need_dparams, keep_saved=False, plan=None):
    """keep_saved: the activation records belong to a launch plan (plans.py) and stay for its next replay; plan: the NetPlan whose
    backward phase is being recorded (it keeps the flat gradient buffer)."""
    lib = L.get()
    st = L.stream_ptr()
    dev = dfeat.device
    N, H, W = shape
    blocks = model._blocks
    P = list(params)
    grads, flat_grads = L.zero_grads(list(params), with_flat=True)
    # parameter index map (same order as effnet_engine.param_list)
    pos = 3
    bidx = []
    for blk in blocks:
        d = {}
        if blk.spec.has_expand:
            d["e"] = pos
            pos += 3
        d["d"] = pos
        d["se"] = pos + 3
        d["p"] = pos + 7
        pos += 10
        bidx.append(d)
    ih = pos
    # Frozen parameters (train.py:159-167 unfreezes only the last k MBConv blocks; everything else has requires_grad = False):
    # their weight-gradient launches are skipped, and the reverse walk stops at the lowest block that still owns a trainable
    # parameter -- nothing below it is touched.
    need = [bool(x) for x in need_dparams]
    first_of = [3 if bi == 0 else (bidx[bi]["e"] if "e" in bidx[bi] else bidx[bi]["d"]) for bi in range(len(blocks))]
    need_below = [any(need[:first_of[bi]]) for bi in range(len(blocks))]        # stem or an earlier block needs a gradient
    lowest = min((bi for bi in range(len(blocks)) if any(need[first_of[bi]:(first_of[bi + 1] if bi + 1 < len(blocks) else ih)])),
                 default=len(blocks))
    if any(need[:3]):
        lowest = -1                                                             # the stem itself is trainable
    run = {"blocks_run": 0, "wgrad_launches": 0, "stem_run": False}
    total_c = arch.STEM_COUT + arch.HEAD_COUT + sum((b.spec.cexp if b.spec.has_expand else 0) + b.spec.cexp + b.spec.cout
                                                    for b in blocks)
    tr = 1 if training else 0
    side = L.SideStream(dev, hold=True)     # (eager launches; a recording plan pins what the side stream reads itself)

    # deterministic mode (MT_DETERMINISTIC / lib.set_deterministic): only kernels whose reductions have a fixed-order form -- the
    # streaming fusions that meet their partial sums with LDS / global atomics step aside for the GEMM (split-K slabs), the
    # row-streaming data gradient and the unfused squeeze-excite stage
    det = L.deterministic()
    pool = _StatsPool(dev, total_c, det)
    slots = -SLOTS if det else SLOTS       # deterministic mode: BatchNorm-backward sums as integer limbs (csrc/common.hpp stat_add)
    expand_fused, wide_wgrad, se_stream, se_fused = (EXPAND_FUSED and not det, WIDE_WGRAD and not det, SE_STREAM and not det,
                                                     SE_FUSED and not det)
    fused_dw = "0" if det else FUSED_DW

    def bn_finalize(bnctx, sums, gidx_gamma, d=None, z=None, rows=0):
        kabc = _new(dev, 3, bnctx.C)
        L.check(lib.mt_bn_bwd_finalize(L.ptr(sums), slots, bnctx.count, L.ptr(P[gidx_gamma]), L.ptr(bnctx.mean_invstd), L.ptr(kabc),
                                       L.ptr(grads[gidx_gamma]), L.ptr(grads[gidx_gamma + 1]), bnctx.C, tr, st), "mt_bn_bwd_finalize")
        return kabc

    def act_bwd(din, z, bnctx, dout, rows, hw, act, gate=None, dpool=None, rowscale=None):
        sums = pool.take(bnctx.C)
        L.check(lib.mt_bn_act_bwd(L.ptr(din), L.ptr(z), L.ptr(bnctx.scale), L.ptr(bnctx.shift), L.ptr(bnctx.mean_invstd), L.ptr(gate),
                                  L.ptr(dpool), L.ptr(rowscale), L.ptr(dout), L.ptr(sums), slots, rows, bnctx.C, hw, act, st),
                "mt_bn_act_bwd")
        return sums

    def conv1x1_bwd(du, z, kabc, w, x_in, rows, cout, cin, gw_idx, need_dx_in, res=None, b_pro=None, epi=None, pl=None):
        """z = x_in . w^T with dz = ka*du+kb*z+kc.  Returns dx_in [rows, cin] (+res) or None.
        epi = (kind, out, kwargs): the data gradient is not stored but consumed by mt_gemm's SE_RED / ACT_BWD epilogue.
        pl = {"w_p": weight planes, "x_p": planes of the forward operand or None}: the late stages' plane path -- dz is evaluated ONCE,
        as planes (mt_bn_bwd_apply_planes), and feeds the data-gradient GEMM (weight planes read along their rows) and, when the
        forward kept its operand as planes, the weight-gradient GEMM."""
        if pl is not None and epi is None:
            dz_p = L.planes_empty(rows, cout, dev)
            L.check(lib.mt_bn_bwd_apply_planes(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(dz_p), rows, cout, st), "mt_bn_bwd_apply_planes")
            wdone = False
            if need[gw_idx] and pl.get("x_p") is not None:
                run["wgrad_launches"] += 1
                side.launch(lambda: L.gemm_planes(L.OP_TN, dz_p, pl["x_p"], cout, cin, rows, Cout=grads[gw_idx].view(cout, cin), ldc=cin,
                                                  epilogue=L.EPI_ATOMIC), reads=(dz_p, pl["x_p"]))
                wdone = True
            dx_in = None
            if need_dx_in:
                dx_in = _new(dev, rows, cin)
                if res is not None:
                    L.gemm_planes(L.OP_NN, dz_p, pl["w_p"], rows, cin, cout, Cout=dx_in, ldc=cin, epilogue=L.EPI_BIAS_RES, R=res, ldr=cin)
                else:
                    L.gemm_planes(L.OP_NN, dz_p, pl["w_p"], rows, cin, cout, Cout=dx_in, ldc=cin)
            if wdone or not need[gw_idx]:
                return dx_in
            need_dx_in, keep_dx = False, dx_in           # the weight gradient alone on the kernels below (its operand was not kept as planes)
        else:
            keep_dx = None
        kw = {}
        if b_pro is not None:
            kw = dict(b_prologue=L.BPRO_BN_SWISH_GATE, b_scale=b_pro[0], b_shift=b_pro[1], b_gate=b_pro[2], b_hw=b_pro[3])
        reads = (du, z, x_in, kabc) + (tuple(b_pro[:3]) if b_pro is not None else ())
        if (expand_fused and b_pro is None and epi is None and need[gw_idx] and need_dx_in and rows >= 100000
                and lib.mt_conv1x1_bwd_fused_supported(cout, cin)):
            # expand convs of stages 1-3: data AND weight gradient in one streaming pass over du (z is folded: 1 instead of 4 passes
            # over the widest tensors of the step; skinny_bwd.hip).  Runs on the main stream: it is the data gradient's critical path.
            run["wgrad_launches"] += 1
            dx_in = _new(dev, rows, cin)
            L.check(lib.mt_conv1x1_bwd_fused(L.ptr(du), L.ptr(kabc), L.ptr(x_in), L.ptr(w), L.ptr(res), L.ptr(dx_in),
                                             L.ptr(grads[gw_idx]), rows, cout, cin, st), "mt_conv1x1_bwd_fused")
            return dx_in
        if not need[gw_idx]:
            pass                                      # frozen weight: no launch
        elif rows >= 100000 and not det and lib.mt_conv1x1_wgrad_supported(cout, cin):
            run["wgrad_launches"] += 1
            # few channels, very many rows: the result stays in MFMA accumulators while the rows stream (skinny_wgrad.hip)
            bp = b_pro if b_pro is not None else (None, None, None, 1)
            side.launch(lambda: L.check(lib.mt_conv1x1_wgrad(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x_in), L.ptr(bp[0]), L.ptr(bp[1]),
                                                             L.ptr(bp[2]), bp[3], L.ptr(grads[gw_idx]), rows, cout, cin,
                                                             L.stream_ptr()), "mt_conv1x1_wgrad"), reads=reads)
        elif wide_wgrad and b_pro is not None and lib.mt_conv1x1_wgrad_wide_supported(cout, cin):
            run["wgrad_launches"] += 1
            # project convs of the late stages: 128-column slabs of the result, both operand transforms applied once (wide_wgrad.hip)
            side.launch(lambda: L.check(lib.mt_conv1x1_wgrad_wide(L.ptr(du), L.ptr(z), L.ptr(kabc), L.ptr(x_in), L.ptr(b_pro[0]),
                                                                  L.ptr(b_pro[1]), L.ptr(b_pro[2]), b_pro[3], L.ptr(grads[gw_idx]), rows,
                                                                  cout, cin, L.stream_ptr()), "mt_conv1x1_wgrad_wide"), reads=reads)
        else:
            run["wgrad_launches"] += 1
            side.launch(lambda: L.gemm(L.OP_TN, du, x_in, grads[gw_idx], cout, cin, rows, cout, cin, cin, prologue=L.PRO_BN_BWD,
                                       epilogue=L.EPI_ATOMIC, split_k=0, A2=z, scale=kabc[0], shift=kabc[1], gate=kabc[2], **kw),
                        reads=reads)
        if not need_dx_in:
            return keep_dx
        if epi is not None:
            for kind, out, kw2 in epi:
                L.gemm(L.OP_NN, du, w, out, rows, cin, cout, cout, cin, cin, prologue=L.PRO_BN_BWD, epilogue=kind, A2=z,
                       scale=kabc[0], shift=kabc[1], gate=kabc[2], **kw2)
            return None
        dx_in = _new(dev, rows, cin)
        if rows >= STREAM_ROWS and lib.mt_conv1x1_rows_supported(cout, cin, 2):
            # data gradient as a streaming kernel: a = dz (BatchNorm backward folded on load), W used transposed
            L.check(lib.mt_conv1x1_rows(L.ptr(du), L.ptr(z), L.ptr(w), cin, 1, L.ptr(kabc[0]), L.ptr(kabc[1]), L.ptr(kabc[2]), 1, 2,
                                        L.ptr(res), L.ptr(dx_in), None, 1, rows, cout, cin, st), "mt_conv1x1_rows")
            return dx_in
        if res is not None:
            L.gemm(L.OP_NN, du, w, dx_in, rows, cin, cout, cout, cin, cin, prologue=L.PRO_BN_BWD, epilogue=L.EPI_BIAS_RES, A2=z,
                   scale=kabc[0], shift=kabc[1], gate=kabc[2], R=res, ldr=cin)
        else:
            L.gemm(L.OP_NN, du, w, dx_in, rows, cin, cout, cout, cin, cin, prologue=L.PRO_BN_BWD, A2=z, scale=kabc[0], shift=kabc[1],
                   gate=kabc[2])
        return dx_in

    # ---- head: feat = swish(bn1(z_h)), z_h = y . Wh^T
    hd = saved["head"]
    s_last = blocks[-1].spec
    M = N * s_last.hout * s_last.hout
    du_h = _new(dev, M, arch.HEAD_COUT)
    sums = act_bwd(dfeat, hd["z"], hd["bn"], du_h, M, 1, 1)
    kabc = bn_finalize(hd["bn"], sums, ih + 1, du_h, hd["z"], M)
    dy = conv1x1_bwd(du_h, hd["z"], kabc, P[ih], hd["y_in"], M, arch.HEAD_COUT, arch.HEAD_CIN, ih, lowest < len(blocks),
                     pl=dict(w_p=hd["w_p"], x_p=hd["y_p"]) if hd.get("w_p") is not None else None)
    del du_h

    # ---- blocks, last to first
    for bi in reversed(range(len(blocks))):
        if bi < lowest:
            break                                     # every parameter from here down is frozen
        run["blocks_run"] += 1
        blk, rec, ix = blocks[bi], saved["blocks"][bi], bidx[bi]
        s = rec["spec"]
        M_in, M_out = N * s.hin * s.hin, N * s.hout * s.hout
        hw = s.hout * s.hout
        # (a) bn2 (+ drop-connect gate): y = bn2(z_p)*dc (+ y_in)
        dc = rec["dc"]
        dyb = _new(dev, M_out, s.cout) if dc is not None else None
        sums = act_bwd(dy, rec["z_p"], rec["bn_p"], dyb, M_out, hw, 0, rowscale=dc)
        dsrc = dyb if dc is not None else dy
        kabc_p = bn_finalize(rec["bn_p"], sums, ix["p"] + 1, dsrc, rec["z_p"], M_out)
        # (b,c) project conv: z_p = (swish(bn1(z_d))*gate) . Wp^T
        bn_d = rec["bn_d"]
        b_pro = (bn_d.scale, bn_d.shift, rec["gate"], hw)
        dgate, dpre2, dpooled = _new(dev, N, s.cexp), _new(dev, N, s.cexp), _new(dev, N, s.cexp)
        dhid = _new(dev, N, s.cse)
        se_scr = _new(dev, lib.mt_se_scratch_floats(N, s.cexp, s.cse))
        se = ix["se"]
        def se_part(parts, _da=None, _rec=rec, _bn=bn_d, _se=se, _s=s, _dg=dgate, _dp=dpre2, _dh=dhid, _dpo=dpooled, _hw=hw, _scr=se_scr):
            L.check(lib.mt_se_bwd(L.ptr(_da), L.ptr(_rec["z_d"]), L.ptr(_bn.scale), L.ptr(_bn.shift), L.ptr(_rec["gate"]),
                                  L.ptr(_rec["hidden"]), L.ptr(_rec["pooled"]), L.ptr(P[_se]), L.ptr(P[_se + 2]), L.ptr(_dg), L.ptr(_dp),
                                  L.ptr(_dh), L.ptr(_dpo), L.ptr(grads[_se]), L.ptr(grads[_se + 1]), L.ptr(grads[_se + 2]),
                                  L.ptr(grads[_se + 3]), N, _hw, _s.cexp, _s.cse, parts, L.ptr(_scr), L.stream_ptr()), "mt_se_bwd")
        if (se_stream and not se_fused and M_out >= 100000
                and lib.mt_se_stage_fused_supported(s.cout, s.cexp, hw)):
            conv1x1_bwd(dsrc, rec["z_p"], kabc_p, P[ix["p"]], rec["z_d"], M_out, s.cout, s.cexp, ix["p"], False, b_pro=b_pro)   # weight gradient only
            def stage(mode, dg, g_, dpo, mi, out, st_):
                L.check(lib.mt_se_stage_fused(L.ptr(dsrc), L.ptr(rec["z_p"]), L.ptr(kabc_p), L.ptr(P[ix["p"]]), L.ptr(rec["z_d"]),
                                              L.ptr(bn_d.scale), L.ptr(bn_d.shift), mode, L.ptr(dg), L.ptr(g_), L.ptr(dpo), L.ptr(mi),
                                              L.ptr(out), L.ptr(st_), slots, M_out, s.cout, s.cexp, hw, st), "mt_se_stage_fused")
            L.zero_(dgate)
            stage(0, dgate, None, None, None, None, None)                       # (c+d) d gate
            se_part(4)                                                            # dgate -> dpooled
            if any(need[se:se + 4]):
                run["wgrad_launches"] += 1
                side.launch(lambda: se_part(2), reads=(dpre2, dhid, rec["hidden"], rec["pooled"]))
            da = _new(dev, M_out, s.cexp)                                         # (c+e) du_d and the bn1 sums
            sums = pool.take(s.cexp)
            stage(1, None, rec["gate"], dpooled, bn_d.mean_invstd, da, sums)
        elif se_fused:
            # (c+d) d gate straight from the accumulators of  dz_p . Wp  (da is never written)
            L.zero_(dgate)
            conv1x1_bwd(dsrc, rec["z_p"], kabc_p, P[ix["p"]], rec["z_d"], M_out, s.cout, s.cexp, ix["p"], True, b_pro=b_pro,
                        epi=[(L.EPI_SE_RED, dgate, dict(C2=rec["z_d"], ldc2=s.cexp, epi=(bn_d.scale, bn_d.shift, None, None, None, hw)))])
            se_part(4)                                # dgate -> dpooled
            if any(need[se:se + 4]):
                run["wgrad_launches"] += 1
                side.launch(lambda: se_part(2), reads=(dpre2, dhid, rec["hidden"], rec["pooled"]))
            # (c+e) the same product again, through gate / pooled gradient / swish' / bn1 sums: du_d
            da = _new(dev, M_out, s.cexp)
            sums = pool.take(s.cexp)
            need_save, need[ix["p"]] = need[ix["p"]], False        # the weight gradient was launched by the first call
            conv1x1_bwd(dsrc, rec["z_p"], kabc_p, P[ix["p"]], rec["z_d"], M_out, s.cout, s.cexp, ix["p"], True, b_pro=b_pro,
                        epi=[(L.EPI_ACT_BWD, da, dict(C2=rec["z_d"], ldc2=s.cexp, stats=sums, stats_slots=slots,
                                                      epi=(bn_d.scale, bn_d.shift, rec["gate"], dpooled, bn_d.mean_invstd, hw)))])
            need[ix["p"]] = need_save
        else:
            da = conv1x1_bwd(dsrc, rec["z_p"], kabc_p, P[ix["p"]], rec["z_d"], M_out, s.cout, s.cexp, ix["p"], True, b_pro=b_pro,
                             pl=dict(w_p=rec["wp_p"], x_p=rec["a_p"]) if rec.get("wp_p") is not None else None)
            # (d) squeeze-excite adjoint
            se_part(1, da)                                # dgate -> dpooled: the data path waits for these
            if any(need[se:se + 4]):
                run["wgrad_launches"] += 1
                side.launch(lambda: se_part(2), reads=(dpre2, dhid, rec["hidden"], rec["pooled"]))     # SE weight gradients: off the path
            # (e) through swish + bn1: du_d (in place over da)
            sums = act_bwd(da, rec["z_d"], bn_d, da, M_out, hw, 1, gate=rec["gate"], dpool=dpooled)
        kabc_d = bn_finalize(bn_d, sums, ix["d"] + 1, da, rec["z_d"], M_out)
        # (f,g) depthwise conv adjoint -> du wrt the dw input's pre-activation (+ its BN sums)
        in_bn = rec["dw_bn"]
        du_in = _new(dev, M_in, s.cexp)
        sums_in = pool.take(s.cexp)
        def dw_part(parts, _da=da, _rec=rec, _kabc=kabc_d, _bn=in_bn, _du_in=du_in, _sums=sums_in, _s=s, _ix=ix):
            if _rec.get("rc"):        # the depthwise input's pre-activation is rebuilt from the block input (csrc/rc.hpp)
                L.check(lib.mt_dwconv_bwd_rc(L.ptr(_da), L.ptr(_rec["z_d"]), L.ptr(_kabc), L.ptr(P[_ix["d"]]), L.ptr(_rec["y_in"]),
                                             L.ptr(_rec["w_e"]), _s.cin, L.ptr(_bn.scale), L.ptr(_bn.shift), L.ptr(_bn.mean_invstd),
                                             L.ptr(_du_in), L.ptr(_sums), slots, L.ptr(grads[_ix["d"]]), N, _s.hin, _s.hin, _s.cexp,
                                             _s.k, _s.s, parts, L.stream_ptr()), "mt_dwconv_bwd_rc")
                return
            L.check(lib.mt_dwconv_bwd(L.ptr(_da), L.ptr(_rec["z_d"]), L.ptr(_kabc), L.ptr(P[_ix["d"]]), L.ptr(_rec["dw_in"]),
                                      L.ptr(_bn.scale), L.ptr(_bn.shift), L.ptr(_bn.mean_invstd), L.ptr(_du_in), L.ptr(_sums), slots,
                                      L.ptr(grads[_ix["d"]]), N, _s.hin, _s.hin, _s.cexp, _s.k, _s.s, parts, 1, None, None, L.stream_ptr()),
                    "mt_dwconv_bwd")
        # algorithmic HBM bytes of the pass: read da, z_d (M_out x cexp each) and the dw input's pre-activation (M_in x cexp: swish'
        # for the data gradient, swish for the weight gradient), write du_in (M_in x cexp)
        need_du_in = need_below[bi] or (s.has_expand and any(need[ix["e"]:ix["e"] + 3]))
        dw_src = rec["y_in"] if rec.get("rc") else rec["dw_in"]          # what the kernels read on the input side
        # algorithmic bytes with the recompute: the block input (cin channels) instead of the expanded pre-activation
        dw_bytes = (4.0 * (s.cexp * (2 * M_out + M_in) + s.cin * M_in) if rec.get("rc") else 4.0 * s.cexp * (2 * M_out + 2 * M_in))
        if not need_du_in or not need[ix["d"]]:
            # a frozen neighbour: only the half that something trainable still needs
            if need[ix["d"]]:
                run["wgrad_launches"] += 1
                side.launch(lambda: dw_part(1), reads=(da, rec["z_d"], kabc_d, dw_src, in_bn.scale, in_bn.shift))
            if need_du_in:
                L.timed("dwconv_dgrad", lambda: dw_part(2), dw_bytes)
        elif not rec.get("rc") and (fused_dw == "1" or (fused_dw == "3" and s.k == 3)):
            run["wgrad_launches"] += 1
            # data AND weight gradient in one pass over da / z_d / the dw input (the separate weight-gradient kernel re-read all three)
            L.timed("dwconv_dgrad", lambda: dw_part(3), 4.0 * s.cexp * (2 * M_out + 2 * M_in))
        else:
            run["wgrad_launches"] += 1
            side.launch(lambda: dw_part(1), reads=(da, rec["z_d"], kabc_d, dw_src, in_bn.scale, in_bn.shift))
            L.timed("dwconv_dgrad", lambda: dw_part(2), dw_bytes)
        del da
        if not need_du_in:
            dy = None
        elif s.has_expand:
            # (h,i,j) bn0 + expand conv: z_e = y_in . We^T
            kabc_e = bn_finalize(in_bn, sums_in, ix["e"] + 1, du_in, rec["dw_in"], M_in)
            z_e = rec["z_e"]
            if z_e is None and not (expand_fused and need[ix["e"]] and need_below[bi] and M_in >= 100000
                                    and lib.mt_conv1x1_bwd_fused_supported(s.cexp, s.cin)):
                # recompute form, but this pass takes a branch that reads the expanded pre-activation (a frozen neighbour): rebuild it
                z_e = _new(dev, M_in, s.cexp)
                L.gemm(L.OP_NT, rec["y_in"], P[ix["e"]], z_e, M_in, s.cexp, s.cin, s.cin, s.cin, s.cexp)
            dy = conv1x1_bwd(du_in, z_e, kabc_e, P[ix["e"]], rec["y_in"], M_in, s.cexp, s.cin, ix["e"], need_below[bi],
                             res=dy if s.skip else None,
                             pl=dict(w_p=rec["we_p"], x_p=rec["y_p"]) if rec.get("we_p") is not None else None)
        else:
            # block 0: the dw input is the stem's activated output
            kabc0 = bn_finalize(in_bn, sums_in, 1, du_in, rec["dw_in"], M_in)
            stem = saved["stem"]
            # the dedicated kernel (LDS-staged outer products) beats the im2col-gather wgrad GEMM here (1.15 vs 2.5 ms at 256 crops):
            # with 3 input channels the gather is scalar
            run["stem_run"] = True
            if need[0]:
                run["wgrad_launches"] += 1
                L.check(lib.mt_stem_conv_wgrad(L.ptr(du_in), L.ptr(stem["z"]), L.ptr(kabc0), L.ptr(stem["x"]), 1 if stem["x"].dtype == torch.uint8 else 0, L.ptr(grads[0]), N, H, W, st),
                        "mt_stem_conv_wgrad")
            dy = None
        del du_in
        if not keep_saved:
            rec.clear()
        side.release_point()

    side.wait()
    if need_dx:
        raise NotImplementedError("gradient w.r.t. the input crops is not part of the MINTIME training path "
                                  "(train.py never sets requires_grad on videos)")
    LAST_RUN.update(run)
    if plan is not None:
        plan.extra["flat_grads"] = flat_grads
    L.grads_ready(model, params, flat_grads)
    out = [g if nd else None for nd, g in zip(need_dparams, grads)]
    return None, out
