"""Plane-operand launch sequences of the Size-Invariant TimeSformer (forward and backward).

Same arithmetic and the same launch order as tsf_engine.tsf_forward / tsf_backward.tsf_backward, with every Linear layer and its
two gradients on mt_gemm_planes (csrc/gemm_planes.hpp): the producer of each GEMM operand writes it ONCE as a blocked bf16 plane
tensor (include/mintime_hip.h) and the forward, data-gradient and weight-gradient GEMMs that consume it stream those planes by
LDS-DMA -- no operand is split inside a GEMM, none is split twice, and the weights need no transposed copies:

    forward    LayerNorm -> xn planes -> QKV GEMM -> qkv (fp32) -> attention -> o planes -> out-proj GEMM (+bias +residual) -> x
               LayerNorm -> xn planes -> FF1 GEMM + GEGLU epilogue -> h planes (+ u fp32 for the backward) -> FF2 GEMM -> x
    backward   dx planes (LayerNorm backward) feed FF2's / out-proj's data AND weight gradients; the GEGLU-backward epilogue emits
               du planes for FF1's two gradients; attention backward -> dqkv planes for the QKV layer's two gradients
    weights    one mt_split_planes_blk_multi launch per step: W planes serve the forward (rows = output features) and, read along
               their rows, the data gradients (reference: size_invariant_timesformer.py:60-76, 109-144 and their autograd)

fp32 tensors that no longer exist: xn, h, du.  Selected by tsf_engine when the split-operand pipe is on (MT_TSF_PLANES=0 keeps the
in-kernel split).
"""
import os

import torch

from . import arch
from . import lib as L

# bumped by the fused optimizers (optim.py): they update parameters through raw pointers, which torch's version counters do not see
WEIGHT_EPOCH = [0]

# LayerNorm backward: parameter-gradient column sums folded into the rows kernel as per-block partials (MT_LN_FOLD=0: the second pass
# over dy / x / dx on the weight-gradient stream)
LN_FOLD = os.environ.get("MT_LN_FOLD", "1") != "0"

_SEL = (2, 3, 7, 8, 12, 14)      # offsets of w_qkv, w_o (time); w_qkv, w_o (space); net.0 / net.3 weights inside a layer's 16 parameters


def eligible(model, M, save):
    if os.environ.get("MT_TSF_PLANES", "1") == "0" or not L.gemm_split_enabled():
        return False
    if os.environ.get("MT_TSF_PRUNE_LAST", "0") != "0" or os.environ.get("MT_WGRAD_DEFER", "0") != "0":
        return False
    if torch.cuda.is_current_stream_capturing() and save:
        return False
    D, inner = model.dim, model.heads * model.dim_head
    return (M >= 512 or dropout_active(model)) and D % 16 == 0 and inner % 16 == 0 and model.dim_head == 64


def dropout_active(model):
    """nn.Dropout is the identity in eval mode (size_invariant_timesformer.py:66-70, 98-101)."""
    return model.training and (float(model.attn_dropout) > 0.0 or float(model.ff_dropout) > 0.0)


def _dropout_mult(model, shape, p, dev):
    """keep / (1 - p) multipliers of one nn.Dropout call; the uniforms come from torch.rand or from model.dropout_uniform."""
    sampler = getattr(model, "dropout_uniform", None)
    u = sampler(shape, dev) if sampler is not None else torch.rand(shape, device=dev, dtype=torch.float32)
    return ((u.to(device=dev, dtype=torch.float32) >= p).float() / (1.0 - p)).contiguous()


def weight_planes(model, params, training):
    """Plane tensors of the 6 Linear weights of every layer, kept on the module (their storage is allocated once) and re-split from
    the fp32 weights at the start of every forward."""
    lib = L.get()
    wts = list(params[5:5 + 16 * model.depth])
    sel = [(li, off) for li in range(model.depth) for off in _SEL]
    ws = [wts[16 * li + off] for li, off in sel]
    ident = tuple(w.data_ptr() for w in ws)
    cache = getattr(model, "_wplanes_cache", None)
    dev = ws[0].device
    if cache is None or cache["ident"] != ident:
        holder, rows, first = {}, [], 0
        for (li, off), w in zip(sel, ws):
            if w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 2:
                raise L.MintimeHipError("plane path needs contiguous fp32 2-D Linear weights")
            t = L.planes_empty(w.shape[0], w.shape[1], dev)
            holder[(li, off)] = t
            rows.append((w.data_ptr(), t.data_ptr(), w.shape[0], w.shape[1], first))
            first += t.shape[1] * t.shape[2]
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        cache = dict(ident=ident, holder=holder, host=host, table=host.to(dev, non_blocking=True), blocks=first, count=len(rows),
                     stamp=None)
        model._wplanes_cache = cache
    # The planes are re-written on EVERY forward (one launch, ~60 us for the 54 matrices): updates that move no version counter
    # (p.data.copy_, dist.broadcast(p.data), a fused optimizer step through raw pointers) cannot leave them stale, and a forward
    # captured into a HIP graph (harness.GraphedEval) holds the split as a node, so its replays read the weights of the replay's time.
    stamp = (tuple(w._version for w in ws), WEIGHT_EPOCH[0])
    if model.__dict__.get("_wplanes_presplit"):      # MT_TSF_CHAINS: tsf_apply split once, before the chains' streams forked
        return cache["holder"], cache["serial"]
    L.check(lib.mt_split_planes_blk_multi(L.ptr(cache["table"]), cache["count"], cache["blocks"], L.stream_ptr()),
            "mt_split_planes_blk_multi")             # (L.ptr: a launch plan being recorded pins the table)
    if cache["stamp"] != stamp:                      # the weights changed since the planes were last written: graphs that saved the
        cache["serial"] = cache.get("serial", 0) + 1          # old serial must not run their backward on the new planes
    cache["stamp"] = stamp
    return cache["holder"], cache["serial"]


def weight_planes_touch(model, params):
    """Bookkeeping of weight_planes() for a replayed forward (the split launch itself is part of the recorded phase): a new serial
    when the weights changed since the planes were last written.  Returns the serial the replayed forward's planes carry."""
    cache = model._wplanes_cache
    wts = list(params[5:5 + 16 * model.depth])
    ws = [wts[16 * li + off] for li in range(model.depth) for off in _SEL]
    stamp = (tuple(w._version for w in ws), WEIGHT_EPOCH[0])
    if cache["stamp"] != stamp:
        cache["serial"] = cache.get("serial", 0) + 1
        cache["stamp"] = stamp
    return cache["serial"]


def check_weight_serial(model, saved):
    cache = getattr(model, "_wplanes_cache", None)
    if cache is None or cache.get("serial") != saved["w_serial"]:
        raise RuntimeError("SizeInvariantTimeSformer: the Linear weights were updated between this graph's forward and its backward "
                           "(their operand planes were rewritten by a later forward): run backward before the optimizer step")


def _new(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


def tsf_forward_planes(model, feat, aux, params, B, F, n, save):
    """Forward launch sequence on plane operands.  Returns (logits, space_att, time_att, saved-dict or None)."""
    from .tsf_engine import _publish_index_flag
    lib = L.get()
    st = L.stream_ptr()
    dev = feat.device
    D, H, dh, C_in = model.dim, model.heads, model.dim_head, model.channels
    inner = H * dh
    N = 1 + F * n
    M = B * N
    eps = arch.LN_EPS
    scale = float(dh) ** -0.5
    it = iter(params)
    w_pe, b_pe, cls, pos_w, size_w = next(it), next(it), next(it), next(it), next(it)
    wp, serial = weight_planes(model, params, save)

    x = _new(dev, B, N, D)
    L.gemm(L.OP_NT, feat, w_pe, x, B * F * n, D, C_in, C_in, C_in, D, bias=b_pe, c_map=(F * n, N, 1))
    L.check(lib.mt_embed_fwd(L.ptr(x), L.ptr(cls), L.ptr(pos_w), L.ptr(size_w), L.ptr(aux.positions), L.ptr(aux.sizes),
                             B, F, n, D, pos_w.shape[0], size_w.shape[0] if size_w is not None else 0, L.ptr(aux.err[0]), st),
            "mt_embed_fwd")
    _publish_index_flag(aux.err)

    saved = {"layers": [], "planes": True, "w_serial": serial} if save else None
    drop = dropout_active(model)
    p_att, p_ff = (float(model.attn_dropout), float(model.ff_dropout)) if drop else (0.0, 0.0)
    want_att = model.require_attention
    s_att = t_att = None
    xn_p = L.planes_empty(M, D, dev)
    o_p = L.planes_empty(M, inner, dev)
    h_p = L.planes_empty(M, 4 * D, dev)
    qkv = _new(dev, M, 3 * inner)
    for li in range(model.depth):
        last = li == model.depth - 1
        rec = {}
        for mode in (0, 1):   # 0 = time, 1 = space
            g, b_, w_qkv, w_o, b_o = next(it), next(it), next(it), next(it), next(it)
            if save:
                xn_p, o_p, qkv = L.planes_empty(M, D, dev), L.planes_empty(M, inner, dev), _new(dev, M, 3 * inner)
                stats = _new(dev, M, 2)
            else:
                stats = None
            L.check(lib.mt_layernorm_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), None, L.ptr(stats), M, D, eps, L.ptr(xn_p), st), "mt_layernorm_fwd")
            L.gemm_planes(L.OP_NT, xn_p, wp[(li, 2 if mode == 0 else 7)], M, 3 * inner, D, Cout=qkv, ldc=3 * inner)
            att = None
            if want_att and last:
                att = _new(dev, B * H, 1, N)
                if mode == 0:
                    t_att = att
                else:
                    s_att = att
            L.check(lib.mt_attn_fwd(L.ptr(qkv), None, L.ptr(att), L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, mode,
                                    scale, L.ptr(o_p), st), "mt_attn_fwd")      # o leaves the attention kernels as planes only
            x_new = _new(dev, B, N, D) if save else x
            dm = None
            if p_att > 0.0:
                # to_out = Sequential(Linear, Dropout): x + (o Wo^T + b) * m   (the residual add moves behind the multiplier)
                dm = _dropout_mult(model, (M, D), p_att, dev)
                y0 = _new(dev, M, D)
                L.gemm_planes(L.OP_NT, o_p, wp[(li, 3 if mode == 0 else 8)], M, D, inner, Cout=y0, ldc=D, bias=b_o)
                L.check(lib.mt_mul_add(L.ptr(y0), L.ptr(dm), L.ptr(x), L.ptr(x_new), M * D, st), "mt_mul_add")
            else:
                L.gemm_planes(L.OP_NT, o_p, wp[(li, 3 if mode == 0 else 8)], M, D, inner, Cout=x_new, ldc=D, epilogue=L.EPI_BIAS_RES,
                              bias=b_o, R=x, ldr=D)
            if save:
                rec[mode] = dict(x=x, stats=stats, xn_p=xn_p, qkv=qkv, o_p=o_p, dm=dm)
            x = x_new
        g, b_, w1, b1, w2, b2 = next(it), next(it), next(it), next(it), next(it), next(it)
        if save:
            xn_p, h_p, stats = L.planes_empty(M, D, dev), L.planes_empty(M, 4 * D, dev), _new(dev, M, 2)
            u = _new(dev, M, 8 * D)
        else:
            stats, u = None, None
        L.check(lib.mt_layernorm_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), None, L.ptr(stats), M, D, eps, L.ptr(xn_p), st), "mt_layernorm_fwd")
        dm = None
        if p_ff > 0.0:
            # net = Linear, GEGLU, Dropout, Linear: h leaves the GEGLU epilogue as fp32, its planes are those of h * m
            dm = _dropout_mult(model, (M, 4 * D), p_ff, dev)
            h = _new(dev, M, 4 * D)
            L.gemm_planes(L.OP_NT, xn_p, wp[(li, 12)], M, 8 * D, D, epilogue=L.EPI_GEGLU, bias=b1, C2=u, ldc2=8 * D, n_half=4 * D,
                          Cout=h, ldc=4 * D)
            L.check(lib.mt_mul_planes(L.ptr(h), L.ptr(dm), L.ptr(h_p), None, M, 4 * D, st), "mt_mul_planes")
            del h
        else:
            L.gemm_planes(L.OP_NT, xn_p, wp[(li, 12)], M, 8 * D, D, epilogue=L.EPI_GEGLU, bias=b1, C2=u, ldc2=8 * D, n_half=4 * D,
                          c_planes=h_p)
        x_new = _new(dev, B, N, D) if save else x
        L.gemm_planes(L.OP_NT, h_p, wp[(li, 14)], M, D, 4 * D, Cout=x_new, ldc=D, epilogue=L.EPI_BIAS_RES, bias=b2, R=x, ldr=D)
        if save:
            rec[2] = dict(x=x, stats=stats, xn_p=xn_p, u=u, h_p=h_p, dm=dm)
            saved["layers"].append(rec)
        x = x_new
    g, b_, w_h, b_h = next(it), next(it), next(it), next(it)
    logits = _new(dev, B, model.num_classes)
    L.check(lib.mt_head_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(w_h), L.ptr(b_h), L.ptr(logits), B, x.shape[1], D, model.num_classes,
                            eps, st), "mt_head_fwd")
    if save:
        saved["x_final"] = x
    return logits, s_att, t_att, saved


def tsf_backward_planes(model, feat, aux, params, dims, saved, dlogits, need_dfeat, need_dparams, keep_saved=False, plan=None):
    """Reverse launch sequence on plane operands: data gradients (NN, the weight planes read along their rows) on the main stream,
    weight gradients (TN, both operands read along their rows, K-range-major split-K + fp32 atomics) and the parameter-gradient sums
    of the LayerNorm backward on the weight-gradient stream.  Every sub-block writes its gradients into fresh buffers that the side
    launches pin, so the main stream waits for nothing inside the layer loop (tsf_backward.py)."""
    lib = L.get()
    st = L.stream_ptr()
    dev = feat.device
    B, F, n = dims
    D, H, dh, C_in = model.dim, model.heads, model.dim_head, model.channels
    inner = H * dh
    N = 1 + F * n
    M = B * N
    eps = arch.LN_EPS
    scale = float(dh) ** -0.5
    check_weight_serial(model, saved)
    wp = model._wplanes_cache["holder"]
    grads, flat_grads = L.zero_grads(list(params), with_flat=True)
    P = list(params)
    idx = len(P)

    def take(k):
        nonlocal idx
        idx -= k
        return idx

    side = L.SideStream(dev, hold=True)     # (eager launches; a recording plan pins what the side stream reads itself)

    def wgrad(a_p, b_p, out, M_, N_, bias_src=None, bias_out=None):
        """dW[M_, N_] += A^T B over the M token rows, on the side stream (+ an optional bias column sum of an fp32 tensor)."""
        def run():
            L.gemm_planes(L.OP_TN, a_p, b_p, M_, N_, M, Cout=out, ldc=N_, epilogue=L.EPI_ATOMIC)
            if bias_out is not None:
                L.check(lib.mt_colsum(L.ptr(bias_src), D, L.RowMap(0, 0, 0), M, D, L.ptr(bias_out), L.stream_ptr()), "mt_colsum")
        return side.launch(run, reads=(a_p, b_p, bias_src))

    def ln_bwd(dxn_, r_, g_, dx_cur, i_g, tgt, skip):
        """dx_new = LN'(dxn) + dx_cur as fp32 (the residual stream's gradient) and as planes (operand of the GEMMs below); the three
        parameter-gradient column sums on the side stream."""
        dx_new = torch.empty_like(dx_cur)
        dx_p = L.planes_empty(M, D, dev)
        x_, st_ = r_["x"], r_["stats"]
        if LN_FOLD and D <= 512:
            # the rows kernel also leaves per-block partial column sums; the weight-gradient stream only adds 6 MB of them up
            nb = lib.mt_layernorm_bwd_rows_blocks(M)
            part = _new(dev, nb, 3, D)
            L.check(lib.mt_layernorm_bwd_rows_sums(L.ptr(dxn_), L.ptr(x_), L.ptr(st_), L.ptr(g_), L.ptr(dx_new), L.ptr(dx_cur), M, D,
                                                   L.ptr(dx_p), L.ptr(part), skip, st), "mt_layernorm_bwd_rows_sums")
            side.launch(lambda: L.check(lib.mt_layernorm_bwd_cols_reduce(L.ptr(part), nb, D, L.ptr(grads[i_g]), L.ptr(grads[i_g + 1]),
                                                                         L.ptr(tgt), L.stream_ptr()), "mt_layernorm_bwd_cols_reduce"),
                        reads=(part,))
            return dx_new, dx_p
        L.check(lib.mt_layernorm_bwd_rows(L.ptr(dxn_), L.ptr(x_), L.ptr(st_), L.ptr(g_), L.ptr(dx_new), L.ptr(dx_cur), M, D,
                                          L.ptr(dx_p), st), "mt_layernorm_bwd_rows")
        side.launch(lambda: L.check(lib.mt_layernorm_bwd_cols(L.ptr(dxn_), L.ptr(x_), L.ptr(st_), L.ptr(dx_new), L.ptr(grads[i_g]),
                                                              L.ptr(grads[i_g + 1]), L.ptr(tgt), skip, M, D, L.stream_ptr()),
                                    "mt_layernorm_bwd_cols"), reads=(dxn_, x_, st_, dx_new))
        return dx_new, dx_p

    # ---- head
    i0 = take(4)
    g, b_, w_h, b_h = P[i0:i0 + 4]
    dx = L.zeros((B, N, D), torch.float32, dev)
    L.check(lib.mt_head_bwd(L.ptr(dlogits), L.ptr(saved["x_final"]), L.ptr(g), L.ptr(b_), L.ptr(w_h), L.ptr(dx), L.ptr(grads[i0]),
                            L.ptr(grads[i0 + 1]), L.ptr(grads[i0 + 2]), L.ptr(grads[i0 + 3]), B, N, D, model.num_classes, eps,
                            st), "mt_head_bwd")
    dx2 = dx.view(M, D)
    dx_p = L.split_planes_blk(dx2, M, D)
    do = torch.empty(M, inner, dtype=torch.float32, device=dev)
    dqkv = torch.empty(M, 3 * inner, dtype=torch.float32, device=dev)     # scratch of mt_attn_bwd (main stream only)

    for li in reversed(range(model.depth)):
        rec = saved["layers"][li]
        # ---- feed-forward: x_out = h W2^T + b2 + x ; h = a*gelu(g) ; [a|g] = LN(x) W1^T + b1
        i0 = take(6)
        g, b_, w1, b1, w2, b2 = P[i0:i0 + 6]
        r = rec[2]
        du_p = L.planes_empty(M, 8 * D, dev)
        dxn = torch.empty(M, D, dtype=torch.float32, device=dev)
        last = li == model.depth - 1
        wgrad(dx_p, r["h_p"], grads[i0 + 4], D, 4 * D, bias_src=dx2 if last else None, bias_out=grads[i0 + 5] if last else None)
        if r.get("dm") is not None:
            # ff-dropout: dh = (dx W2) * m, then GEGLU' as a pass (the fused epilogue has no multiplier); bias gradient = column sums of du
            dh, du_f = torch.empty(M, 4 * D, dtype=torch.float32, device=dev), torch.empty(M, 8 * D, dtype=torch.float32, device=dev)
            L.gemm_planes(L.OP_NN, dx_p, wp[(li, 14)], M, 4 * D, D, Cout=dh, ldc=4 * D)
            L.check(lib.mt_geglu_bwd(L.ptr(dh), L.ptr(r["dm"]), L.ptr(r["u"]), L.ptr(du_p), L.ptr(du_f), M, 4 * D, st), "mt_geglu_bwd")
            side.launch(lambda du_f=du_f, out=grads[i0 + 3]: L.check(lib.mt_colsum(L.ptr(du_f), 8 * D, L.RowMap(0, 0, 0), M, 8 * D,
                                                                                 L.ptr(out), L.stream_ptr()), "mt_colsum"), reads=(du_f,))
            del dh
        else:
            L.gemm_planes(L.OP_NN, dx_p, wp[(li, 14)], M, 4 * D, D, epilogue=L.EPI_GEGLU_BWD, C2=r["u"], ldc2=8 * D, n_half=4 * D,
                          col_sum=grads[i0 + 3], c_planes=du_p)      # net.0.bias gradient = column sums of du, taken in the epilogue
        wgrad(du_p, r["xn_p"], grads[i0 + 2], 8 * D, D)
        L.gemm_planes(L.OP_NN, du_p, wp[(li, 12)], M, D, 8 * D, Cout=dxn, ldc=D)
        # column sums of the new dx -> space to_out.0.bias -- unless that projection sits under a dropout (its own sub-block sums the
        # masked gradient then)
        dx2, dx_p = ln_bwd(dxn, r, g, dx2, i0, None if rec[1].get("dm") is not None else grads[i0 - 1], 0)
        if not keep_saved:
            r.clear()
        # ---- attention blocks: x_out = o Wo^T + bo + x ; o = attn(qkv) ; qkv = LN(x) Wqkv^T
        for mode in (1, 0):
            i0 = take(5)
            g, b_, w_qkv, w_o, b_o = P[i0:i0 + 5]
            r = rec[mode]
            dxn = torch.empty(M, D, dtype=torch.float32, device=dev)
            dy_p = dx_p                                  # gradient w.r.t. the projection's output
            if r.get("dm") is not None:
                # attn-dropout: d(o Wo^T + b) = dx * m -- as planes for both GEMMs, as fp32 for the bias gradient's column sums
                dy_p = L.planes_empty(M, D, dev)
                dy_f = torch.empty(M, D, dtype=torch.float32, device=dev)
                L.check(lib.mt_mul_planes(L.ptr(dx2), L.ptr(r["dm"]), L.ptr(dy_p), L.ptr(dy_f), M, D, st), "mt_mul_planes")
                side.launch(lambda dy_f=dy_f, out=grads[i0 + 4]: L.check(lib.mt_colsum(L.ptr(dy_f), D, L.RowMap(0, 0, 0), M, D, L.ptr(out),
                                                                                     L.stream_ptr()), "mt_colsum"), reads=(dy_f,))
            wgrad(dy_p, r["o_p"], grads[i0 + 3], D, inner)
            L.gemm_planes(L.OP_NN, dy_p, wp[(li, 8 if mode == 1 else 3)], M, inner, D, Cout=do, ldc=inner)
            dqkv_p = L.planes_empty(M, 3 * inner, dev)
            L.check(lib.mt_attn_bwd(L.ptr(r["qkv"]), L.ptr(do), L.ptr(dqkv), L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, mode,
                                    scale, L.ptr(dqkv_p), st), "mt_attn_bwd")    # dqkv (fp32) is working memory here
            wgrad(dqkv_p, r["xn_p"], grads[i0 + 2], 3 * inner, D)
            L.gemm_planes(L.OP_NN, dqkv_p, wp[(li, 7 if mode == 1 else 2)], M, D, 3 * inner, Cout=dxn, ldc=D)
            # the updated dx feeds the sub-block below: time attention's to_out.0.bias (index i0 - 1), the previous layer's
            # net.3.bias (i0 - 1 as well), or -- below layer 0 -- the patch embedding's bias (index 1), which skips the cls rows
            if mode == 0 and li == 0:
                tgt, skip = grads[1], N
            elif mode == 1 and rec[0].get("dm") is not None:
                tgt, skip = None, 0                      # time to_out.0.bias sits under a dropout: summed in its own sub-block
            else:
                tgt, skip = grads[i0 - 1], 0
            dx2, dx_p = ln_bwd(dxn, r, g, dx2, i0, tgt, skip)
            if not keep_saved:
                r.clear()
        side.release_point()

    # ---- embeddings + patch embedding (row-mapped operands: the fp32 GEMM family)
    i0 = take(5)
    w_pe, b_pe, cls, pos_w, size_w = P[i0:i0 + 5]
    dx = dx2.view(B, N, D)
    L.check(lib.mt_embed_bwd(L.ptr(dx), L.ptr(grads[i0 + 2]), L.ptr(grads[i0 + 3]), L.ptr(grads[i0 + 4]), L.ptr(aux.positions),
                             L.ptr(aux.sizes), B, F, n, D, pos_w.shape[0], size_w.shape[0] if size_w is not None else 0, st),
            "mt_embed_bwd")
    tok_map = (F * n, N, 1)
    Mt = B * F * n
    side.launch(lambda: L.gemm(L.OP_TN, dx2, feat, grads[i0], D, C_in, Mt, D, C_in, C_in, epilogue=L.EPI_ATOMIC, split_k=0, a_map=tok_map),
                reads=(dx2, feat))
    dfeat = None
    if need_dfeat:
        dfeat = torch.empty(Mt, C_in, dtype=torch.float32, device=dev)
        L.gemm(L.OP_NN, dx2, w_pe, dfeat, Mt, C_in, D, D, C_in, C_in, a_map=tok_map)
    side.wait()
    assert idx == 0
    if plan is not None:
        plan.extra["flat_grads"] = flat_grads
    L.grads_ready(model, params, flat_grads)
    return dfeat, [gr if gneed else None for gneed, gr in zip(need_dparams, grads)]
