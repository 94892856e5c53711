"""Reverse launch sequence of the Size-Invariant TimeSformer (analytic backward, not torch autograd over torch ops).

Walks the layers last-to-first on the saved buffers of tsf_engine.tsf_forward:
    dgemm (NN) for activations, wgrad (TN, split-K + fp32 atomics) for weights, column sums for biases,
    the attention-core / LayerNorm / embedding / head adjoint kernels of csrc/tsf_bwd.hip.
`dx` is the running gradient of the residual stream and is updated in place.
"""
import os

import torch

from . import arch
from . import lib as L


def _zeros_like_param(p):
    return torch.zeros_like(p, dtype=torch.float32) if p is not None else None


def _splits(M):
    # enough K-splits to fill the chip for the skinny wgrad outputs, chunk kept >= 512 rows
    return max(1, min(32, M // 512))


def tsf_backward(model, feat, aux, params, dims, saved, dlogits, need_dfeat, need_dparams):
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
    sk = _splits(M)
    grads, flat_grads = L.zero_grads(list(params), with_flat=True)
    P = list(params)
    idx = len(P)

    def take(k):
        nonlocal idx
        idx -= k
        return idx

    side = L.SideStream(dev)
    # Deferred weight gradients (MT_WGRAD_DEFER=1, off by default).  Inside the TimeSformer's reverse walk the weight-gradient GEMMs
    # (9 ms of matrix work on the side stream) compete with the data-gradient GEMMs of the critical path for the same matrix cores;
    # the EfficientNet backward that follows is HBM-bound and leaves those cores idle.  With the switch on the 55 launches are only
    # RECORDED here (their operands kept: per-layer du / dqkv buffers and out-of-place residual-stream gradients instead of buffers
    # reused in place, ~4 GB of the 288) and issued after the walk on their own low-priority stream, under the extractor's
    # backward; the main stream joins that stream when the whole autograd pass has finished (engine callback).
    # MEASURED (profiles/r03_wgrad_schedule_experiments.txt): the TimeSformer backward phase drops 20.5 -> 12.9 ms and the extractor
    # backward phase grows 17.3 -> 23.6 ms -- the step is unchanged (54.4 ms), also with the late stream confined to half of the
    # CUs (hipExtStreamCreateWithCUMask) and with its HBM traffic cut by 45 %.  Concurrency only re-distributes a fixed amount of
    # work; the GPU runs the step at ~1250 W with the shader clock throttled to ~2245 of 2400 MHz.  Hence off: it would only delay
    # the TimeSformer gradient bucket of a data-parallel run.
    defer = (os.environ.get("MT_WGRAD_DEFER", "0") != "0" and need_dfeat and side.enabled and M >= 4096
             and getattr(model, "_grads_ready_hook", None) is None and not torch.cuda.is_current_stream_capturing())
    deferred = []
    wT = saved.get("wT")                              # transposed weights (tsf_engine.tsf_forward): data gradients in NT form
    if wT is not None and getattr(model, "_wT_cache", {}).get("serial") != saved.get("wT_serial"):
        wT = None                                     # a later forward rewrote the persistent buffers (two graphs alive): the weights
                                                      # as stored are still this graph's (torch's version counters guard THEM): NN form
    if wT is not None:
        side.wait(saved["wT_ready"])

    def colsum(A, lda, rows, cols, out, amap=(0, 0, 0)):
        L.check(lib.mt_colsum(L.ptr(A), lda, L.RowMap(*amap), rows, cols, L.ptr(out), L.stream_ptr()), "mt_colsum")

    # share of a long-K data gradient that is handed to the side stream (see dgrad_skinny); tuned in-step
    side_share = float(os.environ.get("MT_DGRAD_SIDE_SHARE", "0.25"))

    def dgrad_skinny(dY, Wm, out, K_, WmT=None):
        """out[M,D] = dY[M,K_] . Wm[K_,D]: only 396 output tiles -> K-slices + fp32 atomics onto a zeroed output when K is long.
        The step is bound by the main stream's kernel time while the weight-gradient stream has slack (in-step trace: main
        97 % busy, side 80-90 %), and a split-K sum does not care which stream a slice runs on: the last `side_share` of the
        contraction goes to the side stream.  Returns the side launch's event (or None); the caller waits for it before `out`
        is consumed."""
        if M >= 4096 and K_ >= 1024 and os.environ.get("MT_SKINNY_SPLIT") == "1":
            out.zero_()
            splits = 4 if K_ >= 2048 else 3                      # measured optimum with 64x64 tiles
            k_side = 0
            if side.enabled and side_share > 0 and K_ >= 2048:
                k_side = int(K_ * side_share) // 512 * 512
            k_main = K_ - k_side
            ev = None
            if k_side:
                dY_s, W_s = dY[:, k_main:], Wm[k_main:]
                ev = side.launch(lambda: L.gemm(L.OP_NN, dY_s, W_s, out, M, D, k_side, K_, D, D, epilogue=L.EPI_ATOMIC,
                                                split_k=max(1, round(splits * k_side / K_))), reads=(dY, Wm, out))
            L.gemm(L.OP_NN, dY, Wm, out, M, D, k_main, K_, D, D, epilogue=L.EPI_ATOMIC, split_k=max(1, round(splits * k_main / K_)))
            return ev
        if WmT is not None:
            L.gemm(L.OP_NT, dY, WmT, out, M, D, K_, K_, K_, D)          # WmT = Wm^T [D, K_]
        else:
            L.gemm(L.OP_NN, dY, Wm, out, M, D, K_, K_, D, D)
        return None

    def wgrad(A, Bm, out, M_, N_, K_, lda, ldb, ldc, bias_out=None, **kw):
        """dW (+ db) on the side stream: reads A [K_,M_] and Bm [K_,N_], accumulates into zero-filled grads."""
        def run():
            L.gemm(L.OP_TN, A, Bm, out, M_, N_, K_, lda, ldb, ldc, epilogue=L.EPI_ATOMIC, split_k=0, **kw)
            if bias_out is not None:
                colsum(A, lda, K_, M_, bias_out, kw.get("a_map", (0, 0, 0)))
        if defer:
            deferred.append((run, (A, Bm)))
            return None
        return side.launch(run, reads=(A, Bm))

    def ln_bwd(dxn_, r_, g_, dx_cur, i_g, tgt, skip):
        """dx = LN'(dxn) + dx_cur (+ gamma / beta gradients, + column sums for the bias below).  In place, or -- when weight
        gradients that read dx_cur are still to come -- into a fresh buffer.  With the weight-gradient stream on, only dx is
        computed on the main stream (mt_layernorm_bwd_rows: 48 VGPRs, no LDS -- it co-resides with the weight-gradient GEMMs
        instead of waiting for their blocks to end); the three column sums are parameter gradients and go to the side stream."""
        dx_new = torch.empty_like(dx_cur) if ln_oop else dx_cur
        if ln_split:
            x_, st_ = r_["x"], r_["stats"]
            L.check(lib.mt_layernorm_bwd_rows(L.ptr(dxn_), L.ptr(x_), L.ptr(st_), L.ptr(g_), L.ptr(dx_new), L.ptr(dx_cur), M, D, None, st),
                    "mt_layernorm_bwd_rows")
            side.launch(lambda: L.check(lib.mt_layernorm_bwd_cols(L.ptr(dxn_), L.ptr(x_), L.ptr(st_), L.ptr(dx_new), L.ptr(grads[i_g]),
                                                                  L.ptr(grads[i_g + 1]), L.ptr(tgt), skip, M, D, L.stream_ptr()),
                                        "mt_layernorm_bwd_cols"), reads=(dxn_, x_, st_, dx_new))
            return dx_new
        L.check(lib.mt_layernorm_bwd(L.ptr(dxn_), L.ptr(r_["x"]), L.ptr(r_["stats"]), L.ptr(g_), L.ptr(dx_new), L.ptr(grads[i_g]),
                                     L.ptr(grads[i_g + 1]), M, D, 1, L.ptr(tgt), skip, L.ptr(dx_cur) if ln_oop else None, st),
                "mt_layernorm_bwd")
        return dx_new

    # ---- head
    i0 = take(4)
    g, b_, w_h, b_h = P[i0:i0 + 4]
    pruned = bool(saved.get("pruned_last"))           # tsf_engine: the last layer ran its tail on the cls rows only
    Nh = 1 if pruned else N
    dx = torch.zeros(B, Nh, D, dtype=torch.float32, device=dev)
    L.check(lib.mt_head_bwd(L.ptr(dlogits), L.ptr(saved["x_final"]), L.ptr(g), L.ptr(b_), L.ptr(w_h), L.ptr(dx), L.ptr(grads[i0]),
                            L.ptr(grads[i0 + 1]), L.ptr(grads[i0 + 2]), L.ptr(grads[i0 + 3]), B, Nh, D, model.num_classes, eps,
                            st), "mt_head_bwd")
    dx2 = dx.view(B * Nh, D)
    # Bias gradients of the Linears whose output gradient is the residual stream's (net.3 / to_out.0 / patch embedding) are the
    # column sums of dx2 at that point.  The first one (last layer's net.3.bias) takes a column-sum launch over the head's dx;
    # every later one is emitted by the LayerNorm backward that produced that dx2 (dx_colsum), so no pass re-reads dx2 for it.
    dxn = torch.empty(M, D, dtype=torch.float32, device=dev)
    du = torch.empty(M, 8 * D, dtype=torch.float32, device=dev)
    do = torch.empty(M, inner, dtype=torch.float32, device=dev)
    dqkv = torch.empty(M, 3 * inner, dtype=torch.float32, device=dev)
    # Joins with the weight-gradient stream.  The first version joined the two streams at every sub-block start (its du / dqkv
    # buffers are reused) and waited for the sub-block's first weight gradient before the LayerNorm backward overwrote dx2 in
    # place (MT_TSF_JOIN=1 restores that).  Every cross-queue wait costs the main queue 13-19 us even when the event is long
    # complete (tools/lab/queue_sync_cost.py), and a join stalls it until the side stream has caught up: 54 of them per step.
    # Now each sub-block writes du / dqkv / dx2 into FRESH buffers (the side launches that read the old ones pin them with
    # record_stream), so the main stream waits for nothing inside the layer loop.
    join_each = os.environ.get("MT_TSF_JOIN", "0") == "1" and not defer
    ln_oop = (defer or not join_each) and side.enabled
    ln_split = ln_oop and not defer and os.environ.get("MT_LN_SPLIT", "1") != "0"     # dxn is then read on the side stream too: fresh per sub-block

    for li in reversed(range(model.depth)):
        rec = saved["layers"][li]
        if pruned and li == model.depth - 1:
            # ---- last layer, dead rows pruned: feed-forward and the space attention's tail on the B cls rows (dx2 is [B, D])
            i0 = take(6)
            g, b_, w1, b1, w2, b2 = P[i0:i0 + 6]
            r = rec[2]
            du_c = torch.empty(B, 8 * D, dtype=torch.float32, device=dev)
            dxn_c = torch.empty(B, D, dtype=torch.float32, device=dev)
            wgrad(dx2, r["h"], grads[i0 + 4], D, 4 * D, B, D, 4 * D, 4 * D, bias_out=grads[i0 + 5])
            L.gemm(L.OP_NN, dx2, w2, du_c, B, 4 * D, D, D, 4 * D, 8 * D, epilogue=L.EPI_GEGLU_BWD, C2=r["u"], ldc2=8 * D, n_half=4 * D,
                   col_sum=grads[i0 + 3])
            wgrad(du_c, r["xn"], grads[i0 + 2], 8 * D, D, B, 8 * D, D, D)
            L.gemm(L.OP_NN, du_c, w1, dxn_c, B, D, 8 * D, 8 * D, D, D)
            side.wait()
            L.check(lib.mt_layernorm_bwd(L.ptr(dxn_c), L.ptr(r["x"]), L.ptr(r["stats"]), L.ptr(g), L.ptr(dx2), L.ptr(grads[i0]),
                                         L.ptr(grads[i0 + 1]), B, D, 1, L.ptr(grads[i0 - 1]), 0, None, st), "mt_layernorm_bwd")
            r.clear()
            # space attention: out-projection on the cls rows (row stride N), the cls query's adjoint, then the full QKV tail
            i0 = take(5)
            g, b_, w_qkv, w_o, b_o = P[i0:i0 + 5]
            r = rec[1]
            wgrad(dx2, r["o"], grads[i0 + 3], D, inner, B, D, N * inner, inner)             # o's cls rows: ldb = N * inner
            L.gemm(L.OP_NN, dx2, w_o, do, B, inner, D, D, inner, N * inner)                 # row 0 of each clip's do; the rest is not read
            dqkv.zero_()                                                                    # the patch queries' gradients are zero
            L.check(lib.mt_attn_bwd(L.ptr(r["qkv"]), L.ptr(do), L.ptr(dqkv), L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, 2, scale, None, st),
                    "mt_attn_bwd")
            wgrad(dqkv, r["xn"], grads[i0 + 2], 3 * inner, D, M, 3 * inner, D, D)
            e_dg = dgrad_skinny(dqkv, w_qkv, dxn, 3 * inner, wT[(li, 7)] if wT is not None else None)
            dx_full = torch.zeros(B, N, D, dtype=torch.float32, device=dev)
            dx_full[:, 0, :] = dx2                                                          # the residual path: cls rows only
            dx2 = dx_full.view(M, D)
            if e_dg is not None:
                side.wait(e_dg)
            dx2 = ln_bwd(dxn, r, g, dx2, i0, grads[i0 - 1], 0)
            r.clear()
            # time attention: unchanged
            i0 = take(5)
            g, b_, w_qkv, w_o, b_o = P[i0:i0 + 5]
            r = rec[0]
            side.wait()
            if defer:
                dqkv = torch.empty(M, 3 * inner, dtype=torch.float32, device=dev)
            e_dx = wgrad(dx2, r["o"], grads[i0 + 3], D, inner, M, D, inner, inner)
            if wT is not None:
                L.gemm(L.OP_NT, dx2, wT[(li, 3)], do, M, inner, D, D, D, inner)
            else:
                L.gemm(L.OP_NN, dx2, w_o, do, M, inner, D, D, inner, inner)
            L.check(lib.mt_attn_bwd(L.ptr(r["qkv"]), L.ptr(do), L.ptr(dqkv), L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, 0, scale, None, st),
                    "mt_attn_bwd")
            wgrad(dqkv, r["xn"], grads[i0 + 2], 3 * inner, D, M, 3 * inner, D, D)
            e_dg = dgrad_skinny(dqkv, w_qkv, dxn, 3 * inner, wT[(li, 2)] if wT is not None else None)
            if e_dx is not None and not ln_oop:
                side.wait(e_dx)
            if e_dg is not None:
                side.wait(e_dg)
            tgt, skip = (grads[1], N) if li == 0 else (grads[i0 - 1], 0)
            dx2 = ln_bwd(dxn, r, g, dx2, i0, tgt, skip)
            r.clear()
            continue
        # ---- feed-forward: x_out = h W2^T + b2 + x ; h = a*gelu(g) ; [a|g] = LN(x) W1^T + b1
        i0 = take(6)
        g, b_, w1, b1, w2, b2 = P[i0:i0 + 6]
        r = rec[2]
        if join_each or not ln_oop:
            side.wait()                               # du / dx2 readers of the previous sub-block are done
        if ln_oop:
            du = torch.empty(M, 8 * D, dtype=torch.float32, device=dev)       # the previous one may still be read by a weight gradient
        if ln_split:
            dxn = torch.empty(M, D, dtype=torch.float32, device=dev)
        e_dx = wgrad(dx2, r["h"], grads[i0 + 4], D, 4 * D, M, D, 4 * D, 4 * D,
                     bias_out=grads[i0 + 5] if li == model.depth - 1 else None)
        if wT is not None:
            L.gemm(L.OP_NT, dx2, wT[(li, 14)], du, M, 4 * D, D, D, D, 8 * D, epilogue=L.EPI_GEGLU_BWD, C2=r["u"], ldc2=8 * D,
                   n_half=4 * D, col_sum=grads[i0 + 3])
        else:
            L.gemm(L.OP_NN, dx2, w2, du, M, 4 * D, D, D, 4 * D, 8 * D, epilogue=L.EPI_GEGLU_BWD, C2=r["u"], ldc2=8 * D, n_half=4 * D,
                   col_sum=grads[i0 + 3])             # net.0.bias gradient = column sums of du, taken in the epilogue
        wgrad(du, r["xn"], grads[i0 + 2], 8 * D, D, M, 8 * D, D, D)
        e_dg = dgrad_skinny(du, w1, dxn, 8 * D, wT[(li, 12)] if wT is not None else None)
        if e_dx is not None and not ln_oop:
            side.wait(e_dx)                           # LayerNorm backward updates dx2 in place
        if e_dg is not None:
            side.wait(e_dg)                           # ... and reads dxn, part of which the side stream summed
        dx2 = ln_bwd(dxn, r, g, dx2, i0, grads[i0 - 1], 0)                     # column sums -> space to_out.0.bias
        r.clear()
        # ---- attention blocks: x_out = o Wo^T + bo + x ; o = attn(qkv) ; qkv = LN(x) Wqkv^T
        for mode in (1, 0):
            i0 = take(5)
            g, b_, w_qkv, w_o, b_o = P[i0:i0 + 5]
            r = rec[mode]
            if join_each or not ln_oop:
                side.wait()                           # dqkv / dx2 readers of the previous sub-block are done
            if ln_oop:
                dqkv = torch.empty(M, 3 * inner, dtype=torch.float32, device=dev)    # (the old one is pinned by the launch reading it)
            if ln_split:
                dxn = torch.empty(M, D, dtype=torch.float32, device=dev)
            dq = dqkv
            e_dx = wgrad(dx2, r["o"], grads[i0 + 3], D, inner, M, D, inner, inner)
            if wT is not None:
                L.gemm(L.OP_NT, dx2, wT[(li, 8 if mode == 1 else 3)], do, M, inner, D, D, D, inner)
            else:
                L.gemm(L.OP_NN, dx2, w_o, do, M, inner, D, D, inner, inner)
            L.check(lib.mt_attn_bwd(L.ptr(r["qkv"]), L.ptr(do), L.ptr(dq), L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, mode,
                                    scale, None, st), "mt_attn_bwd")
            wgrad(dq, r["xn"], grads[i0 + 2], 3 * inner, D, M, 3 * inner, D, D)
            e_dg = dgrad_skinny(dq, w_qkv, dxn, 3 * inner, wT[(li, 7 if mode == 1 else 2)] if wT is not None else None)
            if e_dx is not None and not ln_oop:
                side.wait(e_dx)
            if e_dg is not None:
                side.wait(e_dg)
            # the updated dx2 feeds the sub-block below: time attention's to_out.0.bias (index i0 - 1), the previous layer's
            # net.3.bias (i0 - 1 as well: parameter order is ..., w2, b2 | g, b, w_qkv, w_o, b_o | ...), or -- below layer 0 --
            # the patch embedding's bias (index 1), which does not see the cls rows
            if mode == 0 and li == 0:
                tgt, skip = grads[1], N
            else:
                tgt, skip = grads[i0 - 1], 0
            dx2 = ln_bwd(dxn, r, g, dx2, i0, tgt, skip)
            r.clear()

    # ---- embeddings + patch embedding
    i0 = take(5)
    w_pe, b_pe, cls, pos_w, size_w = P[i0:i0 + 5]
    dx = dx2.view(B, N, D)                            # (a fresh buffer per sub-block when the weight gradients are deferred)
    L.check(lib.mt_embed_bwd(L.ptr(dx), L.ptr(grads[i0 + 2]), L.ptr(grads[i0 + 3]), L.ptr(grads[i0 + 4]), L.ptr(aux.positions),
                             L.ptr(aux.sizes), B, F, n, D, pos_w.shape[0], size_w.shape[0] if size_w is not None else 0, st),
            "mt_embed_bwd")
    tok_map = (F * n, N, 1)      # token row r of the feature matrix lives at row (r/(F n))*N + 1 + r%(F n) of dx
    Mt = B * F * n
    side.wait()
    wgrad(dx2, feat, grads[i0], D, C_in, Mt, D, C_in, C_in, a_map=tok_map)     # (bias: emitted by the last LayerNorm backward)
    dfeat = None
    if need_dfeat:
        dfeat = torch.empty(Mt, C_in, dtype=torch.float32, device=dev)
        L.gemm(L.OP_NN, dx2, w_pe, dfeat, Mt, C_in, D, D, C_in, C_in, a_map=tok_map)
    side.wait()
    assert idx == 0
    if deferred:
        late = L.SideStream(dev, name=":deferred-wgrad")
        main = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(main)                            # every operand of the recorded launches has been produced by now
        late.stream.wait_event(ready)
        with torch.cuda.stream(late.stream):
            for run, reads in deferred:
                for t in reads:
                    t.record_stream(late.stream)      # freed by this function's return: the allocator must not hand them out early
                run()
        for g_ in grads:
            if g_ is not None:
                g_.record_stream(late.stream)
                break                                 # (all views of one flat buffer)
        done = torch.cuda.Event()
        done.record(late.stream)
        # end of the WHOLE backward pass (after the extractor's walk): the stream autograd ran on waits for the late launches
        torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream(dev).wait_event(done))
    L.grads_ready(model, params, flat_grads)
    out = []
    for gneed, gr in zip(need_dparams, grads):
        out.append(gr if gneed else None)
    return dfeat, out
