"""Launch sequences (forward and backward) of the Size-Invariant TimeSformer on libmintime_hip.

Python here is plumbing only: it allocates device buffers through torch, passes raw pointers + the current
HIP stream to the C ABI, and wires the result into autograd with ONE torch.autograd.Function for the whole
transformer (inputs: token-major features + every parameter; the backward pass is our own reverse launch
sequence, not torch autograd over torch ops).
"""
import ctypes as C
import os

import torch

from . import arch
from . import lib as L


def _as_tokens(x):
    """[B,F,C,h,w] (any strides) -> token-major [B*F*h*w, C] contiguous.  Zero-copy when x is the NHWC-strided
    view our EfficientNet returns; otherwise one layout copy (the reference always pays it, :227)."""
    b, f, c, h, w = x.shape
    t = x.permute(0, 1, 3, 4, 2)
    if not t.is_contiguous():
        t = t.contiguous()
    return t.reshape(b * f * h * w, c)


class _Aux:
    """Device-side copies of the per-clip side inputs in the dtypes the kernels read."""

    def __init__(self, model, x, mask, identities_mask, size_embedding, positions):
        dev = x.device
        b, f = x.shape[0], x.shape[1]
        if mask is None:
            mask = torch.ones(b, f, dtype=torch.bool, device=dev)
        if identities_mask is None:
            identities_mask = torch.ones(b, f, f, dtype=torch.bool, device=dev)
        self.mask = mask.to(device=dev, dtype=torch.uint8).contiguous()
        self.ident = identities_mask.to(device=dev, dtype=torch.uint8).contiguous()
        self.sizes = None
        rows = model.pos_emb.weight.shape[0]
        if model.enable_size_emb:
            if size_embedding is not None and not size_embedding.is_cuda and size_embedding.numel():
                # the reference's callers leave it on the host (train.py:355): range-check for free, like nn.Embedding would
                lo, hi = int(size_embedding.min()), int(size_embedding.max())
                if lo < 0 or hi >= model.size_emb.weight.shape[0]:
                    raise IndexError(f"size_embedding values must lie in [0, {model.size_emb.weight.shape[0]}); got [{lo}, {hi}]")
            # arrives as a CPU int32 tensor in the reference call sites (train.py:355); moved here like :245 -- through a pinned
            # staging buffer and an async copy: a pageable H2D copy blocks the host until the stream has drained the whole
            # extractor forward, and the GPU then idles (~0.3 ms per step) while the host re-fills the launch queue
            se = size_embedding.to(dtype=torch.int32)
            if not se.is_cuda:
                se = se.contiguous().pin_memory().to(dev, non_blocking=True)
            self.sizes = se.to(device=dev).contiguous()
        self.positions = None
        if model.enable_pos_emb:
            if not positions.is_cuda and positions.numel():
                lo, hi = int(positions.min()), int(positions.max())
                if lo < 0 or hi >= rows:
                    raise IndexError(f"positions must lie in [0, {rows}); got [{lo}, {hi}]")
            self.positions = positions.to(device=dev, dtype=torch.int64).contiguous()
        # device-resident indices cannot be checked without a sync: the embedding kernels clamp them and raise a sticky flag
        # that is looked at when the NEXT forward starts (a deferred device-side assert, like nn.Embedding on a GPU)
        self.err = _index_flag(model, dev)


def _index_flag(model, dev):
    st = model.__dict__.setdefault("_index_flags", {})
    key = str(dev)
    if key in st:
        flag, host, ev = st[key]
        if ev is not None and not torch.cuda.is_current_stream_capturing() and ev.query():
            bad = int(host[0])
            if bad:
                flag.zero_()
                host.zero_()
                what = " and ".join(n for bit, n in ((1, "positions"), (2, "size_embedding")) if bad & bit)
                raise IndexError(f"an earlier SizeInvariantTimeSformer.forward received out-of-range {what} indices "
                                 "(they were clamped into the embedding tables)")
    else:
        st[key] = [torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32).pin_memory(), None]
    return st[key]


def _publish_index_flag(state):
    if torch.cuda.is_current_stream_capturing():
        return
    flag, host, _ = state
    host.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    state[2] = ev


def _new(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


def tsf_forward(model, feat, aux, params, B, F, n, save):
    """Runs the forward launch sequence.  Returns (logits, space_att, time_att, saved-dict or None)."""
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

    x = _new(dev, B, N, D)
    L.gemm(L.OP_NT, feat, w_pe, x, B * F * n, D, C_in, C_in, C_in, D, bias=b_pe, c_map=(F * n, N, 1))
    L.check(lib.mt_embed_fwd(L.ptr(x), L.ptr(cls), L.ptr(pos_w), L.ptr(size_w), L.ptr(aux.positions), L.ptr(aux.sizes),
                             B, F, n, D, pos_w.shape[0], size_w.shape[0] if size_w is not None else 0, L.ptr(aux.err[0]), st),
            "mt_embed_fwd")
    _publish_index_flag(aux.err)

    saved = {"layers": []} if save else None
    if save and M >= 4096 and L.gemm_split_enabled() and os.environ.get("MT_DGRAD_NT", "1") != "0" \
            and not torch.cuda.is_current_stream_capturing():
        # Data gradients contract over a weight's ROW index: with the weight as stored that is a k-major B operand (eight strided
        # dword loads per thread in the split loop); over the transposed weight it is the same k-contiguous NT form as the forward
        # (two float4 loads): -10 % per launch (profiles/r02_gemm_split_lab.txt).  The 36 transposes (48 M floats) run on the
        # weight-gradient stream, which is idle through the forward.
        side = L.SideStream(dev)
        wts = list(params[5:5 + 16 * model.depth])
        sel = [(li, off) for li in range(model.depth) for off in (2, 3, 7, 8, 12, 14)]   # w_qkv, w_o (time); w_qkv, w_o (space); net.0 / net.3
        key = tuple(wts[16 * li + off].data_ptr() for li, off in sel)
        cache = getattr(model, "_wT_cache", None)
        if cache is None or cache["key"] != key:
            # PERSISTENT transposed copies (192 MB at depth 9) and one device table for mt_transpose_multi: no allocation and one
            # launch per step.  (Per-step buffers under the side stream needed record_stream(main), and the caching allocator then
            # queued one event record per buffer on the main queue when they died: 0.3 ms of idle device before the extractor's
            # backward; 54 torch copy kernels cost the host ~2 ms per step.)  Rewritten at the start of every training forward: the
            # side stream waits for everything the main stream has enqueued, i.e. for the previous step's data gradients.
            holder, rows, tiles = {}, [], 0
            for li, off in sel:
                w = wts[16 * li + off]
                if w.dtype != torch.float32 or not w.is_contiguous():
                    raise L.MintimeHipError("transposed-weight cache needs contiguous fp32 Linear weights")
                t = torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=dev)
                holder[(li, off)] = t
                rows.append((w.data_ptr(), t.data_ptr(), w.shape[0], w.shape[1], tiles))
                tiles += ((w.shape[0] + 31) // 32) * ((w.shape[1] + 31) // 32)
            host = torch.tensor(rows, dtype=torch.int64).pin_memory()
            cache = dict(key=key, holder=holder, table=host.to(dev, non_blocking=True), host=host, tiles=tiles, count=len(rows))
            model._wT_cache = cache

        def transpose_all():
            L.check(lib.mt_transpose_multi(cache["table"].data_ptr(), cache["count"], cache["tiles"], L.stream_ptr()),
                    "mt_transpose_multi")
        if side.enabled:       # with MT_SIDE_STREAM=0 the transposes would sit on the critical path: the NN form is used instead
            cache["serial"] = cache.get("serial", 0) + 1       # a later forward rewrites the buffers: backward checks this
            saved["wT"], saved["wT_ready"], saved["wT_serial"] = cache["holder"], side.launch(transpose_all, reads=wts), cache["serial"]
    # Optional (MT_TSF_PRUNE_LAST=1, off by default): dead-row pruning of the LAST layer.  The classification head reads the cls
    # token only (size_invariant_timesformer.py:270-276), so everything the last layer computes for the 392 patch rows AFTER its
    # time attention is never read: the space attention's patch queries and out-projection rows and the whole feed-forward block
    # (the reference computes them and throws them away; their gradients are exactly zero).  With the switch on those rows are not
    # computed: the cls query attends to all keys as before, out-projection / LayerNorm / FF1 / FF2 run on the B cls rows.  Logits,
    # attentions and every gradient are the same numbers (tests/test_gpu_tsf.py); 7 % of the TimeSformer's FLOPs disappear.
    prune_last = os.environ.get("MT_TSF_PRUNE_LAST", "0") != "0"
    want_att = model.require_attention
    s_att = t_att = None
    xn = _new(dev, M, D)
    qkv = _new(dev, M, 3 * inner)
    o = _new(dev, M, inner)
    hbuf = _new(dev, M, 4 * D)
    for li in range(model.depth):
        last = li == model.depth - 1
        rec = {}
        for mode in (0, 1):   # 0 = time, 1 = space
            g, b_, w_qkv, w_o, b_o = next(it), next(it), next(it), next(it), next(it)
            if save:
                xn, qkv, o = _new(dev, M, D), _new(dev, M, 3 * inner), _new(dev, M, inner)
                stats = _new(dev, M, 2)
            else:
                stats = None
            L.check(lib.mt_layernorm_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(xn), L.ptr(stats), M, D, eps, None, st), "mt_layernorm_fwd")
            L.gemm(L.OP_NT, xn, w_qkv, qkv, M, 3 * inner, D, D, D, 3 * inner)
            att = None
            if want_att and last:
                att = _new(dev, B * H, 1, N)
                if mode == 0:
                    t_att = att
                else:
                    s_att = att
            if last and mode == 1 and prune_last:
                # cls query only; out-projection + residual on the B cls rows (row stride N: no gather needed)
                L.check(lib.mt_attn_fwd(L.ptr(qkv), L.ptr(o), L.ptr(att), L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, 2,
                                        scale, None, st), "mt_attn_fwd")
                x_cls = _new(dev, B, D)
                L.gemm(L.OP_NT, o, w_o, x_cls, B, D, inner, N * inner, inner, D, epilogue=L.EPI_BIAS_RES, bias=b_o, R=x, ldr=N * D)
                if save:
                    rec[mode] = dict(x=x, xn=xn, stats=stats, qkv=qkv, o=o)
                x = x_cls
                continue
            L.check(lib.mt_attn_fwd(L.ptr(qkv), L.ptr(o), L.ptr(att), L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, mode,
                                    scale, None, st), "mt_attn_fwd")
            x_new = _new(dev, B, N, D) if save else x
            L.gemm(L.OP_NT, o, w_o, x_new, M, D, inner, inner, inner, D, epilogue=L.EPI_BIAS_RES, bias=b_o, R=x, ldr=D)
            if save:
                rec[mode] = dict(x=x, xn=xn, stats=stats, qkv=qkv, o=o)
            x = x_new
        g, b_, w1, b1, w2, b2 = next(it), next(it), next(it), next(it), next(it), next(it)
        if last and prune_last:
            # feed-forward block on the cls rows only (x is [B, D] here)
            xn_c, h_c, stats_c = _new(dev, B, D), _new(dev, B, 4 * D), (_new(dev, B, 2) if save else None)
            u_c = _new(dev, B, 8 * D) if save else None
            L.check(lib.mt_layernorm_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(xn_c), L.ptr(stats_c), B, D, eps, None, st), "mt_layernorm_fwd")
            L.gemm(L.OP_NT, xn_c, w1, h_c, B, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b1, C2=u_c, ldc2=8 * D, n_half=4 * D)
            x_out = _new(dev, B, 1, D)
            L.gemm(L.OP_NT, h_c, w2, x_out, B, D, 4 * D, 4 * D, 4 * D, D, epilogue=L.EPI_BIAS_RES, bias=b2, R=x, ldr=D)
            if save:
                rec[2] = dict(x=x, xn=xn_c, stats=stats_c, u=u_c, h=h_c)
                saved["layers"].append(rec)
                saved["pruned_last"] = True
            x = x_out
            continue
        if save:
            xn, hbuf, stats = _new(dev, M, D), _new(dev, M, 4 * D), _new(dev, M, 2)
            u = _new(dev, M, 8 * D)
        else:
            stats, u = None, None
        L.check(lib.mt_layernorm_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(xn), L.ptr(stats), M, D, eps, None, st), "mt_layernorm_fwd")
        L.gemm(L.OP_NT, xn, w1, hbuf, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b1, C2=u, ldc2=8 * D, n_half=4 * D)
        # FF2 is skinny (N = 512 -> 396 output tiles on 256 CUs): 5 K-slices accumulated with fp32 atomics onto the residual
        # even out the tail (measured 333 -> 270 us at B = 32).  Training only: atomics make the sum order -- the last bits of
        # the logits -- vary from run to run, and inference (no saved buffers) stays bit-reproducible at every batch size.
        if M >= 4096 and save and os.environ.get("MT_SKINNY_SPLIT") == "1":
            x_new = x.clone() if save else x
            L.gemm(L.OP_NT, hbuf, w2, x_new, M, D, 4 * D, 4 * D, 4 * D, D, epilogue=L.EPI_ATOMIC, bias=b2, split_k=5)
        else:
            x_new = _new(dev, B, N, D) if save else x
            L.gemm(L.OP_NT, hbuf, w2, x_new, M, D, 4 * D, 4 * D, 4 * D, D, epilogue=L.EPI_BIAS_RES, bias=b2, R=x, ldr=D)
        if save:
            rec[2] = dict(x=x, xn=xn, stats=stats, u=u, h=hbuf)
            saved["layers"].append(rec)
        x = x_new
    g, b_, w_h, b_h = next(it), next(it), next(it), next(it)
    logits = _new(dev, B, model.num_classes)
    L.check(lib.mt_head_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(w_h), L.ptr(b_h), L.ptr(logits), B, x.shape[1], D, model.num_classes,
                            eps, st), "mt_head_fwd")        # x: [B, N, D], or [B, 1, D] after dead-row pruning
    if save:
        saved["x_final"] = x
    return logits, s_att, t_att, saved


class _StaticAux:
    """The per-clip side inputs of a recorded forward in static buffers (plans.py): same attributes as _Aux."""

    NAMES = ("mask", "ident", "sizes", "positions")

    def __init__(self, np_, aux):
        from . import plans
        for nm in self.NAMES:
            setattr(self, nm, plans.static_input(np_, "aux_" + nm, getattr(aux, nm)))
        self.err = aux.err

    def refresh(self, np_, aux):
        from . import plans
        for nm in self.NAMES:
            plans.refresh_input(np_, "aux_" + nm, getattr(aux, nm))
        self.err = aux.err


class _TSFFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, aux, dims, feat, *params):
        B, F, n, grad_on = dims
        dims = (B, F, n)
        # needs_input_grad mirrors requires_grad even under torch.no_grad(), and grad mode is always off inside forward():
        # the caller's grad mode comes in through `dims`.  Without it an eval forward would keep every activation and take
        # the training-only split-K + atomics branch.
        save = grad_on and any(ctx.needs_input_grad)
        from . import tsf_planes, plans
        ctx.model, ctx.dims, ctx.params = model, dims, params
        ctx.plan = ctx.token = None
        planes = tsf_planes.eligible(model, B * (1 + F * n), save)
        np_, mode = None, "eager"
        if planes and save and not tsf_planes.dropout_active(model):
            stream = torch.cuda.current_stream(feat.device).cuda_stream
            key = ("tsf", B, F, n, feat.dtype, model.training, model.require_attention, L.deterministic(),
                   tuple(ctx.needs_input_grad[3:]), stream, os.environ.get("MT_PLANES_STREAMK", "0"),
                   tuple(getattr(aux, nm) is None for nm in _StaticAux.NAMES))
            np_, mode = plans.lookup(model, key)
            if mode == "replay" and np_.state_ptrs != plans.state_ptrs(params):
                plans.drop(model, np_)
                np_, mode = None, "eager"
        if mode == "eager":
            if planes:
                logits, s_att, t_att, saved = tsf_planes.tsf_forward_planes(model, feat, aux, params, B, F, n, save)
            elif tsf_planes.dropout_active(model):
                raise NotImplementedError("attn-dropout / ff-dropout > 0 in train mode runs on the plane path only (MT_TSF_PLANES=1, "
                                          "MT_GEMM_SPLIT=1, no MT_TSF_PRUNE_LAST / MT_WGRAD_DEFER, not under stream capture)")
            else:
                logits, s_att, t_att, saved = tsf_forward(model, feat, aux, params, B, F, n, save)
            ctx.aux, ctx.saved, ctx.feat = aux, saved, feat
        elif mode == "record":
            np_.stream = stream
            feat_s = plans.static_input(np_, "feat", feat)
            aux_s = _StaticAux(np_, aux)
            pl = L.Plan()
            try:
                with pl:
                    logits, s_att, t_att, saved = tsf_planes.tsf_forward_planes(model, feat_s, aux_s, params, B, F, n, True)
            except Exception:
                np_.broken = True
                raise
            np_.fwd = pl
            np_.extra.update(saved=saved, outs=(logits, s_att, t_att), aux=aux_s, feat=feat_s)
            np_.state_ptrs = plans.state_ptrs(params)
            plans.own(np_, logits)
            plans.STATS["recorded"] += 1
        else:
            plans.refresh_input(np_, "feat", feat)
            np_.extra["aux"].refresh(np_, aux)
            np_.extra["saved"]["w_serial"] = tsf_planes.weight_planes_touch(model, params)   # the split launch is in the plan
            plans.run(np_.fwd)
            _publish_index_flag(aux.err)
            logits, s_att, t_att = np_.extra["outs"]
        if np_ is not None:
            ctx.plan, ctx.token = np_, np_.begin()
            ctx.aux, ctx.saved, ctx.feat = np_.extra["aux"], np_.extra["saved"], np_.extra["feat"]
            # the user-facing outputs are small ([B, classes], two [(B H), 1, N] maps): hand out copies, so that a caller who keeps
            # predictions on the device across steps does not see the next replay write over them (the eager path returns fresh tensors)
            logits, s_att, t_att = (None if t is None else t.clone() for t in (logits, s_att, t_att))
        outs = [logits]
        if model.require_attention:
            ctx.mark_non_differentiable(s_att, t_att)
            outs += [s_att, t_att]
        return tuple(outs)

    @staticmethod
    def backward(ctx, dlogits, *unused):
        from .tsf_backward import tsf_backward
        from . import plans
        if ctx.saved is None:
            raise RuntimeError("SizeInvariantTimeSformer: backward ran a second time through the same forward; the activation "
                               "buffers are released after the first pass (retain_graph is not supported by the HIP engine)")
        if ctx.saved.get("planes"):
            from .tsf_planes import tsf_backward_planes as tsf_backward
        np_ = ctx.plan
        dlogits = dlogits.contiguous()
        need_df, need_dp = ctx.needs_input_grad[3], ctx.needs_input_grad[4:]
        if np_ is None:
            dfeat, dparams = tsf_backward(ctx.model, ctx.feat, ctx.aux, ctx.params, ctx.dims, ctx.saved, dlogits, need_df, need_dp)
        elif (plans.grads_exist(ctx.params) or torch.cuda.current_stream(dlogits.device).cuda_stream != np_.stream
              or torch.cuda.is_current_stream_capturing()):
            plans.STATS["eager_accumulate"] += 1        # see effnet_engine._EffNetFunction.backward
            dfeat, dparams = tsf_backward(ctx.model, ctx.feat, ctx.aux, ctx.params, ctx.dims, ctx.saved, dlogits, need_df, need_dp,
                                          keep_saved=True)
        elif np_.bwd is None:
            d_s = plans.static_input(np_, "dlogits", dlogits)
            pl = L.Plan()
            try:
                with pl:
                    dfeat, dparams = tsf_backward(ctx.model, ctx.feat, ctx.aux, ctx.params, ctx.dims, ctx.saved, d_s, need_df,
                                                  need_dp, keep_saved=True, plan=np_)
            except Exception:
                np_.broken = True
                raise
            np_.bwd = pl
            np_.extra.update(grads=list(dparams), dfeat=dfeat)
            plans.own(np_, dfeat)
            dparams = plans.fresh_aliases(dparams)
            dfeat = None if dfeat is None else dfeat.detach()
        else:
            from .tsf_planes import check_weight_serial
            check_weight_serial(ctx.model, ctx.saved)
            plans.refresh_input(np_, "dlogits", dlogits)
            plans.run(np_.bwd)
            L.grads_ready(ctx.model, ctx.params, np_.extra["flat_grads"])
            dfeat = None if np_.extra["dfeat"] is None else np_.extra["dfeat"].detach()
            dparams = plans.fresh_aliases(np_.extra["grads"])
        ctx.saved = None
        if np_ is not None:
            np_.release(ctx.token)
            ctx.token = None
        return (None, None, None, dfeat) + tuple(dparams)


_CHAIN_STREAMS = {}


def _chain_stream(dev, i):
    key = (str(dev), i)
    if key not in _CHAIN_STREAMS:
        _CHAIN_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _CHAIN_STREAMS[key]


def _run_chain(model, x, mask, identities_mask, size_embedding, positions, grad_on):
    b, f, c, h, w = x.shape
    aux = _Aux(model, x, mask, identities_mask, size_embedding, positions)
    feat = _as_tokens(x.float())
    return _TSFFunction.apply(model, aux, (b, f, h * w, grad_on), feat, *model._param_list())


def tsf_apply(model, x, mask, identities_mask, size_embedding, positions):
    if not x.is_cuda:
        raise L.MintimeHipError("SizeInvariantTimeSformer (MI355X build) needs device tensors; there is no CPU path")
    b, f, c, h, w = x.shape
    if c != model.channels:
        raise ValueError(f"expected {model.channels} feature channels, got {c}")
    if f != model.num_frames:
        raise ValueError(f"expected num-frames={model.num_frames} face slots per clip, got {f}")
    if h * w != model.num_patches:
        raise ValueError(f"expected {model.num_patches} patches per slot, got {h * w}")
    grad_on = torch.is_grad_enabled()
    chains = int(os.environ.get("MT_TSF_CHAINS", "1"))
    # single-GPU experiment knob: a data-parallel reducer counts ONE grads_ready call per network and step (ddp.py), which the
    # chains would fire once each -- with a reducer installed the clips stay in one chain
    if chains > 1 and b >= 8 * chains and not torch.cuda.is_current_stream_capturing()             and getattr(model, "_grads_ready_hook", None) is None:
        # Clips are independent inside the TimeSformer (no BatchNorm): run `chains` groups of clips as concurrent launch
        # sequences on their own streams.  One sequence leaves the matrix cores idle during its LayerNorm / attention / epilogue
        # phases and the tile tails; a second one fills them (measured in-step: two co-running kernels each stretch ~1.4x, not 2x).
        main = torch.cuda.current_stream(x.device)
        from . import tsf_planes
        presplit = tsf_planes.eligible(model, (b // chains) * (1 + f * h * w), grad_on)
        if presplit:
            # the chains share the weights' operand planes: write them ONCE on the main stream, before the fork (each chain re-writing
            # them while another chain's GEMMs read them was an unsynchronised -- if value-preserving -- write)
            tsf_planes.weight_planes(model, model._param_list(), grad_on)
            model.__dict__["_wplanes_presplit"] = True
        ready = torch.cuda.Event()
        ready.record(main)
        outs = []
        try:
            for ci in range(chains):
                lo, hi = ci * b // chains, (ci + 1) * b // chains
                st = _chain_stream(x.device, ci)
                st.wait_event(ready)
                x.record_stream(st)
                with torch.cuda.stream(st):
                    sl = lambda t: None if t is None else t[lo:hi]
                    outs.append(_run_chain(model, x[lo:hi], sl(mask), sl(identities_mask), sl(size_embedding), sl(positions), grad_on))
        finally:
            model.__dict__.pop("_wplanes_presplit", None)
        for ci in range(chains):
            main.wait_stream(_chain_stream(x.device, ci))
        for o in outs:
            for t in o:
                t.record_stream(main)
        outs = tuple(torch.cat([o[k] for o in outs], dim=0) for k in range(len(outs[0])))
    else:
        outs = _run_chain(model, x, mask, identities_mask, size_embedding, positions, grad_on)
    if model.require_attention:
        return outs[0], [outs[1], outs[2]]       # order [space, time] (reference :271)
    return outs[0]
