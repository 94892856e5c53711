#!/usr/bin/env python3
"""In-model bisect: the eval launch sequence of tsf_engine.tsf_forward with a bit-hash of every op's output, three runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import mintime_amd
from mintime_amd import lib as L, arch, synth, tsf_engine, SizeInvariantTimeSformer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 11
cfg = arch.default_tsf_config(1280, 8)
model = SizeInvariantTimeSformer(config=cfg)
model.load_state_dict(synth.tsf_state(cfg, 0))
model = model.cuda().eval()
feats = synth.features(B, 8, 1280, 1).cuda()
a = synth.clip_inputs(B, 8, 2, 1, ragged=True, with_video=False)
aux = tsf_engine._Aux(model, feats, a["mask"].cuda(), a["identities_mask"].cuda(), a["size_embedding"], a["positions"].cuda())
feat = tsf_engine._as_tokens(feats.float())
params = model._param_list()


import re
SYNC = os.environ.get("SYNC_AFTER", ".*")
_pending = []


def h(t):
    return t


class _Log(list):
    def append(self, item):
        name, t = item
        if re.search(SYNC, name):
            torch.cuda.synchronize()
        if name.endswith("ff2") and name.startswith("L8"):
            list.append(self, (name, int(t.contiguous().view(torch.int32).to(torch.int64).sum())))


INC = os.environ.get("INCLUDE", "")


def run():
    global feat, aux
    if "tokens" in INC:
        feat = tsf_engine._as_tokens(feats.float())
    if "aux" in INC:
        aux = tsf_engine._Aux(model, feats, a["mask"].cuda(), a["identities_mask"].cuda(), a["size_embedding"], a["positions"].cuda())
    lib = L.get(); st = L.stream_ptr(); dev = feat.device
    F, n = 8, 49
    D, H, dh, C_in = model.dim, model.heads, model.dim_head, model.channels
    inner = H * dh; N = 1 + F * n; M = B * N; eps = arch.LN_EPS; scale = float(dh) ** -0.5
    it = iter(params)
    w_pe, b_pe, cls, pos_w, size_w = next(it), next(it), next(it), next(it), next(it)
    log = _Log()
    x = torch.empty(B, N, D, device=dev)
    L.gemm(L.OP_NT, feat, w_pe, x, B * F * n, D, C_in, C_in, C_in, D, bias=b_pe, c_map=(F * n, N, 1))
    log.append(("patch", h(x[:, 1:])))
    L.check(lib.mt_embed_fwd(L.ptr(x), L.ptr(cls), L.ptr(pos_w), L.ptr(size_w), L.ptr(aux.positions), L.ptr(aux.sizes), B, F, n, D,
                             pos_w.shape[0], size_w.shape[0], None, st), "embed")
    log.append(("embed", h(x)))
    xn = torch.empty(M, D, device=dev); qkv = torch.empty(M, 3 * inner, device=dev); o = torch.empty(M, inner, device=dev)
    hbuf = torch.empty(M, 4 * D, device=dev)
    for li in range(model.depth):
        for mode in (0, 1):
            g, b_, w_qkv, w_o, b_o = next(it), next(it), next(it), next(it), next(it)
            L.check(lib.mt_layernorm_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(xn), None, M, D, eps, st), "ln")
            log.append((f"L{li}.{mode}.ln", h(xn)))
            L.gemm(L.OP_NT, xn, w_qkv, qkv, M, 3 * inner, D, D, D, 3 * inner)
            log.append((f"L{li}.{mode}.qkv", h(qkv)))
            L.check(lib.mt_attn_fwd(L.ptr(qkv), L.ptr(o), None, L.ptr(aux.mask), L.ptr(aux.ident), B, H, F, n, mode, scale, st), "attn")
            log.append((f"L{li}.{mode}.attn", h(o)))
            L.gemm(L.OP_NT, o, w_o, x, M, D, inner, inner, inner, D, epilogue=L.EPI_BIAS_RES, bias=b_o, R=x, ldr=D)
            log.append((f"L{li}.{mode}.out", h(x)))
        g, b_, w1, b1, w2, b2 = next(it), next(it), next(it), next(it), next(it), next(it)
        L.check(lib.mt_layernorm_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(xn), None, M, D, eps, st), "ln")
        log.append((f"L{li}.ff.ln", h(xn)))
        L.gemm(L.OP_NT, xn, w1, hbuf, M, 8 * D, D, D, D, 4 * D, epilogue=L.EPI_GEGLU, bias=b1, n_half=4 * D)
        log.append((f"L{li}.ff1", h(hbuf)))
        L.gemm(L.OP_NT, hbuf, w2, x, M, D, 4 * D, 4 * D, 4 * D, D, epilogue=L.EPI_BIAS_RES, bias=b2, R=x, ldr=D)
        log.append((f"L{li}.ff2", h(x)))
    if "head" in INC:
        g, b_, w_h, b_h = next(it), next(it), next(it), next(it)
        logits = torch.empty(B, 1, device=dev)
        L.check(lib.mt_head_fwd(L.ptr(x), L.ptr(g), L.ptr(b_), L.ptr(w_h), L.ptr(b_h), L.ptr(logits), B, N, D, 1, eps, st), "head")
        list.append(log, ("head", int(logits.contiguous().view(torch.int32).to(torch.int64).sum())))
    if "model" in INC:
        with torch.no_grad():
            y = model(feats, mask=a["mask"].cuda(), identities_mask=a["identities_mask"].cuda(), size_embedding=a["size_embedding"],
                      positions=a["positions"].cuda())
        list.append(log, ("model", int(y.contiguous().view(torch.int32).to(torch.int64).sum())))
    return log


logs = [run() for _ in range(3)]
first = None
for i, (name, v) in enumerate(logs[0]):
    if any(l[i][1] != v for l in logs[1:]):
        first = (i, name, [l[i][1] for l in logs])
        break
print([[v for _, v in l] for l in logs])
print(f"B={B}: first differing op: {first}")
if first:
    i = first[0]
    print("ops before:", [n for n, _ in logs[0][max(0, i - 3):i]])
