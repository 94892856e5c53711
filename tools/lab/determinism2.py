#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import mintime_amd
from mintime_amd import lib as L, arch, synth, tsf_engine

torch.manual_seed(0)
lib = L.get()
for B in (11, 16):
    M = B * 393
    print("B", B, "M", M)
    for (N, K, epi, name, inplace) in ((1536, 512, L.EPI_STORE, "qkv", False), (512, 512, L.EPI_BIAS_RES, "outproj", True),
                                       (512, 2048, L.EPI_BIAS_RES, "ff2", True), (4096, 512, L.EPI_GEGLU, "ff1", False)):
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
        outs = []
        for rep in range(4):
            C = torch.ones(M, N if epi != L.EPI_GEGLU else N // 2, device="cuda")
            kw = dict(bias=b)
            if epi == L.EPI_BIAS_RES:
                kw.update(R=C, ldr=N)
            if epi == L.EPI_GEGLU:
                kw.update(n_half=N // 2)
            L.gemm(L.OP_NT, A, W, C, M, N, K, K, K, C.shape[1], epilogue=epi, **kw)
            outs.append(C.clone())
        print(f"  {name:8s}: identical {all(torch.equal(outs[0], o) for o in outs[1:])}", flush=True)
    # patch embedding with row map + embed + LN + attention
    Fr, n, D, H = 8, 49, 512, 8
    feat = torch.randn(B * Fr * n, 1280, device="cuda"); w = torch.randn(D, 1280, device="cuda") * 0.02; b = torch.randn(D, device="cuda")
    aux = synth.clip_inputs(B, Fr, 2, 1, ragged=True, with_video=False)
    mask = aux["mask"].cuda().to(torch.uint8).contiguous(); ident = aux["identities_mask"].cuda().to(torch.uint8).contiguous()
    res = {k: [] for k in ("patch", "ln", "attn_t", "attn_s", "cls_att")}
    g, be = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    qkv = torch.randn(M, 3 * D, device="cuda")
    for rep in range(4):
        x = torch.empty(B, 393, D, device="cuda").fill_(float(rep))      # different garbage each time: the cls rows are not written by the GEMM
        L.gemm(L.OP_NT, feat, w, x, B * Fr * n, D, 1280, 1280, 1280, D, bias=b, c_map=(Fr * n, 393, 1))
        res["patch"].append(x[:, 1:].clone())
        xin = torch.randn(M, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        xn = torch.empty(M, D, device="cuda")
        L.check(lib.mt_layernorm_fwd(L.ptr(xin), L.ptr(g), L.ptr(be), L.ptr(xn), None, M, D, 1e-5, L.stream_ptr()), "ln")
        res["ln"].append(xn.clone())
        for mode, key in ((0, "attn_t"), (1, "attn_s")):
            o = torch.empty(M, D, device="cuda")
            att = torch.empty(B * H, 1, 393, device="cuda")
            L.check(lib.mt_attn_fwd(L.ptr(qkv), L.ptr(o), L.ptr(att), L.ptr(mask), L.ptr(ident), B, H, Fr, n, mode, 0.125, L.stream_ptr()), "attn")
            res[key].append(o.clone())
            if mode == 0:
                res["cls_att"].append(att.clone())
    for k, v in res.items():
        print(f"  {k:8s}: identical {all(torch.equal(v[0], o) for o in v[1:])}", flush=True)
