#!/usr/bin/env python3
"""Which op makes two identical eval forwards differ?  Runs every GEMM form twice and compares bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import mintime_amd
from mintime_amd import lib as L, arch, synth, SizeInvariantTimeSformer

torch.manual_seed(0)
M = 16 * 393
for (N, K, epi, name) in ((1536, 512, L.EPI_STORE, "qkv"), (512, 512, L.EPI_BIAS_RES, "outproj"), (512, 2048, L.EPI_BIAS_RES, "ff2"),
                          (4096, 512, L.EPI_GEGLU, "ff1"), (512, 1280, L.EPI_STORE, "patch")):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    outs = []
    for rep in range(4):
        R = torch.ones(M, N if epi != L.EPI_GEGLU else N // 2, device="cuda")
        C = torch.empty_like(R)
        kw = dict(bias=b)
        if epi == L.EPI_BIAS_RES:
            kw.update(R=R, ldr=N)
        if epi == L.EPI_GEGLU:
            kw.update(n_half=N // 2)
        L.gemm(L.OP_NT, A, W, C, M, N, K, K, K, C.shape[1], epilogue=epi, **kw)
        outs.append(C.clone())
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{name:8s} N={N} K={K}: bit-identical across 4 runs: {same}  maxdiff {max(float((outs[0]-o).abs().max()) for o in outs[1:]):.3e}", flush=True)

cfg = arch.default_tsf_config(1280, 8)
model = SizeInvariantTimeSformer(config=cfg)
model.load_state_dict(synth.tsf_state(cfg, 0))
model = model.cuda().eval()
for B in (2, 8, 11, 16):
    feats = synth.features(B, 8, 1280, 1).cuda()
    aux = synth.clip_inputs(B, 8, 2, 1, ragged=True, with_video=False)
    with torch.no_grad():
        o = [model(feats, mask=aux["mask"].cuda(), identities_mask=aux["identities_mask"].cuda(), size_embedding=aux["size_embedding"],
                   positions=aux["positions"].cuda()).clone() for _ in range(3)]
    print(f"model B={B}: identical {torch.equal(o[0], o[1]) and torch.equal(o[0], o[2])}  maxdiff {float((o[0]-o[1]).abs().max()):.3e}", flush=True)
