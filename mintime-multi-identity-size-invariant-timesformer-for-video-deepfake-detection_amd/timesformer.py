"""SizeInvariantTimeSformer on libmintime_hip (MI355X).

Mirrors the reference module's Python surface -- constructor `SizeInvariantTimeSformer(config=dict,
require_attention=False)`, `forward(x, mask, identities_mask, size_embedding, positions)`,
`no_weight_decay()`, and the exact state-dict keys/shapes (reference
models/size_invariant_timesformer.py:147-276) -- while the arithmetic is a fixed sequence of HIP
kernel launches on token-major buffers:

    features [B*F*49, C] --patch-embed GEMM (+bias, rows remapped past the cls slot)--> x [B, N, 512]
    embed kernel (cls + pos_emb + size_emb gathers)
    9 x { LN -> QKV GEMM -> attention kernel (time, identity-masked) -> out-proj GEMM (+bias +residual)
          LN -> QKV GEMM -> attention kernel (space)                 -> out-proj GEMM (+bias +residual)
          LN -> FF1 GEMM with fused GEGLU epilogue -> FF2 GEMM (+bias +residual) }
    head kernel (LN + Linear on the cls row)

None of the reference's rearrange / chunk / cat / repeat / F.pad copies exist.  Submodules below are
parameter holders only (their own forward is never used).
"""
import torch
from torch import nn

from . import arch
from . import lib as L


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the HIP path is driven by SizeInvariantTimeSformer.forward")


class _Linear(_Holder):
    def __init__(self, fin, fout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout)) if bias else None


class _LayerNorm(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Embedding(_Holder):
    def __init__(self, rows, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(rows, dim))


class _Seq(_Holder):
    """Numbered children like nn.Sequential (keys `0`, `3`, ...), without being callable."""

    def __init__(self, children: dict):
        super().__init__()
        for k, m in children.items():
            self.add_module(str(k), m)


class _PreNorm(_Holder):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = _LayerNorm(dim)


class _Attention(_Holder):
    def __init__(self, dim, dim_head, heads):
        super().__init__()
        inner = dim_head * heads
        self.to_qkv = _Linear(dim, inner * 3, bias=False)
        self.to_out = _Seq({0: _Linear(inner, dim)})


class _FeedForward(_Holder):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = _Seq({0: _Linear(dim, dim * mult * 2), 3: _Linear(dim * mult, dim)})


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class SizeInvariantTimeSformer(nn.Module):
    def __init__(self, *, config, require_attention=False):
        super().__init__()
        m = config["model"]
        self.dim = m["dim"]
        self.num_frames = m["num-frames"]
        self.max_identities = m["max-identities"]
        self.image_size = m["image-size"]
        self.num_classes = m["num-classes"]
        self.patch_size = m["patch-size"]
        self.num_patches = m["num-patches"]
        self.channels = m["channels"]
        self.depth = m["depth"]
        self.heads = m["heads"]
        self.dim_head = m["dim-head"]
        self.attn_dropout = m["attn-dropout"]
        self.ff_dropout = m["ff-dropout"]
        self.shift_tokens = m["shift-tokens"]
        self.enable_size_emb = m["enable-size-emb"]
        self.enable_pos_emb = m["enable-pos-emb"]
        self.require_attention = require_attention
        if self.shift_tokens:
            # the reference itself raises NameError on this path (size_invariant_timesformer.py:189)
            raise NotImplementedError("shift-tokens: True is a dead path in the reference (NameError at :189)")
        if not (0.0 <= float(self.attn_dropout) < 1.0 and 0.0 <= float(self.ff_dropout) < 1.0):
            raise ValueError("attn-dropout / ff-dropout must be in [0, 1)")
        # dropout > 0 (not in the MINTIME configs, accepted like the reference's constructor :89-106): train-mode forwards draw the
        # multipliers with torch.rand -- or `self.dropout_uniform(shape, device)` when set (parity tests feed the reference's draws) --
        # and run on the plane path (tsf_planes.py)
        self.dropout_uniform = None
        if self.dim_head != 64:
            raise NotImplementedError("dim-head must be 64 (the attention kernels are specialised for it)")

        num_positions = self.num_frames * self.channels          # (sic) reference :172 -- keeps state-dict shapes
        self.to_patch_embedding = _Linear(self.channels, self.dim)
        self.cls_token = nn.Parameter(torch.empty(1, self.dim))
        self.pos_emb = _Embedding(num_positions + 1, self.dim)
        if self.enable_size_emb:
            self.size_emb = _Embedding(num_positions + 1, self.dim)
        self.layers = nn.ModuleList([])
        for _ in range(self.depth):
            self.layers.append(nn.ModuleList([
                _PreNorm(self.dim, _Attention(self.dim, self.dim_head, self.heads)),
                _PreNorm(self.dim, _Attention(self.dim, self.dim_head, self.heads)),
                _PreNorm(self.dim, _FeedForward(self.dim)),
            ]))
        self.to_out = _Seq({0: _LayerNorm(self.dim), 1: _Linear(self.dim, self.num_classes)})
        self.reset_parameters()

    # init equivalent to reference :200-214 (parity never relies on its RNG: weights are loaded)
    def reset_parameters(self):
        for mod in self.modules():
            if isinstance(mod, _Linear):
                _trunc_normal_(mod.weight)
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)
            elif isinstance(mod, _LayerNorm):
                nn.init.ones_(mod.weight)
                nn.init.zeros_(mod.bias)
            elif isinstance(mod, _Embedding):
                _trunc_normal_(mod.weight)
        _trunc_normal_(self.cls_token)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_emb", "cls_token", "size_emb"} if self.enable_size_emb else {"pos_emb", "cls_token"}

    # ------------------------------------------------------------------------------------------
    def _param_list(self):
        """Fixed parameter order shared by forward and backward."""
        ps = [self.to_patch_embedding.weight, self.to_patch_embedding.bias, self.cls_token, self.pos_emb.weight]
        ps.append(self.size_emb.weight if self.enable_size_emb else None)
        for (t, s, f) in self.layers:
            for a in (t, s):
                ps += [a.norm.weight, a.norm.bias, a.fn.to_qkv.weight, getattr(a.fn.to_out, "0").weight,
                       getattr(a.fn.to_out, "0").bias]
            ps += [f.norm.weight, f.norm.bias, getattr(f.fn.net, "0").weight, getattr(f.fn.net, "0").bias,
                   getattr(f.fn.net, "3").weight, getattr(f.fn.net, "3").bias]
        ps += [getattr(self.to_out, "0").weight, getattr(self.to_out, "0").bias, getattr(self.to_out, "1").weight,
               getattr(self.to_out, "1").bias]
        return ps

    def forward(self, x, mask=None, identities_mask=None, size_embedding=None, positions=None):
        from .tsf_engine import tsf_apply
        return tsf_apply(self, x, mask, identities_mask, size_embedding, positions)
