"""Static architecture tables for the MINTIME hot path.

EfficientNet-B0 stage table: restated from the block strings at
reference models/efficientnet/efficientnet_pytorch/utils.py:502-510 and the
construction loop at model.py:178-196 (width/depth multiplier 1.0, image 224).
TF-"SAME" padding rule: utils.py:248-276 (pad_total = max((ceil(i/s)-1)*s + k - i, 0),
before = pad_total // 2, after = rest).

Everything here is plain data; both the HIP host code and the CPU oracle read it.
"""
from dataclasses import dataclass
from typing import List

BN_EPS_EFFNET = 1e-3          # utils.py:521
BN_MOMENTUM_EFFNET = 0.01     # 1 - 0.99, model.py:51 / utils.py:520
DROP_CONNECT_RATE = 0.2       # utils.py:522
LN_EPS = 1e-5                 # nn.LayerNorm default, size_invariant_timesformer.py:22


def same_pad(i: int, k: int, s: int):
    """(before, after) zero padding of TF-SAME for input size i, kernel k, stride s."""
    o = -(-i // s)
    total = max((o - 1) * s + k - i, 0)
    return total // 2, total - total // 2


@dataclass(frozen=True)
class MBConv:
    idx: int
    k: int          # depthwise kernel
    s: int          # depthwise stride
    e: int          # expand ratio
    cin: int
    cout: int
    cexp: int       # cin * e
    cse: int        # squeeze width = max(1, int(cin * 0.25))  (model.py:78)
    hin: int
    hout: int
    pad0: int       # TF-SAME pad before (top / left)
    pad1: int       # TF-SAME pad after (bottom / right)
    skip: bool      # residual (model.py:123): stride is the int 1 and cin == cout

    @property
    def has_expand(self):
        return self.e != 1


_STAGES = [  # (repeats, k, s, e, cin, cout)
    (1, 3, 1, 1, 32, 16),
    (2, 3, 2, 6, 16, 24),
    (2, 5, 2, 6, 24, 40),
    (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112),
    (4, 5, 2, 6, 112, 192),
    (1, 3, 1, 6, 192, 320),
]

STEM_CIN, STEM_COUT, STEM_K, STEM_S = 3, 32, 3, 2
HEAD_CIN, HEAD_COUT = 320, 1280
IMAGE_SIZE = 224


def effnet_b0_blocks(image_size: int = IMAGE_SIZE) -> List[MBConv]:
    h = -(-image_size // STEM_S)
    out = []
    idx = 0
    for (r, k, s, e, cin, cout) in _STAGES:
        for j in range(r):
            bs = s if j == 0 else 1
            bcin = cin if j == 0 else cout
            hout = -(-h // bs)
            p0, p1 = same_pad(h, k, bs)
            # first block of a stage carries stride as a list in the reference
            # (utils.py:394) so `stride == 1` is False there even when s == 1
            skip = (j > 0) and bcin == cout
            out.append(MBConv(idx, k, bs, e, bcin, cout, bcin * e, max(1, int(bcin * 0.25)),
                              h, hout, p0, p1, skip))
            h = hout
            idx += 1
    return out


def effnet_b0_state_spec(include_top: bool = True, num_classes: int = 1000):
    """Ordered (key, shape, kind) list matching the reference state_dict (360 entries).

    kind in {stem, expand, dw, se_r_w, se_r_b, se_e_w, se_e_b, project, head,
             bn_w, bn_b, bn_rm, bn_rv, bn_nbt, fc_w, fc_b}
    """
    spec = []

    def bn(prefix, c):
        spec.extend([(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
                     (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
                     (prefix + ".num_batches_tracked", (), "bn_nbt")])

    spec.append(("_conv_stem.weight", (STEM_COUT, STEM_CIN, STEM_K, STEM_K), "stem"))
    bn("_bn0", STEM_COUT)
    for b in effnet_b0_blocks():
        p = f"_blocks.{b.idx}."
        if b.has_expand:
            spec.append((p + "_expand_conv.weight", (b.cexp, b.cin, 1, 1), "expand"))
            bn(p + "_bn0", b.cexp)
        spec.append((p + "_depthwise_conv.weight", (b.cexp, 1, b.k, b.k), "dw"))
        bn(p + "_bn1", b.cexp)
        spec.append((p + "_se_reduce.weight", (b.cse, b.cexp, 1, 1), "se_r_w"))
        spec.append((p + "_se_reduce.bias", (b.cse,), "se_r_b"))
        spec.append((p + "_se_expand.weight", (b.cexp, b.cse, 1, 1), "se_e_w"))
        spec.append((p + "_se_expand.bias", (b.cexp,), "se_e_b"))
        spec.append((p + "_project_conv.weight", (b.cout, b.cexp, 1, 1), "project"))
        bn(p + "_bn2", b.cout)
    spec.append(("_conv_head.weight", (HEAD_COUT, HEAD_CIN, 1, 1), "head"))
    bn("_bn1", HEAD_COUT)
    if include_top:
        spec.append(("_fc.weight", (num_classes, HEAD_COUT), "fc_w"))
        spec.append(("_fc.bias", (num_classes,), "fc_b"))
    return spec


# ---------------------------------------------------------------------------------------------
# Size-Invariant TimeSformer
# ---------------------------------------------------------------------------------------------

def default_tsf_config(channels: int = 1280, num_frames: int = 8):
    """The shipped config/size_invariant_timesformer.yaml:15-32 with the two overrides the
    EfficientNet variant needs (channels 1280, num-frames 8)."""
    return {
        "model": {
            "image-size": 224, "patch-size": 1, "num-classes": 1, "num-patches": 49,
            "num-frames": num_frames, "max-identities": 2, "dim": 512, "depth": 9,
            "dim-head": 64, "channels": channels, "heads": 8, "attn-dropout": 0.0,
            "ff-dropout": 0.0, "shift-tokens": False, "enable-size-emb": True,
            "enable-pos-emb": True, "enable-identity-attention": True,
        },
        "training": {"lr": 0.01, "weight-decay": 0.0001, "bs": 8, "val_bs": 8, "optimizer": "SGD",
                     "scheduler": "cosinelr", "gamma": 0.1, "step-size": 5, "augmentation": "max"},
        "test": {"bs": 1},
    }


def tsf_state_spec(cfg):
    """Ordered (key, shape, kind) for SizeInvariantTimeSformer (size_invariant_timesformer.py:172-198).

    kind in {lin_w, lin_b, qkv_w, ln_w, ln_b, emb, cls}
    """
    m = cfg["model"]
    dim, C, F, depth = m["dim"], m["channels"], m["num-frames"], m["depth"]
    inner = m["heads"] * m["dim-head"]
    npos = F * C + 1
    spec = [("cls_token", (1, dim), "cls"),
            ("to_patch_embedding.weight", (dim, C), "lin_w"),
            ("to_patch_embedding.bias", (dim,), "lin_b"),
            ("pos_emb.weight", (npos, dim), "emb")]
    if m["enable-size-emb"]:
        spec.append(("size_emb.weight", (npos, dim), "emb"))
    for i in range(depth):
        for j in (0, 1):  # 0 = time attention, 1 = spatial attention
            p = f"layers.{i}.{j}."
            spec.append((p + "fn.to_qkv.weight", (3 * inner, dim), "qkv_w"))
            spec.append((p + "fn.to_out.0.weight", (dim, inner), "lin_w"))
            spec.append((p + "fn.to_out.0.bias", (dim,), "lin_b"))
            spec.append((p + "norm.weight", (dim,), "ln_w"))
            spec.append((p + "norm.bias", (dim,), "ln_b"))
        p = f"layers.{i}.2."
        spec.append((p + "fn.net.0.weight", (dim * 8, dim), "lin_w"))
        spec.append((p + "fn.net.0.bias", (dim * 8,), "lin_b"))
        spec.append((p + "fn.net.3.weight", (dim, dim * 4), "lin_w"))
        spec.append((p + "fn.net.3.bias", (dim,), "lin_b"))
        spec.append((p + "norm.weight", (dim,), "ln_w"))
        spec.append((p + "norm.bias", (dim,), "ln_b"))
    spec.append(("to_out.0.weight", (dim,), "ln_w"))
    spec.append(("to_out.0.bias", (dim,), "ln_b"))
    spec.append(("to_out.1.weight", (m["num-classes"], dim), "lin_w"))
    spec.append(("to_out.1.bias", (m["num-classes"],), "lin_b"))
    return spec


# ---------------------------------------------------------------------------------------------
# Xception (config 5 extractor; reference models/xception.py:82-217)
# ---------------------------------------------------------------------------------------------
BN_EPS_XCEPTION = 1e-5        # nn.BatchNorm2d default (xception.py:97)
BN_MOMENTUM_XCEPTION = 0.1

XCEPTION_BLOCKS = [  # (name, cin, cout, reps, stride, start_with_relu, grow_first)   xception.py:113-129
    ("block1", 64, 128, 2, 2, False, True), ("block2", 128, 256, 2, 2, True, True), ("block3", 256, 728, 2, 2, True, True),
    *[(f"block{i}", 728, 728, 3, 1, True, True) for i in range(4, 12)],
    ("block12", 728, 1024, 2, 2, True, False)]


def xception_block_units(cin, cout, reps, grow_first):
    """(cin, cout) of a Block's separable convs in order (xception.py:44-58)."""
    units, filters = [], cin
    if grow_first:
        units.append((cin, cout))
        filters = cout
    units += [(filters, filters)] * (reps - 1)
    if not grow_first:
        units.append((cin, cout))
    return units


def xception_unit_keys(name, start_with_relu, n_units):
    """(separable-conv prefix, bn prefix) per unit: nn.Sequential indices of Block.rep (xception.py:44-66)."""
    out, idx = [], 0
    for u in range(n_units):
        if u > 0 or start_with_relu:
            idx += 1
        out.append((f"{name}.rep.{idx}", f"{name}.rep.{idx + 1}"))
        idx += 2
    return out


def xception_state_spec(num_classes: int = 1):
    """Ordered (key, shape, kind) matching the reference Xception state_dict (276 entries)."""
    spec = []

    def bn(prefix, c):
        spec.extend([(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
                     (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
                     (prefix + ".num_batches_tracked", (), "bn_nbt")])

    spec.append(("conv1.weight", (32, 3, 3, 3), "conv_first"))
    bn("bn1", 32)
    spec.append(("conv2.weight", (64, 32, 3, 3), "conv"))
    bn("bn2", 64)
    for (name, cin, cout, reps, stride, srelu, grow) in XCEPTION_BLOCKS:
        if cout != cin or stride != 1:
            spec.append((name + ".skip.weight", (cout, cin, 1, 1), "pw"))
            bn(name + ".skipbn", cout)
        units = xception_block_units(cin, cout, reps, grow)
        for (sep, bnp), (ci, co) in zip(xception_unit_keys(name, srelu, len(units)), units):
            spec.append((sep + ".conv1.weight", (ci, 1, 3, 3), "dw"))
            spec.append((sep + ".pointwise.weight", (co, ci, 1, 1), "pw"))
            bn(bnp, co)
    for (sep, bnp, ci, co) in (("conv3", "bn3", 1024, 1536), ("conv4", "bn4", 1536, 2048)):
        spec.append((sep + ".conv1.weight", (ci, 1, 3, 3), "dw"))
        spec.append((sep + ".pointwise.weight", (co, ci, 1, 1), "pw"))
        bn(bnp, co)
    spec.append(("fc.weight", (num_classes, 2048), "fc_w"))
    spec.append(("fc.bias", (num_classes,), "fc_b"))
    return spec
