"""CPU ORACLE for the MINTIME hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-PyTorch-CPU (fp32 or fp64) functional restatement of the reference's forward for
  * EfficientNet-B0 feature extractor (reference models/efficientnet/efficientnet_pytorch/model.py:267-288,
    MBConvBlock.forward model.py:89-128, TF-SAME conv utils.py:248-276, swish utils.py:64-80,
    drop_connect utils.py:129-154)
  * SizeInvariantTimeSformer (reference models/size_invariant_timesformer.py:224-276, Attention.forward
    :109-144, attn() :80-87, GEGLU/FeedForward :60-76, PreNorm :18-26)
written from the op graph, operating on a state-dict with the reference's keys.  Backward comes from
torch autograd over these same ops.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
path (the HIP library and its Python host) never does and has no CPU fallback.

PARITY PIN: the reference holds no golden vectors or numerical tests for this path, so
this oracle is pinned against outputs of the reference itself, generated in the build container by
tools/make_golden.py (imports /root/reference, loads the same seeded state-dicts) and committed as
tests/golden/*.npz; tests/test_oracle_golden.py checks every one of them.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

FLT_MAX = torch.finfo(torch.float32).max


# --------------------------------------------------------------------------------------------
# EfficientNet-B0
# --------------------------------------------------------------------------------------------
_B0_STAGES = [  # (repeats, k, s, e, cin, cout)   utils.py:502-510
    (1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
_BN_EPS = 1e-3      # utils.py:521
_BN_MOM = 0.01      # model.py:51
_DROP_CONNECT = 0.2  # utils.py:522


def _b0_blocks():
    out = []
    for (r, k, s, e, cin, cout) in _B0_STAGES:
        for j in range(r):
            out.append(dict(k=k, s=s if j == 0 else 1, e=e, cin=cin if j == 0 else cout, cout=cout,
                            skip=(j > 0)))   # model.py:123 (+ utils.py:394: stage-first stride is a list)
    return out


def _same_conv(x, w, stride, groups=1):
    """TF-SAME zero padding then conv with padding 0 (utils.py:248-276)."""
    ih, iw = x.shape[-2:]
    kh, kw = w.shape[-2:]
    oh, ow = -(-ih // stride), -(-iw // stride)
    ph = max((oh - 1) * stride + kh - ih, 0)
    pw = max((ow - 1) * stride + kw - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    return F.conv2d(x, w, None, stride, 0, 1, groups)


def _swish(x):
    return x * torch.sigmoid(x)   # utils.py:67


class BNState:
    """Collects train-mode running-stat updates (the reference updates module buffers in place)."""

    def __init__(self):
        self.updates = {}


def _bn(x, sd, prefix, training, bn_state: Optional[BNState]):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, _BN_MOM, _BN_EPS)
    rm2, rv2 = rm.detach().clone(), rv.detach().clone()
    y = F.batch_norm(x, rm2, rv2, w, b, True, _BN_MOM, _BN_EPS)
    if bn_state is not None:
        bn_state.updates[prefix + ".running_mean"] = rm2
        bn_state.updates[prefix + ".running_var"] = rv2
    return y


def drop_connect_uniforms(seed: int, n_img: int, drop_connect_rate: float = _DROP_CONNECT):
    """The U[0,1) draws the reference's drop_connect makes in one train-mode forward after torch.manual_seed(seed):
    one torch.rand([N,1,1,1], fp32) per block that has a skip connection and a non-zero rate, in block order
    (model.py:280-282 scales the rate by idx/16; `if drop_connect_rate:` skips the call when it is 0; utils.py:148-150)."""
    torch.manual_seed(seed)
    out = {}
    for i, b in enumerate(_b0_blocks()):
        if b["skip"] and b["cin"] == b["cout"] and drop_connect_rate * float(i) / len(_b0_blocks()):
            out[i] = torch.rand([n_img, 1, 1, 1], dtype=torch.float32)
    return out


def effnet_b0_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, training: bool = False,
                      drop_connect_rate: float = 0.0, bn_state: Optional[BNState] = None,
                      taps: Optional[dict] = None, dc_uniform: Optional[dict] = None):
    """x [N,3,224,224] (any strides) -> features [N,1280,7,7].  model.py:267-288.

    training=True uses batch statistics in every BN (train.py:157).  With drop_connect_rate > 0 (train mode only) the
    per-sample Bernoulli gate of utils.py:129-154 is applied from the uniform draws in dc_uniform ({block index: [N,1,1,1]},
    see drop_connect_uniforms) so that a parity run can feed the same draws to both sides.
    """
    assert drop_connect_rate == 0.0 or not training or dc_uniform is not None, \
        "train-mode drop-connect needs the uniform draws (dc_uniform) to be reproducible"
    x = _swish(_bn(_same_conv(x, sd["_conv_stem.weight"], 2), sd, "_bn0", training, bn_state))  # model.py:276
    if taps is not None:
        taps["stem"] = x
    for i, b in enumerate(_b0_blocks()):
        p = f"_blocks.{i}."
        inp = x
        if b["e"] != 1:                                                    # model.py:98-101
            x = _swish(_bn(F.conv2d(x, sd[p + "_expand_conv.weight"]), sd, p + "_bn0", training, bn_state))
        cexp = x.shape[1]
        x = _same_conv(x, sd[p + "_depthwise_conv.weight"], b["s"], groups=cexp)   # model.py:103
        x = _swish(_bn(x, sd, p + "_bn1", training, bn_state))
        s = F.adaptive_avg_pool2d(x, 1)                                     # model.py:108-113
        s = _swish(F.conv2d(s, sd[p + "_se_reduce.weight"], sd[p + "_se_reduce.bias"]))
        s = F.conv2d(s, sd[p + "_se_expand.weight"], sd[p + "_se_expand.bias"])
        x = torch.sigmoid(s) * x
        x = _bn(F.conv2d(x, sd[p + "_project_conv.weight"]), sd, p + "_bn2", training, bn_state)  # :116-117
        if b["skip"] and b["cin"] == b["cout"]:
            p_drop = drop_connect_rate * float(i) / len(_b0_blocks())      # model.py:280-282
            if training and p_drop:                                         # model.py:125-126, utils.py:141-153
                keep = 1 - p_drop
                binary = torch.floor(keep + dc_uniform[i].to(x.dtype))
                x = x / keep * binary
            x = x + inp                                                     # model.py:127
        if taps is not None:
            taps[f"block{i}"] = x
    x = _swish(_bn(F.conv2d(x, sd["_conv_head.weight"]), sd, "_bn1", training, bn_state))   # model.py:286
    return x


# --------------------------------------------------------------------------------------------
# Xception (config 5 extractor)
# --------------------------------------------------------------------------------------------
_XC_BLOCKS = [  # (name, cin, cout, reps, stride, start_with_relu, grow_first)    models/xception.py:113-129
    ("block1", 64, 128, 2, 2, False, True), ("block2", 128, 256, 2, 2, True, True), ("block3", 256, 728, 2, 2, True, True),
    *[(f"block{i}", 728, 728, 3, 1, True, True) for i in range(4, 12)],
    ("block12", 728, 1024, 2, 2, True, False)]
_XC_BN_EPS, _XC_BN_MOM = 1e-5, 0.1   # nn.BatchNorm2d defaults (xception.py:97)


def _xc_bn(x, sd, prefix, training, bn_state):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, _XC_BN_MOM, _XC_BN_EPS)
    rm2, rv2 = rm.detach().clone(), rv.detach().clone()
    y = F.batch_norm(x, rm2, rv2, w, b, True, _XC_BN_MOM, _XC_BN_EPS)
    if bn_state is not None:
        bn_state.updates[prefix + ".running_mean"] = rm2
        bn_state.updates[prefix + ".running_var"] = rv2
    return y


def _xc_sep(x, sd, prefix):
    """SeparableConv2d (xception.py:17-27): depthwise 3x3 pad 1, then pointwise 1x1; no norm/activation in between."""
    x = F.conv2d(x, sd[prefix + ".conv1.weight"], None, 1, 1, 1, x.shape[1])
    return F.conv2d(x, sd[prefix + ".pointwise.weight"])


def xception_block_units(cin, cout, reps, grow_first):
    """Channel plan of a Block's separable convs (xception.py:44-58)."""
    units = []
    filters = cin
    if grow_first:
        units.append((cin, cout))
        filters = cout
    for _ in range(reps - 1):
        units.append((filters, filters))
    if not grow_first:
        units.append((cin, cout))
    return units


def xception_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, training: bool = False,
                     bn_state: Optional[BNState] = None, taps: Optional[dict] = None, masks: Optional[list] = None):
    """x [N,3,224,224] -> bn4 output [N,2048,7,7] WITHOUT the final ReLU (xception.py:161-203, 215-217)."""
    def relu(t):
        if masks is not None:                    # the piecewise-linear decisions of the forward: ReLU signs, max-pool winners
            masks.append(("relu", (t > 0).detach()))
        return F.relu(t)

    x = relu(_xc_bn(F.conv2d(x, sd["conv1.weight"], None, 2, 0), sd, "bn1", training, bn_state))
    x = relu(_xc_bn(F.conv2d(x, sd["conv2.weight"], None, 1, 0), sd, "bn2", training, bn_state))
    for (name, cin, cout, reps, stride, start_relu, grow_first) in _XC_BLOCKS:
        inp = x
        units = xception_block_units(cin, cout, reps, grow_first)
        # rep = [ReLU, Sep, BN] per unit (first ReLU dropped when not start_with_relu), then MaxPool(3, stride, 1) if stride != 1.
        # Sequential indices: with start_with_relu units sit at rep.1, rep.4, rep.7 (BN at +1); without, at rep.0, rep.3.
        idx = 0
        for u in range(len(units)):
            if u > 0 or start_relu:
                x = relu(x)
                idx += 1
            x = _xc_sep(x, sd, f"{name}.rep.{idx}")
            x = _xc_bn(x, sd, f"{name}.rep.{idx + 1}", training, bn_state)
            idx += 2
        if stride != 1:
            if masks is not None:
                masks.append(("maxpool", F.max_pool2d(x, 3, stride, 1, return_indices=True)[1].detach()))
            x = F.max_pool2d(x, 3, stride, 1)
        if cout != cin or stride != 1:
            skip = _xc_bn(F.conv2d(inp, sd[f"{name}.skip.weight"], None, stride), sd, f"{name}.skipbn", training, bn_state)
        else:
            skip = inp
        x = x + skip
        if taps is not None:
            taps[name] = x
    x = relu(_xc_bn(_xc_sep(x, sd, "conv3"), sd, "bn3", training, bn_state))
    x = _xc_bn(_xc_sep(x, sd, "conv4"), sd, "bn4", training, bn_state)
    return x


# --------------------------------------------------------------------------------------------
# Size-Invariant TimeSformer
# --------------------------------------------------------------------------------------------

def _attn_core(q, k, v, mask=None):
    """size_invariant_timesformer.py:80-87: fill (not add) with -FLT_MAX, softmax, weighted sum."""
    sim = torch.einsum("bid,bjd->bij", q, k)
    if mask is not None:
        sim = sim.masked_fill(~mask, -torch.finfo(sim.dtype).max)
    p = sim.softmax(dim=-1)
    return torch.einsum("bij,bjd->bid", p, v), p


def _attention(x, sd, prefix, heads, dim_head, mode, n, f, frame_mask, cls_mask):
    """Attention.forward (:109-144).  mode 'time' regroups tokens '(b n) f d', 'space' '(b f) n d'."""
    B, N, D = x.shape
    qkv = F.linear(x, sd[prefix + "to_qkv.weight"])                         # :111 (no bias)
    q, k, v = qkv.chunk(3, dim=-1)

    def heads_split(t):                                                     # :112  b n (h d) -> (b h) n d
        return t.reshape(B, N, heads, dim_head).permute(0, 2, 1, 3).reshape(B * heads, N, dim_head)

    q, k, v = map(heads_split, (q, k, v))
    q = q * (dim_head ** -0.5)                                              # :114 (cls query scaled too)
    cls_q, q_ = q[:, :1], q[:, 1:]
    cls_k, k_ = k[:, :1], k[:, 1:]
    cls_v, v_ = v[:, :1], v[:, 1:]
    cls_out, cls_att = _attn_core(cls_q, k, v, cls_mask)                    # :120  cls attends to all N keys
    BH = B * heads

    def regroup(t):                                                         # :122
        t = t.reshape(BH, f, n, dim_head)
        if mode == "time":
            return t.permute(0, 2, 1, 3).reshape(BH * n, f, dim_head)
        return t.reshape(BH * f, n, dim_head)

    q_, k_, v_ = map(regroup, (q_, k_, v_))
    r = q_.shape[0] // BH
    ck = cls_k.repeat_interleave(r, dim=0)                                  # :125-126  (b r) () d
    cv = cls_v.repeat_interleave(r, dim=0)
    k_ = torch.cat((ck, k_), dim=1)                                         # :128-129
    v_ = torch.cat((cv, v_), dim=1)
    out, _ = _attn_core(q_, k_, v_, frame_mask if mode == "time" else None)  # :132
    if mode == "time":                                                      # :135
        out = out.reshape(BH, n, f, dim_head).permute(0, 2, 1, 3).reshape(BH, f * n, dim_head)
    else:
        out = out.reshape(BH, f * n, dim_head)
    out = torch.cat((cls_out, out), dim=1)                                  # :138
    out = out.reshape(B, heads, N, dim_head).permute(0, 2, 1, 3).reshape(B, N, heads * dim_head)  # :141
    out = F.linear(out, sd[prefix + "to_out.0.weight"], sd[prefix + "to_out.0.bias"])            # :144
    return out, cls_att


def tsf_forward(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, mask: torch.Tensor,
                identities_mask: torch.Tensor, size_embedding: torch.Tensor, positions: torch.Tensor,
                require_attention: bool = False, taps: Optional[dict] = None, dropout_masks: Optional[dict] = None):
    """x [B,F,C,7,7] -> logits [B,1] (and [space_cls_att, time_cls_att] of the last layer).  :224-276.
    dropout_masks (train mode with attn-dropout / ff-dropout > 0): {(layer, 0 | 1 | 2): multiplier tensor} -- the keep / (1 - p)
    factors nn.Dropout applied after the time / space attention's output projection (:98-101, [B, N, dim]) and between GEGLU and
    the second feed-forward Linear (:66-70, [B, N, 4 dim]); the draws themselves are the caller's (tests feed the reference's)."""
    m = cfg["model"]
    heads, dim_head, depth, dim = m["heads"], m["dim-head"], m["depth"], m["dim"]
    b, f, c, h, w = x.shape
    n = h * w
    x = x.permute(0, 1, 3, 4, 2).reshape(b, f * n, c)                       # :227  b f c h w -> b (f h w) c
    tok = F.linear(x, sd["to_patch_embedding.weight"], sd["to_patch_embedding.bias"])   # :228
    cls = sd["cls_token"].unsqueeze(0).expand(b, -1, -1)                    # :231
    x = torch.cat((cls, tok), dim=1)                                        # :232
    if m.get("enable-pos-emb", True):
        x = x + F.embedding(positions, sd["pos_emb.weight"])                # :235-236
    else:
        x = x + F.embedding(torch.arange(x.shape[1]), sd["pos_emb.weight"])  # :237-238 (token index, clips share it)
    if m["enable-size-emb"]:                                                # :241-248
        se = size_embedding.to(x.device).repeat_interleave(n, dim=1)
        se = torch.cat((torch.zeros(b, 1, dtype=se.dtype), se), dim=1).int()
        x = x + F.embedding(se, sd["size_emb.weight"])
    if taps is not None:
        taps["tokens"] = x
    fm = mask.unsqueeze(1).expand(b, f, f) & identities_mask               # :252-253
    fm = F.pad(fm, (1, 0), value=True)                                      # :254
    frame_mask = fm.reshape(b, 1, 1, f, f + 1).expand(b, heads, n, f, f + 1).reshape(b * heads * n, f, f + 1)  # :255
    cm = mask.repeat_interleave(n, dim=1)                                   # :259  b (f n)
    cm = F.pad(cm, (1, 0), value=True)                                      # :260
    cls_mask = cm.reshape(b, 1, 1, -1).expand(b, heads, 1, cm.shape[-1]).reshape(b * heads, 1, -1)

    def ln(t, prefix):
        return F.layer_norm(t, (dim,), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)

    cls_rows = []
    for i in range(depth):                                                  # :263-268
        p = f"layers.{i}."
        dm = dropout_masks or {}
        y, t_att = _attention(ln(x, p + "0.norm"), sd, p + "0.fn.", heads, dim_head, "time", n, f, frame_mask, cls_mask)
        if (i, 0) in dm:
            y = y * dm[(i, 0)].to(y.dtype)                                  # to_out = Sequential(Linear, Dropout)  :98-101
        x = x + y
        y, s_att = _attention(ln(x, p + "1.norm"), sd, p + "1.fn.", heads, dim_head, "space", n, f, None, cls_mask)
        if (i, 1) in dm:
            y = y * dm[(i, 1)].to(y.dtype)
        x = x + y
        hdn = F.linear(ln(x, p + "2.norm"), sd[p + "2.fn.net.0.weight"], sd[p + "2.fn.net.0.bias"])
        a, g = hdn.chunk(2, dim=-1)                                         # :61-63 GEGLU, exact-erf gelu
        hdn = a * F.gelu(g)
        if (i, 2) in dm:
            hdn = hdn * dm[(i, 2)].to(hdn.dtype)                            # net = Linear, GEGLU, Dropout, Linear  :66-70
        x = F.linear(hdn, sd[p + "2.fn.net.3.weight"], sd[p + "2.fn.net.3.bias"]) + x
        cls_rows.append(x[:, 0])
    if taps is not None:
        taps["cls_rows"] = torch.stack(cls_rows)
        taps["x_final"] = x
    out = F.linear(ln(x[:, 0], "to_out.0"), sd["to_out.1.weight"], sd["to_out.1.bias"])   # :270-276
    if require_attention:
        return out, [s_att, t_att]                                          # :271 order [space, time]
    return out


# --------------------------------------------------------------------------------------------
# The caller's step (train.py:332-378, test.py:235-247) restated for synthetic inputs
# --------------------------------------------------------------------------------------------

def clip_forward(ef_sd, tsf_sd, cfg, batch, training_extractor=False, require_attention=False,
                 bn_state=None, taps=None, drop_connect_rate=0.0, dc_uniform=None):
    v = batch["videos"]
    b, f, h, w, c = v.shape
    vid = v.reshape(b * f, h, w, c).permute(0, 3, 1, 2)                     # train.py:341 (view; NHWC strides)
    feats = effnet_b0_forward(ef_sd, vid, training=training_extractor, bn_state=bn_state, taps=taps,
                              drop_connect_rate=drop_connect_rate, dc_uniform=dc_uniform)
    if taps is not None:
        taps["features"] = feats
    feats = feats.reshape(b, f, *feats.shape[1:])                           # train.py:354
    return tsf_forward(tsf_sd, cfg, feats, batch["mask"], batch["identities_mask"],
                       batch["size_embedding"], batch["positions"], require_attention, taps)


def bce_with_logits(logits, labels, pos_weight=None):
    """train.py:261,367-368: BCEWithLogitsLoss(pos_weight) on [B,1] logits vs [B,1] float labels."""
    pw = None if pos_weight is None else torch.as_tensor([pos_weight], dtype=logits.dtype)
    return F.binary_cross_entropy_with_logits(logits, labels.reshape(-1, 1).to(logits.dtype), pos_weight=pw)


def to_dtype(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def aggregate_attentions(attentions, heads, num_frames, frames_per_identity, scale_factor=50000):
    """Restatement of reference utils.py:68-96 (numpy): max over batch*heads per token, np.array_split into frames,
    mean*scale, softmax; identity sums with the reference's slicing."""
    import numpy as np
    agg = []
    for att in attentions:
        a = att.squeeze(1).reshape(-1, heads, att.shape[-1])
        agg.append([float(a[:, :, i].max()) for i in range(a.shape[2])])
    agg.append(list(np.sum(np.asarray(agg), axis=0)))
    out = []
    for row in agg:
        chunks = np.array_split(np.asarray(row, dtype=np.float64), num_frames)
        v = np.array([c.mean() * scale_factor for c in chunks])
        e = np.exp(v - v.max())
        out.append(list(e / e.sum()))
    ident = []
    for index, frames in enumerate(frames_per_identity):
        if index == 0:
            ident.append(sum(out[-1][:frames - 1]))
        else:
            ident.append(sum(out[-1][frames_per_identity[index - 1] - 1:frames - 1]))
    return out, ident


# --------------------------------------------------------------------------------------------
# Input-sequence builder (next-row f1): deepfakes_dataset.py:130-186, 216-339; predict.py:183-352
# --------------------------------------------------------------------------------------------
# PARITY PIN (partial): slot assignment (sort + assign_slots) is pinned against the reference's
# DeepFakesDataset.get_sorted_identities run on throw-away directory trees (tests/golden/slots.json, made by
# tools/make_golden.py), and the size-embedding constants (RANGE_SIZE, SIZE_EMB_DICT) against the imported module's
# (tests/golden/f1_constants.json).  The per-clip tensor construction (build_clip_tensors) is PARITY UNPINNED, and it cannot be
# pinned in this container.  The exact lines that block it:
#   deepfakes_dataset.py  __getitem__ (:191-339) is ONE function; between the slot assignment (:216) and the integer tail
#       (mask :280-284, identities_mask :313-320, positions :323-330) it calls cv2.VideoCapture(...).get(3|4) (:250-252: the video
#       area every size bucket is a ratio of), cv2.imread (:257: the face's shape) and the albumentations transform object
#       (:299-308: `transform(image=..., image1=...)`, built from IsotropicResize / PadIfNeeded / Resize at :57-108).
#   predict.py  generate_masks (:254-352) takes the crops in memory, but still calls cv2.VideoCapture(video_path).get (:286-288)
#       for the video area and create_val_transform(...)(image=...) (:321-325) before its integer tail (:329-345); the module's
#       own imports (:2-29: cv2, facenet_pytorch, albumentations, preprocessing.face_detector) are placeholders at best.
# Neither cv2 nor albumentations is installed, there is no network, and giving their calls BEHAVIOUR (a VideoCapture that answers
# get(3), a transform that returns its inputs) would be writing a stand-in for a library -- an oracle pinned against that pins
# nothing.  The integer tail has no function boundary of its own in either file, so it cannot be called on pre-built lists.
# It is therefore restated from the source lines cited below and tested against hand-derived cases (tests/test_sequence_builder.py).

_RANGE_SIZE = 5
SIZE_EMB_DICT = [(1 + i * _RANGE_SIZE, (i + 1) * _RANGE_SIZE) if i != 0 else (0, _RANGE_SIZE) for i in range(20)]   # :30-31


def max_faces_per_identity(num_frames, identities_number):
    """deepfakes_dataset.py:50-53 / predict.py:207-210."""
    table = {1: [num_frames],
             2: [int(num_frames / 2), int(num_frames / 2)],
             3: [int(num_frames / 3), int(num_frames / 3), int(num_frames / 4)],
             4: [int(num_frames / 3), int(num_frames / 3), int(num_frames / 8), int(num_frames / 8)]}
    return table[identities_number]


def sort_identities(infos, ordering=0, max_identities=3):
    """infos: list of (name, mean_side, number_of_faces).  deepfakes_dataset.py:149-158: ordering 0 = by mean face side, 1 = by
    number of faces, both descending and stable (Python's sort keeps ties in input order); then truncate."""
    infos = [list(r) for r in infos]
    if ordering == 0:
        infos = sorted(infos, key=lambda x: x[1], reverse=True)
    elif ordering == 1:
        infos = sorted(infos, key=lambda x: x[2], reverse=True)
    else:
        raise ValueError("random identity order (ordering 2) is not reproducible")
    return infos[:max_identities]


def assign_slots(face_counts, num_frames):
    """Slots per identity from the available face counts of the already sorted/truncated identities
    (deepfakes_dataset.py:160-186 == predict.py:203-243)."""
    n = len(face_counts)
    faces = list(face_counts)
    extra = []
    if n > 1:
        cap = max_faces_per_identity(num_frames, n)
        for i in range(n):
            if faces[i] < cap[i] and i < n - 1:
                faces[i + 1] += cap[i] - faces[i]          # :166 (the next identity's COUNT grows, not its cap)
                extra.append(0)
            elif faces[i] > cap[i]:
                extra.append(faces[i] - cap[i])
                faces[i] = cap[i]
            else:
                extra.append(0)
    else:
        faces[0] = num_frames                              # :173-175
        extra.append(0)
    total = sum(faces)
    if total < num_frames:                                  # :178-190
        for i in range(n):
            need = num_frames - total
            if extra[i] > 0:
                add = min(extra[i], need)
                faces[i] += add
                total += add
                if total == num_frames:
                    break
        if total < num_frames:
            faces[-1] += num_frames - total
    return faces


def select_faces(n_available, max_faces, index=1, variant="dataset"):
    """Indices of the faces kept when an identity has more faces than slots: uniform, alternating start by sample parity
    (deepfakes_dataset.py:236-242); predict.py:277-279 always uses the odd-index rule."""
    import numpy as np
    if n_available <= max_faces:
        return list(range(n_available))
    if variant == "predict" or index % 2:
        idx = np.round(np.linspace(0, n_available - 2, max_faces)).astype(int)
    else:
        idx = np.round(np.linspace(1, n_available - 1, max_faces)).astype(int)
    return [int(i) for i in idx]


def size_bucket(face_h, face_w, video_w, video_h, variant="dataset"):
    """Face/frame area ratio -> bucket 1..20 (deepfakes_dataset.py:246-263; predict.py:285-296 omits the /2 on the face)."""
    import numpy as np
    video_area = video_w * video_h / 2
    face_area = face_h * face_w / 2 if variant == "dataset" else face_h * face_w
    ratio = int(face_area * 100 / video_area)
    hits = [ratio in range(a, b + 1) for (a, b) in SIZE_EMB_DICT]
    return int(np.where(hits)[0][0] + 1)       # IndexError for ratio > 100, like the reference


def build_clip_tensors(identities, num_frames, num_patches=49, video_wh=(1280, 720), variant="dataset",
                       enable_identity_attention=True):
    """identities: per identity dict(slots=int, faces=[(frame_number, face_h, face_w), ...]) AFTER select_faces, in slot order.
    Returns (size_embedding int32 [F], mask bool [F], identities_mask bool [F,F], positions int64 [1+F*49]) -- the non-image
    members of the tuple at deepfakes_dataset.py:339 / predict.py:352.

    Faithful quirk: in the DATASET the padding test at :281 runs after identity_images was already extended to max_faces
    (:268-272), so its mask is all ones even for padded slots; predict.py:301-309 builds the intended mask."""
    mask, sizes, frames = [], [], []
    for ident in identities:
        max_faces = ident["slots"]
        got = ident["faces"][:max_faces]
        id_sizes = [size_bucket(h, w, video_wh[0], video_wh[1], variant) for (_, h, w) in got]
        frames.extend(fr for (fr, _, _) in got)
        if len(got) < max_faces:
            diff = max_faces - len(got)
            id_sizes = id_sizes + [0] * diff                                   # :269
            frames.extend([max(frames) if frames else 0] * diff)               # :271-275 (max over ALL frames so far)
            if variant == "predict" and enable_identity_attention:
                mask.extend([1 if i < max_faces - diff else 0 for i in range(max_faces)])     # predict.py:304
            else:
                mask.extend([1] * max_faces)                                   # dataset :281-284 (see docstring)
        else:
            mask.extend([1] * max_faces)
        sizes.extend(id_sizes)
    assert len(mask) == num_frames, (len(mask), num_frames)
    ident_mask = []
    start = 0
    for ident in identities:                                                   # :315-321
        row = [start <= i < start + ident["slots"] for i in range(num_frames)]
        ident_mask.extend([row] * ident["slots"])
        start += ident["slots"]
    rank = {k: v + 1 for v, k in enumerate(sorted(set(frames)))}              # :324
    positions = [0]                                                            # :329 cls
    for fr in frames:
        p = rank[fr]
        positions.extend(range((p - 1) * num_patches + 1, p * num_patches + 1))   # :327
    return (torch.tensor(sizes).int(), torch.tensor(mask).bool(), torch.tensor(ident_mask).bool(), torch.tensor(positions))
