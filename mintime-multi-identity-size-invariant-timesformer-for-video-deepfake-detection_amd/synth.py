"""Seeded synthetic weights and inputs (machine-stable: numpy Philox, no torch RNG).

Why not the reference's own random init: a random-init EfficientNet in eval() has BN running
stats 0/1 and produces ~1e-9 features -- a parity test on that is vacuous.
These "healthy" weights keep every activation O(1) so a relative-1e-3 comparison means
something; the same state-dict is loaded into the reference (when fixtures are generated),
into the CPU oracle and into the HIP modules, which also exercises key/shape compatibility.

Input contract follows reference deepfakes_dataset.py:339 (the tuple the dataset returns) and
SURVEY.md §8(d).
"""
import numpy as np
import torch

from . import arch


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def _t(a, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a.astype(dtype)))


def effnet_b0_state(seed: int = 0, include_top: bool = True, calibrate: bool = True):
    """Seeded state-dict with the reference EfficientNet-B0 keys/shapes (360 entries).

    calibrate=True sets every BN's running_mean/var from the statistics a seeded 2-crop batch
    produces at that layer (perturbed by ~10 %), the way a trained network's buffers track its
    activations -- so eval-mode activations stay O(1) through all 16 blocks.
    """
    sd = {}
    for i, (key, shape, kind) in enumerate(arch.effnet_b0_state_spec(include_top)):
        g = _rng(seed, 1000 + i)
        if kind == "stem":
            # inputs are raw 0..255 (deepfakes_dataset.py:257): keep z0 O(1)
            w = g.standard_normal(shape) / (np.sqrt(27.0) * 74.0)
        elif kind in ("expand", "head"):
            w = g.standard_normal(shape) * (1.3 / np.sqrt(shape[1]))
        elif kind == "project":
            w = g.standard_normal(shape) * (3.0 / np.sqrt(shape[1]))
        elif kind == "dw":
            w = g.standard_normal(shape) * (1.6 / np.sqrt(shape[2] * shape[3]))
        elif kind in ("se_r_w", "se_e_w"):
            w = g.standard_normal(shape) * (1.5 / np.sqrt(shape[1]))
        elif kind in ("se_r_b", "se_e_b"):
            w = g.standard_normal(shape) * 0.2
        elif kind == "bn_w":
            w = g.uniform(0.5, 1.5, shape)
        elif kind == "bn_b":
            w = g.standard_normal(shape) * 0.1
        elif kind == "bn_rm":
            w = g.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            w = g.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            sd[key] = torch.tensor(0, dtype=torch.int64)
            continue
        elif kind == "fc_w":
            w = g.standard_normal(shape) * 0.01
        elif kind == "fc_b":
            w = np.zeros(shape)
        else:
            raise KeyError(kind)
        sd[key] = _t(w)
    if calibrate:
        _calibration_tools().calibrate_effnet_bn(sd, seed)
    return sd


def _calibration_tools():
    """The BatchNorm-buffer calibration walks the network once with plain torch ops on the host.  That is weight SYNTHESIS for
    tests and benchmarks, not a forward path of the product, so it lives outside the package (tools/calibrate_bn.py)."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module("tools.calibrate_bn")


def xception_state(seed: int = 0, num_classes: int = 1, calibrate: bool = True):
    """Seeded state-dict with the reference Xception keys/shapes (276 entries); BN buffers calibrated like effnet_b0_state."""
    sd = {}
    for i, (key, shape, kind) in enumerate(arch.xception_state_spec(num_classes)):
        g = _rng(seed, 20000 + i)
        if kind == "conv_first":
            w = g.standard_normal(shape) / (np.sqrt(27.0) * 74.0)
        elif kind == "conv":
            w = g.standard_normal(shape) * (1.4 / np.sqrt(shape[1] * 9))
        elif kind == "dw":
            w = g.standard_normal(shape) * (1.4 / 3.0)
        elif kind == "pw":
            w = g.standard_normal(shape) * (1.0 / np.sqrt(shape[1]))
        elif kind == "bn_w":
            w = g.uniform(0.5, 1.5, shape)
        elif kind == "bn_b":
            w = g.standard_normal(shape) * 0.1
        elif kind == "bn_rm":
            w = g.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            w = g.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            sd[key] = torch.tensor(0, dtype=torch.int64)
            continue
        elif kind == "fc_w":
            w = g.standard_normal(shape) * 0.01
        elif kind == "fc_b":
            w = np.zeros(shape)
        else:
            raise KeyError(kind)
        sd[key] = _t(w)
    if calibrate:
        _calibration_tools().calibrate_xception_bn(sd, seed)
    return sd


def tsf_state(cfg, seed: int = 0):
    """Seeded state-dict with the reference SizeInvariantTimeSformer keys/shapes."""
    sd = {}
    for i, (key, shape, kind) in enumerate(arch.tsf_state_spec(cfg)):
        g = _rng(seed, 5000 + i)
        if kind == "lin_w":
            w = g.standard_normal(shape) * 0.03
        elif kind == "qkv_w":
            w = g.standard_normal(shape) * 0.08   # peaky softmax: exercises the attention numerics
        elif kind == "lin_b":
            w = g.standard_normal(shape) * 0.02
        elif kind == "ln_w":
            w = g.uniform(0.5, 1.5, shape)
        elif kind == "ln_b":
            w = g.standard_normal(shape) * 0.1
        elif kind == "emb":
            # only rows [0, F*49] (pos) / [0, 20] (size) are ever read; fill a prefix, zero the rest
            w = np.zeros(shape, dtype=np.float32)
            live = min(shape[0], 2048)
            w[:live] = g.standard_normal((live, shape[1])) * 0.05
        elif kind == "cls":
            w = g.standard_normal(shape) * 0.05
        else:
            raise KeyError(kind)
        sd[key] = _t(w)
    return sd


def identity_split(num_frames: int, num_identities: int):
    """Slots per identity with ample faces (deepfakes_dataset.py:50-53,153-186):
    8/1 -> [8]; 8/2 -> [4,4]; 16/3 -> [7,5,4] ..."""
    table = {(8, 1): [8], (8, 2): [4, 4], (16, 1): [16], (16, 2): [8, 8], (16, 3): [7, 5, 4],
             (32, 1): [32], (32, 2): [16, 16], (32, 3): [14, 10, 8]}
    if (num_frames, num_identities) in table:
        return table[(num_frames, num_identities)]
    base = num_frames // num_identities
    out = [base] * num_identities
    out[0] += num_frames - base * num_identities
    return out


def clip_inputs(batch: int, num_frames: int = 8, num_identities: int = 1, seed: int = 0,
                ragged: bool = False, image_size: int = 224, num_patches: int = 49,
                with_video: bool = True):
    """The tuple the reference dataset yields, for `batch` clips (deepfakes_dataset.py:339).

    videos          [B,F,H,W,3] fp32, integer-valued 0..255 (BGR uint8 crops cast to float)
    size_embedding  [B,F] int32 in 1..20 (0 at padded slots)
    mask            [B,F] bool (False = padded slot)
    identities_mask [B,F,F] bool block-diagonal over identities
    positions       [B,1+F*49] int64: [0] ++ 49 ids per slot of its 1-based temporal rank
    labels          [B] float {0,1}
    ragged=True pads the last slot of every identity (mask False, size 0, zero image,
    temporal rank = previous max) to exercise the masking path.
    """
    F = num_frames
    split = identity_split(F, num_identities)
    g = _rng(seed, 9000)
    out = {}
    if with_video:
        v = g.integers(0, 256, size=(batch, F, image_size, image_size, 3), dtype=np.uint8)
    size = g.integers(1, 21, size=(batch, F)).astype(np.int32)
    mask = np.ones((batch, F), dtype=bool)
    ident = np.zeros((batch, F, F), dtype=bool)
    rank = np.zeros((batch, F), dtype=np.int64)
    s0 = 0
    for n in split:
        ident[:, s0:s0 + n, s0:s0 + n] = True
        rank[:, s0:s0 + n] = np.arange(1, n + 1)
        if ragged and n > 1:
            mask[:, s0 + n - 1] = False
            size[:, s0 + n - 1] = 0
            rank[:, s0 + n - 1] = n - 1
            if with_video:
                v[:, s0 + n - 1] = 0
        s0 += n
    pos = np.zeros((batch, 1 + F * num_patches), dtype=np.int64)
    ar = np.arange(1, num_patches + 1)
    for f in range(F):
        pos[:, 1 + f * num_patches: 1 + (f + 1) * num_patches] = (rank[:, f:f + 1] - 1) * num_patches + ar[None]
    labels = g.integers(0, 2, size=(batch,)).astype(np.float32)
    if with_video:
        out["videos"] = torch.from_numpy(v).float()
    out["size_embedding"] = torch.from_numpy(size)
    out["mask"] = torch.from_numpy(mask)
    out["identities_mask"] = torch.from_numpy(ident)
    out["positions"] = torch.from_numpy(pos)
    out["labels"] = torch.from_numpy(labels)
    return out


def features(batch: int, num_frames: int, channels: int, seed: int = 0, hw: int = 7):
    """Synthetic extractor output [B,F,C,7,7] (for TimeSformer-only cases), O(1) like real features."""
    g = _rng(seed, 9100)
    x = g.standard_normal((batch, num_frames, hw, hw, channels)).astype(np.float32)
    x = np.maximum(x, -0.3) * 0.6   # skewed, mostly-positive like swish outputs
    # stored NHWC, returned as the [B,F,C,H,W] view the reference call site produces (train.py:354)
    return torch.from_numpy(x).permute(0, 1, 4, 2, 3)
