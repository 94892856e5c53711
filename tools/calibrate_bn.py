"""BatchNorm running-statistics calibration for the SYNTHETIC weights (mintime_amd.synth).

Test / benchmark infrastructure, deliberately outside the product package: one fp64 walk of the network with plain torch ops on
the host picks running_mean / running_var consistent with the seeded weights, the way a trained network's buffers track its
activations, so that eval-mode activations stay O(1) and a relative-1e-3 parity test means something.
"""
import numpy as np
import torch

import mintime_amd  # noqa: F401
from mintime_amd import arch


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def calibrate_effnet_bn(sd, seed):
    """Weight-synthesis helper (NOT a forward path of the product): walks the B0 graph once in fp64 on
    the host with plain torch ops to pick running statistics consistent with the seeded weights."""
    import torch.nn.functional as F
    g = _rng(seed, 77)
    x = torch.from_numpy(g.integers(0, 256, size=(2, 3, arch.IMAGE_SIZE, arch.IMAGE_SIZE)).astype(np.float64))
    d = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    layer = [0]

    def bn(z, prefix):
        c = z.shape[1]
        gg = _rng(seed, 200 + layer[0])
        layer[0] += 1
        mu = z.mean(dim=(0, 2, 3))
        var = z.var(dim=(0, 2, 3), unbiased=False)
        rm = mu + var.sqrt() * torch.from_numpy(gg.standard_normal(c) * 0.1)
        rv = var * torch.from_numpy(gg.uniform(0.8, 1.25, c))
        sd[prefix + ".running_mean"] = rm.float()
        sd[prefix + ".running_var"] = rv.float()
        rm, rv = rm.float().double(), rv.float().double()
        sh = (1, c, 1, 1)
        return (z - rm.view(sh)) / torch.sqrt(rv.view(sh) + arch.BN_EPS_EFFNET) * d[prefix + ".weight"].view(sh) \
            + d[prefix + ".bias"].view(sh)

    def sw(t):
        return t * torch.sigmoid(t)

    def same(t, w, s, groups=1):
        p0, p1 = arch.same_pad(t.shape[-1], w.shape[-1], s)
        return F.conv2d(F.pad(t, [p0, p1, p0, p1]), w, None, s, 0, 1, groups)

    x = sw(bn(same(x, d["_conv_stem.weight"], 2), "_bn0"))
    for b in arch.effnet_b0_blocks():
        p = f"_blocks.{b.idx}."
        inp = x
        if b.has_expand:
            x = sw(bn(F.conv2d(x, d[p + "_expand_conv.weight"]), p + "_bn0"))
        x = sw(bn(same(x, d[p + "_depthwise_conv.weight"], b.s, groups=b.cexp), p + "_bn1"))
        s = x.mean(dim=(2, 3), keepdim=True)
        s = sw(F.conv2d(s, d[p + "_se_reduce.weight"], d[p + "_se_reduce.bias"]))
        s = torch.sigmoid(F.conv2d(s, d[p + "_se_expand.weight"], d[p + "_se_expand.bias"]))
        x = bn(F.conv2d(x * s, d[p + "_project_conv.weight"]), p + "_bn2")
        if b.skip:
            x = x + inp
    bn(F.conv2d(x, d["_conv_head.weight"]), "_bn1")


def calibrate_xception_bn(sd, seed):
    """Weight-synthesis helper (NOT a forward path of the product): one fp64 host walk of the Xception graph."""
    import torch.nn.functional as F
    g = _rng(seed, 78)
    x = torch.from_numpy(g.integers(0, 256, size=(2, 3, arch.IMAGE_SIZE, arch.IMAGE_SIZE)).astype(np.float64))
    d = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    layer = [0]

    def bn(z, prefix):
        c = z.shape[1]
        gg = _rng(seed, 300 + layer[0])
        layer[0] += 1
        mu, var = z.mean(dim=(0, 2, 3)), z.var(dim=(0, 2, 3), unbiased=False)
        rm = (mu + var.sqrt() * torch.from_numpy(gg.standard_normal(c) * 0.1)).float()
        rv = (var * torch.from_numpy(gg.uniform(0.8, 1.25, c))).float()
        sd[prefix + ".running_mean"], sd[prefix + ".running_var"] = rm, rv
        sh = (1, c, 1, 1)
        return (z - rm.double().view(sh)) / torch.sqrt(rv.double().view(sh) + arch.BN_EPS_XCEPTION) * d[prefix + ".weight"].view(sh) \
            + d[prefix + ".bias"].view(sh)

    def sep(t, prefix):
        t = F.conv2d(t, d[prefix + ".conv1.weight"], None, 1, 1, 1, t.shape[1])
        return F.conv2d(t, d[prefix + ".pointwise.weight"])

    x = F.relu(bn(F.conv2d(x, d["conv1.weight"], None, 2, 0), "bn1"))
    x = F.relu(bn(F.conv2d(x, d["conv2.weight"], None, 1, 0), "bn2"))
    for (name, cin, cout, reps, stride, srelu, grow) in arch.XCEPTION_BLOCKS:
        inp = x
        units = arch.xception_block_units(cin, cout, reps, grow)
        for u, (sp, bnp) in enumerate(arch.xception_unit_keys(name, srelu, len(units))):
            if u > 0 or srelu:
                x = F.relu(x)
            x = bn(sep(x, sp), bnp)
        if stride != 1:
            x = F.max_pool2d(x, 3, stride, 1)
        if cout != cin or stride != 1:
            x = x + bn(F.conv2d(inp, d[name + ".skip.weight"], None, stride), name + ".skipbn")
        else:
            x = x + inp
    x = F.relu(bn(sep(x, "conv3"), "bn3"))
    bn(sep(x, "conv4"), "bn4")


