"""Xception feature extractor (config 5 / "XS" variant) on libmintime_hip (MI355X).

Python surface of reference models/xception.py: `xception(pretrain_path=None, **kwargs)` -> module whose
`forward(x[N,3,224,224])` returns the bn4 output `[N,2048,7,7]` WITHOUT the final ReLU (xception.py:201-203, 215-217),
with the reference's 276 state-dict keys (conv1, bn1, conv2, bn2, block{1..12}.{skip,skipbn,rep.<i>...}, conv3, bn3, conv4,
bn4, fc).  Submodules are parameter/buffer holders; the arithmetic is the launch sequence in xception_engine.py.
"""
import torch
from torch import nn

from . import arch
from . import lib as L


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the HIP path is driven by Xception.forward")


class _Conv(_Holder):
    def __init__(self, cin, cout, k, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))


class _BatchNorm(_Holder):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.momentum = arch.BN_MOMENTUM_XCEPTION
        self.eps = arch.BN_EPS_XCEPTION


class SeparableConv2d(_Holder):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = _Conv(cin, cin, 3, groups=cin)
        self.pointwise = _Conv(cin, cout, 1)


class _Seq(_Holder):
    def __init__(self, children: dict):
        super().__init__()
        for k, m in children.items():
            self.add_module(str(k), m)


class Block(_Holder):
    def __init__(self, name, cin, cout, reps, stride, start_with_relu, grow_first):
        super().__init__()
        self.cfg = (name, cin, cout, reps, stride, start_with_relu, grow_first)
        if cout != cin or stride != 1:
            self.skip = _Conv(cin, cout, 1)
            self.skipbn = _BatchNorm(cout)
        else:
            self.skip = None
        units = arch.xception_block_units(cin, cout, reps, grow_first)
        children = {}
        self.unit_index = []
        for (sep, bnp), (ci, co) in zip(arch.xception_unit_keys(name, start_with_relu, len(units)), units):
            i_sep, i_bn = int(sep.rsplit(".", 1)[1]), int(bnp.rsplit(".", 1)[1])
            children[i_sep] = SeparableConv2d(ci, co)
            children[i_bn] = _BatchNorm(co)
            self.unit_index.append((i_sep, i_bn, ci, co))
        self.rep = _Seq(children)


class _FC(_Holder):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout))


class Xception(nn.Module):
    def __init__(self, in_channels=3, num_classes=1000, **kwargs):
        super().__init__()
        if in_channels != 3:
            raise NotImplementedError("3 input channels (BGR crops) only")
        self.num_classes = num_classes
        self.conv1 = _Conv(3, 32, 3)
        self.bn1 = _BatchNorm(32)
        self.conv2 = _Conv(32, 64, 3)
        self.bn2 = _BatchNorm(64)
        for cfg in arch.XCEPTION_BLOCKS:
            setattr(self, cfg[0], Block(*cfg))
        self.conv3 = SeparableConv2d(1024, 1536)
        self.bn3 = _BatchNorm(1536)
        self.conv4 = SeparableConv2d(1536, 2048)
        self.bn4 = _BatchNorm(2048)
        self.fc = _FC(2048, num_classes)      # present in the state-dict, unused by forward (xception.py:137, 215-217)
        for m in self.modules():              # xception.py:142-149
            if isinstance(m, _Conv):
                n = m.weight.shape[2] * m.weight.shape[3] * m.weight.shape[0]
                nn.init.normal_(m.weight, 0.0, (2.0 / n) ** 0.5)
        nn.init.kaiming_uniform_(self.fc.weight, a=5 ** 0.5)

    def blocks(self):
        return [getattr(self, cfg[0]) for cfg in arch.XCEPTION_BLOCKS]

    def forward(self, input):
        from .xception_engine import xception_apply
        return xception_apply(self, input)

    def features(self, input):
        """(bn4 output, []) -- the per-block feature list of the reference (xception.py:161-203) is not materialised."""
        return self.forward(input), []


def xception(pretrain_path=None, **kwargs):
    """Factory with the reference's loading rules (xception.py:242-272): strip `module.`, skip unknown keys, tolerate
    shape mismatches with a message."""
    model = Xception(**kwargs)
    if pretrain_path is not None:
        state_dict = torch.load(pretrain_path, map_location="cpu")
        if "state_dict" in state_dict:
            state_dict = state_dict["state_dict"]
        own = model.state_dict()
        for name, param in state_dict.items():
            name = name.replace("module.", "")
            if name in own:
                if isinstance(param, nn.Parameter):
                    param = param.data
                try:
                    own[name].copy_(param)
                except Exception:
                    print(f"While copying the parameter named {name}, whose dimensions in the model are {own[name].size()} and "
                          f"whose dimensions in the checkpoint are {param.size()}.")
        print("Features Extractor checkpoint loaded.")
    return model
