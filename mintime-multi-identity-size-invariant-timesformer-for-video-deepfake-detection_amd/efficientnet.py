"""EfficientNet-B0 feature extractor on libmintime_hip (MI355X).

Python surface of the reference's vendored, modified efficientnet_pytorch
(reference models/efficientnet/efficientnet_pytorch/model.py:138-444): `EfficientNet.from_name`,
`.from_pretrained`, `.load_matching_state_dict`, `.forward(x[N,3,224,224]) -> [N,1280,7,7]` (the feature
map, not logits -- model.py:267-288), train()/eval() semantics of BatchNorm and drop-connect, parameter
names of the form `_blocks.<i>.<sub>` (parsed by train.py:159-167) and the 360 upstream state-dict keys.

The arithmetic is a launch sequence over NHWC buffers (see csrc/effnet_fwd.hip for the layout rules); the
returned tensor is an NHWC-strided view shaped [N,1280,7,7], so the caller's
`rearrange('(b f) c h w -> b f c h w')` (train.py:354) stays a view and the TimeSformer consumes it with
zero copies.  Submodules are parameter/buffer holders only.
"""
import torch
from torch import nn

from . import arch
from . import lib as L

VALID_MODELS = ("efficientnet-b0",)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the HIP path is driven by EfficientNet.forward")


class _Conv(_Holder):
    def __init__(self, cin, cout, k, groups=1, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None


class _BatchNorm(_Holder):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.momentum = arch.BN_MOMENTUM_EFFNET
        self.eps = arch.BN_EPS_EFFNET


class _FC(_Holder):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout))


class MBConvBlock(_Holder):
    """Parameter holder for one MBConv block (model.py:36-87)."""

    def __init__(self, spec: arch.MBConv):
        super().__init__()
        self.spec = spec
        if spec.has_expand:
            self._expand_conv = _Conv(spec.cin, spec.cexp, 1)
            self._bn0 = _BatchNorm(spec.cexp)
        self._depthwise_conv = _Conv(spec.cexp, spec.cexp, spec.k, groups=spec.cexp)
        self._bn1 = _BatchNorm(spec.cexp)
        self._se_reduce = _Conv(spec.cexp, spec.cse, 1, bias=True)
        self._se_expand = _Conv(spec.cse, spec.cexp, 1, bias=True)
        self._project_conv = _Conv(spec.cexp, spec.cout, 1)
        self._bn2 = _BatchNorm(spec.cout)


class EfficientNet(nn.Module):
    def __init__(self, model_name="efficientnet-b0", drop_connect_rate=arch.DROP_CONNECT_RATE, num_classes=1000,
                 include_top=True, image_size=arch.IMAGE_SIZE):
        super().__init__()
        self._check_model_name_is_valid(model_name)
        self.model_name = model_name
        self.drop_connect_rate = float(drop_connect_rate or 0.0)
        self.image_size = image_size
        self._conv_stem = _Conv(arch.STEM_CIN, arch.STEM_COUT, arch.STEM_K)
        self._bn0 = _BatchNorm(arch.STEM_COUT)
        self._blocks = nn.ModuleList([MBConvBlock(s) for s in arch.effnet_b0_blocks(image_size)])
        self._conv_head = _Conv(arch.HEAD_CIN, arch.HEAD_COUT, 1)
        self._bn1 = _BatchNorm(arch.HEAD_COUT)
        if include_top:
            # present in the reference's state-dict, never used by forward (model.py:206-208)
            self._fc = _FC(arch.HEAD_COUT, num_classes)
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, _Conv):
                nn.init.kaiming_uniform_(m.weight, a=5 ** 0.5)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, _FC):
                nn.init.kaiming_uniform_(m.weight, a=5 ** 0.5)

    # ---- reference-compatible constructors -------------------------------------------------------
    @classmethod
    def _check_model_name_is_valid(cls, model_name):
        if model_name not in VALID_MODELS:
            raise ValueError("model_name should be one of: " + ", ".join(VALID_MODELS) +
                             " (MINTIME only uses efficientnet-b0, train.py:121-126)")

    @classmethod
    def from_name(cls, model_name, in_channels=3, **override_params):
        if in_channels != 3:
            raise NotImplementedError("the stem kernel is specialised for 3 input channels (BGR crops)")
        allowed = {"drop_connect_rate", "num_classes", "include_top", "image_size"}
        bad = set(override_params) - allowed
        if bad:
            raise ValueError(f"unsupported override params for the MI355X build: {sorted(bad)}")
        return cls(model_name, **override_params)

    @classmethod
    def from_pretrained(cls, model_name, weights_path=None, advprop=False, in_channels=3, num_classes=1000,
                        **override_params):
        model = cls.from_name(model_name, in_channels=in_channels, num_classes=num_classes, **override_params)
        if weights_path is None:
            raise RuntimeError("from_pretrained without weights_path needs network access (the reference downloads from "
                               "a URL, utils.py:602); pass weights_path=<local .pth>")
        sd = torch.load(weights_path, map_location="cpu")
        model.load_state_dict(sd, strict=(num_classes == 1000))
        return model

    def load_matching_state_dict(self, state_dict):
        """Copy every entry whose (prefix-stripped) name exists here; skip the rest (model.py:368-378)."""
        own = self.state_dict()
        for name, param in state_dict.items():
            if "efficient_net" in name:
                name = name.split("efficient_net.")[1]
            if name not in own:
                continue
            if isinstance(param, nn.Parameter):
                param = param.data
            own[name].copy_(param)

    @classmethod
    def get_image_size(cls, model_name):
        cls._check_model_name_is_valid(model_name)
        return arch.IMAGE_SIZE

    def set_swish(self, memory_efficient=True):
        """Kept for API compatibility: the HIP path always recomputes swish in backward (saves only z)."""
        return None

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, inputs):
        from .effnet_engine import effnet_apply
        return effnet_apply(self, inputs)[0]

    def extract_features(self, inputs):
        return self.forward(inputs)

    def extract_endpoints(self, inputs):
        """reduction_1..6 feature maps (model.py:222-265), as NHWC-strided NCHW views."""
        from .effnet_engine import effnet_apply
        feat, ys = effnet_apply(self, inputs, want_blocks=True)
        endpoints = {}
        prev = None
        for i, y in enumerate(ys):
            if prev is not None and prev.shape[2] > y.shape[2]:
                endpoints[f"reduction_{len(endpoints) + 1}"] = prev
            elif i == len(ys) - 1:
                endpoints[f"reduction_{len(endpoints) + 1}"] = y
            prev = y
        endpoints[f"reduction_{len(endpoints) + 1}"] = feat
        return endpoints
