"""MI355X-native MINTIME hot path: EfficientNet-B0 feature extractor + Size-Invariant TimeSformer.

Host side is Python on PyTorch-ROCm (tensors, autograd plumbing); all arithmetic runs in the
hand-written HIP library `csrc/libmintime_hip.so` (C ABI declared in include/mintime_hip.h).
There is no CPU or eager-PyTorch fallback: every op raises if the library is missing.
"""
import os as _os

# Takes effect only if the HIP runtime has not started yet (see bench.py): keeps the side stream on its own hardware queue once
# RCCL has created its streams.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import arch, synth, lib  # noqa: F401,E402
from . import timesformer, tsf_engine, tsf_backward  # noqa: F401
from . import efficientnet, effnet_engine, effnet_backward  # noqa: F401
from . import ddp, optim, harness, sequence, plans  # noqa: F401
from .timesformer import SizeInvariantTimeSformer  # noqa: F401
from .efficientnet import EfficientNet  # noqa: F401
from . import xception as xception_module, xception_engine  # noqa: F401
from .xception import xception, Xception  # noqa: F401

__all__ = ["arch", "synth", "lib", "timesformer", "tsf_engine", "SizeInvariantTimeSformer", "efficientnet", "effnet_engine", "EfficientNet"]
