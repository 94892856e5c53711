"""Drop-in for reference models/efficientnet/efficientnet_pytorch: `EfficientNet` with from_name /
from_pretrained / load_matching_state_dict / forward -> [N,1280,7,7]; implementation is HIP-backed."""
import mintime_amd as _impl

__version__ = "0.7.1"
EfficientNet = _impl.EfficientNet
VALID_MODELS = _impl.efficientnet.VALID_MODELS

__all__ = ["EfficientNet", "VALID_MODELS"]
