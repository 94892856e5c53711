"""Drop-in for reference models/xception.py: `xception(pretrain_path=None, **kwargs)` (train.py:130-133), HIP-backed."""
import mintime_amd as _impl

xception = _impl.xception
Xception = _impl.Xception

__all__ = ["xception"]
