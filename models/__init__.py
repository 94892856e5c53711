"""Drop-in boundary: the reference's `models` package paths, backed by the MI355X-native implementation.

    from models.efficientnet.efficientnet_pytorch import EfficientNet      (reference train.py:27)
    from models.size_invariant_timesformer import SizeInvariantTimeSformer (reference train.py:28)
"""
