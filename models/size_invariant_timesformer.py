"""Drop-in for reference models/size_invariant_timesformer.py: same import path, class name, constructor and
forward signature; implementation lives in the HIP-backed package."""
import mintime_amd as _impl

SizeInvariantTimeSformer = _impl.SizeInvariantTimeSformer

__all__ = ["SizeInvariantTimeSformer"]
