"""rapid_amd -- MI355X-native cut-detection / consensus-counting engine for Rapid-style membership.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of include/rapid_mi355x.h),
`engine.py` (host-side mirror of the reference's Java interface for this path), `scenarios.py` (seeded synthetic
alert streams) and `parallel.py` (receiver sharding across ranks).  Nothing in this package imports `oracle/`.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
