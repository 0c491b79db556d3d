"""model_matching_amd -- MI355X-native StoCS / Super4PCS matching engine (hot path only).

The product is the C-ABI library ``libstocs_hip.so`` (HIP kernels for gfx950 + C++ host code,
declared in ``include/stocs_hip.h``).  This package is the thin Python plumbing on top of it
(ctypes binding, estimator mirror of the reference's ``stocs::stocs_estimator``, synthetic
workloads).  There is no CPU fallback: importing :mod:`model_matching_amd.capi` without the built
library raises.
"""
__all__ = ["capi", "estimator", "synth"]
