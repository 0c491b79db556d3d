"""model_matching_amd -- MI355X-native StoCS / Super4PCS matching engine (hot path only).

The product is the C-ABI library ``libstocs_hip.so`` (HIP kernels for gfx950 + C++ host code,
declared in ``include/stocs_hip.h``).  This package is the thin Python plumbing on top of it
(ctypes binding, estimator mirror of the reference's ``stocs::stocs_estimator``, synthetic
workloads).  There is no CPU fallback: importing :mod:`model_matching_amd.capi` without the built
library raises.
"""
__all__ = ["capi", "estimator", "synth"]


def pin_host_thread_pools():
    """For harnesses (tests, bench.py, tools/): the Python side around the library does a few tiny numpy calls between GPU calls.  On
    the GPU boxes a process sees 256 CPUs while its cgroup has a CPU quota of 16: OpenBLAS sizes its worker pool by the former, the
    spinning workers exhaust the latter, and the kernel freezes the WHOLE process for the rest of a 100 ms scheduler period --
    the sporadic 65-80 ms pause of rounds 1-3 (profiles/r03_stall_root_cause.json).  One BLAS thread is plenty there.
    Importing the package does NOT do this by itself (a host application keeps its BLAS pools): a harness opts in by calling this
    function or by setting STOCS_PIN_BLAS=1 before the import; STOCS_KEEP_BLAS_THREADS=1 overrides both (tools/stall_watch.py
    reproduces the pause that way)."""
    import os
    if os.environ.get("STOCS_KEEP_BLAS_THREADS") == "1":
        return
    for v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(v, "1")            # for libraries loaded after this point
    try:
        import threadpoolctl                      # for a BLAS that numpy has already loaded
        threadpoolctl.threadpool_limits(limits=1, user_api="blas")
    except Exception:
        pass


def _opt_in():
    import os
    if os.environ.get("STOCS_PIN_BLAS") == "1":
        pin_host_thread_pools()


_opt_in()
