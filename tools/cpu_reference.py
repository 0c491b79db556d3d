#!/usr/bin/env python3
"""The reference's CPU path over the WHOLE path, for the measurement tools (tools/trials.py --cpu-reference, tools/frame_latency.py
--cpu-reference): the oracle's restated run_stocs_estimation (oracle/stocs_oracle.cpp orc_run_mode; single-threaded like the reference)
reports the three spans the reference's driver prints -- base sampling / congruent sets + transforms / verification
(src/stocs_match_one_object.cpp:80-105,110-151,156-163).  A measurement helper: nothing in the product path imports it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup's quota (the GPU boxes show 256 and grant 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def spans(cloud, mode=0, edge_map=None, attempts=100, max_sets=200, seed=1234, full_attempts=100):
    """One oracle trial of `attempts` base attempts on `cloud` = (scene pos, nrm, prob, pixel, model pos, nrm); spans scaled to
    `full_attempts` when fewer were run (a Cm trial takes the CPU about a minute)."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.set_index_build_threads(usable_cpus())
    t = time.perf_counter()
    orc = pyoracle.Oracle(*cloud, build_index=True)
    t_idx = time.perf_counter() - t
    if mode and edge_map is not None:
        orc.set_edge_map(edge_map)
    t = time.perf_counter()
    r = orc.run(seed, attempts, max_sets, instance_mode=bool(mode))
    dt = time.perf_counter() - t
    sc = float(full_attempts) / max(attempts, 1)
    return {"kind": "port", "cores": 1, "cpu_model": cpu_model(), "attempts_run": attempts, "scaled_to_attempts": full_attempts,
            "index_build_s_not_in_any_span": t_idx,
            "spans_ms_sample_congruent_verify": [r.t_sample_s * sc * 1e3, r.t_congruent_s * sc * 1e3, r.t_verify_s * sc * 1e3],
            "trial_ms": dt * sc * 1e3, "trials_per_s": 1.0 / max(dt * sc, 1e-12),
            "bases_quads_candidates_of_the_run": [int(r.n_bases), int(r.n_quads_total), int(r.n_candidates)],
            "candidate_poses_per_s_phases_2_4": r.n_candidates / max(r.t_congruent_s + r.t_verify_s, 1e-12),
            "best_lcp": float(r.best_lcp)}
