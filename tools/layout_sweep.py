#!/usr/bin/env python3
"""Where does the centre-sorted (dense) list layout start to pay?  One scene density per call, under the environment's STOCS_GRID_DIV /
STOCS_GRID_DENSE (measurement switches of ctx.hip); prints the kernel time of one scoring launch.
usage: [STOCS_GRID_DIV=d STOCS_GRID_DENSE=0|1] python tools/layout_sweep.py <scene points> <model points> <lattice m> [candidates]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

nS, nM, lattice = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
m = synth.make_model(nM)
s = synth.make_scene(m, nS, lattice=lattice)
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
cs = est.get_scene_centroid().astype(np.float64); cm = est.get_model_centroid().astype(np.float64)
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), K)
dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(K * 4)
est.dev_upload(dT, T)
est.score_device(dT, K, dL); est.sync()
o = np.zeros(K, np.float32); est.dev_download(dL, o)
ms = min(est.time_score_kernel(dT, K, dL, 5) for _ in range(3))
print(nS, nM, lattice, "DIV", os.environ.get("STOCS_GRID_DIV"), "DENSE", os.environ.get("STOCS_GRID_DENSE"), "ms", round(ms, 4), "crc", int(np.bitwise_xor.reduce(o.view(np.uint32))), flush=True)
