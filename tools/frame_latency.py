#!/usr/bin/env python3
"""Per-frame latency of the whole path on the reference's three example frames (raw data fixtures under
tests/golden/): uint16 depth + class-probability image -> scene cloud (stocs_ingest_scene) -> stocs_ctx_set_scene
(upload, centroid shift, GPU brick grid; the model and its PPF index stay) -> one StoCS trial of 100 base
attempts (sampling, congruent sets, <= 200 transforms per base, verification) -> best pose.
usage: python tools/frame_latency.py [frames] [--cpu-reference] | --stream [contexts] [frames] [trials per frame]"""
import json
import os
import sys
import time

# The host-side bookkeeping around the trials is a few tiny numpy calls: keep the BLAS / OpenMP pools to one thread.  On the GPU boxes
# the process sees 256 CPUs but its cgroup has a quota of 16; OpenBLAS sizes its pool by the former, its spinning workers exhaust the
# latter, and the kernel then freezes the whole process for the rest of a 100 ms period -- the sporadic 65-80 ms "stall" of rounds
# 1-3 (profiles/r03_stall_root_cause.json).  Must happen before numpy is imported.
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd.estimator import StocsEstimator, ingest_scene, preprocess_model  # noqa: E402
from pose_check import depth_agreement, pose_matrix_from_colmajor16  # noqa: E402  (tools/pose_check.py)


def stream_mode(argv):
    """--stream [contexts] [frames] [trials per frame]: a frame STREAM -- ingest, scene grid and trial(s) of consecutive frames overlap because
    consecutive frames go to different contexts, each driven by its own host thread (the library calls release the GIL; a context owns its
    streams).  Every frame's result is compared with the same frame run alone on a fresh sequential pass; throughput = frames / wall clock."""
    from concurrent.futures import ThreadPoolExecutor
    n_ctx = int(argv[0]) if argv else 3
    n_frames = int(argv[1]) if len(argv) > 1 else 240
    n_trials = int(argv[2]) if len(argv) > 2 else 1
    out = {}
    for name in ("ycb_024_bowl", "linemod_obj_06"):
        raw = np.load(os.path.join(ROOT, "tests", "golden", "example_%s_raw.npz" % name))
        K = [float(x) for x in raw["K"]]
        depth, cprob, dscale = np.ascontiguousarray(raw["depth"]), np.ascontiguousarray(raw["prob"]), float(raw["depth_scale"])
        mpos, mnrm = preprocess_model(raw["model_raw"], float(raw["normal_radius"]), float(raw["model_voxel"]), float(raw["model_scale"]))
        pos, nrm, prob, pix = ingest_scene(depth, cprob, K, dscale)
        # frames differ (a camera moves): frame f sees the depth image shifted by f % 7 columns -- same size, different clouds and results
        frames = [np.ascontiguousarray(np.roll(depth, f % 7, axis=1)) for f in range(7)]

        def one_frame(est, f):
            p_, n_, pr_, px_ = ingest_scene(frames[f % 7], cprob, K, dscale)
            est.set_scene(p_, n_, pr_, px_)
            if n_trials == 1:
                est.sample_bases(100 + f, 100, mode=0, dispersion=0.9)
                est.find_congruent_all(); est.make_transforms(200, 100 + f)
                lcp, idx, pose = est.compute_best_transform()
                return float(lcp), int(idx), pose.tobytes()
            res = est.run_trials([1000 * f + t for t in range(n_trials)], 100, mode=0, dispersion=0.9)
            best = max(res, key=lambda x: x["best_lcp"])
            return float(best["best_lcp"]), int(best["best_index"]), best["best_pose"].tobytes()

        ests = [StocsEstimator(pos, nrm, prob, pix, mpos, mnrm, build_index=True) for _ in range(n_ctx)]
        for e in ests:
            for f in range(3):
                one_frame(e, f)                  # warm: arenas, workspaces
        t0 = time.perf_counter()
        seq = [one_frame(ests[0], f) for f in range(n_frames)]
        t_seq = time.perf_counter() - t0

        def worker(k):
            return [(f, one_frame(ests[k], f)) for f in range(k, n_frames, n_ctx)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(n_ctx) as ex:
            parts = list(ex.map(worker, range(n_ctx)))
        t_par = time.perf_counter() - t0
        got = dict(x for part in parts for x in part)
        same = all(got[f] == seq[f] for f in range(n_frames))
        out[name] = {"contexts_on_threads": n_ctx, "frames": n_frames, "trials_per_frame": n_trials,
                     "sequential_frames_per_s": n_frames / t_seq, "stream_frames_per_s": n_frames / t_par,
                     "every_frame_identical_to_the_sequential_pass": bool(same)}
        for e in ests:
            e.close()
    print(json.dumps(out, indent=1))


def main():
    if "--stream" in sys.argv[1:]:
        return stream_mode([a for a in sys.argv[1:] if a != "--stream"])
    argv = [a for a in sys.argv[1:] if a != "--cpu-reference"]
    with_cpu = "--cpu-reference" in sys.argv[1:]      # also time the reference's CPU path (oracle, one core) on each frame: +6 s
    frames = int(argv[0]) if argv else 6
    out = {}
    for name in ("ycb_024_bowl", "linemod_obj_06", "packed_dove"):
        raw = np.load(os.path.join(ROOT, "tests", "golden", "example_%s_raw.npz" % name))
        fix = np.load(os.path.join(ROOT, "tests", "golden", "example_%s.npz" % name))
        K = [float(x) for x in raw["K"]]
        depth, cprob, dscale = np.ascontiguousarray(raw["depth"]), np.ascontiguousarray(raw["prob"]), float(raw["depth_scale"])   # npz entries decompress on access
        mpos, mnrm = preprocess_model(raw["model_raw"], float(raw["normal_radius"]), float(raw["model_voxel"]), float(raw["model_scale"]))
        pos, nrm, prob, pix = ingest_scene(depth, cprob, K, dscale)
        mode = 1 if "edge_map" in fix.files else 0
        edge = np.ascontiguousarray(fix["edge_map"]) if mode else None
        t = time.perf_counter()
        est = StocsEstimator(pos, nrm, prob, pix, mpos, mnrm, build_index=True)
        est.sync()
        t_ctx = (time.perf_counter() - t) * 1e3
        rows, agree = [], []
        for f in range(frames):
            t0 = time.perf_counter()
            pos, nrm, prob, pix = ingest_scene(depth, cprob, K, dscale)
            t1 = time.perf_counter()
            est.set_scene(pos, nrm, prob, pix)
            if mode:
                est.set_edge_map(edge)
            t2 = time.perf_counter()
            est.sample_bases(100 + f, 100, mode=mode, dispersion=0.9)
            t3 = time.perf_counter()
            nq = est.find_congruent_all()
            nc = est.make_transforms(200, 100 + f)
            lcp, idx, pose = est.compute_best_transform()
            t4 = time.perf_counter()
            rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, int(nq), int(nc), float(lcp)])
            # (outside the clock) the winner against the frame's own depth image and class-probability map: tools/pose_check.py
            agree.append(depth_agreement(pose_matrix_from_colmajor16(pose), mpos, mnrm, depth, cprob, K, dscale) if idx >= 0 else None)
        # 64 independent trials of the same frame in one set of launches (stocs_run_trials), the best of them against the same evidence
        est.run_trials(list(range(1000, 1064)), 100, mode=mode, dispersion=0.9)          # (sizes the arenas of the batched form)
        tb = time.perf_counter()
        res = est.run_trials(list(range(1000, 1064)), 100, mode=mode, dispersion=0.9)
        t_batch = (time.perf_counter() - tb) * 1e3
        best = max(res, key=lambda x: x["best_lcp"])
        batch = {"trials": 64, "ms": t_batch, "trials_per_s": 64e3 / t_batch, "best_lcp_of_the_batch": best["best_lcp"],
                 "median_best_lcp_of_its_trials": float(np.median([x["best_lcp"] for x in res])),
                 "winner_vs_the_frames_own_depth_and_class_maps": depth_agreement(pose_matrix_from_colmajor16(best["best_pose"]), mpos, mnrm, depth, cprob, K, dscale)
                 if best["best_index"] >= 0 else None}
        r = np.array(rows[1:])   # the first frame warms the arenas up
        out[name] = {"mode": "instance" if mode else "class", "scene_points": int(len(pos)), "model_points": int(len(mpos)),
                     "context_create_incl_index_ms": t_ctx,
                     "median_ms": {"ingest": float(np.median(r[:, 0])), "set_scene": float(np.median(r[:, 1])), "sample_100_bases": float(np.median(r[:, 2])),
                                   "congruent+transforms+verify": float(np.median(r[:, 3])), "frame_total": float(np.median(r[:, 4]))},
                     "winner_vs_the_frames_own_depth_and_class_maps": agree[1:],
                     "batch_of_64_trials_in_one_set_of_launches": batch,
                     "quads": [int(x) for x in r[:, 5]], "candidates": [int(x) for x in r[:, 6]], "best_lcp": [float(x) for x in r[:, 7]]}
        if with_cpu:
            import cpu_reference   # tools/cpu_reference.py
            out[name]["cpu_reference_one_trial"] = cpu_reference.spans((pos, nrm, prob, pix, mpos, mnrm), mode, edge, 100, 200, 100, 100)
        est.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
