"""SURVEY.md 8(f) rows 1-2 on the GPU: scene ingest and model preprocessing (ingest.hip) against the
numpy restatement (oracle/ingest_oracle.py) on the raw inputs of the reference's three examples.
Parity with the reference itself is unpinned for these rows (PCL / OpenCV arithmetic absent)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["linemod_obj_06", "packed_dove", "ycb_024_bowl"]


def _rows(a):
    return {tuple(r) for r in np.ascontiguousarray(a).view(np.uint32).reshape(len(a), -1).tolist()}


@pytest.mark.parametrize("name", NAMES)
def test_scene_ingest_equals_numpy_restatement(name):
    from model_matching_amd.estimator import ingest_scene
    raw = np.load(os.path.join(GOLD, "example_%s_raw.npz" % name))
    fix = np.load(os.path.join(GOLD, "example_%s.npz" % name))
    pos, nrm, prob, pix = ingest_scene(raw["depth"], raw["prob"], raw["K"], float(raw["depth_scale"]), normal_method=1)   # the fixtures' plane-fit normals
    # same points in the same (ascending voxel index) order: voxel centroids, class probabilities and pixels bit for bit,
    # normals to the last digits (the 3x3 eigen solvers differ in their roundings)
    assert len(pos) == len(fix["scene_pos"])
    assert np.array_equal(pos, fix["scene_pos"]) and np.array_equal(prob, fix["scene_prob"]) and np.array_equal(pix, fix["scene_pixel"])
    assert np.abs(nrm.astype(np.float64) - fix["scene_nrm"].astype(np.float64)).max() < 1e-6
    assert (prob >= np.float32(0.1)).all() and pos[:, 2].min() > 0 and pos[:, 2].max() <= 2.0
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-5 and ((nrm * pos).sum(1) <= 0).all()   # toward the camera


@pytest.mark.parametrize("name", NAMES)
def test_scene_ingest_with_depth_gradient_normals_equals_numpy_restatement(name):
    """The default since round 3: surface normals by the published LINEMOD method (what the reference asks OpenCV for,
    rgbd.cpp:203) instead of the plane fit.  GPU against oracle/ingest_oracle.py::depth_normals_gradient on the raw images of the
    three examples: the same points (a point is dropped when its pixel has no normal), normals to float rounding.  Parity with
    OpenCV's own implementation stays unpinned (library absent)."""
    from model_matching_amd.estimator import ingest_scene
    from oracle.ingest_oracle import ingest_scene as ingest_ref
    raw = np.load(os.path.join(GOLD, "example_%s_raw.npz" % name))
    pos, nrm, prob, pix = ingest_scene(raw["depth"], raw["prob"], raw["K"], float(raw["depth_scale"]))
    rpos, rnrm, rprob, rpix = ingest_ref(raw["depth"], raw["prob"], raw["K"], float(raw["depth_scale"]), normal_method=0)
    assert len(pos) == len(rpos) > 1000
    assert np.array_equal(pos, rpos) and np.array_equal(prob, rprob) and np.array_equal(pix, rpix)
    assert np.abs(nrm.astype(np.float64) - rnrm.astype(np.float64)).max() < 2e-6
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-5 and (nrm[:, 2] <= 0).all()      # unit length, toward the camera
    # the two estimators look at the same surfaces through different windows (8 neighbours at +-5 pixels against a 5x5 patch) of
    # depth images quantised to 0.1 - 1 mm: they point the same way, and differ by 15-20 degrees in the median on these frames
    fix = np.load(os.path.join(GOLD, "example_%s.npz" % name))
    key = {tuple(p): i for i, p in enumerate(fix["scene_pixel"].tolist())}
    both = [(i, key[tuple(p)]) for i, p in enumerate(pix.tolist()) if tuple(p) in key]
    assert len(both) > 0.8 * min(len(pix), len(fix["scene_pixel"]))
    a = nrm[[i for i, _ in both]]; b = fix["scene_nrm"][[j for _, j in both]]
    ang = np.degrees(np.arccos(np.clip((a * b).sum(1), -1, 1)))
    assert np.median(ang) < 30.0 and (ang < 90.0).mean() > 0.9


@pytest.mark.parametrize("name", NAMES)
def test_model_preprocess_equals_numpy_restatement(name):
    from model_matching_amd.estimator import preprocess_model
    raw = np.load(os.path.join(GOLD, "example_%s_raw.npz" % name))
    fix = np.load(os.path.join(GOLD, "example_%s.npz" % name))
    pos, nrm = preprocess_model(raw["model_raw"], float(raw["normal_radius"]), float(raw["model_voxel"]), float(raw["model_scale"]))
    assert len(pos) == len(fix["model_pos"])
    assert np.abs(pos - fix["model_pos"]).max() <= 1e-7 * max(1.0, np.abs(pos).max()) * 10
    ang = np.degrees(np.arccos(np.clip((nrm * fix["model_nrm"]).sum(1), -1, 1)))
    assert np.percentile(ang, 99) < 0.05 and (ang < 1.0).mean() > 0.995      # eigen solvers differ only on near-degenerate patches


def test_ingest_then_match_runs_end_to_end():
    """raw depth/probability images + raw model vertices -> pose, all stages on the GPU."""
    from model_matching_amd.estimator import StocsEstimator, ingest_scene, preprocess_model
    raw = np.load(os.path.join(GOLD, "example_packed_dove_raw.npz"))
    spos, snrm, sprob, spix = ingest_scene(raw["depth"], raw["prob"], raw["K"], float(raw["depth_scale"]))
    mpos, mnrm = preprocess_model(raw["model_raw"], float(raw["normal_radius"]), float(raw["model_voxel"]), float(raw["model_scale"]))
    est = StocsEstimator(spos, snrm, sprob, spix, mpos, mnrm, build_index=True)
    valid, ids, inv = est.sample_bases(7, 100)
    assert valid.sum() > 30 and est.find_congruent_all() > 1000
    assert est.make_transforms(200, 7) > 500
    lcp, idx, pose = est.compute_best_transform()
    assert idx >= 0 and lcp > 0.15


def test_icp_refines_a_perturbed_pose():
    """stocs_icp_point_to_plane vs the numpy restatement, and as a refinement step: a pose that is off by
    4 mm / 3 degrees is pulled back onto the model."""
    from model_matching_amd import synth
    from model_matching_amd.estimator import icp_point_to_plane
    from oracle.ingest_oracle import icp as icp_ref
    m = synth.make_model(3000, seed=77)
    rng = np.random.default_rng(5)
    ang = np.deg2rad(3.0)
    R = synth._rot_axis_angle(np.array([[0.3, -0.5, 0.8]]), np.array([ang]))[0]
    t = np.array([0.004, -0.002, 0.001])
    seg_idx = rng.permutation(len(m.pos))[:1200]
    src = (m.pos[seg_idx].astype(np.float64) @ R.T + t + rng.normal(0, 0.0003, (1200, 3))).astype(np.float32)
    T_gpu, nc = icp_point_to_plane(src, m.pos, m.nrm, 5, 0.035)
    T_ref, nc_ref = icp_ref(src, m.pos, m.nrm, 5, 0.035)
    assert nc == nc_ref == 1200
    assert np.abs(T_gpu - T_ref).max() < 2e-5
    aligned = src.astype(np.float64) @ T_gpu[:3, :3].T.astype(np.float64) + T_gpu[:3, 3]
    before = np.linalg.norm(src - m.pos[seg_idx], axis=1).mean()
    after = np.linalg.norm(aligned - m.pos[seg_idx], axis=1).mean()
    assert before > 0.003 and after < 0.0008
    # too few correspondences: identity comes back, no crash
    far = src + np.float32(5.0)
    T_far, nfar = icp_point_to_plane(far, m.pos, m.nrm, 5, 0.035)
    assert nfar == 0 and np.array_equal(T_far, np.eye(4, dtype=np.float32))


def test_a_frame_stream_over_contexts_on_threads_equals_the_frames_one_after_the_other():
    """The upstream rows and the path together, as a frame stream runs them: depth image -> stocs_ingest_scene -> stocs_ctx_set_scene (grid
    build) -> one trial, with consecutive frames on DIFFERENT contexts driven by different host threads, so that frame k + 1's ingest and grid
    build overlap frame k's trial (tools/frame_latency.py --stream: 1 100 frames/s against 540 on one context).  Every frame's pose must be the
    one the same frame gives alone (reference: one process per frame, src/stocs_match_one_object.cpp:187-215; src/rgbd.cpp:179-281)."""
    from concurrent.futures import ThreadPoolExecutor
    from model_matching_amd.estimator import StocsEstimator, ingest_scene, preprocess_model
    raw = np.load(os.path.join(GOLD, "example_ycb_024_bowl_raw.npz"))
    K = [float(x) for x in raw["K"]]
    depth, cprob, dscale = np.ascontiguousarray(raw["depth"]), np.ascontiguousarray(raw["prob"]), float(raw["depth_scale"])
    mpos, mnrm = preprocess_model(raw["model_raw"], float(raw["normal_radius"]), float(raw["model_voxel"]), float(raw["model_scale"]))
    frames = [np.ascontiguousarray(np.roll(depth, f, axis=1)) for f in range(4)]     # four different frames

    def one_frame(est, f):
        p, n, pr, px = ingest_scene(frames[f % 4], cprob, K, dscale)
        est.set_scene(p, n, pr, px)
        est.sample_bases(100 + f, 100, mode=0, dispersion=0.9)
        nq = est.find_congruent_all(); nc = est.make_transforms(200, 100 + f)
        lcp, idx, pose = est.compute_best_transform()
        return len(p), int(nq), int(nc), float(lcp), int(idx), pose.tobytes()

    p0, n0, pr0, px0 = ingest_scene(depth, cprob, K, dscale)
    ests = [StocsEstimator(p0, n0, pr0, px0, mpos, mnrm, build_index=True) for _ in range(3)]
    seq = [one_frame(ests[0], f) for f in range(12)]
    assert len({s[:4] for s in seq}) >= 4                                      # the frames really differ
    with ThreadPoolExecutor(3) as ex:
        parts = list(ex.map(lambda k: [(f, one_frame(ests[k], f)) for f in range(k, 12, 3)], range(3)))
    got = dict(x for part in parts for x in part)
    assert all(got[f] == seq[f] for f in range(12))
    for e in ests:
        e.close()
