"""The patch test of the scoring kernels (lcp.hip: 64-point steps of the model whose bounding sphere is out of reach of
the scene are skipped after one look-up in a distance field).  The reference walks every model point of every candidate
(stocs.cpp:1016-1035), so the test must never change a result: scores are compared BITWISE with the test switched off,
per-point matches with the oracle, and the two tables the test relies on are checked for what the kernel assumes of them
(spheres contain their points; field values never exceed the true distance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LCP_TOL = 1e-5


def _setup(name, oracle_lib=None):
    from model_matching_amd import synth
    from model_matching_amd.estimator import StocsEstimator
    m, s, k = synth.workload(name)
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    orc = oracle_lib.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False) if oracle_lib else None
    cs = est.get_scene_centroid().astype(np.float64); cm = est.get_model_centroid().astype(np.float64)
    Tgt = synth.centred_gt(s.T_gt, cs, cm)
    return m, s, k, est, orc, Tgt


def _random_poses(rng, n, centre, spread):
    """rigid transforms (column-major 16 floats) that put the model anywhere within `spread` of `centre`"""
    from model_matching_amd import synth
    T = np.zeros((n, 4, 4))
    for i in range(n):
        T[i, :3, :3] = synth.random_rotation(rng)
        T[i, :3, 3] = centre + rng.uniform(-spread, spread, 3)
        T[i, 3, 3] = 1.0
    return np.ascontiguousarray(T.transpose(0, 2, 1).reshape(n, 16).astype(np.float32))


@pytest.mark.parametrize("name", ["tiny", "small", "dense"])
def test_spheres_contain_their_points_and_field_is_a_lower_bound(name):
    from scipy.spatial import cKDTree
    m, s, k, est, _, _ = _setup(name)
    patches, perm, g, dist = est.cull_state()
    nM = len(m.pos)
    assert len(patches) == (nM + 63) // 64 and sorted(perm.tolist()) == list(range(nM))
    mp = (m.pos - est.get_model_centroid()).astype(np.float32).astype(np.float64)   # centroid_shift in float, as the library does
    for j, (cx, cy, cz, r) in enumerate(patches.astype(np.float64)):
        pts = mp[perm[64 * j: 64 * j + 64]]
        assert np.linalg.norm(pts - [cx, cy, cz], axis=1).max() <= r + 1e-7
    assert dist is not None and dist.shape == (g["dims"][2], g["dims"][1], g["dims"][0])
    sp = (s.pos - est.get_scene_centroid()).astype(np.float32).astype(np.float64)
    # the box reaches cap + one cell beyond the scene on every side
    o = g["origin"].astype(np.float64); hi = o + np.array(g["dims"]) * g["g"]
    assert np.all(o <= sp.min(0) - g["cap"] - 0.999 * g["g"]) and np.all(hi >= sp.max(0) + g["cap"] + 0.999 * g["g"])
    tree = cKDTree(sp)
    rng = np.random.default_rng(3)
    idx = np.stack([rng.integers(0, g["dims"][a], 200000) for a in range(3)], axis=1)
    centres = o + (idx + 0.5) * g["g"]
    true_d, _ = tree.query(centres)
    got = dist[idx[:, 2], idx[:, 1], idx[:, 0]].astype(np.float64)
    assert np.all(got <= np.minimum(true_d, g["cap"]) + 1e-9)                     # never above the truth
    assert np.all(np.minimum(true_d, g["cap"]) - got <= 2e-6 + 1e-6 * true_d)     # and exact up to the rounding margin
    assert (got < g["cap"]).mean() > 0.001


@pytest.mark.parametrize("name", ["tiny", "small", "dense"])
def test_scores_are_bitwise_the_same_with_and_without_the_patch_test(name, oracle_lib):
    from model_matching_amd import synth
    m, s, k, est, orc, Tgt = _setup(name, oracle_lib)
    rng = np.random.default_rng(11)
    near = synth.make_candidates(Tgt, 256)
    anywhere = _random_poses(rng, 256, Tgt[:3, 3], 0.25)
    # transforms that are not rigid: scaled, sheared, mirrored, flattened (the sphere bound has to hold for them too)
    odd = near[:64].reshape(64, 4, 4).copy()   # [k, col, row]
    for i in range(64):
        A = odd[i, :3, :3].T.astype(np.float64)
        kind = i % 4
        if kind == 0: A = A * rng.uniform(0.3, 2.5)
        elif kind == 1: A = A @ (np.eye(3) + rng.uniform(-0.6, 0.6, (3, 3)))
        elif kind == 2: A = A @ np.diag([1.0, -1.0, 1.0])
        else: A = A @ np.diag([1.0, 1.0, 1e-3])
        odd[i, :3, :3] = A.T.astype(np.float32)
    odd = odd.reshape(64, 16)
    # not finite / absurd
    bad = near[:8].copy()
    bad[0, 12] = np.nan; bad[1, 0] = np.inf; bad[2, 13] = -np.inf; bad[3, 14] = 3e38; bad[4, 5] = 1e30; bad[5, :] = 0.0; bad[6, 12:15] = [1e6, -1e6, 1e6]
    T = np.concatenate([near, anywhere, odd, bad])
    est.set_option("lcp_cull", 0)
    off = est.score_transforms(T)
    est.set_option("lcp_cull", 2)
    on = est.score_transforms(T)
    assert np.array_equal(off.view(np.uint32), on.view(np.uint32))
    ref = orc.lcp_batch(T[: 512 + 64], nthreads=4)
    assert np.abs(on[: 512 + 64] - ref).max() <= LCP_TOL
    assert (on[:256] > 0).any() and (on[256:512] == 0).any()
    # the other way of launching (one wavefront per candidate), and the processing order, see the same steps
    est.set_option("lcp_split", 0)
    assert np.array_equal(est.score_transforms(T).view(np.uint32), off.view(np.uint32))
    est.set_option("lcp_split", 1)
    # per-point results with the test on: skipped steps report "no neighbour, not counted" exactly where the oracle does
    for c in [0, 3, 255, 256, 300, 511, 512, 513, 514, 515]:
        hg, cg = est.lcp_detail(T[c])
        ho, co = orc.lcp_detail(T[c])
        same = hg == ho
        assert np.all((hg[~same] >= 0) & (ho[~same] >= 0))   # only exact-distance ties may differ (Q11)
        if same.all():
            assert np.array_equal(cg, co)
    est.set_option("lcp_cull", 0)
    hg0, cg0 = est.lcp_detail(T[300])
    est.set_option("lcp_cull", 2)
    hg1, cg1 = est.lcp_detail(T[300])
    assert np.array_equal(hg0, hg1) and np.array_equal(cg0, cg1)


def test_metric_size_bitwise_and_default_policy(oracle_lib):
    """Cm (20 000 / 5 000): 36 864 candidates of the bench batch + 4 096 poses anywhere in the scene (2e8 point queries per call), test
    on / off bitwise equal; the default policy (field filled once the scene has seen 1e9 point queries) allocates nothing when it
    engages and changes no score; a new scene starts over."""
    from model_matching_amd import synth, capi
    m, s, k, est, orc, Tgt = _setup("Cm", oracle_lib)
    rng = np.random.default_rng(5)
    T = np.concatenate([synth.make_candidates(Tgt, 36864), _random_poses(rng, 4096, Tgt[:3, 3], 0.3)])
    runs = [est.score_transforms(T)]          # default policy, first call on this scene: not yet
    a0 = capi.load().stocs_device_alloc_count()
    for _ in range(6):                        # the fifth call crosses 1e9: the field is filled and used from there on
        runs.append(est.score_transforms(T))
    assert capi.load().stocs_device_alloc_count() == a0
    est.set_option("lcp_cull", 0)
    off = est.score_transforms(T)
    for got in runs:
        assert np.array_equal(got.view(np.uint32), off.view(np.uint32))
    ref = orc.lcp_batch(T[:512], nthreads=8)
    assert np.abs(runs[-1][:512] - ref).max() <= LCP_TOL
    # a new frame (the same cloud shifted): scores follow the scene, the field is rebuilt for it
    est.set_option("lcp_cull", 2)
    shift = np.array([0.013, -0.007, 0.021], np.float32)
    est.set_scene(s.pos + shift, s.nrm, s.prob, s.pixel)
    moved = est.score_transforms(T[:8192])
    est.set_option("lcp_cull", 0)
    assert np.array_equal(moved.view(np.uint32), est.score_transforms(T[:8192]).view(np.uint32))
    # (the centred scene is the same cloud up to the float rounding of the shift: the scores stay close to the old ones)
    assert np.abs(moved - off[:8192]).max() < 0.05


def test_threshold_option_and_early_fill_on_a_frame_stream():
    """stocs_set_option("lcp_cull_after"): the threshold of the default policy, in millions of point queries; and when the frame
    BEFORE the current one crossed it, stocs_ctx_set_scene fills the new frame's field at once on the auxiliary stream (the first
    scoring call waits for it through an event).  Scores never change: compared bitwise with the test off."""
    from model_matching_amd import capi, synth
    m, s, k, est, _, Tgt = _setup("small")
    T = synth.make_candidates(Tgt, 4096)
    _, _, _, dist0 = est.cull_state(with_field=False)
    est.set_option("lcp_cull", 0)
    off = est.score_transforms(T)
    est.set_option("lcp_cull", 1)
    est.set_option("lcp_cull_after", 0)                       # from the first call on
    assert np.array_equal(est.score_transforms(T).view(np.uint32), off.view(np.uint32))
    est.set_option("lcp_cull_after", 1)                       # one million point queries: this batch alone crosses it
    for shift in ([0.004, 0.0, -0.003], [0.0, 0.006, 0.002], [-0.005, 0.001, 0.0]):      # three more frames of the stream
        est.set_scene(s.pos + np.array(shift, np.float32), s.nrm, s.prob, s.pixel)         # (the frame before was warm: filled here, on the auxiliary stream)
        got = est.score_transforms(T)
        est.set_option("lcp_cull", 0)
        ref = est.score_transforms(T)
        est.set_option("lcp_cull", 1)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), shift
        assert ref.max() > 0.05
    # two frames in a row without any scoring in between: the pending fill of the first is waited for before its memory is recycled
    est.set_scene(s.pos, s.nrm, s.prob, s.pixel)
    est.set_scene(s.pos, s.nrm, s.prob, s.pixel)
    assert np.array_equal(est.score_transforms(T).view(np.uint32), off.view(np.uint32))
    with pytest.raises(capi.StocsError):
        est.set_option("lcp_cull_after", -1)


@pytest.mark.parametrize("name,n", [("Cm", 2048), ("C5", 256)])
def test_step_lists_longer_than_one_ballot(name, n):
    """A wavefront tests 64 steps per ballot: with one wavefront per candidate (lcp_split 0) the 79 steps of the Cm model take two
    rounds, the 782 of the C5 model thirteen (four with the default four wavefronts per candidate).  Scores bitwise equal in
    all four combinations of the split and the patch test."""
    from model_matching_amd import synth
    m, s, k, est, _, Tgt = _setup(name)
    T = synth.make_candidates(Tgt, n)
    outs = {}
    for split in (1, 0):
        for cull in (0, 2):
            est.set_option("lcp_split", split); est.set_option("lcp_cull", cull)
            outs[(split, cull)] = est.score_transforms(T)
    base = outs[(1, 0)]
    assert base.max() > 0.3
    for key, v in outs.items():
        assert np.array_equal(v.view(np.uint32), base.view(np.uint32)), key
