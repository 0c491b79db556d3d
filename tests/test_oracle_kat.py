"""Known-answer tests that pin the CPU oracle (oracle/stocs_oracle.cpp).

The reference (kuwt/model_matching) holds no tests or golden vectors, so the
expected values here are derived by hand from the reference source lines cited in each test.
"""
import ctypes as C
import math

import numpy as np
import pytest


def test_closest_bin(oracle_lib):
    # rgbd.cpp:85-97: nearest multiple, remainder >= 3 (of 5) goes up
    cb = oracle_lib.closest_bin
    assert [cb(v, 5) for v in (0, 1, 2, 3, 4, 5, 7, 8, 12, 13, 178, 180)] == [0, 0, 0, 5, 5, 5, 5, 10, 10, 15, 180, 180]


def test_ppf_axis_aligned(oracle_lib):
    # rgbd.cpp:99-121: u = p1 - p2; f0 = int(|u|*1000); angles via atan2(|n x u|, n.u) in degrees
    k = oracle_lib.ppf_compute([0, 0, 0], [0, 0, 1], [0.1, 0, 0], [0, 0, 1])
    assert k.tolist() == [100, 90, 90, 0]
    # u is not negated for the second normal => PPF(a,b) != PPF(b,a) in general
    a = oracle_lib.ppf_compute([0, 0, 0], [1, 0, 0], [0.1, 0, 0], [0, 0, 1])   # u=(-.1,0,0): n1.u<0 -> 180
    b = oracle_lib.ppf_compute([0.1, 0, 0], [0, 0, 1], [0, 0, 0], [1, 0, 0])   # u=(+.1,0,0)
    assert a.tolist() == [100, 180, 90, 90]
    assert b.tolist() == [100, 90, 0, 90]
    # truncation then rounding: |u| = 0.0079 -> int 7 -> bin 5 ; 0.0081 -> 8 -> 10
    assert oracle_lib.ppf_compute([0, 0, 0], [0, 0, 1], [0.0079, 0, 0], [0, 0, 1])[0] == 5
    assert oracle_lib.ppf_compute([0, 0, 0], [0, 0, 1], [0.0081, 0, 0], [0, 0, 1])[0] == 10
    # 45 degree case
    s = math.sqrt(0.5)
    assert oracle_lib.ppf_compute([0, 0, 0], [s, 0, s], [0, 0, -0.05], [0, 0, 1])[1:].tolist() == [45, 0, 45]


def test_ppf_float_vs_double_interpretation(oracle_lib, tiny):
    """`atan2(float,float)` resolves to the double or float overload depending on the toolchain;
    the two readings may differ only when an angle sits within one float ulp of an integer degree."""
    m, s, k, o = tiny
    rng = np.random.default_rng(5)
    n = oracle_lib.normalize_rows(s.nrm)
    ii = rng.integers(0, len(s.pos), 20000)
    jj = rng.integers(0, len(s.pos), 20000)
    diff = 0
    for i, j in zip(ii, jj):
        if i == j:
            continue
        a = oracle_lib.ppf_compute(s.pos[i], n[i], s.pos[j], n[j], mode=0)
        b = oracle_lib.ppf_compute(s.pos[i], n[i], s.pos[j], n[j], mode=1)
        diff += int((a != b).any())
    assert diff <= 2   # flip rate <= 1e-4 per feature; recorded in DESIGN.md


def test_index_literal_equals_query_form(oracle_lib):
    """rgbd.cpp:123-154 (128-key insertion) == store-once + 128-offset lookup, incl. result order."""
    from model_matching_amd import synth
    m = synth.make_model(70, seed=99)
    nrm = oracle_lib.normalize_rows(m.nrm)
    lit = oracle_lib.Index(m.pos, nrm, literal=True)
    qf = oracle_lib.Index(m.pos, nrm, literal=False)
    assert qf.num_pairs() == 70 * 69
    rng = np.random.default_rng(0)
    keys = set()
    for i in range(70):
        for j in range(70):
            if i != j:
                f = oracle_lib.ppf_compute(m.pos[i], nrm[i], m.pos[j], nrm[j])
                for d0 in (-5, 0):
                    keys.add((f[0] + d0, f[1] + 5 * int(rng.integers(-2, 2)), f[2] + 5 * int(rng.integers(-2, 2)),
                              f[3] + 5 * int(rng.integers(-2, 2))))
    for _ in range(300):
        keys.add((int(rng.integers(0, 40)) * 5, int(rng.integers(-1, 38)) * 5, int(rng.integers(-1, 38)) * 5,
                  int(rng.integers(-1, 38)) * 5))
    n_nonempty = 0
    for key in sorted(keys):
        a = lit.lookup(key)
        b = qf.lookup(key)
        assert a.shape == b.shape and (a == b).all(), key
        assert lit.exists(key) == qf.exists(key)
        n_nonempty += len(a) > 0
    assert n_nonempty > 100
    # keys with distance bin <= 5 mm or a negative angle bin are never stored (rgbd.cpp:136)
    assert len(qf.lookup((5, 90, 90, 0))) == 0 and len(lit.lookup((5, 90, 90, 0))) == 0
    assert len(qf.lookup((50, -5, 90, 0))) == 0


def test_kdtree_equals_brute_force(oracle_lib, tiny):
    m, s, k, o = tiny
    pos = o.scene_centred()
    rng = np.random.default_rng(1)
    sq = np.float32(0.005) * np.float32(0.005)
    nhit = 0
    for t in range(3000):
        base = pos[rng.integers(0, len(pos))]
        q = (base + rng.normal(0, 0.003, 3)).astype(np.float32)
        a = o.nn(q, sq)
        b, ties = o.nn_brute(q, sq)
        if ties == 0:
            assert a == b
        else:
            assert (a >= 0) == (b >= 0)
        nhit += a >= 0
    assert 500 < nhit < 3000
    # a scene point queries itself at distance 0; inclusive radius (kdtree.h:424 `<=`)
    assert o.nn(pos[17], np.float32(1e-12)) == 17
    # sqdist = 0: the root test `qnode.sq < cl_dist` (kdtree.h:410) is 0 < 0 -> nothing is visited
    assert o.nn(pos[17], np.float32(0.0)) == -1
    far = np.array([10, 10, 10], np.float32)
    assert o.nn(far, sq) == -1


def test_segment_invariants(oracle_lib):
    L = oracle_lib.lib()
    fp = C.POINTER(C.c_float)
    def call(p1, p2, q1, q2):
        arrs = [np.array(v, np.float32) for v in (p1, p2, q1, q2)]
        i1, i2 = C.c_double(), C.c_double()
        d = L.orc_segment_distance_and_invariants(*[a.ctypes.data_as(fp) for a in arrs], C.byref(i1), C.byref(i2))
        return d, i1.value, i2.value
    # two segments crossing at their midpoints, 1 cm apart in z  (stocs.cpp:155-222)
    d, i1, i2 = call([-1, 0, 0], [1, 0, 0], [0, -1, 0.01], [0, 1, 0.01])
    assert abs(d - 0.01) < 1e-7 and i1 == 0.5 and i2 == 0.5
    # crossing at 1/4 and 3/4
    d, i1, i2 = call([0, 0, 0], [4, 0, 0], [1, -3, 0], [1, 1, 0])
    assert d < 1e-6 and abs(i1 - 0.25) < 1e-7 and abs(i2 - 0.75) < 1e-7
    # parallel segments: f < kSmallNumber branch -> s1 = 0
    d, i1, i2 = call([0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0])
    assert i1 == 0.0 and abs(d - 1.0) < 1e-7


def test_try_sampled_base_orders_crossing_diagonals(oracle_lib, tiny):
    m, s, k, o = tiny
    ok, ids, inv = o.sample_class_base(7, 3)
    if ok:
        ok2, ids2, inv2 = o.try_sampled_base(ids)
        assert ok2 and sorted(ids2.tolist()) == sorted(ids.tolist())


def test_rigid_transform_recovers_known_pose(oracle_lib, tiny):
    """stocs.cpp:270-361: frames from the first three correspondences, T = Tr(c1) R Tr(-c2)."""
    from model_matching_amd import synth
    m, s, k, _ = tiny
    rng = np.random.default_rng(3)
    R = synth.random_rotation(rng)
    t = np.array([0.3, -0.2, 0.9])
    scene = (m.pos.astype(np.float64) @ R.T + t).astype(np.float32)
    o = oracle_lib.Oracle(scene, (m.nrm.astype(np.float64) @ R.T).astype(np.float32), np.ones(len(scene), np.float32),
                          None, m.pos, m.nrm, build_index=False)
    ids = np.array([5, 77, 140, 201], np.int32)
    ok, T, P = o.rigid_transform(ids, ids)   # scene point i corresponds to model point i
    assert ok
    P = P.reshape(4, 4).T
    assert np.abs(P[:3, :3] - R).max() < 2e-5
    assert np.abs(P[:3, 3] - t).max() < 2e-5
    # the centred transform maps centred model onto centred scene
    T = T.reshape(4, 4).T
    mc, sc = o.model_centred(), o.scene_centred()
    assert np.abs(mc @ T[:3, :3].T + T[:3, 3] - sc).max() < 2e-5
    # degenerate triple (two identical model points): rejected (deliberate divergence Q2)
    ok, _, _ = o.rigid_transform(ids, np.array([5, 5, 140, 201], np.int32))
    assert not ok


def test_lcp_identity_is_mean_class_probability(oracle_lib):
    """stocs.cpp:1006-1041 with T = I on a cloud against itself: every point finds itself
    (distance 0, angle 0) so the score is sum(class_prob)/|M|."""
    from model_matching_amd import synth
    m = synth.make_model(300, seed=5)
    prob = np.linspace(0.2, 1.0, 300).astype(np.float32)
    o = oracle_lib.Oracle(m.pos, m.nrm, prob, None, m.pos, m.nrm, build_index=False)
    T = np.eye(4, dtype=np.float32).reshape(16)
    hit, counted = o.lcp_detail(T)
    assert (hit >= 0).all()
    expect = np.float32(0)
    cp = o.scene_class_prob()
    for i in range(300):
        if counted[i]:
            expect = np.float32(expect + cp[hit[i]])
    assert o.lcp(T) == np.float32(expect / np.float32(300))
    # Q7: n.n of a float-normalised normal can round to 1.0000001 -> acos = NaN -> NOT counted
    n = oracle_lib.normalize_rows(m.nrm)
    d = n[:, 0] * n[:, 0] + (n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2])
    assert (counted.astype(bool) == (d <= np.float32(1.0))).all()
    assert 150 < counted.sum() < 300
    # flipped normals: angle 180 > 30 -> nothing counted (no abs / flip, stocs.cpp:1030 commented out)
    o2 = oracle_lib.Oracle(m.pos, -m.nrm, prob, None, m.pos, m.nrm, build_index=False)
    assert o2.lcp(T) == 0.0


def test_normal_angle_predicates(oracle_lib):
    L = oracle_lib.lib()
    c30 = math.cos(math.radians(30))
    assert L.orc_normal_compatible(1.0) == 1
    assert L.orc_normal_compatible(np.float32(c30 + 1e-4)) == 1
    assert L.orc_normal_compatible(np.float32(c30 - 1e-4)) == 0
    assert L.orc_normal_compatible(np.float32(1.0000001)) == 0     # acos -> NaN -> not counted (Q7)
    assert L.orc_normal_compatible(-1.0) == 0
    # internal angle: min(a, 180-a) < 30 -> reject (stocs.cpp:428-429,440)
    assert L.orc_internal_angle_reject(np.float32(c30 + 1e-4), 30.0) == 1
    assert L.orc_internal_angle_reject(np.float32(c30 - 1e-4), 30.0) == 0
    assert L.orc_internal_angle_reject(np.float32(-c30 - 1e-4), 30.0) == 1
    assert L.orc_internal_angle_reject(0.0, 30.0) == 0
    assert L.orc_internal_angle_reject(np.float32(1.0000001), 30.0) == 0   # NaN never rejects


def test_normalset_constants(oracle_lib):
    L = oracle_lib.lib()
    # normalset.hpp:178-180: alpha = acos(cos), perimeter = 2*pi*atan(alpha), nb = 2*ceil(perimeter*7/2)
    assert L.orc_cone_samples(0.0) == 46          # alpha = 90 deg
    assert L.orc_cone_samples(1.0) == 0           # alpha = 0 -> no samples -> no neighbours (Q9)
    a = math.acos(0.5)
    assert L.orc_cone_samples(0.5) == 2 * math.ceil(2 * math.pi * math.atan(a) * 7 / 2)
    gd, eg, cell = C.c_int(), C.c_int(), C.c_float()
    L.orc_normalset_params(0.0294, C.byref(gd), C.byref(eg), C.byref(cell))   # normalset.h:114-122
    assert (gd.value, eg.value, cell.value) == (5, 32, 1 / 32)
    L.orc_normalset_params(0.03125, C.byref(gd), C.byref(eg), C.byref(cell))
    assert eg.value == 32
    L.orc_normalset_params(0.0313, C.byref(gd), C.byref(eg), C.byref(cell))
    assert eg.value == 16
    # direction cell: int((n/2+0.5)/(1/7+1e-5)) per axis, x fastest (normalset.h:100-104, utils.h:139-148)
    def cell_of(n):
        n = np.array(n, np.float32)
        return L.orc_index_normal(n.ctypes.data_as(C.POINTER(C.c_float)))
    assert cell_of([0, 0, 1]) == 6 * 49 + 3 * 7 + 3
    assert cell_of([1, 0, 0]) == 3 * 49 + 3 * 7 + 6
    assert cell_of([-1, 0, 0]) == 3 * 49 + 3 * 7 + 0


def test_draw(oracle_lib):
    L = oracle_lib.lib()
    w = np.array([0, 0.5, 0, 0.25, 0.25, 0], np.float32)
    fp = w.ctypes.data_as(C.POINTER(C.c_float))
    counts = np.zeros(6, int)
    for k in range(4000):
        counts[L.orc_draw(fp, 6, L.orc_rng(42, 0, k))] += 1
    assert counts[0] == counts[2] == counts[5] == 0
    assert abs(counts[1] / 4000 - 0.5) < 0.03 and abs(counts[3] / 4000 - 0.25) < 0.03
    assert L.orc_draw(fp, 6, 0) == 1 and L.orc_draw(fp, 6, 2 ** 64 - 1) == 4
    z = np.zeros(5, np.float32)
    assert L.orc_draw(z.ctypes.data_as(C.POINTER(C.c_float)), 5, 123) == -1
    assert L.orc_rng(1, 2, 3) == L.orc_rng(1, 2, 3) != L.orc_rng(1, 2, 4)


def test_class_base_sampling_invariants(oracle_lib, tiny):
    m, s, k, o = tiny
    pos = o.scene_centred()
    n_ok = 0
    for attempt in range(30):
        ok, ids, inv = o.sample_class_base(11, attempt)
        ok_again, ids_again, inv_again = o.sample_class_base(11, attempt)
        assert ok == ok_again and (ids == ids_again).all() and (inv == inv_again).all()   # seeded
        if not ok:
            continue
        n_ok += 1
        assert len(set(ids.tolist())) == 4
        assert 0 <= inv[0] <= 1 and 0 <= inv[1] <= 1
        d = np.linalg.norm(pos[ids][:, None] - pos[ids][None], axis=-1)
        assert d[np.triu_indices(4, 1)].min() > 0.005
    assert n_ok >= 15


def test_congruent_sets_contain_ground_truth_like_quads(oracle_lib, tiny):
    """Every returned quad's two pairs must come from the two PPF lookups (stocs.cpp:780-786,864)."""
    m, s, k, o = tiny
    nrm = oracle_lib.normalize_rows(s.nrm)
    pos = o.scene_centred()
    found = 0
    for attempt in range(12):
        ok, ids, inv = o.sample_class_base(5, attempt)
        if not ok:
            continue
        quads = o.find_congruent(ids, float(inv[0]), float(inv[1]))
        if len(quads) == 0:
            continue
        found += 1
        k1 = oracle_lib.ppf_compute(pos[ids[0]], nrm[ids[0]], pos[ids[1]], nrm[ids[1]])
        k2 = oracle_lib.ppf_compute(pos[ids[2]], nrm[ids[2]], pos[ids[3]], nrm[ids[3]])
        P = set(map(tuple, o.index_lookup(k1).tolist()))
        Q = set(map(tuple, o.index_lookup(k2).tolist()))
        assert all((q[0], q[1]) in P and (q[2], q[3]) in Q for q in quads.tolist())
        # std::set ordering by (P index, Q index) == lexicographic in the four ids
        assert quads.tolist() == sorted(quads.tolist())
        assert len(set(map(tuple, quads.tolist()))) == len(quads)
    assert found >= 3


def test_full_run_recovers_pose(oracle_lib, tiny):
    m, s, k, o = tiny
    r = o.run(1234, 100, 200)
    assert r.n_bases > 50 and r.n_candidates > 500 and r.best_index >= 0
    P = np.array(r.best_pose16).reshape(4, 4).T
    R_err = P[:3, :3].T @ s.T_gt[:3, :3]
    ang = math.degrees(math.acos(min(1.0, (np.trace(R_err) - 1) / 2)))
    assert ang < 3.0 and np.linalg.norm(P[:3, 3] - s.T_gt[:3, 3]) < 0.005
    T, Pc, b = o.candidates()
    lcp = o.lcp_batch(T)
    i, sc = oracle_lib.best(lcp)
    assert i == r.best_index and sc == r.best_lcp
    assert oracle_lib.best(np.zeros(5, np.float32)) == (-1, 0.0)      # all-zero -> no pose (Q18)
    assert oracle_lib.best(np.array([0.2, 0.5, 0.5], np.float32))[0] == 1  # first maximum wins


def test_greedy_clustering(oracle_lib):
    """pose_clustering.cpp:79-121."""
    from model_matching_amd import synth
    I = np.eye(4)
    def pose(rz_deg, tx):
        T = I.copy()
        a = math.radians(rz_deg)
        T[:3, :3] = [[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]]
        T[0, 3] = tx
        return T.T.reshape(16)
    poses = np.array([pose(0, 0), pose(2, 0.001), pose(40, 0), pose(0, 0.2), pose(1, 0.0005)], np.float32)
    lcp = np.array([0.9, 0.8, 0.7, 0.6, 0.1], np.float32)
    sym = np.zeros(3, np.float32)
    keep = oracle_lib.greedy_clustering(poses, lcp, 0.5, 0.9, 10, 0.01, 5.0, sym)
    assert keep.tolist() == [0, 2, 3]          # 1 suppressed by 0; 4 below the acceptance fraction
    r, t = oracle_lib.pose_diff(poses[2], poses[0], sym)
    assert abs(r - 40) < 1e-3 and t == 0
    # 360-degree symmetry about z ignores yaw
    keep = oracle_lib.greedy_clustering(poses, lcp, 0.5, 0.9, 10, 0.01, 5.0, np.array([0, 0, 360], np.float32))
    assert keep.tolist() == [0, 3]
    # stop once size > maximum_pose_count (sic: one more than the count is kept)
    keep = oracle_lib.greedy_clustering(poses, lcp, 0.0, 0.9, 1, 0.01, 5.0, sym)
    assert len(keep) == 2


def test_depth_gradient_normals_recover_a_tilted_plane():
    """Known answer for the restated LINEMOD normals (oracle/ingest_oracle.py::depth_normals_gradient, reference rgbd.cpp:203): the
    depth image of a tilted plane n . X = c, rendered through a pin-hole camera in raw units of 0.1 mm, must give the plane's own
    normal (pointed at the camera) at every interior pixel, up to the quantisation of the 16-bit depth; border pixels, zero
    depth and pixels whose 8 neighbours all lie across a depth step give no normal."""
    from oracle.ingest_oracle import depth_normals_gradient
    fx, cx, fy, cy = 600.0, 160.0, 600.0, 120.0
    W, H = 320, 240
    n = np.array([0.2, -0.3, -1.0]); n /= np.linalg.norm(n)        # toward the camera: z < 0
    c = n @ np.array([0.0, 0.0, 0.9])                               # the plane passes through (0, 0, 0.9 m)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    ray = np.stack([(jj - cx) / fx, (ii - cy) / fy, np.ones_like(jj, float)], -1)
    z = c / (ray @ n)                                               # depth along the optical axis
    depth = np.round(z * 10000.0).astype(np.uint16)
    N = depth_normals_gradient(depth, (fx, cx, fy, cy))
    inner = N[5:H - 6, 5:W - 6]
    assert np.isfinite(inner).all() and np.isnan(N[:5]).all() and np.isnan(N[:, :5]).all() and np.isnan(N[H - 6:]).all() and np.isnan(N[:, W - 6:]).all()
    ang = np.degrees(np.arccos(np.clip(inner.reshape(-1, 3).astype(np.float64) @ n, -1, 1)))
    assert ang.max() < 1.0 and np.median(ang) < 0.2
    assert np.abs(np.linalg.norm(inner, axis=-1) - 1).max() < 1e-6 and (inner[..., 2] <= 0).all()
    # a hole (zero depth) and an isolated spike 1 cm in front of the plane: no admissible neighbour -> no normal at the spike
    d2 = depth.copy(); d2[100, 100] = 0; d2[60, 200] -= 100
    N2 = depth_normals_gradient(d2, (fx, cx, fy, cy))
    assert np.isnan(N2[60, 200]).all()
    assert np.isfinite(N2[100, 105]).all()                          # the hole's neighbours drop it (|delta| > 50) and keep their normal
