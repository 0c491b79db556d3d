"""Dominance-pruned candidate lists of the scene grid (grid.hip, round 5): a listed point that another listed point beats over the
whole (widened) cell box can never be the answer of KdTree::doQueryRestrictedClosestIndex (reference include/super4pcs/accelerators/
kdtree.h:394-459) for a position in that cell, and is dropped.  The claim is exactness: the nearest neighbour of every query, its
index and the tie rule are what they were -- so scores are compared BITWISE between a context with pruned lists (the default) and one
with the layouts of rounds 2-4 (STOCS_GRID_PRUNE=0, which also keeps the early-exit kernels 31 / 39 covered), per-point matches against
the oracle, on the synthetic workloads and on scenes built to sit on the margins: exact lattices (equal distances everywhere), exact
duplicates, points on cell boundaries, millimetre units, far-off coordinates."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, cloud, **kw):
    """(pruned context, unpruned context) on the same clouds."""
    from model_matching_amd.estimator import StocsEstimator
    monkeypatch.setenv("STOCS_GRID_PRUNE", "1")
    a = StocsEstimator(*cloud, build_index=False, **kw)
    monkeypatch.setenv("STOCS_GRID_PRUNE", "0")
    b = StocsEstimator(*cloud, build_index=False, **kw)
    monkeypatch.delenv("STOCS_GRID_PRUNE")
    return a, b


@pytest.mark.parametrize("name", ["tiny", "small", "dense"])
def test_pruned_lists_give_the_scores_of_the_full_lists_bit_for_bit(name, oracle_lib, monkeypatch):
    from model_matching_amd import synth
    m, s, k = synth.workload(name)
    cloud = (s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm)
    pr, un = _pair(monkeypatch, cloud)
    orc = oracle_lib.Oracle(*cloud, build_index=False)
    cs, cm = orc.centroids()
    T = synth.make_candidates(synth.centred_gt(s.T_gt, cs.astype(np.float64), cm.astype(np.float64)), max(k, 1024))
    got_p, got_u = pr.score_transforms(T), un.score_transforms(T)
    assert np.array_equal(got_p.view(np.uint32), got_u.view(np.uint32)) and got_p.max() > 0.05
    ref = orc.lcp_batch(T[:256], nthreads=8)
    assert np.abs(got_p[:256] - ref).max() <= 1e-5
    for c in list(range(8)) + [int(np.argmax(got_p))]:
        hp, cp = pr.lcp_detail(T[c]); hu, cu = un.lcp_detail(T[c])
        assert np.array_equal(hp, hu) and np.array_equal(cp, cu), c           # the same scene point for every model point
        ho, co = orc.lcp_detail(T[c])
        same = hp == ho
        assert same.mean() > 0.999 and np.array_equal(cp[same], co[same]), c  # (only exact-distance ties may differ from the kd-tree, Q11)
    # the early-exit kernels of rounds 2-4 stay in the library as cross-checks: on the unpruned dense grid they still run, same scores
    if name == "dense":
        for v in (0, 31, 39):
            un.set_option("lcp_variant", v)
            assert np.array_equal(un.score_transforms(T).view(np.uint32), got_u.view(np.uint32)), v
        un.set_option("lcp_variant", 99)
    for v in (0, 24):                                                        # the plain lane-per-query kernel reads the pruned lists too
        pr.set_option("lcp_variant", v)
        assert np.array_equal(pr.score_transforms(T).view(np.uint32), got_p.view(np.uint32)), v
    pr.close(); un.close()


def _scene(pos, seed=0):
    rng = np.random.default_rng(seed)
    n = len(pos)
    nrm = np.tile(np.array([0.0, 0.0, 1.0], np.float32), (n, 1))
    prob = rng.uniform(0.2, 1.0, n).astype(np.float32)
    pix = np.stack([np.arange(n) // 640, np.arange(n) % 640], 1).astype(np.int32)
    return pos.astype(np.float32), nrm, prob, pix


def _identity_like(rng, n, max_t):
    """n pure translations of at most max_t per axis (column-major 4x4), the first the identity."""
    T = np.tile(np.eye(4, dtype=np.float32).T.reshape(16), (n, 1))
    T[1:, 12:15] = rng.uniform(-max_t, max_t, (n - 1, 3)).astype(np.float32)
    return T


@pytest.mark.parametrize("unit,offset", [(1.0, 0.0), (1000.0, 0.0), (1.0, 3.0)])
def test_margins_lattices_duplicates_and_cell_boundaries(unit, offset, oracle_lib, monkeypatch):
    """Scenes on which distances tie exactly and queries sit on cell faces: an exact lattice (every midpoint is equally far from 2, 4
    or 8 points), exact duplicates of scene points (the larger index must win), a tight cluster, and model points that ARE scene
    points or lattice midpoints, moved by translations of whole and half lattice steps.  In metres, in millimetres (epsilon 5), and far
    from the origin (float spacing 2.4e-7 at 3 m)."""
    from model_matching_amd.estimator import StocsEstimator  # noqa: F401
    rng = np.random.default_rng(11)
    a = 0.0015625 * unit                                                     # lattice step: a power-of-two fraction, exact in float
    g = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(3), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    lat = g * a
    dup = lat[rng.integers(0, len(lat), 300)]                                # exact duplicates (later indices)
    clu = lat[777] + rng.normal(0, 0.0002 * unit, (400, 3))                  # 400 points inside a fraction of a cell
    far = lat[::7] + np.array([0.0, 0.0, 0.004 * unit])                      # a second sheet 4 mm above: inside epsilon of the first
    pos = np.concatenate([lat, dup, clu, far]) + offset
    mids = lat[rng.integers(0, len(lat), 1500)] + a * rng.integers(0, 2, (1500, 3)) * 0.5     # lattice points and edge / face / body midpoints
    rnd = lat[rng.integers(0, len(lat), 1500)] + rng.uniform(-0.006, 0.006, (1500, 3)) * unit
    mpos = (np.concatenate([lat[rng.integers(0, len(lat), 1000)], mids, rnd]) + offset).astype(np.float32)
    mnrm = np.tile(np.array([0.0, 0.0, 1.0], np.float32), (len(mpos), 1))
    sp, sn, spr, spx = _scene(pos)
    cloud = (sp, sn, spr, spx, mpos, mnrm)
    from model_matching_amd import capi
    prm = capi.default_params()
    prm.distance_threshold = 0.005 * unit
    pr, un = _pair(monkeypatch, cloud, params=prm)
    # the estimator centres both clouds (centroid_shift, stocs.cpp:943-964): candidates that undo the two shifts put the model back on the scene
    cs, cm = pr.get_scene_centroid().astype(np.float64), pr.get_model_centroid().astype(np.float64)
    T = _identity_like(rng, 64, 0.002 * unit)
    steps = rng.integers(-4, 5, (32, 3)) * (a * 0.5)                          # whole and half lattice steps
    T[32:, 12:15] = steps.astype(np.float32)
    T[:, 12:15] += (cm - cs).astype(np.float32)
    got_p, got_u = pr.score_transforms(T), un.score_transforms(T)
    assert np.array_equal(got_p.view(np.uint32), got_u.view(np.uint32)) and got_p.max() > 0.2
    n_hits = 0
    for c in (0, 1, 2, 33, 34, 40, 63):
        hp, cp = pr.lcp_detail(T[c]); hu, cu = un.lcp_detail(T[c])
        assert np.array_equal(hp, hu) and np.array_equal(cp, cu), c
        n_hits += int((hp >= 0).sum())
    assert n_hits > 5000
    # against the brute-force answer under the product's tie rule (smallest float distance, then largest index), identity candidate
    hp, _ = pr.lcp_detail(T[0])
    q = (mpos.astype(np.float32) - pr.get_model_centroid().astype(np.float32))
    q = q + T[0, 12:15]
    spc = sp - pr.get_scene_centroid().astype(np.float32)
    eps2 = np.float32(0.005 * unit) ** 2
    bad = 0
    for i in rng.integers(0, len(q), 400):
        d = ((q[i][None, :] - spc) ** 2).astype(np.float32)
        d2 = d[:, 0] + (d[:, 1] + d[:, 2])
        j = np.nonzero(d2 == d2.min())[0].max() if d2.min() <= eps2 else -1
        bad += int(j != hp[i])
    assert bad <= 4          # (numpy's float evaluation of x*x + (y*y + z*z) is the kernel's up to FMA-free rounding: equal here; a handful of slack for the translation's rounding)
    pr.close(); un.close()


def test_pruning_statistics_and_cell_edge_choice(monkeypatch, capfd):
    """What the pruning is for: at C5's density the stored lists are a fraction of the points within r of a cell, and the level choice
    settles at eps/2 (rounds 3-4: eps/4 with early exit)."""
    from model_matching_amd import synth
    from model_matching_amd.estimator import StocsEstimator
    m, s, k = synth.workload("dense")
    monkeypatch.setenv("STOCS_DEBUG_TIMING", "1")
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    err = capfd.readouterr().err
    line = [l for l in err.splitlines() if l.startswith("[stocs grid]")][-1]
    assert "dominance-pruned" in line and "cell edge eps/2" in line, line
    import re
    kept, dilated = [float(x) for x in re.search(r"\(([0-9.]+) per non-empty cell; ([0-9.]+) within r", line).groups()]
    assert kept < 0.4 * dilated and kept < 12.0, line
    est.close()


def test_config5_pruned_lists_give_the_same_16384_scores(monkeypatch):
    """BASELINE config 5 at its full size (200 000-point scene, 50 000-point model, 16 384 candidates): the pruned grid (cell edge eps/2, ~8 entries
    per list) against the grid of rounds 3-4 (eps/4, centre-sorted lists of ~30 entries with early exit) -- every score bit for bit."""
    from model_matching_amd import synth
    m, s, k = synth.workload("C5")
    pr, un = _pair(monkeypatch, (s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm))
    cs, cm = pr.get_scene_centroid().astype(np.float64), pr.get_model_centroid().astype(np.float64)
    T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k, seed=synth.SEED_CAND)
    got_p, got_u = pr.score_transforms(T), un.score_transforms(T)
    assert np.array_equal(got_p.view(np.uint32), got_u.view(np.uint32)) and got_p.max() > 0.05
    best = int(np.argmax(got_p))
    hp, cp = pr.lcp_detail(T[best]); hu, cu = un.lcp_detail(T[best])
    assert np.array_equal(hp, hu) and np.array_equal(cp, cu) and (hp >= 0).sum() > 5000
    pr.close(); un.close()
