"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/stocs_hip.h
declares, and fails loudly (no fallback) when no HIP device is present.  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    from model_matching_amd import capi as m
    if not os.path.exists(m.LIB_PATH):
        g.build()
    return m


def test_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "stocs_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(stocs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = C.CDLL(capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    capi.load()


def test_defaults_match_reference_driver(capi):
    p = capi.default_params()
    # stocs_match_one_object.cpp:7-17, stocs.cpp:368-370,1032
    assert abs(p.distance_threshold - 0.005) < 1e-9 and p.ppf_tr_discretization == 5 and p.ppf_rot_discretization == 5
    assert p.number_of_bases == 100 and p.maximum_congruent_sets == 200
    assert abs(p.plane_threshold - 0.015) < 1e-9 and abs(p.min_distance_base - 0.01) < 1e-9
    assert p.internal_angle_threshold == 30 and p.lcp_normal_angle == 30
    assert capi.load().stocs_version().startswith(b"stocs_hip")


def test_no_device_fails_loudly(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = capi.load()
    p = capi.default_params()
    a, pa = capi.f32(np.zeros((4, 3)))
    w, pw = capi.f32(np.ones(4))
    h = C.c_void_p()
    rc = L.stocs_ctx_create(C.byref(p), pa, pa, pw, None, 4, pa, pa, 4, 0, -1, C.byref(h))
    assert rc == -2 and not h.value            # STOCS_ERR_NO_DEVICE, never a CPU fallback
    assert b"no CPU fallback" in L.stocs_last_error()


def test_pack_best_is_order_preserving(capi):
    L = capi.load()
    k = lambda s, i: L.stocs_pack_best(C.c_float(s), i)
    assert k(0.5, 10) > k(0.4, 0)
    assert k(0.5, 3) > k(0.5, 4)           # lowest id wins ties (first maximum, stocs.cpp:994)
    assert k(0.0, 0) < k(1e-30, 2 ** 32 - 1)
    s, i = C.c_float(), C.c_uint32()
    L.stocs_unpack_best(k(0.25, 1234), C.byref(s), C.byref(i))
    assert s.value == 0.25 and i.value == 1234


def test_ppf_host_matches_oracle(capi, oracle_lib, tiny):
    """Host side of the shared PPF routine (deterministic double atan2) == oracle on libm."""
    m, s, k, o = tiny
    L = capi.load()
    n = oracle_lib.normalize_rows(s.nrm)
    rng = np.random.default_rng(3)
    out = np.zeros(4, np.int32)
    for _ in range(20000):
        i, j = rng.integers(0, len(s.pos), 2)
        if i == j:
            continue
        a, pa = capi.f32(s.pos[i]); b, pb = capi.f32(n[i]); c, pc = capi.f32(s.pos[j]); d, pd = capi.f32(n[j])
        assert L.stocs_ppf_compute_host(pa, pb, pc, pd, 5, 5, out.ctypes.data_as(capi._ip)) == 0
        assert out.tolist() == oracle_lib.ppf_compute(s.pos[i], n[i], s.pos[j], n[j]).tolist()
    # exact special angles
    for (p1, n1, p2, n2, want) in [([0, 0, 0], [0, 0, 1], [0.1, 0, 0], [0, 0, 1], [100, 90, 90, 0]),
                                   ([0, 0, 0], [1, 0, 0], [0.1, 0, 0], [0, 0, 1], [100, 180, 90, 90]),
                                   ([0.1, 0, 0], [0, 0, 1], [0, 0, 0], [1, 0, 0], [100, 90, 0, 90])]:
        a, pa = capi.f32(p1); b, pb = capi.f32(n1); c, pc = capi.f32(p2); d, pd = capi.f32(n2)
        L.stocs_ppf_compute_host(pa, pb, pc, pd, 5, 5, out.ctypes.data_as(capi._ip))
        assert out.tolist() == want


def test_cluster_poses_matches_oracle(capi, oracle_lib):
    """stocs_cluster_poses (host function of the library) == oracle greedy_clustering."""
    from model_matching_amd import synth
    from model_matching_amd.estimator import cluster_poses
    rng = np.random.default_rng(7)
    n = 300
    poses = np.zeros((n, 16), np.float32)
    for i in range(n):
        T = np.eye(4)
        T[:3, :3] = synth.random_rotation(rng) if i % 3 else np.eye(3)
        T[:3, 3] = rng.normal(0, 0.02, 3)
        poses[i] = T.T.reshape(16)
    lcp = rng.random(n).astype(np.float32)
    for sym in ([0, 0, 0], [0, 0, 360], [180, 90, 0]):
        sym = np.array(sym, np.float32)
        a = cluster_poses(poses, lcp, 0.3, float(lcp.max()), 20, 0.03, 25.0, sym)
        b = oracle_lib.greedy_clustering(poses, lcp, 0.3, float(lcp.max()), 20, 0.03, 25.0, sym)
        assert a.tolist() == b.tolist() and len(a) > 3


def test_header_is_plain_c(tmp_path):
    """include/stocs_hip.h is the FFI boundary: it must compile as C99 (no C++ types, extern "C" guards in place)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "stocs_hip.h"\nint main(void){ stocs_params p; stocs_default_params(&p); return (int)sizeof(p) == 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr


def test_facade_header_compiles_with_the_reference_language_level(tmp_path):
    """include/stocs.hpp (the drop-in stocs::stocs_estimator) must build with a plain host compiler at the reference's
    language level (-std=c++11, reference CMakeLists.txt:6-7); no HIP headers are needed on the caller's side."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text('#include "stocs.hpp"\nint main(){ return 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr


def test_cone_filter_never_disagrees_with_the_reference_arithmetic(capi):
    """cone_cells.h: the join kernels colour direction cells with a cheap filtered evaluation and fall back to the
    reference's float arithmetic (normalset.hpp:178-204) whenever a sample is within 5e-5 of a cell boundary.  Host
    evaluation of both over random cone queries -- a quarter of them near the anti-parallel case, where the quaternion
    of setFromTwoVectors is furthest from unit length: the bitsets must be identical, and the fallback must be rare."""
    L = capi.load()
    rng = np.random.default_rng(20261004)
    ex, ke = (C.c_uint32 * 11)(), (C.c_uint32 * 11)()
    ns, nu = C.c_int(), C.c_int()
    tot = und = 0
    for i in range(60000):
        n = rng.normal(size=3).astype(np.float32)
        if i % 4 == 0:
            n[:2] *= np.float32(10.0 ** rng.uniform(-6, -1)); n[2] = -abs(n[2])
        n /= np.linalg.norm(n)
        assert L.stocs_cone_cells_host(n.ctypes.data_as(capi._fp), C.c_float(rng.uniform(-1, 1)), ex, ke, C.byref(ns), C.byref(nu)) == 0
        assert list(ex) == list(ke), (n, i)
        tot += ns.value; und += nu.value
    assert tot > 2_000_000 and und < 1e-3 * tot
    # alpha = 0 -> no samples (Q9); cos > 1 -> NaN alpha -> no samples
    n = np.array([0, 0, 1], np.float32)
    for ca in (1.0, 1.0000001):
        assert L.stocs_cone_cells_host(n.ctypes.data_as(capi._fp), C.c_float(ca), ex, ke, C.byref(ns), C.byref(nu)) == 0
        assert ns.value == 0 and not any(ex) and not any(ke)
