"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/stocs_hip.h
declares, and fails loudly (no fallback) when no HIP device is present.  No GPU compute here."""
import ctypes as C
import os
import re
import subprocess

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


def test_facade_keeps_the_reference_signatures(tmp_path):
    """include/stocs.hpp + include/pose_clustering.hpp against the reference's declarations (stocs.hpp:18-30, 80-149,
    182-191; point3d.hpp; pose_clustering.hpp:9-28): member-function pointers of the reference's exact types must bind,
    at the reference's language level.  The restated caller (tests/cpp/reference_call_sequence.cpp) must compile and link
    against the C-ABI library unchanged (the build does that: apps/stocs_single_percall)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sig.cpp"
    src.write_text(r"""
#include <pose_clustering.hpp>
#include <stocs.hpp>
typedef stocs::stocs_estimator E;
bool (E::*f1)(std::vector<int>&, float&, float&) = &E::sample_class_base;
bool (E::*f2)(std::vector<int>&, float&, float&, std::vector<Point3D>&, float, int) = &E::sample_instance_base;
bool (E::*f3)(std::vector<int>&, float, float, std::vector<Quadrilateral>*) = &E::find_congruent_sets_on_model;
bool (E::*f4)(std::vector<int>&, Quadrilateral&, int) = &E::get_rigid_transform_from_congruent_pair;
Scalar (E::*f5)(const MatrixType&) = &E::compute_alignment_score_for_rigid_transform;
void (E::*f6)() = &E::compute_best_transform;
void (E::*f7)() = &E::kdtree_initialize;
void (E::*f8)() = &E::centroid_shift;
VectorType (E::*f9)() = &E::get_scene_centroid;
std::vector<PoseCandidate*> (E::*f10)() = &E::get_pose_candidates;
Scalar (E::*f11)() = &E::get_best_score;
PoseCandidate* (E::*f12)() = &E::get_best_pose;
void (E::*f13)() = &E::visualize_best_pose;
void (E::*f14)(std::string, PPFMapType&) = &E::load_object_info;
void (E::*f15)(std::string, std::string, std::string, std::string, std::vector<float>, float, float, float, std::string) = &E::load_scene_info;
void (*g1)(std::string, float, float, float, float, float, float, std::string, std::string) = &stocs::pre_process_model;
void (*g2)(std::string, PPFMapType&) = &rgbd::load_ppf_map;
void (*g3)(std::vector<PoseCandidate*>&, float, float, int, float, float, VectorType, std::vector<PoseCandidate*>&) = &clustering::greedy_clustering;
// the constructor of stocs.hpp:18-30, argument for argument
E* make(std::string s, PPFMapType& m, std::vector<float> k) { return new E(s, m, s, s, s, s, s, k, 640, 480, 1e-4f, 1.0f, 0.005f, 0.005f, 5, 5, 0.0f, 0.1f); }
int main() { Point3D p(1, 2, 3); p.set_normal(VectorType(0, 0, 2)); Quadrilateral q(1, 2, 3, 4); PoseCandidate c(MatrixType(), 0.5f, 3);
             return (p.normal()[2] == 1.0f && q[2] == 3 && c.base_index == 3 && c.transform(1, 1) == 1.0f && kLargeNumber > 1e8f) ? 0 : 1; }
""")
    exe = tmp_path / "sig"
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                        "-L", os.path.join(root, "model_matching_amd"), "-lstocs_hip", "-Wl,-rpath," + os.path.join(root, "model_matching_amd"), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(exe)]).returncode == 0
    for app in ("stocs_single", "model_preprocess", "stocs_single_percall"):
        assert os.path.exists(os.path.join(root, "model_matching_amd", "apps", app)), app
    r = subprocess.run([os.path.join(root, "model_matching_amd", "apps", "stocs_single_percall")], capture_output=True, text=True)
    assert "Enter scene path and object name as arguments!" in r.stdout


def test_caller_gets_the_standard_headers_the_reference_headers_give(tmp_path):
    """A caller of the reference includes ONLY <stocs.hpp> and <pose_clustering.hpp> (stocs_match_one_object.cpp:1-2) and
    still uses std::ofstream (:173-174), struct stat (:89-90), system() (:207-208), std::chrono + micro (:80-105),
    std::map and std::vector: the reference's headers hand those over transitively (point3d.hpp:4-7 <vector> <iostream>
    <fstream> <array>; rgbd.hpp:4,23 <chrono> and the std::map of PPFMapType; OpenCV/PCL pull in <cstdlib> and
    <sys/stat.h>).  The facade must hand over the same set.  This TU is written fresh: the 18-argument constructor of
    stocs.hpp:18-30 and the pose-file write of stocs_match_one_object.cpp:171-180, at the reference's -std=c++11."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "caller.cpp"
    src.write_text(r"""
#include <stocs.hpp>
#include <pose_clustering.hpp>
static std::map<std::string, int> seen;
static int write_pose(stocs::stocs_estimator& e, const std::string& file) {
    auto start = std::chrono::high_resolution_clock::now();
    e.compute_best_transform();
    auto finish = std::chrono::high_resolution_clock::now();
    long long us = std::chrono::duration_cast<micro>(finish - start).count();
    std::cout << "verify took " << us << " microseconds" << std::endl;
    PoseCandidate* best = e.get_best_pose();
    if (best != NULL) {
        std::ofstream out;
        out.open(file, std::ofstream::out);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) out << best->transform(r, c) << (r == 2 && c == 3 ? "" : " ");
        out << std::endl;
        out.close();
        return 0;
    }
    std::cout << "no pose found" << std::endl;
    return 1;
}
int run(std::string scene, std::string object, PPFMapType& map) {
    std::vector<float> k = {1066.778f, 312.986f, 1067.487f, 241.310f};
    std::string edge = scene + "/probability_maps/edge.png";
    struct stat buffer;
    bool instance = stat(edge.c_str(), &buffer) == 0;
    if (system(("mkdir -p " + scene + "/dbg").c_str()) != 0) return 2;
    stocs::stocs_estimator e("models/" + object + "/model_search.ply", map, scene + "/rgb.png", scene + "/depth.png",
                             scene + "/probability_maps/" + object + ".png", edge, scene + "/dbg", k, 640, 480,
                             1 / 10000.0f, 1.0f, 0.005f, 0.005f, 5, 5, 0.0f, 0.10f);
    seen[object] = instance ? 1 : 0;
    std::array<int, 4> ids = {{0, 1, 2, 3}};
    std::stringstream name;
    name << scene << "/best_pose_candidate_" << object << ".txt";
    return write_pose(e, name.str()) + ids[0];
}
int main() { return 0; }
""")
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the reference's own callers, where the reference tree is present (this container; absent on the GPU box)
    for caller in ("src/stocs_match_one_object.cpp", "src/model_preprocess.cpp"):
        path = os.path.join("/root/reference", caller)
        if os.path.exists(path):
            r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(root, "include"), path], capture_output=True, text=True)
            assert r.returncode == 0, caller + "\n" + r.stderr


def test_png_and_ply_files_round_trip(capi, tmp_path):
    """stocs_png_read against PIL-written 8/16-bit images (every PNG row filter occurs in PIL's output), stocs_ply_write /
    stocs_ply_read round trip incl. binary_little_endian input, and the error paths."""
    from PIL import Image
    L = capi.load()
    rng = np.random.default_rng(5)

    def read_png(path):
        w, h, c, b = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert L.stocs_png_read(str(path).encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(b), None, 0) == 0, L.stocs_last_error()
        a = np.zeros((h.value, w.value, c.value), np.uint16 if b.value == 16 else np.uint8)
        assert L.stocs_png_read(str(path).encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(b), a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
        assert L.stocs_png_read(str(path).encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(b), a.ctypes.data_as(C.c_void_p), a.nbytes - 1) == -4
        return a
    yy, xx = np.mgrid[0:97, 0:131]
    imgs = {"g16": ((yy * 300 + xx * 7) % 65536).astype(np.uint16), "g16n": rng.integers(0, 65536, (97, 131)).astype(np.uint16),
            "g8": ((yy + xx) % 256).astype(np.uint8), "rgb8": rng.integers(0, 256, (97, 131, 3)).astype(np.uint8)}
    for k, a in imgs.items():
        Image.fromarray(a).save(tmp_path / (k + ".png"))
        got = read_png(tmp_path / (k + ".png"))
        assert np.array_equal(got.reshape(a.shape), a), k
    (tmp_path / "bad.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"0" * 40)
    w = C.c_int()
    assert L.stocs_png_read(str(tmp_path / "bad.png").encode(), C.byref(w), C.byref(w), C.byref(w), C.byref(w), None, 0) == -1
    assert L.stocs_png_read(str(tmp_path / "missing.png").encode(), C.byref(w), C.byref(w), C.byref(w), C.byref(w), None, 0) == -1
    # PLY
    pos = rng.normal(size=(257, 3)).astype(np.float32); nrm = rng.normal(size=(257, 3)).astype(np.float32)
    f = str(tmp_path / "a.ply").encode()
    assert L.stocs_ply_write(f, pos.ctypes.data_as(capi._fp), nrm.ctypes.data_as(capi._fp), 257, C.c_float(1.0)) == 0
    n, hn = C.c_int(), C.c_int()
    assert L.stocs_ply_read(f, None, None, 0, C.byref(n), C.byref(hn)) == 0 and (n.value, hn.value) == (257, 1)
    p2 = np.zeros_like(pos); n2 = np.zeros_like(nrm)
    assert L.stocs_ply_read(f, p2.ctypes.data_as(capi._fp), n2.ctypes.data_as(capi._fp), 257, C.byref(n), C.byref(hn)) == 0
    assert np.array_equal(p2, pos) and np.array_equal(n2, nrm)
    assert L.stocs_ply_read(f, p2.ctypes.data_as(capi._fp), None, 100, C.byref(n), C.byref(hn)) == -4
    with open(tmp_path / "b.ply", "wb") as fh:   # binary little endian with an extra property in front
        fh.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 257\nproperty uchar tag\nproperty float x\nproperty float y\nproperty float z\n"
                 b"property float nx\nproperty float ny\nproperty float nz\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
        rec = np.zeros(257, dtype=[("t", "u1"), ("p", "<f4", 3), ("n", "<f4", 3)]); rec["t"] = 9; rec["p"] = pos; rec["n"] = nrm
        fh.write(rec.tobytes())
    p3 = np.zeros_like(pos); n3 = np.zeros_like(nrm)
    assert L.stocs_ply_read(str(tmp_path / "b.ply").encode(), p3.ctypes.data_as(capi._fp), n3.ctypes.data_as(capi._fp), 257, C.byref(n), C.byref(hn)) == 0
    assert np.array_equal(p3, pos) and np.array_equal(n3, nrm) and hn.value == 1
    (tmp_path / "c.ply").write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nend_header\n1\n2\n")
    assert L.stocs_ply_read(str(tmp_path / "c.ply").encode(), None, None, 0, C.byref(n), C.byref(hn)) == -1


def test_weight_fix_bits_equal_the_double_formula(tmp_path):
    """weight_fix (stocs_math.h) reads trunc(w * 2^32) off the float's bits; the draws are defined by the double formula
    (uint64_t)((double)w * 2^32).  Every 61st of the 2^32 bit patterns plus all patterns around the exponent edges."""
    src = tmp_path / "wf.cpp"
    src.write_text(r'''
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "stocs_math.h"
static uint64_t ref(float w) {
    if (!(w > 0.0f)) return 0;
    double s = (double)w * 4294967296.0;
    if (s >= 1.8446744073709552e19) return 0xFFFFFFFFFFFFFFFFull;
    return (uint64_t)s;
}
int main() {
    unsigned long long bad = 0, n = 0;
    for (uint64_t i = 0; i < (1ull << 32); i += 61) { uint32_t b = (uint32_t)i; float w; memcpy(&w, &b, 4); bad += ref(w) != stocs::weight_fix(w); ++n; }
    for (uint32_t e = 0; e < 256; ++e)
        for (int d = -2048; d <= 2048; ++d) { uint32_t b = (e << 23) + (uint32_t)d; float w; memcpy(&w, &b, 4); bad += ref(w) != stocs::weight_fix(w); ++n; }
    printf("%llu %llu\n", bad, n);
    return bad != 0;
}
''')
    exe = tmp_path / "wf"
    inc = os.path.join(ROOT, "model_matching_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++11", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert int(out.stdout.split()[1]) > 70_000_000


def test_model_patch_order_on_the_host(capi):
    """The order in which the scoring kernels walk the model and the bounding spheres of its 64-point steps (ctx.hip, host code;
    the patch test of lcp.hip relies on them): a permutation; every sphere contains its points (in the float-centred frame the
    library scores in); the patches are compact (median radius well below the Morton order's); sizes that are not a multiple of 64."""
    import ctypes as C
    import os
    from model_matching_amd import synth
    L = capi.load()
    for n in (5000, 1000, 130, 64, 63, 1):
        m = synth.make_model(max(n, 400), seed=21 + n)
        pos = np.ascontiguousarray(m.pos[:n], np.float32)
        npat = (n + 63) // 64
        def order():
            perm = np.zeros(n, np.int32); pat = np.zeros((npat, 4), np.float32)
            assert L.stocs_model_patch_order(pos.ctypes.data_as(capi._fp), n, perm.ctypes.data_as(capi._ip), pat.ctypes.data_as(capi._fp)) == 0
            return perm, pat
        perm, pat = order()
        assert sorted(perm.tolist()) == list(range(n))
        c = np.zeros(3, np.float32)
        for p in pos:                                   # centroid_shift: sequential float sums (stocs.cpp:943-964)
            c = (c + p).astype(np.float32)
        cen = (pos - (c / np.float32(n)).astype(np.float32)).astype(np.float32).astype(np.float64)
        for j in range(npat):
            pts = cen[perm[64 * j: 64 * j + 64]]
            assert np.linalg.norm(pts - pat[j, :3].astype(np.float64), axis=1).max() <= float(pat[j, 3]) + 1e-7
        if n == 5000:
            os.environ["STOCS_MODEL_ORDER"] = "morton"
            try:
                _, pat_m = order()
            finally:
                del os.environ["STOCS_MODEL_ORDER"]
            assert np.median(pat[:, 3]) < 0.85 * np.median(pat_m[:, 3])
            assert pat[:, 3].max() < 0.6 * pat_m[:, 3].max()


def test_stream_audit_selftest():
    """STOCS_DEBUG_STREAMS (model_matching_amd/csrc/stream_audit.h): the host-side happens-before checker of the library's two-stream
    sections, on canned sequences -- the fork / join pattern the library uses passes, every missing event edge is reported."""
    import ctypes as C
    from model_matching_amd import capi
    L = capi.load()
    msg = C.create_string_buffer(512)
    assert L.stocs_debug_stream_audit_selftest(0, msg, 512) == 0 and msg.value == b""
    expect = {1: b"read after write", 2: b"read after write", 3: b"recycled while", 4: b"write after read"}
    for scenario, text in expect.items():
        assert L.stocs_debug_stream_audit_selftest(scenario, msg, 512) >= 1, scenario
        assert text in msg.value, (scenario, msg.value)


def test_bench_contract_line_is_small_and_parses():
    """bench.py prints ONE compact JSON line (< 4 KB, scalars and short lists only) as its last stdout line; the per-run tables,
    step records and counter nests go to bench_details.json.  (Round 4's line had grown to 34 KB and the driver recorded `parsed:
    null`.)  The canned input is the full record of round 4's own run of the driver's command."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_driver_command.json")))
    assert len(json.dumps(full)) > 20000                       # the record that did not parse
    full["cpu_baseline_pipeline"] = {"cpu_pipeline_ms": [650.123456, 55000.1, 14000.9], "cpu_pipeline_poses_per_s_phases_2_4": 115.123456789,
                                     "cpu_pipeline_attempts": 8, "cpu_pipeline_cores": 1, "ycb_cpu_trial_ms": [1373.0, 16.0, 41.0],
                                     "ycb_cpu_trials_per_s": 0.7, "ycb_gpu_trial_ms": [0.09, 0.3, 0.1, 0.1], "ycb_gpu_trials_per_s_single": 1700.0,
                                     "ycb_gpu_trials_per_s_batch64": 22000.0, "cm_cpu_spans_s": [1, 2, 3]}
    full["details_file"] = "bench_details.json"
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 4096, len(line)
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "oracle_check", "cpu_pipeline_ms", "cpu_pipeline_poses_per_s_phases_2_4",
              "pipeline_poses_per_s_phases_2_4", "pipeline_ms", "batched_trials_per_s"):
        assert k in got, k
    assert got["value"] == pytest.approx(full["value"], rel=1e-5) and got["config"]["workload"].startswith("Cm")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "needed_frac", "binding_unit", "c5_kernel_ms", "c5_traffic_over_needed"):
        assert k in got["roofline"], k
    assert set(got["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert got["roofline"]["unit"] == "GB/s" and got["roofline"]["frac"] == pytest.approx(got["roofline"]["achieved"] / got["roofline"]["peak"], rel=1e-4)

    def flat(v, depth=0):                                       # nothing nested beyond one object of scalars / short lists
        if isinstance(v, dict):
            assert depth < 2
            return all(flat(x, depth + 1) for x in v.values())
        if isinstance(v, list):
            return len(v) <= 8 and all(not isinstance(x, (dict, list)) for x in v)
        return True
    assert flat(got)
    # a record without the optional sections (N > 1 ranks, --no-pipeline ...) still gives the contract keys
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")}
    bare["roofline"] = {k: full["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    got = json.loads(json.dumps(bench.compact_line(bare)))
    assert got["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5) and "cpu_baseline" not in got
