"""The C++ driver (model_matching_amd/apps/stocs_single, host code on the façade include/stocs.hpp)
end to end on the GPU: same phases / output file as the reference's stocs_single."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "model_matching_amd", "apps", "stocs_single")


@pytest.mark.gpu
def test_stocs_single_recovers_pose(tmp_path):
    from model_matching_amd import synth, cloudio
    m, s, k = synth.workload("tiny")
    cloudio.write_stcl(tmp_path / "scene.stcl", s.pos, s.nrm, s.prob, s.pixel)
    cloudio.write_stcl(tmp_path / "model.stcl", m.pos, m.nrm)
    out = tmp_path / "best_pose_candidate_obj.txt"
    r = subprocess.run([APP, str(tmp_path / "scene.stcl"), str(tmp_path / "model.stcl"), "--seed", "1234", "--out", str(out),
                        "--dbg", str(tmp_path), "--cluster", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for needle in ("Sampled ", " bases in ", "found ", " congruent sets in ", "evaluated transforms in ", "microseconds"):
        assert needle in r.stdout
    assert "clustered hypotheses:" in r.stdout and "cluster 0: candidate" in r.stdout
    ply = (tmp_path / "best_pose.ply").read_text().splitlines()
    assert ply[0] == "ply" and ("element vertex %d" % len(m.pos)) in ply[:4] and (tmp_path / "scene.ply").exists()
    vals = np.array(out.read_text().split(), float)
    assert vals.shape == (12,)                      # 3x4 row-major, stocs_match_one_object.cpp:171-180
    P = vals.reshape(3, 4)
    dR = P[:, :3].T @ s.T_gt[:3, :3]
    assert math.degrees(math.acos(min(1.0, (np.trace(dR) - 1) / 2))) < 3.0
    assert np.linalg.norm(P[:, 3] - s.T_gt[:3, 3]) < 0.005
    # .stcl round trip
    pos, nrm, prob, pix = cloudio.read_stcl(tmp_path / "scene.stcl")
    assert np.array_equal(pos, s.pos) and np.array_equal(prob, s.prob) and np.array_equal(pix, s.pixel)


def test_stocs_single_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(APP):
        import __graft_entry__ as g
        g.build()
    from model_matching_amd import synth, cloudio
    m = synth.make_model(50, seed=3)
    cloudio.write_stcl(tmp_path / "a.stcl", m.pos, m.nrm, np.ones(50, np.float32))
    r = subprocess.run([APP, str(tmp_path / "a.stcl"), str(tmp_path / "a.stcl")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_trial_sharding_tool_single_and_two_ranks():
    """tools/trials.py (BASELINE config 4): the 2-rank run (sharing the one GPU, gloo rehearsal) must pick
    the same winner as the single-process run over the same trials."""
    import json, socket, sys
    env = dict(os.environ, PYTHONPATH=ROOT)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trials.py"), "--trials", "6", "--seed", "3"], capture_output=True, text=True,
                         timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert r1["mode"] == "instance" and r1["best_lcp"] > 0.05 and r1["candidates_verified"] > 1000  # decayed class probabilities (Q8) keep instance-mode scores low
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env2 = dict(env, STOCS_BENCH_REHEARSAL="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "tools", "trials.py"), "--trials", "6", "--seed", "3"],
                         capture_output=True, text=True, timeout=900, env=env2, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    r2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert r2["n_gpus"] == 2 and r2["rehearsal"] is True
    assert (r2["best_lcp"], r2["best_trial"], r2["best_candidate"]) == (r1["best_lcp"], r1["best_trial"], r1["best_candidate"])
    assert r2["best_pose_row_major_3x4"] == r1["best_pose_row_major_3x4"] and r2["candidates_verified"] == r1["candidates_verified"]


@pytest.mark.gpu
def test_stocs_single_instance_mode_on_packed_dove(tmp_path):
    """edge map present -> the driver takes the sample_instance_base path (stocs_match_one_object.cpp:90)."""
    from model_matching_amd import cloudio
    d = np.load(os.path.join(ROOT, "tests", "golden", "example_packed_dove.npz"))
    cloudio.write_stcl(tmp_path / "scene.stcl", d["scene_pos"], d["scene_nrm"], d["scene_prob"], d["scene_pixel"])
    cloudio.write_stcl(tmp_path / "model.stcl", d["model_pos"], d["model_nrm"])
    d["edge_map"].astype(np.uint8).tofile(tmp_path / "edge.u8")
    out = tmp_path / "pose.txt"
    r = subprocess.run([APP, str(tmp_path / "scene.stcl"), str(tmp_path / "model.stcl"), "--edge", str(tmp_path / "edge.u8"), "--seed", "5",
                        "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Sampled " in r.stdout and "Transforms to verify: " in r.stdout
    n_tf = int(r.stdout.split("Transforms to verify: ")[1].split()[0])
    score = float(r.stdout.split("maximum score: ")[1].split()[0])
    assert n_tf > 500 and score > 0.03
    P = np.array(out.read_text().split(), float).reshape(3, 4)
    assert abs(np.linalg.det(P[:, :3]) - 1.0) < 1e-3 and 0.2 < P[2, 3] < 1.5      # a rotation, in front of the camera
