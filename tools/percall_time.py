#!/usr/bin/env python3
"""Phase timings of the reference's per-call sequence (tests/cpp/reference_call_sequence.cpp -> apps/stocs_single_percall)
next to the batched driver (apps/stocs_single) on the three example frames, file formats and layout of the reference.
usage: python tools/percall_time.py"""
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_driver_gpu as T  # noqa: E402


def main():
    out = {}
    for name in ("ycb_024_bowl", "linemod_obj_06", "packed_dove"):
        with tempfile.TemporaryDirectory() as td:
            raw, fix, obj, scene, repo = T._write_example_tree(Path(td), name)
            K = [float(x) for x in raw["K"]]
            pre = subprocess.run([T.PRE, obj, "--repo", str(repo), "--voxel", repr(float(raw["model_voxel"])), "--normal-radius", repr(float(raw["normal_radius"])),
                                  "--model-scale", repr(float(raw["model_scale"]))], capture_output=True, text=True, timeout=300)
            assert pre.returncode == 0, pre.stderr
            env = dict(os.environ, STOCS_REPO_PATH=str(repo), STOCS_INTRINSICS=",".join(repr(k) for k in K), STOCS_DEPTH_SCALE=repr(float(raw["depth_scale"])), STOCS_SEED="7")
            rows = {}
            for rep in range(2):
                pc = subprocess.run([T.PERCALL, str(scene), obj], capture_output=True, text=True, timeout=600, env=env)
                assert pc.returncode == 0, pc.stdout + pc.stderr
                rows["percall_run%d" % rep] = [l for l in pc.stdout.splitlines() if "microseconds" in l or "candidates" in l]
                r = subprocess.run([T.APP, str(scene), obj, "--repo", str(repo), "--intrinsics", ",".join(repr(k) for k in K), "--depth-scale", repr(float(raw["depth_scale"])),
                                    "--seed", "7"], capture_output=True, text=True, timeout=300)
                assert r.returncode == 0, r.stdout + r.stderr
                rows["batched_run%d" % rep] = [l for l in r.stdout.splitlines() if "microseconds" in l or l.startswith("summary:")]
            out[name] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
