"""world_size-2 gloo test of the N > 1 path (trial sharding + best-score all-reduce + pose
broadcast).  No GPU here, so the oracle stands in as the per-rank scorer; what is under test is the
sharding and the reduction, which are backend independent."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack_best_matches_c_abi():
    from model_matching_amd import capi, dist
    import ctypes as C
    L = capi.load()
    for lcp, gid in [(0.5, 3), (0.25, 70000), (1e-30, 2 ** 32 - 1), (0.999, 0)]:
        assert dist.pack_best(lcp, gid) == L.stocs_pack_best(C.c_float(lcp), gid)
        s, i = dist.unpack_best(dist.pack_best(lcp, gid))
        assert s == np.float32(lcp) and i == gid
    assert dist.pack_best(0.5, 3) > dist.pack_best(0.5, 4) > dist.pack_best(0.4, 0)
    # scores that are not positive pack to 0 = "no pose" on both sides (a rank whose candidates all scored 0 never wins)
    for bad in (0.0, -1.0, float("nan")):
        assert dist.pack_best(bad, 5) == 0 == L.stocs_pack_best(C.c_float(bad), 5)


def test_sharding_covers_everything_once():
    from model_matching_amd import dist
    for n in (0, 1, 7, 64, 100):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = dist.shard_range(n, r, w)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            att = sorted(sum((dist.shard_attempts(n, r, w) for r in range(w)), []))
            assert att == list(range(n))


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from model_matching_amd import synth
    from model_matching_amd import dist as sd
    from oracle import pyoracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m, s, k = synth.workload("tiny")
    orc = pyoracle.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    cs, cm = orc.centroids()
    T = synth.make_candidates(synth.centred_gt(s.T_gt, cs.astype(np.float64), cm.astype(np.float64)), 96)
    lo, hi = sd.shard_range(len(T), rank, world)
    lcp = orc.lcp_batch(T[lo:hi])                      # this rank's shard only
    i = int(np.argmax(lcp)) if len(lcp) else 0
    best, gid = sd.allreduce_best(float(lcp[i]) if len(lcp) else 0.0, lo + i)
    owner = next(r for r in range(world) if sd.shard_range(len(T), r, world)[0] <= gid < sd.shard_range(len(T), r, world)[1])
    pose = sd.broadcast_pose(T[gid] if rank == owner else np.zeros(16, np.float32), owner)
    full = orc.lcp_batch(T)                            # reference answer computed redundantly
    gi, gs = pyoracle.best(full)
    assert gid == gi and best == gs, (gid, gi, best, gs)
    assert np.array_equal(pose, T[gi])
    # all-zero scores -> no pose (Q18)
    assert sd.allreduce_best(0.0, 5) == (0.0, -1)
    # ties: the lowest global id wins on every rank
    assert sd.allreduce_best(0.5, 10 + rank)[1] == 10
    dist.barrier()
    if rank == 0:
        print("DIST_OK", gid, best)
    dist.destroy_process_group()
""")


def test_two_rank_gloo_argmax(tmp_path, oracle_lib):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "DIST_OK" in out.stdout


LAUNCHED = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    from model_matching_amd import dist as sd
    n, outdir = int(sys.argv[1]), sys.argv[2]
    sd.launch_ranks_if_needed(n, os.path.abspath(__file__), sys.argv[1:])   # parent: becomes the launcher and exits with the ranks' code
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        assert (dist.get_rank(), dist.get_world_size()) == (rank, world)
        dist.barrier()
        dist.destroy_process_group()
    open(os.path.join(outdir, "rank_%%d_of_%%d" %% (rank, world)), "w").close()
    if len(sys.argv) > 3 and sys.argv[3] == "fail" and rank == 1:
        sys.exit(7)
""")


def test_gpus_n_without_a_launcher_starts_n_ranks(tmp_path):
    """What bench.py / tools/trials.py do with `--gpus N` when nothing launched them as ranks: N rank children through
    torch.distributed.run (not one rank under an N-GPU label), return code propagated; N = 1 stays one plain process."""
    script = tmp_path / "launched.py"
    script.write_text(LAUNCHED % ROOT)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for n, extra, want_rc0 in ((2, [], True), (1, [], True), (2, ["fail"], False)):
        out = tmp_path / ("out_%d_%s" % (n, "".join(extra)))
        out.mkdir()
        r = subprocess.run([sys.executable, str(script), str(n), str(out)] + extra, capture_output=True, text=True, timeout=300, env=env)
        assert (r.returncode == 0) == want_rc0, r.stderr[-2000:]
        assert sorted(os.listdir(out)) == ["rank_%d_of_%d" % (k, n) for k in range(n)]
