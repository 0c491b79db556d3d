#!/usr/bin/env python3
"""Config 4 of BASELINE.json: T independent StoCS trial streams (each a full run: 100 base attempts ->
congruent sets -> <= 200 transforms per base -> verification) sharded over the ranks, one process per
GPU, combined by ONE 8-byte RCCL max all-reduce of the packed (score, trial, candidate) key and a
64-byte broadcast of the winner's pose.  Instance-mode sampling (edge map present) is sequential inside a
trial, so sharding is across trials.

  python tools/trials.py --example packed_dove --trials 64
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/trials.py --trials 64
(STOCS_BENCH_REHEARSAL=1 lets several ranks share GPU 0 over gloo -- rehearsal of the code path only.)"""
import argparse
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
os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side (importing the library itself has no side effects)

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--example", default="packed_dove", help="reference example fixture, or synth:<workload> (e.g. synth:Cm)")
    ap.add_argument("--trials", type=int, default=64)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--bases", type=int, default=100)
    ap.add_argument("--max-sets", type=int, default=200)
    ap.add_argument("--streams", type=int, default=1, help="trial streams in flight per GPU (one context + HIP stream + host thread each)")
    ap.add_argument("--batch", type=int, default=0, help="B > 0: the rank's trials go through stocs_run_trials in batches of B -- all trials of a batch in ONE set of launches "
                                                         "(0: one trial after the other through the single-trial calls)")
    ap.add_argument("--cpu-reference", type=int, default=0, metavar="ATTEMPTS", help="A > 0: also time the reference's CPU path (the oracle's run_stocs_estimation, one core) on the same input, "
                                                                                      "A base attempts scaled to --bases (a Cm trial takes the CPU about a minute: use 8 there, --bases on the example frames)")
    ap.add_argument("--gpus", type=int, default=0, help="N > 1 without a launcher: start N rank processes (one per GPU) as children and relay their output")
    args = ap.parse_args()
    from model_matching_amd import dist as sd
    sd.launch_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if args.gpus and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("WORLD_SIZE=%s does not match --gpus %d" % (os.environ.get("WORLD_SIZE", "1"), args.gpus))
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    rehearsal = os.environ.get("STOCS_BENCH_REHEARSAL") == "1"
    import torch
    import torch.distributed as dist
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo") if rehearsal else dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from model_matching_amd.estimator import StocsEstimator
    if args.example.startswith("synth:"):
        from model_matching_amd import synth
        m_, s_, _ = synth.workload(args.example.split(":", 1)[1])
        d = None
        cloud = (s_.pos, s_.nrm, s_.prob, s_.pixel, m_.pos, m_.nrm)
        mode = 0
    else:
        d = np.load(os.path.join(ROOT, "tests", "golden", "example_%s.npz" % args.example))
        cloud = (d["scene_pos"], d["scene_nrm"], d["scene_prob"], d["scene_pixel"], d["model_pos"], d["model_nrm"])
        mode = 1 if "edge_map" in d.files else 0
    if args.bases * args.max_sets > 65535 or args.trials > 65535:
        # global candidate id = (trial << 16) | index: both parts must fit 16 bits for the packed 32-bit id of the all-reduce
        raise SystemExit("trials.py: bases * max-sets and trials must each stay below 65536 (packed (trial, candidate) id)")
    lo, hi = sd.shard_range(args.trials, rank, world)
    best = (0.0, -1, None)
    n_cand = 0
    # one-time set-up (contexts, scene grids, model index) happens BEFORE the clock starts: it is per-GPU state that a
    # serving process keeps, and inside the timed span it would be a serial fraction of every rank (10-35 ms against ~0.12 s
    # of work per rank at 64 trials / 8 GPUs)
    n_streams = max(1, min(args.streams, hi - lo))
    t_setup = time.perf_counter()
    ests = [StocsEstimator(*cloud, build_index=True, device=local_rank) for _ in range(n_streams)]
    if mode:
        for est in ests:
            est.set_edge_map(d["edge_map"])
    # one untimed trial per context sizes its arenas (the only device allocations a context ever makes), as a serving
    # process would have done long before the trials that count
    for est in ests:
        est.reset_trial()
        est.sample_bases(args.seed + 10**6, args.bases, mode=mode, dispersion=0.9)
        est.find_congruent_all()
        est.make_transforms(args.max_sets, args.seed + 10**6)
        est.compute_best_transform()
        est.sync()
    setup_s = time.perf_counter() - t_setup
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()

    phase_s = [[0.0] * 4 for _ in range(n_streams)]

    def run_trials(k):
        # stream k of this rank takes trials lo+k, lo+k+S, ...; the library calls release the GIL
        est, b, nc = ests[k], (0.0, -1, None), 0
        for t in range(lo + k, hi, n_streams):
            t0 = time.perf_counter()
            est.reset_trial()                                                # fresh class prior per trial (instance mode decays it)
            est.sample_bases(args.seed + t, args.bases, mode=mode, dispersion=0.9)
            t1 = time.perf_counter()
            est.find_congruent_all()
            t2 = time.perf_counter()
            nc += est.make_transforms(args.max_sets, args.seed + t)
            t3 = time.perf_counter()
            lcp, idx, pose = est.compute_best_transform()
            t4 = time.perf_counter()
            for i, d_ in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                phase_s[k][i] += d_
            gid = (t << 16) | idx
            if idx >= 0 and (lcp > b[0] or (lcp == b[0] and gid < b[1])):
                b = (lcp, gid, pose.copy())
        return b, nc

    def run_batched():
        # the rank's trials lo .. hi in batches of args.batch: every trial of a batch shares every launch (stocs_run_trials)
        est, b, nc = ests[0], (0.0, -1, None), 0
        for b0 in range(lo, hi, args.batch):
            ts = list(range(b0, min(b0 + args.batch, hi)))
            res = est.run_trials([args.seed + t for t in ts], args.bases, mode=mode, dispersion=0.9, max_per_base=args.max_sets)
            tm = dict(est.last_call_timing(3))
            for i, key in enumerate(("sampling", "congruent", "transforms", "verification")):
                phase_s[0][i] += sum(v for k, v in tm.items() if k.startswith(key)) * 1e-3
            for t, r in zip(ts, res):
                nc += r["n_candidates"]
                gid = (t << 16) | r["best_index"]
                if r["best_index"] >= 0 and (r["best_lcp"] > b[0] or (r["best_lcp"] == b[0] and gid < b[1])):
                    b = (r["best_lcp"], gid, r["best_pose"].copy())
        return b, nc

    if args.batch > 0:
        run_batched()                      # untimed: sizes the arenas of the batched form (as the single trial above did for the other)
        est0 = ests[0]; est0.sync()
        for k in range(4):
            phase_s[0][k] = 0.0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        results = [run_batched()]
    elif n_streams == 1:
        results = [run_trials(0)]
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(n_streams) as ex:
            results = list(ex.map(run_trials, range(n_streams)))
    for b, nc in results:                                                    # lowest (trial, candidate) id wins ties
        n_cand += nc
        if b[1] >= 0 and (b[0] > best[0] or (b[0] == best[0] and b[1] < best[1])):
            best = b
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dev = "cpu" if rehearsal or world == 1 else "cuda"
    g_lcp, g_id = sd.allreduce_best(best[0], best[1] if best[1] >= 0 else 0, device=dev)
    owner = 0
    if world > 1:
        mine = torch.tensor([rank if (best[1] == g_id and g_id >= 0) else -1], dtype=torch.int64, device=dev)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        owner = max(int(mine.item()), 0)
        tt = torch.tensor([dt, float(n_cand)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0].item())
        cc = torch.tensor([float(n_cand)], dtype=torch.float64, device=dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        n_cand = int(cc.item())
    pose = sd.broadcast_pose(best[2] if (rank == owner and best[2] is not None) else np.zeros(16, np.float32), owner, device=dev)
    # what the collective saw: every rank reports itself (rank, device, trials taken) through the same backend
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(local_rank), "trials": [lo, hi], "setup_s": setup_s}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    cpu_ref = None
    if rank == 0 and args.cpu_reference > 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cpu_reference
        cpu_ref = cpu_reference.spans(cloud, mode, d["edge_map"] if mode else None, min(args.cpu_reference, args.bases), args.max_sets, args.seed, args.bases)
        cpu_ref["gpu_over_cpu_trials_per_s"] = (args.trials / dt) / cpu_ref["trials_per_s"]
    if rank == 0:
        print(json.dumps({"cpu_reference": cpu_ref, "world_size": world, "backend": (dist.get_backend() if world > 1 else None), "ranks": ranks,
                          "setup_seconds_outside_the_timed_span_rank0": setup_s, "example": args.example, "mode": "instance" if mode else "class", "trials": args.trials, "streams_per_gpu": n_streams, "batch": args.batch,
                          "last_call_steps_ms": {name: [[lab, round(ms, 4)] for lab, ms in ests[0].last_call_timing(w)]
                                                 for w, name in ((0, "find_congruent_all"), (1, "make_transforms"), (2, "verify_all"), (3, "run_trials"))}, "n_gpus": world, "rehearsal": rehearsal,
                          "seconds": dt, "trials_per_s": args.trials / dt,
                          "rank0_phase_seconds_sample_congruent_transforms_verify": [sum(p[i] for p in phase_s) for i in range(4)], "candidates_verified": n_cand, "candidates_per_s": n_cand / dt,
                          "best_lcp": g_lcp, "best_trial": (g_id >> 16) if g_id >= 0 else -1, "best_candidate": (g_id & 0xFFFF) if g_id >= 0 else -1,
                          "best_pose_row_major_3x4": [float(pose.reshape(4, 4).T[r, c]) for r in range(3) for c in range(4)]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
