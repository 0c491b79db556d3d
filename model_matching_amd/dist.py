"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The hot path shards naturally -- StoCS trials / base attempts /
candidate batches are independent given the read-only scene, model and index, which are replicated
on every GPU -- so there is NO data-path collective.  The only exchange is the arg-max of
compute_best_transform (reference src/stocs.cpp:982-1004) across ranks: one 8-byte max all-reduce
of the packed (score, candidate id) key, then a 64-byte broadcast of the winner's pose."""
from __future__ import annotations

import os
import socket
import struct
import subprocess
import sys

import numpy as np


def launch_ranks_if_needed(n_ranks: int, script: str, argv) -> None:
    """`script --gpus N` started as ONE plain process (no WORLD_SIZE in the environment) must not quietly run one rank and
    label the result N GPUs.  Called before anything imports torch or touches the GPU: starts the N rank processes as
    CHILDREN (python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), relays their output -- rank 0
    prints the one JSON line -- and exits with their return code.  Never os.exec: on this pool a process must not replace
    itself once a GPU runtime may be loaded.  Returns immediately for N <= 1 or inside a rank process."""
    if n_ranks <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool: RCCL needs it across processes
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def shard_range(n: int, rank: int, world: int):
    """Contiguous block partition of n independent units (attempts / candidates) over ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_attempts(n_attempts: int, rank: int, world: int):
    """Round-robin assignment of base attempts: attempt g runs on rank g mod world."""
    return list(range(rank, n_attempts, world))


def pack_best(lcp: float, global_id: int) -> int:
    """Same key as stocs_pack_best (include/stocs_hip.h): max wins, lowest id wins ties, scores <= 0
    never win.  Positive float bit patterns are order-preserving and below 2^31, so the key fits a
    signed 64-bit integer (what RCCL / gloo reduce)."""
    if not lcp > 0:
        return 0    # "no pose" (stocs.cpp:987-998): a rank whose candidates all scored 0 must not win the reduction
    bits = struct.unpack("<I", struct.pack("<f", lcp))[0]
    return (bits << 32) | (0xFFFFFFFF - (global_id & 0xFFFFFFFF))


def unpack_best(key: int):
    bits = (key >> 32) & 0xFFFFFFFF
    return struct.unpack("<f", struct.pack("<I", bits))[0], 0xFFFFFFFF - (key & 0xFFFFFFFF)


def allreduce_best(local_lcp: float, local_global_id: int, device="cpu"):
    """Returns (best_lcp, best_global_id) over all ranks; (0.0, -1) when no rank has a positive score."""
    import torch
    import torch.distributed as dist
    key = pack_best(local_lcp, local_global_id) if local_lcp > 0 else 0
    t = torch.tensor([key], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    key = int(t.item())
    if key == 0:
        return 0.0, -1
    return unpack_best(key)


def broadcast_pose(pose16, owner_rank: int, device="cpu"):
    """64-byte broadcast of the winner's camera-frame pose from the rank that owns it."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(pose16, np.float32).reshape(16).copy(), device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=owner_rank)
    return t.cpu().numpy()
