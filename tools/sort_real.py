#!/usr/bin/env python3
"""The library's segmented sort (csrc/sort32.hip) on the pair lists of REAL trials -- the key skew and base lengths the congruent-set phase
actually sees, which tools/sort_bench.py's uniform keys do not have.  Step 1 (measurement build): a trial batch is run with
STOCS_DUMP_SORT, which writes every unsorted list; step 2: each list is replayed through stocs_debug_sort_pairs (own sort, segmented; rocPRIM
over all bits) with device timing.
usage: python tools/sort_real.py [--example synth:Cm] [--trials 16] [--keep DIR]"""
import argparse, ctypes as C, glob, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--example", default="synth:Cm")
ap.add_argument("--trials", type=int, default=16)
ap.add_argument("--keep", default=None)
ap.add_argument("--replay", default=None, help="directory of dumps written earlier")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
d = args.replay or args.keep or tempfile.mkdtemp(prefix="sortdump_")
os.makedirs(d, exist_ok=True)
if not args.replay:
    env = dict(os.environ, STOCS_DUMP_SORT=os.path.join(d, "list"))
    code = ("import sys, os, runpy; sys.path.insert(0, %r); from model_matching_amd import capi; "
            "capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), 'libstocs_hip_tools.so'); "
            "sys.argv = ['trials.py', '--example', %r, '--trials', %r, '--batch', %r]; runpy.run_path(%r, run_name='__main__')"
            % (ROOT, args.example, str(args.trials), str(args.trials), os.path.join(ROOT, "tools", "trials.py")))
    subprocess.run([sys.executable, "-c", code], env=env, check=True, stdout=subprocess.DEVNULL)
from model_matching_amd import capi  # noqa: E402
L = capi.load()
u32p = C.POINTER(C.c_uint32)
seen = set()
for f in sorted(glob.glob(os.path.join(d, "list_*.bin")), key=lambda s: int(s.rsplit("_", 1)[1][:-4])):
    raw = np.fromfile(f, dtype=np.uint32)
    n, n_seg, cell_bits, end_bit = (int(x) for x in raw[:4])
    if (n, n_seg) in seen or n < 100000:
        continue
    seen.add((n, n_seg))
    keys, vals, off = raw[4:4 + n].copy(), raw[4 + n:4 + 2 * n].copy(), raw[4 + 2 * n:4 + 2 * n + n_seg + 1].copy()
    rec = {"file": os.path.basename(f), "n": n, "bases": n_seg, "cell_bits": cell_bits, "end_bit": end_bit}
    lens = np.diff(off.astype(np.int64))
    rec["base_len_max"] = int(lens.max()); rec["base_len_median"] = int(np.median(lens[lens > 0])) if (lens > 0).any() else 0
    cells = keys & np.uint32((1 << cell_bits) - 1)
    u, cnt = np.unique(keys, return_counts=True)
    rec["distinct_keys"] = int(len(u)); rec["pairs_per_key_mean"] = round(float(cnt.mean()), 1); rec["pairs_per_key_max"] = int(cnt.max())
    for digit in (0, 8):
        h = np.bincount((cells >> digit) & 255, minlength=256)
        rec["digit%d_top_share" % digit] = round(float(h.max()) / n, 4)
    ko = np.empty(n, np.uint32); vo = np.empty(n, np.uint32)
    ms = C.c_float(0)
    capi.check(L.stocs_debug_sort_pairs(-1, keys.ctypes.data_as(u32p), vals.ctypes.data_as(u32p), n, cell_bits, 1, args.reps, ko.ctypes.data_as(u32p), vo.ctypes.data_as(u32p),
                                        C.byref(ms), off.ctypes.data_as(u32p), n_seg))
    rec["own_ms"] = round(ms.value, 4)
    order = np.argsort(keys.astype(np.uint64), kind="stable")
    rec["own_correct"] = bool(np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order]))
    capi.check(L.stocs_debug_sort_pairs(-1, keys.ctypes.data_as(u32p), vals.ctypes.data_as(u32p), n, end_bit, 0, args.reps, None, None, C.byref(ms), None, 0))
    rec["rocprim_ms"] = round(ms.value, 4)
    # the same list with the cells of every base redrawn uniformly: what the skew costs
    rng = np.random.default_rng(1)
    uk = (keys & ~np.uint32((1 << cell_bits) - 1)) | rng.integers(0, 1 << cell_bits, n, dtype=np.uint32)
    capi.check(L.stocs_debug_sort_pairs(-1, uk.ctypes.data_as(u32p), vals.ctypes.data_as(u32p), n, cell_bits, 1, args.reps, None, None, C.byref(ms), off.ctypes.data_as(u32p), n_seg))
    rec["own_uniform_cells_ms"] = round(ms.value, 4)
    rec["own_GBps_16B_per_pair_and_pass"] = round(n * 16.0 * ((cell_bits + 7) // 8) / (rec["own_ms"] * 1e-3) / 1e9, 1)
    print(json.dumps(rec), flush=True)
