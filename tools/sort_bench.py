#!/usr/bin/env python3
"""Device time of one sort of (u32, u32) pairs: the library's own onesweep (csrc/sort32.hip) against rocPRIM's radix_sort_pairs, at the sizes
and key widths of the congruent-set phase (single Cm trial: ~2 M survivors per list, 15 bits (P) and 22 bits (Q); a 40-trial piece: ~80 M, 16
and 28 bits).  usage: python tools/sort_bench.py [n ...]"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_matching_amd import capi  # noqa: E402
L = capi.load()
u32p = C.POINTER(C.c_uint32)
sizes = [int(a) for a in sys.argv[1:]] or [500000, 2000000, 8000000, 80000000]
out = []
for n in sizes:
    rng = np.random.default_rng(n)
    for bits in ((15, 22) if n <= 8000000 else (16, 28)):
        keys = rng.integers(0, 1 << bits, n, dtype=np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        rec = {"n": n, "key_bits": bits}
        for which, name in ((0, "rocprim"), (1, "own")):
            ms = C.c_float(0)
            capi.check(L.stocs_debug_sort_pairs(-1, keys.ctypes.data_as(u32p), vals.ctypes.data_as(u32p), n, bits, which, 10, None, None, C.byref(ms), None, 0))
            rec[name + "_ms"] = round(ms.value, 4)
        # the form the congruent-set phase uses: 100 (single trial) or 4 000 (a 40-trial piece) base segments of uneven length, sorted by the 15 / 16 cell
        # bits alone -- against rocPRIM over all significant bits of (base, cell), which is what rounds 2-4 ran
        n_seg = 100 if n <= 8000000 else 4000
        cb = 15 if n <= 8000000 else 16
        w = rng.pareto(0.8, n_seg) + 0.01
        lens = np.floor(w / w.sum() * n).astype(np.int64); lens[int(np.argmax(lens))] += n - int(lens.sum())
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        if bits > cb:
            skeys = (np.repeat(np.arange(n_seg, dtype=np.uint32), lens) << np.uint32(cb)) | (keys & np.uint32((1 << cb) - 1))
            ms = C.c_float(0)
            capi.check(L.stocs_debug_sort_pairs(-1, skeys.ctypes.data_as(u32p), vals.ctypes.data_as(u32p), n, cb, 1, 10, None, None, C.byref(ms), off.ctypes.data_as(u32p), n_seg))
            rec["own_segmented_%d_bases_%d_cell_bits_ms" % (n_seg, cb)] = round(ms.value, 4)
            ms = C.c_float(0)
            capi.check(L.stocs_debug_sort_pairs(-1, skeys.ctypes.data_as(u32p), vals.ctypes.data_as(u32p), n, bits, 0, 10, None, None, C.byref(ms), None, 0))
            rec["rocprim_same_keys_all_bits_ms"] = round(ms.value, 4)
        passes = (bits + 7) // 8
        rec["own_GBps_16B_per_pair_and_pass"] = round(n * 16.0 * passes / (rec["own_ms"] * 1e-3) / 1e9, 1)
        rec["rocprim_over_own"] = round(rec["rocprim_ms"] / rec["own_ms"], 3)
        out.append(rec)
        print(json.dumps(rec), flush=True)
