"""The library's own stable radix sort of (u32 key, u32 value) pairs (csrc/sort32.hip) -- the sort behind the pair lists of the
congruent-set phase, where the reference keeps a pointer grid of per-cell vectors filled one pair at a time (IndexedNormalSet::addElement,
reference include/super4pcs/accelerators/normalset.hpp:114-131; src/stocs.cpp:806-866): a run of equal keys is a cell's vector in insertion
order, so the sort must be STABLE.  Checked against numpy's stable argsort and against rocPRIM, bit for bit, on the key shapes of the path."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(keys, vals, end_bit, which, reps=1, seg_off=None):
    from model_matching_amd import capi
    L = capi.load()
    k = np.ascontiguousarray(keys, np.uint32); v = np.ascontiguousarray(vals, np.uint32)
    ko = np.zeros_like(k); vo = np.zeros_like(v)
    ms = C.c_float(0)
    u32p = C.POINTER(C.c_uint32)
    so = None if seg_off is None else np.ascontiguousarray(seg_off, np.uint32)
    capi.check(L.stocs_debug_sort_pairs(-1, k.ctypes.data_as(u32p), v.ctypes.data_as(u32p), len(k), end_bit, which, reps,
                                        ko.ctypes.data_as(u32p), vo.ctypes.data_as(u32p), C.byref(ms),
                                        None if so is None else so.ctypes.data_as(u32p), 0 if so is None else len(so) - 1))
    return ko, vo, ms.value


@pytest.mark.parametrize("n,end_bit", [(0, 8), (1, 1), (63, 5), (4096, 8), (4097, 15), (100003, 22), (1 << 20, 28), (3000017, 22), (2500000, 15), (777777, 32), (50000, 9), (50000, 17)])
def test_own_sort_is_stable_and_equals_numpy_and_rocprim(n, end_bit):
    rng = np.random.default_rng(n + end_bit)
    mask = np.uint64((1 << end_bit) - 1)
    keys = (rng.integers(0, 1 << 32, n, dtype=np.uint64) & mask).astype(np.uint32)
    if n > 1000:                                   # the key shapes of the path: long runs of one (base, cell), a few heavy cells, bits above end_bit set
        keys[: n // 3] = np.sort(keys[: n // 3])
        keys[n // 2: n // 2 + n // 8] = keys[0]
        keys |= (rng.integers(0, 2, n, dtype=np.uint32) << np.uint32(31)) if end_bit < 32 else np.uint32(0)
    vals = np.arange(n, dtype=np.uint32)[::-1].copy()
    ko, vo, _ = _sort(keys, vals, end_bit, 1)
    order = np.argsort(keys & np.uint32(mask), kind="stable")
    assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])
    kr, vr, _ = _sort(keys, vals, end_bit, 0)
    assert np.array_equal(ko, kr) and np.array_equal(vo, vr)


def test_all_keys_equal_and_all_distinct():
    n = 300000
    vals = np.arange(n, dtype=np.uint32)
    ko, vo, _ = _sort(np.full(n, 12345, np.uint32), vals, 22, 1)
    assert np.array_equal(vo, vals) and (ko == 12345).all()             # one run: the order of the input
    keys = np.random.default_rng(5).permutation(n).astype(np.uint32)
    ko, vo, _ = _sort(keys, vals, 19, 1)
    assert np.array_equal(ko, np.arange(n, dtype=np.uint32)) and np.array_equal(keys[vo], ko)


@pytest.mark.parametrize("n,n_seg,cell_bits", [(200000, 7, 15), (3000000, 100, 15), (1500000, 4000, 16), (50000, 300, 9), (4096 * 3, 3, 8)])
def test_segmented_by_base_equals_a_full_sort_of_base_and_cell(n, n_seg, cell_bits):
    """The form the congruent-set phase uses: the list is base-major, every base's stretch is sorted by its cell bits alone (two passes) --
    the result must be the stable sort by the whole (base, cell) key.  Segment lengths as uneven as the bases of a trial (a dozen bases carry
    most of the entries, many are short, some are empty); bits above the cell bits hold the base and must be ignored by the passes."""
    rng = np.random.default_rng(n_seg * 31 + cell_bits)
    w = rng.pareto(0.8, n_seg) + 0.01
    w[rng.integers(0, n_seg, max(1, n_seg // 10))] = 0.0                      # empty bases
    lens = np.floor(w / w.sum() * n).astype(np.int64)
    lens[int(np.argmax(lens))] += n - int(lens.sum())
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    base = np.repeat(np.arange(n_seg, dtype=np.uint32), lens)
    cell = rng.integers(0, 1 << cell_bits, n, dtype=np.uint32)
    cell[: n // 4] = np.sort(cell[: n // 4]) >> np.uint32(3) << np.uint32(3)   # long runs of one cell
    keys = (base << np.uint32(cell_bits)) | cell
    vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    ko, vo, _ = _sort(keys, vals, cell_bits, 1, seg_off=off)
    order = np.argsort(keys, kind="stable")                                   # (base-major input: sorting the whole key = sorting every segment by cell)
    assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])
