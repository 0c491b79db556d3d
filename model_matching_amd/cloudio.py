"""Flat cloud files (.stcl) consumed by the C++ driver model_matching_amd/apps/stocs_single.cpp.

Layout: 8-byte magic "STOCSCL1", int32 n, int32 flags (bit 0: class probability present, bit 1:
pixels present), float32 pos[3n], float32 nrm[3n], [float32 prob[n]], [int32 pixel[2n]].
These replace the reference's PLY / PNG / Boost-archive inputs (out of the hot-path scope)."""
import struct

import numpy as np

MAGIC = b"STOCSCL1"


def write_stcl(path, pos, nrm, prob=None, pixel=None):
    pos = np.ascontiguousarray(pos, np.float32)
    nrm = np.ascontiguousarray(nrm, np.float32)
    n = len(pos)
    flags = (1 if prob is not None else 0) | (2 if pixel is not None else 0)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<ii", n, flags))
        f.write(pos.tobytes())
        f.write(nrm.tobytes())
        if prob is not None:
            f.write(np.ascontiguousarray(prob, np.float32).tobytes())
        if pixel is not None:
            f.write(np.ascontiguousarray(pixel, np.int32).tobytes())


def read_stcl(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not a .stcl file: %s" % path)
        n, flags = struct.unpack("<ii", f.read(8))
        pos = np.frombuffer(f.read(12 * n), np.float32).reshape(n, 3).copy()
        nrm = np.frombuffer(f.read(12 * n), np.float32).reshape(n, 3).copy()
        prob = np.frombuffer(f.read(4 * n), np.float32).copy() if flags & 1 else None
        pixel = np.frombuffer(f.read(8 * n), np.int32).reshape(n, 2).copy() if flags & 2 else None
    return pos, nrm, prob, pixel
