#!/usr/bin/env python
"""Fixture for the LZ4-frame reader of slr_sfs_amd.io: a motion field pickled and compressed by the SYSTEM
liblz4 (LZ4F_compressFrame, an implementation independent of the reader), as the reference's CLAW
`.pth` motion files are (utils/utils.py:111-115 reads them with lz4framed.decompress + pickle.loads).
Run in the build container (needs liblz4.so.1); writes tests/golden/motion_lz4.pth."""
import ctypes
import os
import pickle

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def motion(seed=21, h=96, w=160):
    """Smooth field quantised to 1/64 px (compressible, like real optical flow), float32 [1,2,h,w]."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    p = rng.uniform(0, 6.28, 2)
    u = 2.0 * np.sin(2 * np.pi * x / w + p[0]) * (x > 0.3 * w)
    v = 1.5 * np.cos(2 * np.pi * y / h + p[1]) * (x > 0.3 * w)
    return (np.round(np.stack([u, v])[None] * 64) / 64).astype(np.float32)


if __name__ == "__main__":
    L = ctypes.CDLL("liblz4.so.1")
    L.LZ4F_compressFrameBound.restype = ctypes.c_size_t
    L.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    L.LZ4F_compressFrame.restype = ctypes.c_size_t
    L.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.LZ4F_isError.restype = ctypes.c_uint
    L.LZ4F_isError.argtypes = [ctypes.c_size_t]
    raw = pickle.dumps(motion(), protocol=4)
    cap = L.LZ4F_compressFrameBound(len(raw), None)
    dst = ctypes.create_string_buffer(cap)
    n = L.LZ4F_compressFrame(dst, cap, raw, len(raw), None)
    assert not L.LZ4F_isError(n)
    out = os.path.join(ROOT, "tests", "golden", "motion_lz4.pth")
    open(out, "wb").write(dst.raw[:n])
    print("wrote", out, n, "bytes from", len(raw))
