"""TEST INFRASTRUCTURE - ctypes handle on oracle/_ref/libsdf_ref.so: the reference's OWN SDF voxelisation kernel
(reference sdf/sdf/csrc/sdf_cuda_kernel.cu:21-304, compiled unmodified for the host by oracle/Makefile, launch geometry
of :307-335 replayed by oracle/sdf_ref_driver.cpp).  This is what pins oracle/sdf_np.py and the HIP op mvfit_sdf.

The .so is built in the build container (where /root/reference exists) and travels to the GPU box with the
snapshot; ``available()`` is False when it is missing.  Never imported by the shipped package."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_ref', 'libsdf_ref.so')
_lib = None


def available() -> bool:
    return os.path.isfile(LIB_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        for name in ('ref_sdf_f32', 'ref_sdf_f64'):
            fn = getattr(_lib, name)
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
            fn.restype = C.c_int
    return _lib


def sdf(faces, vertices, grid_size, all_voxels=False, dtype=np.float32):
    """phi[B,G,G,G] exactly as ``sdf.SDF().forward(faces, vertices, grid_size)`` produces it (sdf.py:21-26): phi starts as
    zeros, num_faces = faces.shape[0] (the launcher's faces.size(0), :314).  ``faces`` [num_faces, 3] int."""
    lib = _load()
    f = np.ascontiguousarray(np.asarray(faces).reshape(-1, 3), np.int32)
    v = np.ascontiguousarray(vertices, dtype)
    B, nv = v.shape[0], v.shape[1]
    phi = np.zeros((B, grid_size, grid_size, grid_size), dtype)
    fn = lib.ref_sdf_f32 if dtype == np.float32 else lib.ref_sdf_f64
    rc = fn(phi.ctypes.data, f.ctypes.data, v.ctypes.data, B, f.shape[0], nv, grid_size, 1 if all_voxels else 0)
    assert rc == 0
    return phi
