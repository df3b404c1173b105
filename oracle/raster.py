"""oracle/raster.py -- TEST INFRASTRUCTURE. ctypes wrapper over oracle/raster_ref.c (see its header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraster_ref.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "raster_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.nmr_rasterize_fim_wim.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                               ctypes.c_float, ip, fp, fp]
        _lib.nmr_rasterize_fim_wim.restype = ctypes.c_int
        _lib.nmr_face_inverse.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.nmr_face_inverse.restype = None
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def rasterize_fim_wim(faces, image_size=256, near=0.1, far=100.0):
    """faces: (bs, nf, 3, 3) float32 -> fim (bs,is,is) int32, wim (bs,is,is,3), depth (bs,is,is).
    Same defaults as rasterize.py:8-13; outputs already flipped as rasterize.py:334-338."""
    faces = np.ascontiguousarray(faces, dtype=np.float32)
    bs, nf = faces.shape[:2]
    fim = np.empty((bs, image_size, image_size), np.int32)
    wim = np.empty((bs, image_size, image_size, 3), np.float32)
    depth = np.empty((bs, image_size, image_size), np.float32)
    rc = lib().nmr_rasterize_fim_wim(_fp(faces), bs, nf, image_size, near, far,
                                     fim.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fp(wim), _fp(depth))
    if rc != 0:
        raise MemoryError("oracle rasteriser allocation failed")
    return fim, wim, depth


def face_inverse(faces, image_size=256):
    faces = np.ascontiguousarray(faces, dtype=np.float32)
    bs, nf = faces.shape[:2]
    inv = np.zeros_like(faces)
    lib().nmr_face_inverse(_fp(faces), _fp(inv), bs, nf, image_size)
    return inv
