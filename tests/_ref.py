"""ctypes driver for oracle/_ref/libref.so -- the reference's own hot-path headers (tools.hpp, preintegration.hpp, voxel_map.hpp),
unmodified, compiled where they lie by `make -C oracle ref` (oracle/ref_capi.cpp; Eigen/PCL/ROS API shim under oracle/shim/).
Test infrastructure only.  It is a second instance of tests/_oracle.py bound to that library: same classes, same call shapes.

    from tests import _ref
    R = _ref.backend()          # None when libref.so is neither prebuilt nor buildable (no /root/reference)
    f = R.Oracle(win_size)      # LidarFactor + Lidar_BA_Optimizer of the reference
"""
import ctypes as C
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref.so")
DROPIN_LIB = os.path.join(ROOT, "oracle", "_ref", "libdropin.so")
_BACKEND = None
_DROPIN = None


def available():
    return backend() is not None


def _load(name, path, target):
    spec = importlib.util.spec_from_file_location(name, os.path.join(_HERE, "_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LIB_PATH = path
    mod.MAKE_TARGET = target
    mod.PARTIAL = True
    try:
        mod.lib()
    except (OSError, Exception):
        return None
    return mod


def dropin():
    """The reference's OctoTree / OctreeGBA / call sequences compiled on top of include/vxba_voxel_map.hpp (`make -C oracle dropin`):
    same wrapper again, but every LidarFactor / optimizer call lands in libvxba.so -- needs the GPU."""
    global _DROPIN
    if _DROPIN is None:
        mod = _load("tests._dropin_backend", DROPIN_LIB, "dropin")
        if mod is not None:
            L = mod.lib()
            L.vxo_backend.restype = C.c_char_p
            mod.BACKEND_NAME = L.vxo_backend().decode()
        _DROPIN = mod or False
    return _DROPIN or None


def backend():
    global _BACKEND
    if _BACKEND is not None:
        return _BACKEND or None
    mod = _load("tests._ref_backend", REF_LIB, "ref")
    if mod is None:
        _BACKEND = False
        return None
    L = mod.lib()
    L.vxo_backend.restype = C.c_char_p
    mod.BACKEND_NAME = L.vxo_backend().decode()
    f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    L.vxo_localmap_match.argtypes = [C.c_void_p, C.c_int64, f64p, f64p, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), f64p,
                                     np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")]
    _BACKEND = mod
    return mod
