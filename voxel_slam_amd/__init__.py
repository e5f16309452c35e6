"""Importable alias for the product package, which lives in ``voxel-slam_amd/``.

The directory name mandated for the package contains a hyphen and cannot be
imported directly; this stub points ``__path__`` at it and executes its
``__init__.py`` so ``import voxel_slam_amd`` (and submodules) resolve there.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "voxel-slam_amd")
__path__[:] = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f, _os
