"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/vxba.h
declares, and fails LOUDLY without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    so = os.path.join(ROOT, "voxel-slam_amd", "csrc", "libvxba.so")
    if not os.path.exists(so):
        g.build()
    from voxel_slam_amd import vxba
    return vxba.load_library()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "vxba.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vxba_[a-z0-9_]+)\s*\(", hdr)) - {"vxba_allreduce_fn", "vxba_bcast_fn", "vxba_hess_fn", "vxba_resid_fn"})


def test_library_exports_every_declared_symbol(lib):
    from voxel_slam_amd import vxba
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"libvxba.so does not export {s}"
    assert sorted(vxba.EXPORTS) == syms     # the Python mirror binds exactly the header's surface


def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a gfx950 device the product must refuse to run rather than compute on the CPU."""
    from voxel_slam_amd import vxba
    h = C.c_void_p()
    rc = lib.vxba_create(5, 0, C.byref(h))
    if rc == 0:            # a GPU is present (GPU box): creation works, nothing more to assert here
        lib.vxba_destroy(h)
        pytest.skip("GPU present")
    assert rc in (2, 3) and not h.value
    with pytest.raises(vxba.VxbaError):
        vxba.LidarFactor(5)
    with pytest.raises(vxba.VxbaError):
        vxba.plane_fit([[1.0] * 10])


def test_product_does_not_touch_the_oracle():
    """Nothing under voxel-slam_amd/ may import, link or execute anything under oracle/."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "voxel-slam_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"oracle/|liboracle|vxo_|_oracle", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
