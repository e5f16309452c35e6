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


def _device_kernel_metadata():
    """name -> dict(vgpr_count, private_segment_fixed_size, vgpr_spill_count) of every gfx950 kernel inside libvxba.so, read with the
    ROCm llvm tools from a scratch copy (llvm-objdump --offloading writes the code objects next to its input)."""
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("ROCm llvm tools not installed")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "libvxba.so")
        shutil.copy(os.path.join(ROOT, "voxel-slam_amd", "csrc", "libvxba.so"), so)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=td)
        for fn in sorted(os.listdir(td)):
            if "gfx950" not in fn:
                continue
            notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(td, fn)], check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+-?\s*\.(name|vgpr_count|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k == "name":
                    cur = out.setdefault(v, {})
                else:
                    cur[k] = int(v)
    return out


def test_hessian_sweep_kernels_do_not_spill():
    """Every instantiation of the Hessian sweep (window sizes 1..10, instrumented and mixed-precision builds) runs two waves per SIMD
    out of registers alone: no scratch (private segment 0), no spilled VGPRs.  Round 3 shipped k3_hessian_kernel<5> with 12 bytes of
    scratch unnoticed; this is the build-time check."""
    meta = _device_kernel_metadata()
    k3 = {n: m for n, m in meta.items() if "k3_hessian_kernel" in n or "k23_fused_kernel" in n}   # ... and the fused residual + Hessian launch (round 6)
    assert len(k3) >= 60, sorted(k3)            # 2 kernels x 10 window sizes x (plain, instrumented, mixed)
    bad = {n: m for n, m in k3.items() if m.get("private_segment_fixed_size", -1) != 0 or m.get("vgpr_spill_count", -1) != 0 or m.get("vgpr_count", 999) > 256}
    assert not bad, bad


def test_no_kernel_of_the_library_uses_scratch_memory():
    """No hand-written kernel of libvxba.so spills registers or keeps anything in scratch (private segment 0).  Round 4 found three that did without
    anyone noticing: map_margi_kernel and lio_plane_update_kernel had no launch bound (compiled for 1024 threads: 128 registers, 129 / 69 spilled),
    and in map_subdivide_wave_kernel the optimiser's common-code sinking had put two accumulators of the fold behind a pointer phi -- a scratch round
    trip per folded point, most of that kernel's time (vxba_map.hip is compiled with -mllvm -simplifycfg-sink-common=false since).  rocPRIM's own
    kernels (sorts, scans) are not ours to judge."""
    meta = _device_kernel_metadata()
    ours = {n: m for n, m in meta.items() if "rocprim" not in n.lower()}
    assert len(ours) > 150, len(ours)
    bad = {n: m for n, m in ours.items() if m.get("private_segment_fixed_size", -1) != 0 or m.get("vgpr_spill_count", -1) != 0}
    assert not bad, bad


def test_hba_window_schedule_is_upstreams(lib):
    """vxba_hba_num_windows / vxba_hba_window (no GPU needed) against a literal replay of thd_globalmapping's loop (voxelslam.cpp:2498-2575): keyframes
    arrive one by one, a window runs whenever localID holds wdsize of them and mgsize are popped, and the closing iteration (total_ba == 1) runs on
    whatever localID still holds -- the case the round-5 pass left out (advisor).  The Python twin used by the orchestration must agree."""
    import ctypes as C
    from voxel_slam_amd import hba

    def upstream(K, wd, mg):
        local, wins = [], []
        for k in range(K):
            local.append(k)
            if len(local) < wd:
                continue
            wins.append((local[0], len(local)))
            local = local[mg:] if mg <= len(local) else []
        if local:                                  # the closing iteration: no size test (:2519-2523)
            wins.append((local[0], len(local)))
        return wins

    for K in range(1, 60):
        for wd, mg in ((10, 5), (6, 3), (2, 1), (5, 5), (4, 2), (10, 9), (3, 1)):
            want = upstream(K, wd, mg)
            n = lib.vxba_hba_num_windows(K, wd, mg, 1)
            got = []
            for w in range(n):
                f0, c = C.c_int(), C.c_int()
                assert lib.vxba_hba_window(K, wd, mg, 1, w, C.byref(f0), C.byref(c)) == 0
                got.append((f0.value, c.value))
            assert got == want == hba.windows(K, wd, mg), (K, wd, mg, got, want)
            full = [x for x in want if x[1] == wd and x[0] + wd <= K and x[0] % mg == 0][: (K - wd) // mg + 1 if K >= wd else 0]
            assert hba.windows(K, wd, mg, tail=False) == full and lib.vxba_hba_num_windows(K, wd, mg, 0) == len(full)
    assert lib.vxba_hba_num_windows(0, 10, 5, 1) == 0 and lib.vxba_hba_num_windows(5, 10, 5, 0) == 0
