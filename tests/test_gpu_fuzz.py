"""A short run of the randomised GPU-vs-oracle sweep (scripts/fuzz_parity.py: random window sizes, voxel counts, incidences, fix
clusters, perturbations, map depths -- LiDAR LM narrow / wide / mixed precision, LiDAR-inertial shells, the gravity
variant, odometry, both batch voxelisations, down-sampling, the per-leaf plane producers)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_sweep_has_no_mismatch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "20250410", "66"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "66 cases, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-1500:]
