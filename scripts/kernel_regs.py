#!/usr/bin/env python3
"""Register / scratch use of the sweep kernels: compiles vxba_kernels.hip to gfx950 assembly (no GPU needed) and prints the
.vgpr_count / .agpr_count / spill / scratch metadata of every kernel whose name contains one of the given substrings."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "voxel-slam_amd", "csrc", "vxba_kernels.hip")
out = "/tmp/vxba_kernels_regs.s"
if not os.environ.get("REUSE_ASM"):
  subprocess.run(["/opt/rocm/bin/hipcc", "-I/opt/rocm/include", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=14",
                "-S", "--cuda-device-only", "-o", out, src] + [a for a in sys.argv[1:] if a.startswith("-D")], check=True, stderr=subprocess.DEVNULL)
keys = [a for a in sys.argv[1:] if not a.startswith("-D")] or ["k3_hessian", "k2_residual", "k3_finalize"]
s = open(out).read()
for b in s.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if not any(k in name for k in keys):
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void vxk::", "")
    g = lambda key: re.search(key + r":\s+(\d+)", b).group(1)
    print("%-60s agpr %3s vgpr %3s spill %3s scratch %4s" % (dem, b.splitlines()[0].strip(), g(".vgpr_count"), g(".vgpr_spill_count"), g(".private_segment_fixed_size")))
