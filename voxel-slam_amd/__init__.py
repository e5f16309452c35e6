"""voxel_slam_amd -- MI355X-native local-mapping LiDAR bundle adjustment (the Voxel-SLAM hot path).

Host-side mirror of the reference ``LidarFactor`` / ``Lidar_BA_Optimizer`` interface
(VoxelSLAM/src/voxel_map.hpp:109-444) over the C-ABI library ``libvxba.so``
(include/vxba.h) whose kernels are hand-written HIP for gfx950.  There is no CPU
fallback: importing :mod:`voxel_slam_amd.vxba` and creating a factor fails loudly if
the HIP library is missing or no GPU is present.
"""
__version__ = "0.1.0"
