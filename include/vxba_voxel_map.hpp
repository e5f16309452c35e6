// vxba_voxel_map.hpp -- the drop-in: `LidarFactor`, `Lidar_BA_Optimizer`, `LI_BA_Optimizer`, `LI_BA_OptimizerGravity` under the
// reference's own names, signatures and public members, implemented on the C ABI (vxba.h / libvxba.so, MI355X).
//
// It REPLACES the block VoxelSLAM/src/voxel_map.hpp:108-864 (from "// The LiDAR BA factor in optimization" up to, not including,
// "// 10 scans merge into a keyframe"): delete that block and put `#include "vxba_voxel_map.hpp"` in its place -- everything the
// block needs is defined above it (tools.hpp: PointCluster, IMUST, DIM; preintegration.hpp: IMU_PRE; Eigen), everything below it
// (OctoTree::tras_opt / margi, cut_voxel, ...) and every call site in voxelslam.cpp / loop_refine.hpp keeps compiling as written:
//
//   voxelslam.cpp:623-655    motion_init: voxhess.clear(); voxhess.win_size = ..; tras_opt(voxhess); voxhess.plvec_voxels.size();
//                            LI_BA_OptimizerGravity opt_lsv; opt_lsv.damping_iter(x_buf, voxhess, imu_pre_buf, resis, hess, 3);
//                            for(Eigen::Matrix3d &iter: voxhess.eig_vectors) ...
//   voxelslam.cpp:1609-1669  local mapping: multi_recut(.., voxhess, ..); LI_BA_Optimizer opt_lsv; opt_lsv.damping_iter(x_buf, voxhess,
//                            imu_pre_buf, &hess);  multi_margi(.., voxhess, ..)  [OctoTree::margi reads vox_opt.pcr_adds[opt_state] etc.]
//   voxelslam.cpp:2378-2384  HBA: LidarFactor voxhess(wdsize); OctreeGBA_multi_recut(oct_map, voxhess, thread_num);
//                            Lidar_BA_Optimizer opt_lsv; opt_lsv.thd_num = thread_num; opt_lsv.damping_iter(xs, voxhess, &hess, resis, up, is_display);
//   loop_refine.hpp:387      vox_opt.push_voxel(pcrs, pcr_fix, coe, eig_value, eig_vector, pcr_add);
//   loop_refine.hpp:499-534  vector<LidarFactor> vec_voxhess(thd_num, voxhess);  voxhess.X.insert(voxhess.X.end(), other.X.begin(), other.X.end())
//
// tests/test_gpu_dropin.py does exactly that substitution on a build-time copy of the reference header (oracle/Makefile `dropin`) and
// runs the reference's OctoTree, OctreeGBA and call sequences on top of these classes against the untouched reference (oracle/_ref).
//
// Threading: the reference fans sweeps out over `thd_num` std::threads on voxel sub-ranges; here one call covers [0, V) on the GPU
// (`thd_num` is accepted and ignored).  No CPU fallback: failures of the C ABI throw std::runtime_error(vxba_last_error()).
#pragma once
#include <deque>
#include <vector>

#include "vxba_lidar_factor.hpp"
#include "vxba_li_optimizer.hpp"

// file-scope names of the replaced block that the rest of the reference uses (voxel_map.hpp:446, 448; voxelslam.cpp:822)
double imu_coef = 1e-4;
#ifndef DVEL
#define DVEL 6
#endif

// voxel_map.hpp:109-290
class LidarFactor : public vxba::LidarFactorT<PointCluster, IMUST, Eigen::Vector3d, Eigen::Matrix3d, Eigen::MatrixXd, Eigen::VectorXd> {
  typedef vxba::LidarFactorT<PointCluster, IMUST, Eigen::Vector3d, Eigen::Matrix3d, Eigen::MatrixXd, Eigen::VectorXd> Base;
 public:
  LidarFactor(int _w) : Base(_w) {}
  LidarFactor(const LidarFactor& o) : Base(o) {}
  LidarFactor& operator=(const LidarFactor& o) { Base::operator=(o); return *this; }
};

// voxel_map.hpp:293-444
class Lidar_BA_Optimizer {
 public:
  int win_size, jac_leng, thd_num = 2;

  // :298-335 (one GPU sweep over all voxels instead of thd_num host threads)
  double divide_thread(std::vector<IMUST>& x_stats, LidarFactor& voxhess, Eigen::MatrixXd& Hess, Eigen::VectorXd& JacT) {
    double residual = 0;
    voxhess.acc_evaluate2(x_stats, 0, (int)voxhess.plvec_voxels.size(), Hess, JacT, residual);
    return residual;
  }
  // :337-365
  double only_residual(std::vector<IMUST>& x_stats, LidarFactor& voxhess) {
    double residual1 = 0;
    voxhess.evaluate_only_residual(x_stats, 0, (int)voxhess.plvec_voxels.size(), residual1);
    return residual1;
  }
  // :367-442
  bool damping_iter(std::vector<IMUST>& x_stats, LidarFactor& voxhess, Eigen::MatrixXd* hess, std::vector<double>& resis, int max_iter = 3,
                    bool is_display = false) {
    win_size = voxhess.win_size;
    jac_leng = win_size * 6;
    return voxhess.damping_iter(x_stats, hess, resis, max_iter, is_display);
  }
};

// voxel_map.hpp:450-655
class LI_BA_Optimizer : public vxba::LI_BA_OptimizerT<IMUST, IMU_PRE, Eigen::MatrixXd, Eigen::VectorXd, LidarFactor> {
 public:
  void damping_iter(std::vector<IMUST>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor, Eigen::MatrixXd* hess) {
    this->imu_coef = ::imu_coef;
    vxba::LI_BA_OptimizerT<IMUST, IMU_PRE, Eigen::MatrixXd, Eigen::VectorXd, LidarFactor>::damping_iter(x_stats, voxhess, imus_factor, hess);
  }
  double divide_thread(std::vector<IMUST>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor, Eigen::MatrixXd& Hess, Eigen::VectorXd& JacT) {
    this->imu_coef = ::imu_coef;
    return vxba::LI_BA_OptimizerT<IMUST, IMU_PRE, Eigen::MatrixXd, Eigen::VectorXd, LidarFactor>::divide_thread(x_stats, voxhess, imus_factor, Hess, JacT);
  }
  double only_residual(std::vector<IMUST>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor) {
    this->imu_coef = ::imu_coef;
    return vxba::LI_BA_OptimizerT<IMUST, IMU_PRE, Eigen::MatrixXd, Eigen::VectorXd, LidarFactor>::only_residual(x_stats, voxhess, imus_factor);
  }
};

// voxel_map.hpp:659-864
class LI_BA_OptimizerGravity : public vxba::LI_BA_OptimizerGravityT<IMUST, IMU_PRE, Eigen::MatrixXd, Eigen::VectorXd, LidarFactor> {
 public:
  void damping_iter(std::vector<IMUST>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor, std::vector<double>& resis,
                    Eigen::MatrixXd* hess, int max_iter = 2) {
    this->imu_coef = ::imu_coef;
    vxba::LI_BA_OptimizerGravityT<IMUST, IMU_PRE, Eigen::MatrixXd, Eigen::VectorXd, LidarFactor>::damping_iter(x_stats, voxhess, imus_factor, resis, hess, max_iter);
  }
};
