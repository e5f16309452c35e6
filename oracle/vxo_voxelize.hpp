// TEST INFRASTRUCTURE ONLY (see vxo_linalg.hpp header).  PINNED against the reference's own code compiled here (oracle/_ref/libref.so, `make -C oracle ref`: tests/test_ref_pin.py):
// OctreeGBA::cut_voxel + OctreeGBA_multi_recut from the unmodified loop_refine.hpp (test_octree_gba_voxelisation_matches_the_reference).
//
// CPU restatement of the batch factor construction of the hierarchical / global BA:
//   OctreeGBA::cut_voxel   loop_refine.hpp:446-476   world point -> root voxel (float quotient, "-1 if negative", truncation)
//   OctreeGBA::push        loop_refine.hpp:317-322
//   OctreeGBA::subdivide   loop_refine.hpp:324-356   octant by strict '>' against the node centre, float quarter lengths
//   OctreeGBA::recut       loop_refine.hpp:358-405   N > 10, plane_judge (:311-315), >= 2 observing frames, lambda0/lambda1 <= 0.12
//   call site              voxelslam.cpp:2374-2379
// Output: the voxels that become BA factors, each with its W body-frame clusters, the world cluster and its
// eigen-decomposition (what recut hands to LidarFactor::push_voxel), in a canonical order (the reference's order is the
// iteration order of an unordered_map and is not defined).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>

#include "vxo_ba.hpp"

namespace vxo {

struct VoxelizeParams {
  double voxel_size = 1.0;        // gba_voxel_size
  int max_layer = 2;              // max_layer
  int min_points = 10;            // recut: N <= 10 -> return
  double min_eigen_value = 0.01;  // gba_min_eigen_value
  double eigen_ratio[4] = {1.0 / 16, 1.0 / 16, 1.0 / 16, 1.0 / 16};   // gba_eigen_value_array (already inverted, voxelslam.cpp:2490)
  double factor_ratio_max = 0.12; // lambda0 / lambda1 > 0.12 -> return
  // OctoTree's variant (motion_init's map build: cut_voxel for every scan, then recut + tras_opt, voxelslam.cpp:606-625;
  // OctoTree::recut voxel_map.hpp:1148-1194: N <= min_point[layer] -> no plane, no subdivision; tras_opt :1308-1333: no frame count)
  int min_points_layer[4] = {0, 0, 0, 0};   // > 0: min_point[layer]
  int min_frames = 2;                       // OctreeGBA: `if(exi <= 1) return` (loop_refine.hpp:371-376); OctoTree: 0
};

struct FactorVoxel {
  uint64_t node_id;               // canonical id: [x:16 | y:16 | z:16 | path:9 | pad:4 | layer:3], coordinates offset by 32768
  std::vector<PointCluster> pcrs; // W body-frame clusters
  PointCluster pcr_add;           // world cluster
  V3 eig_value;
  M3 eig_vector;
};

struct GbaNode {
  std::vector<std::vector<V3>> locals, worlds;
  PointCluster pcr_add;
  int layer = 0, wdsize = 0;
  std::unique_ptr<GbaNode> leaves[8];
  double voxel_center[3] = {0, 0, 0};
  float quater_length = 0;
  uint64_t root48 = 0, path = 0;
  GbaNode(int l, int w) : locals(w), worlds(w), layer(l), wdsize(w) {}

  void push(int ord, const V3& local, const V3& world) {
    locals[ord].push_back(local);
    worlds[ord].push_back(world);
    pcr_add.push(world);
  }
  void subdivide() {
    for (int i = 0; i < wdsize; i++)
      for (size_t j = 0; j < locals[i].size(); j++) {
        const V3& pw = worlds[i][j];
        int xyz[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++)
          if (pw[k] > voxel_center[k]) xyz[k] = 1;
        const int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
        if (!leaves[leafnum]) {
          leaves[leafnum].reset(new GbaNode(layer + 1, wdsize));
          GbaNode& c = *leaves[leafnum];
          for (int k = 0; k < 3; k++) c.voxel_center[k] = voxel_center[k] + (2 * xyz[k] - 1) * quater_length;
          c.quater_length = quater_length / 2;
          c.root48 = root48;
          c.path = path | ((uint64_t)leafnum << (3 * (2 - layer)));     // layer-1 octant in bits 8..6, layer-2 in 5..3, layer-3 in 2..0
        }
        leaves[leafnum]->push(i, locals[i][j], pw);
      }
  }
  void recut(const VoxelizeParams& p, std::vector<FactorVoxel>& out) {
    if (pcr_add.N <= (p.min_points_layer[layer] > 0 ? p.min_points_layer[layer] : p.min_points)) return;
    V3 eig_value; M3 eig_vector;
    eig_sym3(pcr_add.cov(), eig_value, eig_vector);
    const bool is_plane = eig_value[0] < p.min_eigen_value && (eig_value[0] / eig_value[2]) < p.eigen_ratio[layer];
    if (is_plane) {
      int exi = 0;
      for (int i = 0; i < wdsize; i++)
        if (!locals[i].empty()) exi++;
      if (exi < p.min_frames) return;
      if (eig_value[0] / eig_value[1] > p.factor_ratio_max) return;
      FactorVoxel fv;
      fv.node_id = (root48 << 16) | (path << 7) | (uint64_t)layer;
      fv.pcrs.resize(wdsize);
      for (int i = 0; i < wdsize; i++)
        for (const V3& v : locals[i]) fv.pcrs[i].push(v);
      fv.pcr_add = pcr_add;
      fv.eig_value = eig_value;
      fv.eig_vector = eig_vector;
      out.push_back(fv);
      return;
    } else if (layer >= p.max_layer) {
      return;
    }
    subdivide();
    for (int i = 0; i < 8; i++)
      if (leaves[i]) leaves[i]->recut(p, out);
  }
};

// returns false if a point falls outside the +-32768-voxel range the canonical id can hold
inline bool voxelize(int W, const std::vector<std::vector<V3>>& clouds_local, const std::vector<Pose>& xs, const VoxelizeParams& p,
                     std::vector<FactorVoxel>& out) {
  std::map<uint64_t, std::unique_ptr<GbaNode>> feat_map;
  for (int i = 0; i < W; i++)
    for (const V3& local : clouds_local[i]) {
      const V3 world = xs[i].R * local + xs[i].p;
      int64_t pos[3];
      for (int j = 0; j < 3; j++) {
        float loc = world[j] / p.voxel_size;
        if (loc < 0) loc -= 1;
        pos[j] = (int64_t)loc;
        if (pos[j] < -32768 || pos[j] > 32767) return false;
      }
      const uint64_t root48 = ((uint64_t)(pos[0] + 32768) << 32) | ((uint64_t)(pos[1] + 32768) << 16) | (uint64_t)(pos[2] + 32768);
      auto it = feat_map.find(root48);
      if (it == feat_map.end()) {
        std::unique_ptr<GbaNode> ot(new GbaNode(0, W));
        for (int j = 0; j < 3; j++) ot->voxel_center[j] = (0.5 + pos[j]) * p.voxel_size;
        ot->quater_length = p.voxel_size / 4.0;
        ot->root48 = root48;
        it = feat_map.emplace(root48, std::move(ot)).first;
      }
      it->second->push(i, local, world);
    }
  out.clear();
  for (auto& kv : feat_map) kv.second->recut(p, out);
  std::sort(out.begin(), out.end(), [](const FactorVoxel& a, const FactorVoxel& b) { return a.node_id < b.node_id; });
  return true;
}

}  // namespace vxo
