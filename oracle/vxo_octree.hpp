// TEST INFRASTRUCTURE ONLY (see vxo_linalg.hpp header).  PINNED against the reference's own code compiled here (oracle/_ref/libref.so, `make -C oracle ref`: tests/test_ref_pin.py):
// the reference's OctoTree / SlideWindow / cut_voxel / recut / margi / tras_opt driven scan by scan beside this restatement
// (test_local_map_evolves_like_the_reference_octree: the trees agree leaf for leaf).
//
// CPU restatement of the incremental side of the local map, i.e. what the local-mapping thread does to
// `surf_map` / `surf_map_slide` scan by scan (voxelslam.cpp:1592-1700):
//   SlideWindow                      VoxelSLAM/src/voxel_map.hpp:896-930
//   OctoTree::push / push_fix        :969-1013      allocate / allocate_fix :1021-1072   fix_divide :1074-1094   subdivide :1096-1116
//   OctoTree::plane_update           :1118-1146     recut :1148-1194    margi :1196-1305    tras_opt :1308-1333    clear_slwd :1482-1500
//   cut_voxel_multi                  :1545-1639     (root voxel by the float-typed index; points grouped per root, pushed in scan order)
//   multi_recut / multi_margi        VoxelSLAM/src/voxelslam.cpp:1396-1453, 1313-1393   (thread fan-out; `if(g_size < thd_num) return` kept)
//   the ring `mp` of window slots    voxelslam.cpp:1683-1687
// The SlideWindow pool (`sws`) only recycles memory (a window is cleared before it goes back), so a node simply owns its window here.
// The reference walks `unordered_map`s in hash order; this restatement walks roots in order of first insertion -- the order of the
// factor voxels differs, their content does not.
#pragma once
#include <cstdio>
#include <stdexcept>
#include <unordered_set>

#include "vxo_ba.hpp"
#include "vxo_lio.hpp"

namespace vxo {

struct LocalMapParams {
  double voxel_size = 1.0;
  int max_layer = 2;
  double min_point[4] = {5, 5, 5, 5};                     // Eigen::Vector4d min_point (voxel_map.hpp:83, voxelslam.cpp:812)
  double min_eigen_value = 0.0025;
  double plane_eigen_value_thre[4] = {1.0 / 4, 1.0 / 4, 1.0 / 4, 1.0 / 4};
  int max_points = 100;                                   // voxel_map.hpp:86
  int win_size = 10;
  int thread_num = 5;
};

struct TreeCtx {
  const LocalMapParams& prm;
  const std::vector<int>& mp;
};

struct TreeNode {
  // SlideWindow (voxel_map.hpp:896-930): per window slot the body-frame points (kept only above the finest layer) and their cluster
  bool has_sw = false;
  std::vector<std::vector<PointVar>> points;
  std::vector<PointCluster> pcrs_local;

  PointCluster pcr_add;
  double cov_add[81];
  PointCluster pcr_fix;
  std::vector<PointVar> point_fix;
  int layer, octo_state = 0, wdsize;
  std::unique_ptr<TreeNode> leaves[8];
  double voxel_center[3] = {0, 0, 0};
  float quater_length = 0;
  Plane plane;
  bool isexist = false;
  V3 eig_value = zero3();
  M3 eig_vector = zero33();
  int last_num = 0, opt_state = -1;

  TreeNode(int l, int w) : layer(l), wdsize(w) { for (double& x : cov_add) x = 0; }

  void sw_open() {
    if (!has_sw) { has_sw = true; points.assign(wdsize, {}); pcrs_local.assign(wdsize, PointCluster()); }
  }
  void sw_release() { has_sw = false; points.clear(); pcrs_local.clear(); }

  void add_bf_var(const PointVar& pv, const V3& vec) {
    double Bi[81];
    bf_var(pv.var, vec, Bi);
    for (int k = 0; k < 81; k++) cov_add[k] += Bi[k];
  }

  // :969-993
  void push(int ord, const PointVar& pv, const V3& pw, TreeCtx& c) {
    sw_open();
    if (!isexist) isexist = true;
    const int mord = c.mp[ord];
    if (layer < c.prm.max_layer) points[mord].push_back(pv);
    pcrs_local[mord].push(pv.pnt);
    pcr_add.push(pw);
    add_bf_var(pv, pw);
  }
  // :995-1003
  void push_fix(const PointVar& pv, TreeCtx& c) {
    if (layer < c.prm.max_layer) point_fix.push_back(pv);
    pcr_fix.push(pv.pnt);
    pcr_add.push(pv.pnt);
    add_bf_var(pv, pv.pnt);
  }
  // :1015-1019
  bool plane_judge(const V3& ev, TreeCtx& c) const { return ev[0] < c.prm.min_eigen_value && (ev[0] / ev[2]) < c.prm.plane_eigen_value_thre[layer]; }

  TreeNode* child_for(const V3& p) {   // :1029-1042 (and its three copies)
    int xyz[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++)
      if (p[k] > voxel_center[k]) xyz[k] = 1;
    const int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
    if (!leaves[leafnum]) {
      leaves[leafnum].reset(new TreeNode(layer + 1, wdsize));
      for (int k = 0; k < 3; k++) leaves[leafnum]->voxel_center[k] = voxel_center[k] + (2 * xyz[k] - 1) * quater_length;
      leaves[leafnum]->quater_length = quater_length / 2;
    }
    return leaves[leafnum].get();
  }

  // :1021-1046
  void allocate(int ord, const PointVar& pv, const V3& pw, TreeCtx& c) {
    if (octo_state == 0) push(ord, pv, pw, c);
    else child_for(pw)->allocate(ord, pv, pw, c);
  }
  // :1074-1094
  void fix_divide(TreeCtx& c) {
    for (const PointVar& pv : point_fix) child_for(pv.pnt)->push_fix(pv, c);
  }
  // :1096-1116
  void subdivide(int si, const Pose& xx, TreeCtx& c) {
    const std::vector<PointVar> pts = points[c.mp[si]];
    for (const PointVar& pv : pts) {
      V3 pw = xx.R * pv.pnt + xx.p;
      child_for(pw)->push(si, pv, pw, c);
    }
  }
  // :1118-1146
  void do_plane_update() {
    double cen[3], nrm[3];
    vxo::plane_update(pcr_add.P, pcr_add.v, (double)pcr_add.N, eig_value, eig_vector, cov_add, cen, nrm, plane.plane_var, plane.radius);
    plane.center = v3(cen[0], cen[1], cen[2]);
    plane.normal = v3(nrm[0], nrm[1], nrm[2]);
  }

  // :1148-1194
  void recut(int win_count, const std::vector<Pose>& x_buf, TreeCtx& c) {
    if (octo_state == 0) {
      if (layer >= 0) {
        opt_state = -1;
        if (pcr_add.N <= c.prm.min_point[layer]) { plane.is_plane = false; return; }
        if (!isexist || !has_sw) return;
        eig_sym3(pcr_add.cov(), eig_value, eig_vector);
        plane.is_plane = plane_judge(eig_value, c);
        if (plane.is_plane) return;
        else if (layer >= c.prm.max_layer) return;
      }
      if (pcr_fix.N != 0) {
        fix_divide(c);
        std::vector<PointVar>().swap(point_fix);
      }
      for (int i = 0; i < win_count; i++) subdivide(i, x_buf[i], c);
      sw_release();
      octo_state = 1;
    }
    for (int i = 0; i < 8; i++)
      if (leaves[i]) leaves[i]->recut(win_count, x_buf, c);
  }

  // :1196-1305
  void margi(int win_count, int mgsize, const std::vector<Pose>& x_buf, const LidarFactor& vox_opt, TreeCtx& c) {
    if (octo_state == 0 && layer >= 0) {
      if (!isexist || !has_sw) return;
      std::vector<PointCluster> pcrs_world(wdsize);
      if (opt_state >= (int)vox_opt.pcr_adds.size()) throw std::runtime_error("margi: opt_state beyond the factor");   // exit(0) upstream
      if (opt_state >= 0) {
        pcr_add = vox_opt.pcr_adds[opt_state];
        eig_value = vox_opt.eig_values[opt_state];
        eig_vector = vox_opt.eig_vectors[opt_state];
        opt_state = -1;
        for (int i = 0; i < mgsize; i++)
          if (pcrs_local[c.mp[i]].N != 0) cluster_transform(pcrs_world[i], pcrs_local[c.mp[i]], x_buf[i]);
      } else {
        pcr_add = pcr_fix;
        for (int i = 0; i < win_count; i++)
          if (pcrs_local[c.mp[i]].N != 0) {
            cluster_transform(pcrs_world[i], pcrs_local[c.mp[i]], x_buf[i]);
            pcr_add += pcrs_world[i];
          }
        if (plane.is_plane) eig_sym3(pcr_add.cov(), eig_value, eig_vector);
      }
      if (pcr_fix.N < c.prm.max_points && plane.is_plane)
        if (pcr_add.N - last_num >= 5 || last_num <= 10) {
          do_plane_update();
          last_num = pcr_add.N;
        }
      if (pcr_fix.N < c.prm.max_points) {
        for (int i = 0; i < mgsize; i++)
          if (pcrs_world[i].N != 0) {
            pcr_fix += pcrs_world[i];
            for (PointVar pv : points[c.mp[i]]) {
              pv.pnt = x_buf[i].R * pv.pnt + x_buf[i].p;
              point_fix.push_back(pv);
            }
          }
      } else {
        for (int i = 0; i < mgsize; i++)
          if (pcrs_world[i].N != 0) pcr_add -= pcrs_world[i];
        if (point_fix.size() != 0) std::vector<PointVar>().swap(point_fix);
      }
      for (int i = 0; i < mgsize; i++)
        if (pcrs_local[c.mp[i]].N != 0) {
          pcrs_local[c.mp[i]].clear();
          points[c.mp[i]].clear();
        }
      isexist = !(pcr_fix.N >= pcr_add.N);
    } else {
      isexist = false;
      for (int i = 0; i < 8; i++)
        if (leaves[i]) {
          leaves[i]->margi(win_count, mgsize, x_buf, vox_opt, c);
          isexist = isexist || leaves[i]->isexist;
        }
    }
  }

  // :1308-1333
  void tras_opt(LidarFactor& vox_opt, TreeCtx& c) {
    if (octo_state == 0) {
      if (layer >= 0 && isexist && plane.is_plane && has_sw) {
        if (eig_value[0] / eig_value[1] > 0.12) return;
        std::vector<PointCluster> pcrs(wdsize);
        for (int i = 0; i < wdsize; i++) pcrs[i] = pcrs_local[c.mp[i]];
        opt_state = (int)vox_opt.plvec_voxels.size();
        vox_opt.push_voxel(pcrs, pcr_fix, 1.0, eig_value, eig_vector, pcr_add);
      }
    } else {
      for (int i = 0; i < 8; i++)
        if (leaves[i]) leaves[i]->tras_opt(vox_opt, c);
    }
  }

  // :1482-1500
  void clear_slwd() {
    if (octo_state != 0)
      for (int i = 0; i < 8; i++)
        if (leaves[i]) leaves[i]->clear_slwd();
    if (has_sw) sw_release();
  }
};

// One leaf of the map as the tests look at it.
struct LeafView {
  uint64_t node_id;   // the batch voxeliser's canonical id (vxo_voxelize.hpp): [x:16 | y:16 | z:16 | path:9 | pad:4 | layer:3]
  const TreeNode* node;
};

class LocalMap {
 public:
  LocalMapParams prm;
  std::vector<int> mp;
  std::unordered_map<LocKey, std::unique_ptr<TreeNode>, LocHash> surf_map;
  std::vector<LocKey> root_order;                      // first-insertion order of surf_map
  std::unordered_set<LocKey, LocHash> slide_set;       // surf_map_slide
  std::vector<LocKey> slide_order;

  explicit LocalMap(const LocalMapParams& p) : prm(p), mp(p.win_size) { for (int i = 0; i < p.win_size; i++) mp[i] = i; }

  // cut_voxel_multi (voxel_map.hpp:1545-1639): scan `ord` = win_count - 1; pv.pnt body frame, pv.var already in the world frame
  // (pvec_update), pwld the world points
  void cut_voxel(int ord, const std::vector<PointVar>& pvec, const std::vector<V3>& pwld) {
    TreeCtx c{prm, mp};
    std::unordered_map<TreeNode*, std::vector<int>> map_pvec;
    std::vector<TreeNode*> touched;                    // stands in for the hash order of map_pvec
    for (size_t i = 0; i < pvec.size(); i++) {
      const LocKey position = voxel_of(pwld[i], prm.voxel_size);
      TreeNode* ot;
      auto iter = surf_map.find(position);
      if (iter != surf_map.end()) {
        ot = iter->second.get();
        ot->isexist = true;
      } else {
        ot = new TreeNode(0, prm.win_size);
        ot->voxel_center[0] = (0.5 + position.x) * prm.voxel_size;
        ot->voxel_center[1] = (0.5 + position.y) * prm.voxel_size;
        ot->voxel_center[2] = (0.5 + position.z) * prm.voxel_size;
        ot->quater_length = prm.voxel_size / 4.0;
        surf_map[position].reset(ot);
        root_order.push_back(position);
      }
      if (slide_set.insert(position).second) slide_order.push_back(position);
      auto& lst = map_pvec[ot];
      if (lst.empty()) touched.push_back(ot);
      lst.push_back((int)i);
    }
    if ((int)touched.size() < prm.thread_num) return;   // :1603-1605: fewer touched roots than threads -> nothing is pushed
    for (TreeNode* ot : touched)
      for (int k : map_pvec[ot]) ot->allocate(ord, pvec[k], pwld[k], c);
  }

  // multi_recut (voxelslam.cpp:1396-1453): recut of every voxel of the slide map, then tras_opt into the factor
  void recut(int win_count, const std::vector<Pose>& xs, LidarFactor& voxopt) {
    TreeCtx c{prm, mp};
    if ((int)slide_order.size() < prm.thread_num) return;
    for (const LocKey& k : slide_order) surf_map[k]->recut(win_count, xs, c);
    for (const LocKey& k : slide_order) surf_map[k]->tras_opt(voxopt, c);
  }

  // multi_margi (voxelslam.cpp:1313-1393): margi(win_count, 1, ...) on the slide map, then voxels without live content leave it
  void margi(int win_count, const std::vector<Pose>& xs, const LidarFactor& voxopt) {
    TreeCtx c{prm, mp};
    if ((int)slide_order.size() < prm.thread_num) return;
    for (const LocKey& k : slide_order) surf_map[k]->margi(win_count, 1, xs, voxopt, c);
    std::vector<LocKey> keep;
    for (const LocKey& k : slide_order) {
      TreeNode* n = surf_map[k].get();
      if (n->isexist) keep.push_back(k);
      else { n->clear_slwd(); slide_set.erase(k); }
    }
    slide_order.swap(keep);
  }

  // voxelslam.cpp:1683-1687
  void slide(int mgsize) {
    for (int i = 0; i < prm.win_size; i++) {
      mp[i] += mgsize;
      if (mp[i] >= prm.win_size) mp[i] -= prm.win_size;
    }
  }

  // every node with octo_state == 0, roots in insertion order, children in octant order
  bool leaves(std::vector<LeafView>& out) const {
    bool in_range = true;
    for (const LocKey& k : root_order) {
      if (k.x < -32768 || k.x > 32767 || k.y < -32768 || k.y > 32767 || k.z < -32768 || k.z > 32767) { in_range = false; continue; }
      const uint64_t root48 = ((uint64_t)(k.x + 32768) << 32) | ((uint64_t)(k.y + 32768) << 16) | (uint64_t)(k.z + 32768);
      walk(surf_map.at(k).get(), root48, 0, out);
    }
    return in_range;
  }

 private:
  static void walk(const TreeNode* n, uint64_t root48, uint64_t path, std::vector<LeafView>& out) {
    if (n->octo_state == 0) { out.push_back(LeafView{(root48 << 16) | (path << 7) | (uint64_t)n->layer, n}); return; }
    for (int i = 0; i < 8; i++)
      if (n->leaves[i]) walk(n->leaves[i].get(), root48, path | ((uint64_t)i << (3 * (2 - n->layer))), out);
  }
};

}  // namespace vxo
