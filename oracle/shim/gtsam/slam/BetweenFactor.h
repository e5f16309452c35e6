// TEST INFRASTRUCTURE ONLY -- see gtsam/geometry/Pose3.h in this directory.  Not GTSAM.
#pragma once
#include "../geometry/Pose3.h"
namespace gtsam {
template <typename T> class BetweenFactor : public NonlinearFactor {
 public:
  BetweenFactor(Key, Key, const T&, const noiseModel::Diagonal::shared_ptr&) {}
};
}  // namespace gtsam
