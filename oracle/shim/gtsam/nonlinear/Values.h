// TEST INFRASTRUCTURE ONLY -- see gtsam/geometry/Pose3.h in this directory.  Not GTSAM.
#pragma once
#include "../geometry/Pose3.h"
