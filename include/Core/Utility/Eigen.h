// Core/Utility/Eigen.h -- the two typedefs of O3D/Core/Utility/Eigen.h:36-37
// that the 6x6 normal-equation interface is expressed in.
#pragma once
#include <Eigen/Core>

namespace Eigen {
typedef Eigen::Matrix<double, 6, 6> Matrix6d;
typedef Eigen::Matrix<double, 6, 1> Vector6d;
}  // namespace Eigen
