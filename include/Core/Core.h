// Core/Core.h -- minimal stand-alone stand-in for the Open3D 0.3.0 umbrella
// header, covering ONLY what the ICP registration path touches, so callers
// written against the reference (`#include "constrained_ICP.h"` +
// `open3d::RegistrationICP(...)`, src/evaluation.cpp:11,260-271;
// src/annotation.cpp:46-56) compile unchanged on a machine without Open3D.
// Where the real Open3D is installed, put ITS include directory first: the
// adapter (visma_icp_open3d.hpp) is written against either set of types.
#pragma once
#define VISMA_ICP_STANDALONE_OPEN3D_TYPES 1

#include "Utility/Eigen.h"
#include "Geometry/PointCloud.h"
#include "Registration/TransformationEstimation.h"
#include "Registration/Registration.h"
