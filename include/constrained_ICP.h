// constrained_ICP.h -- drop-in for the reference header of the same name
// (include/constrained_ICP.h:1-34): callers keep
//
//     #include "constrained_ICP.h"
//     ...
//     open3d::RegistrationICP(*model, *scene, threshold, init,
//         open3d::cicp::TransformationEstimationPointToPoint4DoF(),
//         open3d::ICPConvergenceCriteria());
//
// and get the MI355X path.  "Core/Core.h" resolves to the real Open3D umbrella
// when its include directory comes first, else to the minimal stand-alone set
// shipped next to this file; visma_icp_open3d.hpp then declares the estimator
// class and the GPU-backed RegistrationICP driver on whichever types it found.
#pragma once

#ifdef VISMA_ICP_OPEN3D_NO_UMBRELLA
// (an Open3D source tree that CMake has not configured has no Open3DConfig.h for its umbrella header to include: the
//  headers of the path, one by one -- tests/cpp/build_shim.py compiles the drivers against the reference's tree so)
#include <Core/Geometry/PointCloud.h>
#include <Core/Registration/Registration.h>
#include <IO/ClassIO/PointCloudIO.h>
#else
#include "Core/Core.h"
#endif
#include "visma_icp_open3d.hpp"
