// open3d_registration_interpose.cpp -- link-time substitution of Open3D's registration entry points.
//
// The reference's callers call  open3d::RegistrationICP(...)  /  open3d::EvaluateRegistration(...)  directly
// (src/evaluation.cpp:260-271; declared at O3D/Core/Registration/Registration.h:96-107, defined in Open3D's libCore,
// Registration.cpp:98-186).  This translation unit DEFINES those two functions -- same namespace, same signatures,
// hence the same mangled names -- and forwards them to the MI355X path (visma_icp_open3d.hpp -> the C ABI of
// visma_icp.h).  Built into a shared library that the application links AHEAD of Open3D's Core library (or that is
// LD_PRELOADed), it makes the dynamic linker bind the callers' references to these definitions: src/evaluation.cpp
// and example_evaluate run their ICP on the GPU without a changed source line ("optionally also providing
// the open3d::RegistrationICP symbol for link-time substitution").  Everything else of Open3D (PointCloud, I/O, the
// estimator classes and their virtual ComputeTransformation / ComputeRMSE) stays Open3D's.
//
// Compile it against the application's OWN Open3D headers and with the application's Eigen storage order: the mangled
// name carries Eigen::Matrix4d's options (`Eigen::Matrix<double, 4, 4, 0, 4, 4>` column-major, `..., 1, 4, 4>` under
// -DEIGEN_DEFAULT_TO_ROW_MAJOR, which VISMA's CMakeLists.txt:11-12 sets), and open3d::PointCloud's layout is whatever
// those headers say.  INTEGRATION.md section 1b has the build and link lines; tests/cpp/build_shim.py builds both
// storage orders against /root/reference/thirdparty/Open3D/src and tests/test_interpose.py runs the column-major one
// against the compiled reference standing in for libCore.
#include <Core/Geometry/PointCloud.h>
#include <Core/Registration/Registration.h>

#include "visma_icp_open3d.hpp"

#include <atomic>

namespace {
std::atomic<int> g_calls(0);   // how often the substituted entry points ran (a caller can check that the substitution took)
}

extern "C" __attribute__((visibility("default"))) int visma_open3d_interpose_calls() { return g_calls.load(); }

namespace open3d {

// Registration.h:96-99 (default argument lives in the header's declaration)
RegistrationResult EvaluateRegistration(const PointCloud &source, const PointCloud &target,
                                        double max_correspondence_distance, const Eigen::Matrix4d &transformation)
{
    ++g_calls;
    return cicp::EvaluateRegistration(source, target, max_correspondence_distance, transformation);
}

// Registration.h:102-107
RegistrationResult RegistrationICP(const PointCloud &source, const PointCloud &target, double max_correspondence_distance,
                                   const Eigen::Matrix4d &init, const TransformationEstimation &estimation,
                                   const ICPConvergenceCriteria &criteria)
{
    ++g_calls;
    return cicp::RegistrationICP(source, target, max_correspondence_distance, init, estimation, criteria);
}

}  // namespace open3d
