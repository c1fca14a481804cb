// Core/Registration/TransformationEstimation.h (stand-alone header set of the MI355X ICP shim)
//
// The estimator plugin interface that open3d::RegistrationICP accepts, with the names and call
// signatures of Open3D 0.3.0 so that VISMA's call sites and user plugins compile unchanged.
// Nothing here computes: the two stock estimators get their bodies in visma_icp_open3d.hpp, on
// top of the C ABI (statistics from the GPU, closed-form / Gauss-Newton solve on the host).
#pragma once
#include <Eigen/Core>
#include <vector>

namespace open3d {

class PointCloud;
using CorrespondenceSet = std::vector<Eigen::Vector2i>;   // (source index, target index) pairs

enum class TransformationEstimationType { Unspecified = 0, PointToPoint = 1, PointToPlane = 2, ColoredICP = 3 };

// The plugin interface: what open3d::RegistrationICP calls on an estimator
// (O3D/Core/Registration/TransformationEstimation.h:51-66).
class TransformationEstimation {
public:
    virtual ~TransformationEstimation() = default;
    virtual TransformationEstimationType GetTransformationEstimationType() const = 0;
    virtual double ComputeRMSE(const PointCloud &source, const PointCloud &target,
                               const CorrespondenceSet &corres) const = 0;
    virtual Eigen::Matrix4d ComputeTransformation(const PointCloud &source, const PointCloud &target,
                                                  const CorrespondenceSet &corres) const = 0;
};

// p' = c R p + t minimising sum |q - p'|^2 over the pairs (Umeyama); c = 1 unless with_scaling_.
class TransformationEstimationPointToPoint : public TransformationEstimation {
public:
    explicit TransformationEstimationPointToPoint(bool with_scaling = false) : with_scaling_(with_scaling) {}
    TransformationEstimationType GetTransformationEstimationType() const override { return kind; }
    double ComputeRMSE(const PointCloud &source, const PointCloud &target,
                       const CorrespondenceSet &corres) const override;
    Eigen::Matrix4d ComputeTransformation(const PointCloud &source, const PointCloud &target,
                                          const CorrespondenceSet &corres) const override;
    bool with_scaling_;
    static constexpr TransformationEstimationType kind = TransformationEstimationType::PointToPoint;
};

// One Gauss-Newton step on sum ((q - p') . n_q)^2; needs normals on the target cloud.
class TransformationEstimationPointToPlane : public TransformationEstimation {
public:
    TransformationEstimationType GetTransformationEstimationType() const override { return kind; }
    double ComputeRMSE(const PointCloud &source, const PointCloud &target,
                       const CorrespondenceSet &corres) const override;
    Eigen::Matrix4d ComputeTransformation(const PointCloud &source, const PointCloud &target,
                                          const CorrespondenceSet &corres) const override;
    static constexpr TransformationEstimationType kind = TransformationEstimationType::PointToPlane;
};

}  // namespace open3d
