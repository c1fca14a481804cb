// Core/Registration/TransformationEstimation.h -- the estimator plugin
// interface of the path (shape of O3D/Core/Registration/
// TransformationEstimation.h:38-111).  The two stock estimators are declared
// here and implemented in visma_icp_open3d.hpp on top of the C ABI's
// statistics / solve functions.
#pragma once
#include <Eigen/Core>
#include <vector>

namespace open3d {

class PointCloud;

typedef std::vector<Eigen::Vector2i> CorrespondenceSet;

enum class TransformationEstimationType {
    Unspecified = 0,
    PointToPoint = 1,
    PointToPlane = 2,
    ColoredICP = 3,
};

class TransformationEstimation {
public:
    TransformationEstimation() {}
    virtual ~TransformationEstimation() {}
    virtual TransformationEstimationType GetTransformationEstimationType() const = 0;
    virtual double ComputeRMSE(const PointCloud &source, const PointCloud &target,
                               const CorrespondenceSet &corres) const = 0;
    virtual Eigen::Matrix4d ComputeTransformation(const PointCloud &source,
                                                  const PointCloud &target,
                                                  const CorrespondenceSet &corres) const = 0;
};

class TransformationEstimationPointToPoint : public TransformationEstimation {
public:
    TransformationEstimationPointToPoint(bool with_scaling = false) : with_scaling_(with_scaling) {}
    ~TransformationEstimationPointToPoint() override {}
    TransformationEstimationType GetTransformationEstimationType() const override
    {
        return TransformationEstimationType::PointToPoint;
    }
    inline double ComputeRMSE(const PointCloud &source, const PointCloud &target,
                              const CorrespondenceSet &corres) const override;
    inline Eigen::Matrix4d ComputeTransformation(const PointCloud &source, const PointCloud &target,
                                                 const CorrespondenceSet &corres) const override;
    bool with_scaling_ = false;
};

class TransformationEstimationPointToPlane : public TransformationEstimation {
public:
    TransformationEstimationPointToPlane() {}
    ~TransformationEstimationPointToPlane() override {}
    TransformationEstimationType GetTransformationEstimationType() const override
    {
        return TransformationEstimationType::PointToPlane;
    }
    inline double ComputeRMSE(const PointCloud &source, const PointCloud &target,
                              const CorrespondenceSet &corres) const override;
    inline Eigen::Matrix4d ComputeTransformation(const PointCloud &source, const PointCloud &target,
                                                 const CorrespondenceSet &corres) const override;
};

}  // namespace open3d
