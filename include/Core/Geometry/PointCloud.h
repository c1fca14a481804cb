// Core/Geometry/PointCloud.h -- the data carrier of the path
// (shape of O3D/Core/Geometry/PointCloud.h:42-89: AoS f64 points/normals/colors).
#pragma once
#include <Eigen/Core>
#include <memory>
#include <vector>

#include "KDTreeSearchParam.h"

namespace open3d {

class PointCloud {
public:
    PointCloud() {}
    virtual ~PointCloud() {}

    void Clear() { points_.clear(); normals_.clear(); colors_.clear(); }
    bool IsEmpty() const { return !HasPoints(); }
    bool HasPoints() const { return points_.size() > 0; }
    bool HasNormals() const { return points_.size() > 0 && normals_.size() == points_.size(); }
    bool HasColors() const { return points_.size() > 0 && colors_.size() == points_.size(); }

    // Rigid/affine motion of points (w = 1) and normals (w = 0); the 4th row of
    // the matrix is ignored, as pinned by the reference's PointCloud.Transform test.
    void Transform(const Eigen::Matrix4d &T)
    {
        for (auto &p : points_) {
            const Eigen::Vector3d q = T.block<3, 3>(0, 0) * p + T.block<3, 1>(0, 3);
            p = q;
        }
        for (auto &n : normals_) {
            const Eigen::Vector3d m = T.block<3, 3>(0, 0) * n;
            n = m;
        }
    }

    PointCloud &operator+=(const PointCloud &o)
    {
        const bool keep_n = (!HasPoints() || HasNormals()) && o.HasNormals();
        const bool keep_c = (!HasPoints() || HasColors()) && o.HasColors();
        if (!keep_n) normals_.clear(); else normals_.insert(normals_.end(), o.normals_.begin(), o.normals_.end());
        if (!keep_c) colors_.clear(); else colors_.insert(colors_.end(), o.colors_.begin(), o.colors_.end());
        points_.insert(points_.end(), o.points_.begin(), o.points_.end());
        return *this;
    }

    std::vector<Eigen::Vector3d> points_;
    std::vector<Eigen::Vector3d> normals_;
    std::vector<Eigen::Vector3d> colors_;
};

/// Down-sample with a voxel grid (shape of O3D/Core/Geometry/PointCloud.h:101-105);
/// implemented on the GPU in visma_icp_open3d.hpp.
inline std::shared_ptr<PointCloud> VoxelDownSample(const PointCloud &input, double voxel_size);

/// Normals from the covariance of each point's neighbours (shape of
/// O3D/Core/Geometry/PointCloud.h:140-146); on the GPU, in visma_icp_open3d.hpp.  Existing normals
/// keep their sign.
inline bool EstimateNormals(PointCloud &cloud, const KDTreeSearchParam &search_param = KDTreeSearchParamKNN());
/// (O3D/Core/Geometry/PointCloud.h:148-158) host loops, in visma_icp_open3d.hpp
inline bool OrientNormalsToAlignWithDirection(PointCloud &cloud,
                                              const Eigen::Vector3d &orientation_reference = Eigen::Vector3d(0.0, 0.0, 1.0));
inline bool OrientNormalsTowardsCameraLocation(PointCloud &cloud,
                                               const Eigen::Vector3d &camera_location = Eigen::Vector3d::Zero());

}  // namespace open3d
