// Core/Registration/Registration.h -- criteria / result types and the
// RegistrationICP entry point of the path (shape of
// O3D/Core/Registration/Registration.h:46-107).  In this stand-alone header set
// open3d::RegistrationICP forwards to the MI355X driver
// open3d::cicp::RegistrationICP (visma_icp_open3d.hpp).
#pragma once
#include <Eigen/Core>

#include "TransformationEstimation.h"

namespace open3d {

class ICPConvergenceCriteria {
public:
    ICPConvergenceCriteria(double relative_fitness = 1e-6, double relative_rmse = 1e-6,
                           int max_iteration = 30)
        : relative_fitness_(relative_fitness), relative_rmse_(relative_rmse),
          max_iteration_(max_iteration) {}
    double relative_fitness_;
    double relative_rmse_;
    int max_iteration_;
};

class RegistrationResult {
public:
    RegistrationResult(const Eigen::Matrix4d &transformation = Eigen::Matrix4d::Identity())
        : transformation_(transformation), inlier_rmse_(0.0), fitness_(0.0) {}
    Eigen::Matrix4d transformation_;
    CorrespondenceSet correspondence_set_;
    double inlier_rmse_;
    double fitness_;
};

inline RegistrationResult EvaluateRegistration(
    const PointCloud &source, const PointCloud &target, double max_correspondence_distance,
    const Eigen::Matrix4d &transformation = Eigen::Matrix4d::Identity());

inline RegistrationResult RegistrationICP(
    const PointCloud &source, const PointCloud &target, double max_correspondence_distance,
    const Eigen::Matrix4d &init = Eigen::Matrix4d::Identity(),
    const TransformationEstimation &estimation = TransformationEstimationPointToPoint(false),
    const ICPConvergenceCriteria &criteria = ICPConvergenceCriteria());

}  // namespace open3d
