// mesh_refine_driver.cpp -- feh::gpu::ICPRefinement(scene, models, ...) (both clouds made on the device) against
// the same steps through the separate shim calls, the way src/evaluation.cpp:248-271 strings them together.
// Self-contained: builds three box meshes, a scan, runs both paths, prints MESH_REFINE_OK when they agree.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "constrained_ICP.h"
#include "visma_geometry.hpp"

using namespace open3d;

struct Model {   // the fields of the reference's feh::Model that ICPRefinement touches
    Eigen::Matrix<double, Eigen::Dynamic, 3> V_;
    Eigen::Matrix<int, Eigen::Dynamic, 3> F_;
    Eigen::Matrix4d model_to_scene_;
};

static Model box(double sx, double sy, double sz, double ox, double oy, double oz, double yaw, double tx)
{
    Model m;
    m.V_.resize(8, 3);
    int r = 0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int c = 0; c < 2; c++, r++) { m.V_(r, 0) = a * sx + ox; m.V_(r, 1) = b * sy + oy; m.V_(r, 2) = c * sz + oz; }
    const int f[12][3] = {{0, 1, 3}, {0, 3, 2}, {4, 6, 7}, {4, 7, 5}, {0, 4, 5}, {0, 5, 1}, {2, 3, 7}, {2, 7, 6}, {0, 2, 6}, {0, 6, 4}, {1, 5, 7}, {1, 7, 3}};
    m.F_.resize(12, 3);
    for (int i = 0; i < 12; i++)
        for (int k = 0; k < 3; k++) m.F_(i, k) = f[i][k];
    m.model_to_scene_ = Eigen::Matrix4d::Identity();
    m.model_to_scene_(0, 0) = std::cos(yaw); m.model_to_scene_(0, 2) = std::sin(yaw);
    m.model_to_scene_(2, 0) = -std::sin(yaw); m.model_to_scene_(2, 2) = std::cos(yaw);
    m.model_to_scene_(0, 3) = tx; m.model_to_scene_(1, 3) = 0.1 * tx;
    return m;
}

int main()
{
    std::vector<Model> models;
    models.push_back(box(0.6, 0.4, 0.5, 0.0, 0.0, 0.0, 0.3, 0.0));
    models.push_back(box(0.3, 0.9, 0.2, 0.1, -0.2, 0.05, -0.5, 0.8));
    models.push_back(box(0.5, 0.5, 0.5, 0.2, -0.4, 0.05, 1.1, -0.7));
    const int samples = 20000;
    const double voxel = 0.02, max_distance = 0.05;
    const uint64_t seed = 5;
    // ---- the scan: the models seen again, denser, after a small motion
    auto scan = std::make_shared<PointCloud>();
    for (size_t k = 0; k < models.size(); k++) {
        PointCloud part;
        part.points_ = feh::gpu::SamplePointCloudFromMesh<double>(models[k].V_, models[k].F_, 5 * samples, feh::gpu::SamplingMode::Surface, 100 + k);
        part.Transform(models[k].model_to_scene_);
        *scan += part;
    }
    Eigen::Matrix4d T_gt = Eigen::Matrix4d::Identity();
    T_gt(0, 3) = 0.012; T_gt(1, 3) = -0.007; T_gt(2, 3) = 0.009;
    scan->Transform(T_gt);
    const Eigen::Matrix4d T0 = Eigen::Matrix4d::Identity();
    // ---- (A) one call at a time, as the reference's ICPRefinement is written
    auto scene_est = std::make_shared<PointCloud>();
    for (size_t k = 0; k < models.size(); k++) {
        auto model_ptr = std::make_shared<PointCloud>();
        model_ptr->points_ = feh::gpu::SamplePointCloudFromMesh<double>(models[k].V_, models[k].F_, samples, feh::gpu::SamplingMode::Surface, seed + k);
        model_ptr->Transform(models[k].model_to_scene_);
        *scene_est += *model_ptr;
    }
    auto scene = cicp::VoxelDownSample(*scan, voxel);
    RegistrationResult ra = cicp::RegistrationICP(*scene_est, *scene, max_distance, T0);
    // ---- (B) both clouds made on the device
    PointCloud est_b;
    RegistrationResult rb = feh::gpu::ICPRefinement(*scan, models, T0, samples, voxel, max_distance, feh::gpu::SamplingMode::Surface, seed, &est_b);
    // ---- (C) the reference's container
    std::unordered_map<int, Model> src;
    for (size_t k = 0; k < models.size(); k++) src[(int)k] = models[k];
    RegistrationResult rc = feh::gpu::ICPRefinementMap(*scan, src, T0, samples, voxel, max_distance, feh::gpu::SamplingMode::Surface, seed);
    int bad = 0;
    if (est_b.points_.size() != scene_est->points_.size()) { std::printf("sizes differ %zu %zu\n", est_b.points_.size(), scene_est->points_.size()); bad++; }
    else
        for (size_t i = 0; i < est_b.points_.size(); i++)
            if (!(est_b.points_[i] == scene_est->points_[i])) { if (bad < 5) std::printf("point %zu differs\n", i); bad++; }
    if (!(ra.transformation_ == rb.transformation_)) { std::printf("transformations differ\n"); bad++; }
    if (ra.correspondence_set_.size() != rb.correspondence_set_.size() || ra.fitness_ != rb.fitness_ || ra.inlier_rmse_ != rb.inlier_rmse_) { std::printf("statistics differ\n"); bad++; }
    if ((rb.transformation_ - T_gt).cwiseAbs().maxCoeff() > 5e-3) { std::printf("did not converge to the motion\n"); bad++; }
    if (rc.correspondence_set_.size() < scene_est->points_.size() / 2 || (rc.transformation_ - T_gt).cwiseAbs().maxCoeff() > 5e-3) { std::printf("map variant off\n"); bad++; }
    std::printf("K = %zu of %zu, fitness %.4f, rmse %.6f\n", rb.correspondence_set_.size(), est_b.points_.size(), rb.fitness_, rb.inlier_rmse_);
    std::printf(bad ? "MESH_REFINE_FAILED\n" : "MESH_REFINE_OK\n");
    return bad ? 1 : 0;
}
