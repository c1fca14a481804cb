// ref_harness.cpp -- C-ABI harness around the REAL reference implementation.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code; it is
// compiled TOGETHER WITH the reference's own sources where they lie under
// /root/reference (recipe: oracle/Makefile, target `ref`) into
// oracle/_ref/libvisma_ref.so, which is git-ignored.  It exposes the Open3D
// 0.3.0 registration path that VISMA's estimator plugs into
// (O3D = thirdparty/Open3D/src):
//   O3D/Core/Registration/Registration.cpp:141-186   RegistrationICP
//   O3D/Core/Registration/Registration.cpp:129-139   EvaluateRegistration
//   O3D/Core/Registration/TransformationEstimation.cpp:35-103
//   O3D/Core/Geometry/PointCloud.cpp:75-87,122-142
//   O3D/Core/Utility/Eigen.cpp:58-68,88-106
//   O3D/Core/Geometry/DownSample.cpp:179-220        VoxelDownSample
//   O3D/Core/Geometry/EstimateNormals.cpp:114-153   EstimateNormals
//
// NOTE on VISMA's own src/constrained_ICP.cpp: it includes "Core/Core.h",
// which includes the CMake-GENERATED "../Open3DConfig.h"; that header does
// not exist in the tree, so that one file is unbuildable here without
// writing a stand-in (not allowed).  Its two methods (src/constrained_ICP.cpp
// :13-23, :25-37) are textually the same statements as Open3D's
// TransformationEstimationPointToPoint (TransformationEstimation.cpp:35-59),
// which IS compiled below and is what `estimator == 1` runs.
#include <Core/Geometry/KDTreeSearchParam.h>
#include <Core/Geometry/PointCloud.h>
#include <Core/Registration/Registration.h>
#include <Core/Registration/TransformationEstimation.h>
#include <Core/Utility/Eigen.h>

#include <cstdint>
#include <cstring>
#include <memory>

using open3d::PointCloud;

namespace {

void fill_cloud(PointCloud &pc, const double *xyz, int64_t n,
                const double *normals)
{
    pc.points_.resize((size_t)n);
    for (int64_t i = 0; i < n; i++)
        pc.points_[i] = Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    if (normals) {
        pc.normals_.resize((size_t)n);
        for (int64_t i = 0; i < n; i++)
            pc.normals_[i] = Eigen::Vector3d(normals[3 * i], normals[3 * i + 1],
                                             normals[3 * i + 2]);
    }
}

Eigen::Matrix4d from_rowmajor(const double T[16])
{
    Eigen::Matrix4d M;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) M(i, j) = T[i * 4 + j];
    return M;
}

void to_rowmajor(const Eigen::Matrix4d &M, double T[16])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) T[i * 4 + j] = M(i, j);
}

open3d::CorrespondenceSet make_corr(const int32_t *corr, int64_t k)
{
    open3d::CorrespondenceSet cs((size_t)k);
    for (int64_t i = 0; i < k; i++)
        cs[i] = Eigen::Vector2i(corr[2 * i], corr[2 * i + 1]);
    return cs;
}

std::unique_ptr<open3d::TransformationEstimation> make_est(int estimator,
                                                           int with_scaling)
{
    if (estimator == 2)
        return std::unique_ptr<open3d::TransformationEstimation>(
            new open3d::TransformationEstimationPointToPlane());
    return std::unique_ptr<open3d::TransformationEstimation>(
        new open3d::TransformationEstimationPointToPoint(with_scaling != 0));
}

void export_result(const open3d::RegistrationResult &r, int64_t ns,
                   double T_out[16], double *fitness, double *rmse,
                   int32_t *idx_out, int64_t *k_out)
{
    to_rowmajor(r.transformation_, T_out);
    *fitness = r.fitness_;
    *rmse = r.inlier_rmse_;
    *k_out = (int64_t)r.correspondence_set_.size();
    if (idx_out) {
        for (int64_t i = 0; i < ns; i++) idx_out[i] = -1;
        for (const auto &c : r.correspondence_set_) idx_out[c[0]] = c[1];
    }
}

}  // namespace

extern "C" {

// estimator: 1 = point-to-point (== VISMA 4DoF class arithmetic), 2 = plane
int ref_registration_icp(const double *src, int64_t ns, const double *src_normals,
                         const double *tgt, int64_t nt, const double *tgt_normals,
                         double max_dist, const double init[16], int estimator,
                         int with_scaling, double rel_fitness, double rel_rmse,
                         int max_iter, double T_out[16], double *fitness,
                         double *rmse, int32_t *idx_out, int64_t *k_out)
{
    PointCloud s, t;
    fill_cloud(s, src, ns, src_normals);
    fill_cloud(t, tgt, nt, tgt_normals);
    auto est = make_est(estimator, with_scaling);
    open3d::RegistrationResult r = open3d::RegistrationICP(
        s, t, max_dist, from_rowmajor(init), *est,
        open3d::ICPConvergenceCriteria(rel_fitness, rel_rmse, max_iter));
    export_result(r, ns, T_out, fitness, rmse, idx_out, k_out);
    return 0;
}

int ref_evaluate_registration(const double *src, int64_t ns, const double *tgt,
                              int64_t nt, double max_dist, const double T[16],
                              double *fitness, double *rmse, int32_t *idx_out,
                              int64_t *k_out)
{
    PointCloud s, t;
    fill_cloud(s, src, ns, nullptr);
    fill_cloud(t, tgt, nt, nullptr);
    open3d::RegistrationResult r =
        open3d::EvaluateRegistration(s, t, max_dist, from_rowmajor(T));
    double Tout[16];
    export_result(r, ns, Tout, fitness, rmse, idx_out, k_out);
    return 0;
}

void ref_transform_points(double *xyz, int64_t n, double *normals,
                          const double T[16])
{
    PointCloud p;
    fill_cloud(p, xyz, n, normals);
    p.Transform(from_rowmajor(T));
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            xyz[3 * i + a] = p.points_[i](a);
            if (normals) normals[3 * i + a] = p.normals_[i](a);
        }
}

// O3D/Core/Geometry/DownSample.cpp:179-220, in the reference's own output order
int64_t ref_voxel_down_sample(const double *xyz, const double *normals, const double *colors,
                              int64_t n, double voxel_size, double *out_xyz, double *out_normals,
                              double *out_colors)
{
    PointCloud p;
    fill_cloud(p, xyz, n, normals);
    if (colors) {
        p.colors_.resize((size_t)n);
        for (int64_t i = 0; i < n; i++)
            p.colors_[i] = Eigen::Vector3d(colors[3 * i], colors[3 * i + 1], colors[3 * i + 2]);
    }
    std::shared_ptr<PointCloud> o = open3d::VoxelDownSample(p, voxel_size);
    const int64_t m = (int64_t)o->points_.size();
    for (int64_t i = 0; i < m; i++)
        for (int a = 0; a < 3; a++) {
            out_xyz[3 * i + a] = o->points_[i](a);
            if (normals && out_normals) out_normals[3 * i + a] = o->normals_[i](a);
            if (colors && out_colors) out_colors[3 * i + a] = o->colors_[i](a);
        }
    return m;
}

void ref_nn_distance(const double *src, int64_t ns, const double *tgt,
                     int64_t nt, double *dist)
{
    PointCloud s, t;
    fill_cloud(s, src, ns, nullptr);
    fill_cloud(t, tgt, nt, nullptr);
    std::vector<double> d = open3d::ComputePointCloudToPointCloudDistance(s, t);
    std::memcpy(dist, d.data(), d.size() * sizeof(double));
}

double ref_compute_rmse(const double *src, int64_t ns, const double *tgt,
                        int64_t nt, const double *tgt_normals,
                        const int32_t *corr, int64_t k, int estimator)
{
    PointCloud s, t;
    fill_cloud(s, src, ns, nullptr);
    fill_cloud(t, tgt, nt, tgt_normals);
    return make_est(estimator, 0)->ComputeRMSE(s, t, make_corr(corr, k));
}

void ref_compute_transformation(const double *src, int64_t ns, const double *tgt,
                                int64_t nt, const double *tgt_normals,
                                const int32_t *corr, int64_t k, int estimator,
                                int with_scaling, double T_out[16])
{
    PointCloud s, t;
    fill_cloud(s, src, ns, nullptr);
    fill_cloud(t, tgt, nt, tgt_normals);
    to_rowmajor(make_est(estimator, with_scaling)
                    ->ComputeTransformation(s, t, make_corr(corr, k)),
                T_out);
}

int ref_solve_jacobian_system(const double JTJ[36], const double JTr[6],
                              double T_out[16])
{
    Eigen::Matrix6d A;
    Eigen::Vector6d b;
    for (int i = 0; i < 6; i++) {
        b(i) = JTr[i];
        for (int j = 0; j < 6; j++) A(i, j) = JTJ[i * 6 + j];
    }
    bool ok;
    Eigen::Matrix4d X;
    std::tie(ok, X) = open3d::SolveJacobianSystemAndObtainExtrinsicMatrix(A, b);
    to_rowmajor(X, T_out);
    return ok ? 1 : 0;
}

void ref_vector6d_to_matrix4d(const double x[6], double T_out[16])
{
    Eigen::Vector6d v;
    for (int i = 0; i < 6; i++) v(i) = x[i];
    to_rowmajor(open3d::TransformVector6dToMatrix4d(v), T_out);
}

// open3d::EstimateNormals; search_type 0 KNN(knn) | 1 Radius(radius) | 2 Hybrid(radius, knn);
// normals_in may be NULL (the cloud has no normals)
void ref_estimate_normals(const double *xyz, int64_t n, const double *normals_in, int search_type,
                          int knn, double radius, double *out)
{
    PointCloud pc;
    fill_cloud(pc, xyz, n, normals_in);
    if (search_type == 0) open3d::EstimateNormals(pc, open3d::KDTreeSearchParamKNN(knn));
    else if (search_type == 1) open3d::EstimateNormals(pc, open3d::KDTreeSearchParamRadius(radius));
    else open3d::EstimateNormals(pc, open3d::KDTreeSearchParamHybrid(radius, knn));
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) out[3 * i + a] = pc.normals_[i](a);
}

}  // extern "C"
