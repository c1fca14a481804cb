// shim_driver.cpp -- exercises the C++/Eigen shim the way the reference's
// callers do (src/evaluation.cpp:260-271, src/annotation.cpp:35-61).
// Usage: shim_driver <mode> <in.bin> <out.bin>
//   in : int64 ns, int64 nt, double radius, int32 iters, int32 level, double init[16],
//        ns*3 doubles, nt*3 doubles, [nt*3 target normals, ns*3 source normals]
//   out: double T[16], fitness, rmse, K, extra
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "constrained_ICP.h"
#include "visma_geometry.hpp"

using namespace open3d;

// a user-defined estimator plugin: the generic path must keep working
class MyEstimator : public TransformationEstimation {
public:
    TransformationEstimationType GetTransformationEstimationType() const override
    {
        return TransformationEstimationType::Unspecified;
    }
    double ComputeRMSE(const PointCloud &s, const PointCloud &t, const CorrespondenceSet &c) const override
    {
        return inner.ComputeRMSE(s, t, c);
    }
    Eigen::Matrix4d ComputeTransformation(const PointCloud &s, const PointCloud &t,
                                          const CorrespondenceSet &c) const override
    {
        ++calls;
        return inner.ComputeTransformation(s, t, c);
    }
    cicp::TransformationEstimationPointToPoint4DoF inner;
    mutable int calls = 0;
};

// a class DERIVED from a stock estimator that overrides the solve: the override must be the one
// that runs (the reference always calls the virtual method, Registration.cpp:172-173)
class DerivedFromStock : public TransformationEstimationPointToPoint {
public:
    Eigen::Matrix4d ComputeTransformation(const PointCloud &s, const PointCloud &t,
                                          const CorrespondenceSet &c) const override
    {
        ++calls;
        return TransformationEstimationPointToPoint::ComputeTransformation(s, t, c);
    }
    mutable int calls = 0;
};

static void read_cloud(FILE *f, std::vector<Eigen::Vector3d> &v, int64_t n)
{
    v.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        double p[3];
        if (fread(p, sizeof(double), 3, f) != 3) { std::fprintf(stderr, "short read\n"); std::exit(2); }
        v[i] = Eigen::Vector3d(p[0], p[1], p[2]);
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    FILE *f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    int64_t ns, nt; double radius; int32_t iters, level; double init_rm[16];
    if (fread(&ns, 8, 1, f) != 1 || fread(&nt, 8, 1, f) != 1 || fread(&radius, 8, 1, f) != 1 ||
        fread(&iters, 4, 1, f) != 1 || fread(&level, 4, 1, f) != 1 || fread(init_rm, 8, 16, f) != 16)
        return 2;
    auto model = std::make_shared<PointCloud>();
    auto scene = std::make_shared<PointCloud>();
    read_cloud(f, model->points_, ns);
    read_cloud(f, scene->points_, nt);
    if (mode == "plane") { read_cloud(f, scene->normals_, nt); read_cloud(f, model->normals_, ns); }
    std::fclose(f);
    Eigen::Matrix4d init;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) init(i, j) = init_rm[i * 4 + j];

    RegistrationResult result;
    double extra = 0.0;
    try {
        if (mode == "icp4dof") {            // src/annotation.cpp:51-56 call shape
            result = open3d::RegistrationICP(*model, *scene, radius, init,
                                             open3d::cicp::TransformationEstimationPointToPoint4DoF(),
                                             open3d::ICPConvergenceCriteria(0.0, 0.0, iters));
        } else if (mode == "default") {     // src/evaluation.cpp:267-270 call shape
            result = open3d::RegistrationICP(*model, *scene, radius, init);
        } else if (mode == "plane") {       // src/evaluation.cpp:261-265
            result = open3d::RegistrationICP(*model, *scene, radius, init,
                                             open3d::TransformationEstimationPointToPlane(),
                                             open3d::ICPConvergenceCriteria(0.0, 0.0, iters));
        } else if (mode == "plugin") {
            MyEstimator est;
            result = open3d::RegistrationICP(*model, *scene, radius, init, est,
                                             open3d::ICPConvergenceCriteria(0.0, 0.0, iters));
            extra = est.calls;
        } else if (mode == "derived") {
            DerivedFromStock est;
            result = open3d::RegistrationICP(*model, *scene, radius, init, est,
                                             open3d::ICPConvergenceCriteria(0.0, 0.0, iters));
            extra = est.calls;
        } else if (mode == "sweep") {       // src/annotation.cpp:29-64
            RegistrationResult best;
            result.transformation_ = cicp::RegisterModelToScene(*model, *scene, level, radius, false, &best);
            result.fitness_ = best.fitness_; result.inlier_rmse_ = best.inlier_rmse_;
            result.correspondence_set_ = best.correspondence_set_;
        } else if (mode == "refine") {      // src/evaluation.cpp:258-271: down-sample the scene, then ICP
            result = cicp::ICPRefinement(*scene, *model, init, /*voxel*/ 0.05, radius, false);
            extra = (double)open3d::VoxelDownSample(*scene, 0.05)->points_.size();
        } else if (mode == "normals_plane") {
            // clouds without normals: estimate them (hybrid search: radius = 2 x the ICP radius, 30 neighbours),
            // orient them, then the point-to-plane estimator -- what a caller does before
            // src/evaluation.cpp:261-265 when its PLY carries no normals
            if (!open3d::EstimateNormals(*scene, open3d::KDTreeSearchParamHybrid(2.0 * radius, 30))) return 4;
            if (!open3d::EstimateNormals(*model, open3d::KDTreeSearchParamKNN(level))) return 4;
            open3d::OrientNormalsToAlignWithDirection(*scene);
            open3d::OrientNormalsTowardsCameraLocation(*model, Eigen::Vector3d(0.0, 0.0, 10.0));
            if (!scene->HasNormals() || !model->HasNormals()) return 5;
            extra = scene->normals_[scene->normals_.size() / 2](2);
            result = open3d::RegistrationICP(*model, *scene, radius, init,
                                             open3d::TransformationEstimationPointToPlane(),
                                             open3d::ICPConvergenceCriteria(0.0, 0.0, iters));
        } else if (mode == "evaluate") {
            result = open3d::EvaluateRegistration(*model, *scene, radius, init);
        } else if (mode == "mesh") {        // src/evaluation.cpp:320 MeasureSurfaceError / :252 sampling
            // "model" rows are the vertices, "scene" rows the faces (indices stored as doubles);
            // the target mesh is the source moved by init; level = number of samples
            Eigen::Matrix<double, Eigen::Dynamic, 3> V(ns, 3), Vt(ns, 3);
            Eigen::Matrix<int, Eigen::Dynamic, 3> F(nt, 3);
            for (int64_t i = 0; i < ns; i++) {
                const Eigen::Vector3d q = init.block<3, 3>(0, 0) * model->points_[i] + init.block<3, 1>(0, 3);
                for (int c = 0; c < 3; c++) { V(i, c) = model->points_[i](c); Vt(i, c) = q(c); }
            }
            for (int64_t i = 0; i < nt; i++)
                for (int c = 0; c < 3; c++) F(i, c) = (int)scene->points_[i](c);
            struct Opt { struct V { int v; int asInt() const { return v; } }; int n;
                         V operator[](const char *) const { return V{n}; } } opt{level};
            const auto m = feh::gpu::MeasureSurfaceError(V, F, Vt, F, opt);         // reference call shape
            const auto m2 = feh::gpu::MeasureSurfaceError(V, F, Vt, F, level, feh::gpu::SamplingMode::Surface, 0);
            if (m.mean_ != m2.mean_ || m.max_ != m2.max_) return 4;
            result.transformation_.setZero();
            result.transformation_(0, 0) = m.mean_; result.transformation_(0, 1) = m.std_;
            result.transformation_(0, 2) = m.median_; result.transformation_(0, 3) = m.min_;
            result.transformation_(1, 0) = m.max_;
            const auto pts = feh::gpu::SamplePointCloudFromMesh(V, F, level, feh::gpu::SamplingMode::Reference, 3);
            extra = (double)pts.size();
            std::vector<double> d;
            for (const auto &q : pts) d.push_back(q.norm());
            result.transformation_(1, 1) = feh::gpu::ComputeErrorMetric(d).median_;
        } else if (mode == "io") {          // host-only: src/evaluation.cpp:124 / core/utils.cpp:125 loaders
            // argv[2] still holds the (unused) clouds; the files come from the environment
            const char *ply = std::getenv("SHIM_PLY"), *objf = std::getenv("SHIM_OBJ");
            PointCloud pc;
            if (!ply || !objf || !open3d::ReadPointCloudFromPLY(ply, pc)) return 4;
            Eigen::Matrix<double, Eigen::Dynamic, 3> V;
            Eigen::Matrix<int, Eigen::Dynamic, 3> F;
            if (!feh::gpu::LoadMesh(objf, V, F)) return 5;
            if (open3d::ReadPointCloudFromPLY("/nonexistent.ply", pc)) return 6;    // false, message on stderr
            double pcd_n = 0.0, pcd_nn = 0.0, pcd_x = 0.0;
            if (const char *pcd = std::getenv("SHIM_PCD")) {                        // FilePCD.cpp:727-742
                PointCloud pd;
                if (!open3d::ReadPointCloudFromPCD(pcd, pd)) return 7;
                if (open3d::ReadPointCloudFromPCD("/nonexistent.pcd", pd)) return 8;
                open3d::ReadPointCloudFromPCD(pcd, pd);
                pcd_n = (double)pd.points_.size();
                pcd_nn = (double)pd.normals_.size();
                pcd_x = pd.points_.empty() ? 0.0 : pd.points_.back()(0);
            }
            open3d::ReadPointCloudFromPLY(ply, pc);
            result.transformation_.setZero();
            result.transformation_(0, 0) = (double)pc.points_.size();
            result.transformation_(0, 1) = (double)pc.normals_.size();
            result.transformation_(0, 2) = (double)pc.colors_.size();
            result.transformation_(0, 3) = pc.points_.empty() ? 0.0 : pc.points_.back()(2);
            result.transformation_(1, 0) = (double)V.rows();
            result.transformation_(1, 1) = (double)F.rows();
            result.transformation_(1, 2) = V.rows() ? V(V.rows() - 1, 1) : 0.0;
            result.transformation_(1, 3) = F.rows() ? (double)F(F.rows() - 1, 2) : 0.0;
            result.transformation_(2, 0) = pc.colors_.empty() ? 0.0 : pc.colors_[0](1);
            result.transformation_(2, 1) = pcd_n; result.transformation_(2, 2) = pcd_nn;
            result.transformation_(2, 3) = pcd_x;
        } else if (mode == "estimator") {   // host-only: explicit correspondences, no GPU needed
            CorrespondenceSet cs;
            for (int64_t i = 0; i < ns; i++) cs.push_back(Eigen::Vector2i((int)i, (int)((i * 7919) % nt)));
            cicp::TransformationEstimationPointToPoint4DoF est(level != 0);
            result.transformation_ = est.ComputeTransformation(*model, *scene, cs);
            result.inlier_rmse_ = est.ComputeRMSE(*model, *scene, cs);
            result.correspondence_set_ = cs;
        } else {
            return 2;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "shim_driver: %s\n", e.what());
        return 3;
    }
    FILE *o = std::fopen(argv[3], "wb");
    double out[20];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = result.transformation_(i, j);
    out[16] = result.fitness_; out[17] = result.inlier_rmse_;
    out[18] = (double)result.correspondence_set_.size(); out[19] = extra;
    std::fwrite(out, sizeof(double), 20, o);
    std::fclose(o);
    return 0;
}
