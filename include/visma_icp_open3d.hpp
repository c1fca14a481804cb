// visma_icp_open3d.hpp -- header-only C++/Eigen adapter between Open3D-shaped
// callers and the C ABI (visma_icp.h).
//
// It is written against the NAMES the reference's callers use
// (open3d::PointCloud::points_/normals_, CorrespondenceSet, RegistrationResult,
// ICPConvergenceCriteria, TransformationEstimation and its enum), so it
// compiles against the real Open3D 0.3.0 headers or against the stand-alone
// set in include/Core/.  Include one of them first.
//
//   open3d::cicp::RegistrationICP      same signature as open3d::RegistrationICP
//                                      (O3D/Core/Registration/Registration.h:102-107)
//   open3d::cicp::EvaluateRegistration (Registration.h:96-99)
//   open3d::cicp::RegisterModelToScene feh::RegisterModelToScene
//                                      (include/tool.h:40-42, src/annotation.cpp:29-64)
//   open3d::cicp::ICPRefinement        the ICP call of feh::ICPRefinement
//                                      (src/evaluation.cpp:258-271)
//
// No Eigen type crosses a binary boundary: the C ABI takes raw row-major
// doubles, so a translation unit built with -DEIGEN_DEFAULT_TO_ROW_MAJOR (as
// VISMA's CMakeLists.txt:11-12 does) and one built without it can both use
// this header.  Infrastructure failures (no GPU, HIP error) throw
// std::runtime_error; argument errors follow the reference (message on stderr,
// RegistrationResult(init) returned, Registration.cpp:148-157).
#pragma once

#include <Eigen/Core>
#include <cmath>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <vector>

#include "visma_icp.h"
#include "visma_io.h"

namespace open3d {
namespace cicp {

// The plugin of include/constrained_ICP.h:14-30, made concrete: the reference
// class never overrides the pure virtual GetTransformationEstimationType()
// (O3D/Core/Registration/TransformationEstimation.h:58-59) and so cannot be
// instantiated against the vendored Open3D; this one can.
class TransformationEstimationPointToPoint4DoF : public TransformationEstimation {
public:
    TransformationEstimationPointToPoint4DoF(bool with_scaling = false)
        : with_scaling_(with_scaling) {}
    ~TransformationEstimationPointToPoint4DoF() override {}
    TransformationEstimationType GetTransformationEstimationType() const override
    {
        return TransformationEstimationType::PointToPoint;
    }
    inline double ComputeRMSE(const PointCloud &source, const PointCloud &target,
                              const CorrespondenceSet &corres) const override;
    inline Eigen::Matrix4d ComputeTransformation(const PointCloud &source,
                                                 const PointCloud &target,
                                                 const CorrespondenceSet &corres) const override;
    bool with_scaling_ = false;
};

namespace detail {

inline void to_rowmajor(const Eigen::Matrix4d &M, double T[16])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) T[i * 4 + j] = M(i, j);
}

inline Eigen::Matrix4d from_rowmajor(const double T[16])
{
    Eigen::Matrix4d M;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) M(i, j) = T[i * 4 + j];
    return M;
}

// One context per host thread (the ABI is thread-compatible, not thread-safe).
class ThreadContext {
public:
    ~ThreadContext() { if (ctx_) visma_icp_destroy(ctx_); }
    visma_icp_ctx *get()
    {
        if (!ctx_) {
            int device = 0;
            if (const char *e = std::getenv("VISMA_ICP_DEVICE")) device = std::atoi(e);
            const int rc = visma_icp_create(&ctx_, device);
            if (rc != VISMA_ICP_OK)
                throw std::runtime_error(std::string("visma_icp_create failed: ") +
                                         visma_icp_last_error(nullptr));
            // drop-in callers want the reference's correspondences: f64 search for the cloud
            // sizes where one flipped near-tie would show (VISMA_ICP_SEARCH_PRECISION=0/1/2 overrides)
            int prec = 1;
            if (const char *e = std::getenv("VISMA_ICP_SEARCH_PRECISION")) prec = std::atoi(e);
            visma_icp_set_search_precision(ctx_, prec);
        }
        return ctx_;
    }
    static ThreadContext &instance()
    {
        static thread_local ThreadContext tc;
        return tc;
    }

private:
    visma_icp_ctx *ctx_ = nullptr;
};

inline void check(visma_icp_ctx *ctx, int rc, const char *what)
{
    if (rc != VISMA_ICP_OK)
        throw std::runtime_error(std::string(what) + ": " + visma_icp_last_error(ctx));
}

// std::vector<Eigen::Vector3d> is contiguous AoS f64 with stride 3
inline const double *xyz(const std::vector<Eigen::Vector3d> &v)
{
    return v.empty() ? nullptr : v[0].data();
}

// `radius`: the max_correspondence_distance of the registration that follows (> 0: the library builds its search
// structure on the GPU while it is still staging the source, visma_icp_set_radius_hint)
template <typename Cloud>
inline visma_icp_ctx *upload(const Cloud &source, const Cloud &target, bool normals, double radius = 0.0)
{
    visma_icp_ctx *ctx = ThreadContext::instance().get();
    if (radius > 0.0) check(ctx, visma_icp_set_radius_hint(ctx, radius), "visma_icp_set_radius_hint");
    check(ctx, visma_icp_set_clouds_f64(ctx, xyz(source.points_), (int64_t)source.points_.size(), 3,
                                        xyz(target.points_), (int64_t)target.points_.size(), 3),
          "visma_icp_set_clouds_f64");
    if (normals)
        check(ctx, visma_icp_set_target_normals_f64(ctx, xyz(target.normals_),
                                                    (int64_t)target.normals_.size(), 3),
              "visma_icp_set_target_normals_f64");
    return ctx;
}

template <typename Result>
inline void fill_result(visma_icp_ctx *ctx, const visma_icp_result &r, size_t ns, Result &out)
{
    out.transformation_ = from_rowmajor(r.transformation);
    out.fitness_ = r.fitness;
    out.inlier_rmse_ = r.inlier_rmse;
    std::vector<int32_t> si(ns ? ns : 1), ti(ns ? ns : 1);
    int64_t k = 0;
    check(ctx, visma_icp_get_correspondences(ctx, si.data(), ti.data(), nullptr, &k),
          "visma_icp_get_correspondences");
    out.correspondence_set_.resize((size_t)k);
    for (int64_t i = 0; i < k; i++) out.correspondence_set_[i] = Eigen::Vector2i(si[i], ti[i]);
}

// 38 statistics of explicit correspondences, accumulated on the host in f64
// (same layout the reduction kernel produces; see visma_icp.h).
template <typename Cloud, typename Corr>
inline void host_stats(const Cloud &source, const Cloud &target, const Corr &corres, bool plane,
                       double st[VISMA_ICP_NSTATS])
{
    double JTJ[6][6] = {{0}}, JTr[6] = {0}, M[3][3] = {{0}}, r2 = 0.0;
    for (const auto &c : corres) {
        const Eigen::Vector3d &p = source.points_[c[0]];
        const Eigen::Vector3d &q = target.points_[c[1]];
        const int rows = plane ? 1 : 3;
        for (int k = 0; k < rows; k++) {
            Eigen::Vector3d n = plane ? Eigen::Vector3d(target.normals_[c[1]]) : Eigen::Vector3d::Zero();
            if (!plane) n[k] = 1.0;
            const Eigen::Vector3d a(p[1] * n[2] - p[2] * n[1], p[2] * n[0] - p[0] * n[2],
                                    p[0] * n[1] - p[1] * n[0]);  // p x n
            const double J[6] = {a[0], a[1], a[2], n[0], n[1], n[2]};
            const double r = (p - q).dot(n);
            for (int i = 0; i < 6; i++) {
                for (int j = 0; j < 6; j++) JTJ[i][j] += J[i] * J[j];
                JTr[i] += J[i] * r;
            }
            r2 += r * r;
        }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) M[i][j] += q[i] * p[j];
    }
    int o = 0;
    st[o++] = (double)corres.size();
    st[o++] = r2;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) st[o++] = JTJ[i][j];
    for (int i = 0; i < 6; i++) st[o++] = JTr[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) st[o++] = M[i][j];
}

template <typename Cloud, typename Corr>
inline Eigen::Matrix4d host_update(const Cloud &s, const Cloud &t, const Corr &corres, bool plane,
                                   bool with_scaling)
{
    if (corres.empty()) return Eigen::Matrix4d::Identity();
    double st[VISMA_ICP_NSTATS], T[16];
    host_stats(s, t, corres, plane, st);
    visma_icp_solve_from_stats(st, plane ? VISMA_ICP_SOLVER_GN_EULER : VISMA_ICP_SOLVER_KABSCH,
                               with_scaling ? 1 : 0, T);
    return from_rowmajor(T);
}

template <typename Cloud, typename Corr>
inline double host_rmse_point_to_point(const Cloud &s, const Cloud &t, const Corr &corres)
{
    if (corres.empty()) return 0.0;
    double e = 0.0;
    for (const auto &c : corres) e += (s.points_[c[0]] - t.points_[c[1]]).squaredNorm();
    return std::sqrt(e / (double)corres.size());
}

}  // namespace detail

// ---------------------------------------------------------------------------
// One NN pass at a given transformation: fitness / inlier_rmse / correspondences.
inline RegistrationResult EvaluateRegistration(
    const PointCloud &source, const PointCloud &target, double max_correspondence_distance,
    const Eigen::Matrix4d &transformation = Eigen::Matrix4d::Identity())
{
    RegistrationResult result(transformation);
    if (max_correspondence_distance <= 0.0) return result;
    visma_icp_ctx *ctx = detail::upload(source, target, false, max_correspondence_distance);
    double T[16];
    detail::to_rowmajor(transformation, T);
    visma_icp_result r;
    // zero iterations of the loop == exactly one NN pass at T
    detail::check(ctx, visma_icp_run(ctx, T, max_correspondence_distance, 0, 0.0, 0.0,
                                     VISMA_ICP_SOLVER_KABSCH, 0, &r), "visma_icp_run");
    detail::fill_result(ctx, r, source.points_.size(), result);
    return result;
}

// ---------------------------------------------------------------------------
// The ICP driver.  Point-to-point estimators (Open3D's stock one and the 4DoF
// class above -- identical arithmetic) and the point-to-plane estimator run
// fully on the GPU.  Any OTHER subclass of TransformationEstimation still
// works: the NN passes run on the GPU and the plugin's own virtual
// ComputeTransformation is called on the host each iteration, exactly as
// Registration.cpp:169-184 does.
inline RegistrationResult RegistrationICP(
    const PointCloud &source, const PointCloud &target, double max_correspondence_distance,
    const Eigen::Matrix4d &init = Eigen::Matrix4d::Identity(),
    const TransformationEstimation &estimation = TransformationEstimationPointToPoint4DoF(false),
    const ICPConvergenceCriteria &criteria = ICPConvergenceCriteria())
{
    if (max_correspondence_distance <= 0.0) {
        std::fprintf(stderr, "Error: Invalid max_correspondence_distance.\n");
        return RegistrationResult(init);
    }
    const TransformationEstimationType type = estimation.GetTransformationEstimationType();
    const bool plane = type == TransformationEstimationType::PointToPlane;
    if (plane && (!source.HasNormals() || !target.HasNormals())) {
        std::fprintf(stderr, "Error: TransformationEstimationPointToPlane requires "
                             "pre-computed normal vectors.\n");
        return RegistrationResult(init);
    }
    RegistrationResult result(init);
    visma_icp_ctx *ctx = detail::upload(source, target, plane, max_correspondence_distance);
    double T[16];
    detail::to_rowmajor(init, T);
    visma_icp_result r;

    // The built-in solves replace ComputeTransformation only for EXACTLY the three stock
    // estimators.  A user class derived from one of them may override the virtual methods: it
    // takes the generic plugin loop below, which calls them like the reference does
    // (Registration.cpp:172-173).
    const std::type_info &dyn = typeid(estimation);
    const auto *four = dyn == typeid(TransformationEstimationPointToPoint4DoF)
                           ? static_cast<const TransformationEstimationPointToPoint4DoF *>(&estimation) : nullptr;
    const auto *p2p = dyn == typeid(TransformationEstimationPointToPoint)
                          ? static_cast<const TransformationEstimationPointToPoint *>(&estimation) : nullptr;
    const auto *p2l = dyn == typeid(TransformationEstimationPointToPlane)
                          ? static_cast<const TransformationEstimationPointToPlane *>(&estimation) : nullptr;
    if (four || p2p) {
        const bool scaling = four ? four->with_scaling_ : p2p->with_scaling_;
        detail::check(ctx, visma_icp_run(ctx, T, max_correspondence_distance, criteria.max_iteration_,
                                         criteria.relative_fitness_, criteria.relative_rmse_,
                                         VISMA_ICP_SOLVER_KABSCH, scaling ? 1 : 0, &r),
                      "visma_icp_run");
        detail::fill_result(ctx, r, source.points_.size(), result);
        return result;
    }
    if (p2l) {
        detail::check(ctx, visma_icp_run_point_to_plane(ctx, T, max_correspondence_distance,
                                                        criteria.max_iteration_,
                                                        criteria.relative_fitness_,
                                                        criteria.relative_rmse_, &r),
                      "visma_icp_run_point_to_plane");
        detail::fill_result(ctx, r, source.points_.size(), result);
        return result;
    }

    // Generic plugin: GPU NN passes + the plugin's host-side solve.
    Eigen::Matrix4d transformation = init;
    PointCloud pcd = source;
    if (!init.isIdentity()) pcd.Transform(init);
    auto nn = [&](RegistrationResult &out) {
        detail::to_rowmajor(transformation, T);
        detail::check(ctx, visma_icp_run(ctx, T, max_correspondence_distance, 0, 0.0, 0.0,
                                         VISMA_ICP_SOLVER_KABSCH, 0, &r), "visma_icp_run");
        detail::fill_result(ctx, r, source.points_.size(), out);
        out.transformation_ = transformation;
    };
    nn(result);
    for (int i = 0; i < criteria.max_iteration_; i++) {
        const Eigen::Matrix4d update =
            estimation.ComputeTransformation(pcd, target, result.correspondence_set_);
        transformation = update * transformation;
        pcd.Transform(update);
        const double bf = result.fitness_, br = result.inlier_rmse_;
        nn(result);
        if (std::abs(bf - result.fitness_) < criteria.relative_fitness_ &&
            std::abs(br - result.inlier_rmse_) < criteria.relative_rmse_)
            break;
    }
    return result;
}

// feh::RegisterModelToScene (src/annotation.cpp:29-64) with its JSON options
// as plain arguments: rotation_level yaw initialisations about +Y, a full ICP
// from each, the first result with strictly the most correspondences wins.
inline Eigen::Matrix4d RegisterModelToScene(const PointCloud &model, const PointCloud &scene,
                                            int rotation_level, double distance_threshold,
                                            bool point_to_plane = false,
                                            RegistrationResult *best_out = nullptr)
{
    RegistrationResult best;
    const bool plane_ready = point_to_plane && model.HasNormals() && scene.HasNormals();
    if ((!point_to_plane || plane_ready) && rotation_level > 0 && distance_threshold > 0.0) {
        // all levels in one library call (the sweep is advanced on the GPU), either estimator
        visma_icp_ctx *ctx = detail::upload(model, scene, point_to_plane, distance_threshold);
        visma_icp_result b;
        int level = -1;
        const ICPConvergenceCriteria c;
        if (point_to_plane)
            detail::check(ctx, visma_icp_run_yaw_sweep_point_to_plane(ctx, rotation_level, distance_threshold,
                                                                      c.max_iteration_, c.relative_fitness_,
                                                                      c.relative_rmse_, &b, &level, nullptr),
                          "visma_icp_run_yaw_sweep_point_to_plane");
        else
            detail::check(ctx, visma_icp_run_yaw_sweep(ctx, rotation_level, distance_threshold,
                                                       c.max_iteration_, c.relative_fitness_,
                                                       c.relative_rmse_, VISMA_ICP_SOLVER_KABSCH, &b,
                                                       &level, nullptr),
                          "visma_icp_run_yaw_sweep");
        best.transformation_ = detail::from_rowmajor(b.transformation);
        best.fitness_ = b.fitness;
        best.inlier_rmse_ = b.inlier_rmse;
        if (best_out) {
            // re-evaluate at the winning transform to materialise its correspondences
            *best_out = level >= 0 ? cicp::EvaluateRegistration(model, scene, distance_threshold,
                                                          best.transformation_)
                                   : best;
        }
        return best.transformation_;
    }
    const double interval = 2.0 * M_PI / rotation_level;
    for (int i = 0; i < rotation_level; ++i) {
        const double a = interval * i, c = std::cos(a), s = std::sin(a);
        Eigen::Matrix4d init = Eigen::Matrix4d::Identity();
        init(0, 0) = c; init(0, 2) = s; init(2, 0) = -s; init(2, 2) = c;
        RegistrationResult r;
        if (point_to_plane)
            r = cicp::RegistrationICP(model, scene, distance_threshold, init,
                                TransformationEstimationPointToPlane(), ICPConvergenceCriteria());
        else
            r = cicp::RegistrationICP(model, scene, distance_threshold, init,
                                TransformationEstimationPointToPoint4DoF(), ICPConvergenceCriteria());
        if (r.correspondence_set_.size() > best.correspondence_set_.size()) best = r;
    }
    if (best_out) *best_out = best;
    return best.transformation_;
}

// The per-object loop of feh::AnnotationTool (src/annotation.cpp:103-168) for MANY (model, scan) pairs at
// once: what the loop body hands to RegisterModelToScene, collected first, registered together by the
// library's native work queue (visma_icp_run_corpus: one host thread per GPU pulls chunks of pairs, each
// chunk's rotation_level yaw starts run as one batch), the loop's T3 per pair returned in order.
// `devices`: one queue worker (context, own stream) per entry -- the SAME GPU may be listed several times, and should:
// while one worker packs and uploads its next chunk, the others' searches have the device (three on one GPU: ~1.5x one).
// Empty = three workers on device 0.  Point-to-point estimator (ICP.point_to_plane = false).
inline std::vector<Eigen::Matrix4d> RegisterModelsToScenes(
    const std::vector<std::pair<std::shared_ptr<PointCloud>, std::shared_ptr<PointCloud>>> &model_scan_pairs,
    int rotation_level, double distance_threshold, const std::vector<int> &devices = std::vector<int>(),
    std::vector<RegistrationResult> *best_out = nullptr)
{
    const size_t n = model_scan_pairs.size();
    std::vector<Eigen::Matrix4d> T(n, Eigen::Matrix4d::Identity());
    if (best_out) best_out->assign(n, RegistrationResult());
    if (n == 0 || rotation_level <= 0 || !(distance_threshold > 0.0)) return T;
    std::vector<visma_icp_corpus_item> items(n);
    for (size_t i = 0; i < n; i++) {
        const PointCloud &m = *model_scan_pairs[i].first, &sc = *model_scan_pairs[i].second;
        // std::vector<Eigen::Vector3d> is AoS f64 with stride 3 (static_assert in detail::upload)
        items[i].model_xyz = m.points_.empty() ? nullptr : m.points_[0].data();
        items[i].n_model = (int64_t)m.points_.size();
        items[i].scene_xyz = sc.points_.empty() ? nullptr : sc.points_[0].data();
        items[i].n_scene = (int64_t)sc.points_.size();
    }
    std::vector<visma_icp_ctx *> ctxs;
    const std::vector<int> devs = devices.empty() ? std::vector<int>(3, 0) : devices;   // (three workers on GPU 0)
    auto destroy_all = [&]() { for (visma_icp_ctx *c : ctxs) visma_icp_destroy(c); };
    for (int d : devs) {
        visma_icp_ctx *c = nullptr;
        if (visma_icp_create(&c, d) != VISMA_ICP_OK) { destroy_all(); throw std::runtime_error("visma_icp_create failed (no gfx950 GPU?)"); }
        ctxs.push_back(c);
    }
    const ICPConvergenceCriteria crit;
    visma_icp_corpus_params p;
    p.level = rotation_level; p.max_dist = distance_threshold; p.max_iter = crit.max_iteration_;
    p.rel_fitness = crit.relative_fitness_; p.rel_rmse = crit.relative_rmse_; p.solver = VISMA_ICP_SOLVER_KABSCH; p.chunk = 0;
    std::vector<visma_icp_corpus_result> res(n);
    char err[512];
    const int rc = visma_icp_run_corpus(ctxs.data(), (int)ctxs.size(), items.data(), (int64_t)n, &p, nullptr, res.data(), err, sizeof(err));
    destroy_all();
    if (rc != VISMA_ICP_OK) throw std::runtime_error(std::string("visma_icp_run_corpus: ") + err);
    for (size_t i = 0; i < n; i++) {
        T[i] = detail::from_rowmajor(res[i].best.transformation);
        if (best_out) {
            RegistrationResult &b = (*best_out)[i];
            b.transformation_ = T[i];
            b.fitness_ = res[i].best.fitness;
            b.inlier_rmse_ = res[i].best.inlier_rmse;
        }
    }
    return T;
}

// open3d::VoxelDownSample (O3D/Core/Geometry/DownSample.cpp:179-220) on the GPU:
// same points / normals / colours, bit for bit; voxels come out in ascending
// (ix,iy,iz) order instead of the reference's hash-map iteration order.
inline std::shared_ptr<PointCloud> VoxelDownSample(const PointCloud &input, double voxel_size)
{
    auto output = std::make_shared<PointCloud>();
    const int64_t n = (int64_t)input.points_.size();
    if (voxel_size <= 0.0 || n == 0) return output;
    visma_icp_ctx *ctx = detail::ThreadContext::instance().get();
    const bool hn = input.HasNormals(), hc = input.HasColors();
    output->points_.resize((size_t)n);
    if (hn) output->normals_.resize((size_t)n);
    if (hc) output->colors_.resize((size_t)n);
    int64_t m = 0;
    detail::check(ctx, visma_icp_voxel_down_sample(
                           ctx, detail::xyz(input.points_), n, hn ? detail::xyz(input.normals_) : nullptr,
                           hc ? detail::xyz(input.colors_) : nullptr, voxel_size,
                           output->points_[0].data(), hn ? output->normals_[0].data() : nullptr,
                           hc ? output->colors_[0].data() : nullptr, &m),
                  "visma_icp_voxel_down_sample");
    output->points_.resize((size_t)m);
    if (hn) output->normals_.resize((size_t)m);
    if (hc) output->colors_.resize((size_t)m);
    return output;
}

// open3d::EstimateNormals (O3D/Core/Geometry/EstimateNormals.cpp:114-153) on the GPU: the three
// KDTreeFlann searches, FastEigen3x3, (0,0,1) for fewer than 3 neighbours, the sign of existing normals
// kept.  What a caller runs before the point-to-plane estimator on clouds without normals
// (Registration.cpp:152-157 returns the initial transform otherwise).
inline bool EstimateNormals(PointCloud &cloud, const KDTreeSearchParam &search_param = KDTreeSearchParamKNN())
{
    const int64_t n = (int64_t)cloud.points_.size();
    const bool has_normal = cloud.HasNormals();
    int type = 0, knn = 0;
    double radius = 0.0;
    switch (search_param.GetSearchType()) {
    case KDTreeSearchParam::SearchType::Knn:
        knn = static_cast<const KDTreeSearchParamKNN &>(search_param).knn_;
        break;
    case KDTreeSearchParam::SearchType::Radius:
        type = 1;
        radius = static_cast<const KDTreeSearchParamRadius &>(search_param).radius_;
        break;
    case KDTreeSearchParam::SearchType::Hybrid:
        type = 2;
        radius = static_cast<const KDTreeSearchParamHybrid &>(search_param).radius_;
        knn = static_cast<const KDTreeSearchParamHybrid &>(search_param).max_nn_;
        break;
    }
    std::vector<Eigen::Vector3d> out((size_t)n);
    if (n > 0) {
        visma_icp_ctx *ctx = detail::ThreadContext::instance().get();
        detail::check(ctx, visma_icp_estimate_normals(ctx, detail::xyz(cloud.points_), n,
                                                      has_normal ? detail::xyz(cloud.normals_) : nullptr, type, knn,
                                                      radius, out[0].data()),
                      "visma_icp_estimate_normals");
    }
    cloud.normals_.swap(out);
    return true;
}

// O3D/Core/Geometry/EstimateNormals.cpp:155-174
inline bool OrientNormalsToAlignWithDirection(PointCloud &cloud,
                                              const Eigen::Vector3d &orientation_reference = Eigen::Vector3d(0.0, 0.0, 1.0))
{
    for (auto &normal : cloud.normals_) {
        if (normal.norm() == 0.0) normal = orientation_reference;
        else if (normal.dot(orientation_reference) < 0.0) normal *= -1.0;
    }
    return true;
}

// O3D/Core/Geometry/EstimateNormals.cpp:176-204
inline bool OrientNormalsTowardsCameraLocation(PointCloud &cloud,
                                               const Eigen::Vector3d &camera_location = Eigen::Vector3d::Zero())
{
    const size_t n = cloud.HasNormals() ? cloud.points_.size() : 0;
    for (size_t i = 0; i < n; i++) {
        const Eigen::Vector3d towards = camera_location - cloud.points_[i];
        Eigen::Vector3d &normal = cloud.normals_[i];
        if (normal.norm() == 0.0) {
            normal = towards;
            if (normal.norm() == 0.0) normal = Eigen::Vector3d(0.0, 0.0, 1.0);
            else normal.normalize();
        } else if (normal.dot(towards) < 0.0) {
            normal *= -1.0;
        }
    }
    return true;
}

// open3d::ReadPointCloudFromPCD (O3D/IO/FileFormat/FilePCD.cpp:727-742): ascii, binary and
// binary_compressed files; the reader's values bit for bit (include/visma_io.h).
inline bool ReadPointCloudFromPCD(const std::string &filename, PointCloud &pointcloud)
{
    visma_io_cloud c;
    if (visma_io_read_pcd(filename.c_str(), &c) != VISMA_IO_OK) {
        std::fprintf(stderr, "Read PCD failed: %s\n", visma_io_last_error());
        return false;
    }
    pointcloud.points_.resize((size_t)c.n);
    pointcloud.normals_.resize((size_t)c.n_normals);
    pointcloud.colors_.resize((size_t)c.n_colors);
    for (int64_t i = 0; i < c.n; i++) pointcloud.points_[(size_t)i] = Eigen::Vector3d(c.xyz[3 * i], c.xyz[3 * i + 1], c.xyz[3 * i + 2]);
    for (int64_t i = 0; i < c.n_normals; i++)
        pointcloud.normals_[(size_t)i] = Eigen::Vector3d(c.normals[3 * i], c.normals[3 * i + 1], c.normals[3 * i + 2]);
    for (int64_t i = 0; i < c.n_colors; i++)
        pointcloud.colors_[(size_t)i] = Eigen::Vector3d(c.colors[3 * i], c.colors[3 * i + 1], c.colors[3 * i + 2]);
    visma_io_free_cloud(&c);
    return true;
}

// open3d::ReadPointCloudFromPLY (O3D/IO/FileFormat/FilePLY.cpp:206-264): the scene and scan
// clouds of both callers (src/evaluation.cpp:124,211; src/annotation.cpp:76-157).  Same
// points / normals / colours as the rply-based reader; false (and a message on stderr) on
// failure, like the reference.
inline bool ReadPointCloudFromPLY(const std::string &filename, PointCloud &pointcloud)
{
    visma_io_cloud c;
    if (visma_io_read_ply(filename.c_str(), &c) != VISMA_IO_OK) {
        std::fprintf(stderr, "Read PLY failed: %s\n", visma_io_last_error());
        return false;
    }
    pointcloud.points_.resize((size_t)c.n);
    pointcloud.normals_.resize((size_t)c.n_normals);
    pointcloud.colors_.resize((size_t)c.n_colors);
    for (int64_t i = 0; i < c.n; i++) pointcloud.points_[(size_t)i] = Eigen::Vector3d(c.xyz[3 * i], c.xyz[3 * i + 1], c.xyz[3 * i + 2]);
    for (int64_t i = 0; i < c.n_normals; i++)
        pointcloud.normals_[(size_t)i] = Eigen::Vector3d(c.normals[3 * i], c.normals[3 * i + 1], c.normals[3 * i + 2]);
    for (int64_t i = 0; i < c.n_colors; i++)
        pointcloud.colors_[(size_t)i] = Eigen::Vector3d(c.colors[3 * i], c.colors[3 * i + 1], c.colors[3 * i + 2]);
    visma_io_free_cloud(&c);
    return true;
}

// feh::ICPRefinement (src/evaluation.cpp:258-271) from the down-sampling on:
// scene = VoxelDownSample(scene, voxel_size); RegistrationICP(scene_est, scene, ...).
inline RegistrationResult ICPRefinement(const PointCloud &scene_raw, const PointCloud &scene_est,
                                        const Eigen::Matrix4d &T_scene_src, double voxel_size,
                                        double max_distance, bool use_point_to_plane)
{
    if (!use_point_to_plane && voxel_size > 0.0 && max_distance > 0.0 && !scene_raw.points_.empty() &&
        !scene_est.points_.empty()) {
        // the scene goes up once, is down-sampled on the device and becomes the target where it lies
        // (visma_icp_set_clouds_f64_voxel_target): the down-sampled cloud never crosses PCIe.  The result is the
        // one of the two separate steps below; its correspondence_set_ indexes the voxels in ascending order.
        visma_icp_ctx *ctx = detail::ThreadContext::instance().get();
        int64_t nt = 0;
        detail::check(ctx, visma_icp_set_clouds_f64_voxel_target(ctx, detail::xyz(scene_est.points_), (int64_t)scene_est.points_.size(), 3,
                                                                 detail::xyz(scene_raw.points_), (int64_t)scene_raw.points_.size(), 3,
                                                                 voxel_size, &nt),
                      "visma_icp_set_clouds_f64_voxel_target");
        const ICPConvergenceCriteria c;
        double T[16];
        detail::to_rowmajor(T_scene_src, T);
        visma_icp_result r;
        detail::check(ctx, visma_icp_run(ctx, T, max_distance, c.max_iteration_, c.relative_fitness_, c.relative_rmse_,
                                         VISMA_ICP_SOLVER_KABSCH, 0, &r), "visma_icp_run");
        RegistrationResult result(T_scene_src);
        detail::fill_result(ctx, r, scene_est.points_.size(), result);
        return result;
    }
    const std::shared_ptr<PointCloud> scene = cicp::VoxelDownSample(scene_raw, voxel_size);
    if (use_point_to_plane)
        return cicp::RegistrationICP(scene_est, *scene, max_distance, T_scene_src,
                                     TransformationEstimationPointToPlane());
    return cicp::RegistrationICP(scene_est, *scene, max_distance, T_scene_src);
}

// The registration call alone (scene already down-sampled by the caller).
inline RegistrationResult ICPRefinement(const PointCloud &scene, const PointCloud &scene_est,
                                        const Eigen::Matrix4d &T_scene_src, double max_distance,
                                        bool use_point_to_plane)
{
    if (use_point_to_plane)
        return cicp::RegistrationICP(scene_est, scene, max_distance, T_scene_src,
                               TransformationEstimationPointToPlane());
    return cicp::RegistrationICP(scene_est, scene, max_distance, T_scene_src);
}

// ---- 4DoF estimator methods (explicit-correspondence entry points) ----------
inline double TransformationEstimationPointToPoint4DoF::ComputeRMSE(
    const PointCloud &source, const PointCloud &target, const CorrespondenceSet &corres) const
{
    return detail::host_rmse_point_to_point(source, target, corres);
}

inline Eigen::Matrix4d TransformationEstimationPointToPoint4DoF::ComputeTransformation(
    const PointCloud &source, const PointCloud &target, const CorrespondenceSet &corres) const
{
    return detail::host_update(source, target, corres, false, with_scaling_);
}

}  // namespace cicp

#ifdef VISMA_ICP_STANDALONE_OPEN3D_TYPES
// Stand-alone header set: give the stock names their bodies too.
inline double TransformationEstimationPointToPoint::ComputeRMSE(
    const PointCloud &s, const PointCloud &t, const CorrespondenceSet &c) const
{
    return cicp::detail::host_rmse_point_to_point(s, t, c);
}
inline Eigen::Matrix4d TransformationEstimationPointToPoint::ComputeTransformation(
    const PointCloud &s, const PointCloud &t, const CorrespondenceSet &c) const
{
    return cicp::detail::host_update(s, t, c, false, with_scaling_);
}
// NB the reference's point-to-plane ComputeRMSE assigns `err = r * r` instead of
// accumulating (TransformationEstimation.cpp:70), i.e. it returns
// sqrt(r_last^2 / K).  RegistrationICP never calls it; we keep its value.
inline double TransformationEstimationPointToPlane::ComputeRMSE(
    const PointCloud &s, const PointCloud &t, const CorrespondenceSet &c) const
{
    if (c.empty() || !t.HasNormals()) return 0.0;
    const auto &l = c.back();
    const double r = (s.points_[l[0]] - t.points_[l[1]]).dot(t.normals_[l[1]]);
    return std::sqrt(r * r / (double)c.size());
}
inline Eigen::Matrix4d TransformationEstimationPointToPlane::ComputeTransformation(
    const PointCloud &s, const PointCloud &t, const CorrespondenceSet &c) const
{
    if (c.empty() || !t.HasNormals()) return Eigen::Matrix4d::Identity();
    return cicp::detail::host_update(s, t, c, true, false);
}
inline std::shared_ptr<PointCloud> VoxelDownSample(const PointCloud &input, double voxel_size)
{
    return cicp::VoxelDownSample(input, voxel_size);
}
inline RegistrationResult EvaluateRegistration(const PointCloud &source, const PointCloud &target,
                                               double max_correspondence_distance,
                                               const Eigen::Matrix4d &transformation)
{
    return cicp::EvaluateRegistration(source, target, max_correspondence_distance, transformation);
}
inline bool ReadPointCloudFromPLY(const std::string &filename, PointCloud &pointcloud)
{
    return cicp::ReadPointCloudFromPLY(filename, pointcloud);
}
inline bool ReadPointCloudFromPCD(const std::string &filename, PointCloud &pointcloud)
{
    return cicp::ReadPointCloudFromPCD(filename, pointcloud);
}
inline bool EstimateNormals(PointCloud &cloud, const KDTreeSearchParam &search_param)
{
    return cicp::EstimateNormals(cloud, search_param);
}
inline bool OrientNormalsToAlignWithDirection(PointCloud &cloud, const Eigen::Vector3d &orientation_reference)
{
    return cicp::OrientNormalsToAlignWithDirection(cloud, orientation_reference);
}
inline bool OrientNormalsTowardsCameraLocation(PointCloud &cloud, const Eigen::Vector3d &camera_location)
{
    return cicp::OrientNormalsTowardsCameraLocation(cloud, camera_location);
}
inline RegistrationResult RegistrationICP(const PointCloud &source, const PointCloud &target,
                                          double max_correspondence_distance,
                                          const Eigen::Matrix4d &init,
                                          const TransformationEstimation &estimation,
                                          const ICPConvergenceCriteria &criteria)
{
    return cicp::RegistrationICP(source, target, max_correspondence_distance, init, estimation,
                                 criteria);
}
#endif

}  // namespace open3d
