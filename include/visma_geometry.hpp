// visma_geometry.hpp -- GPU versions of the mesh steps either side of ICP, with
// the names and argument meaning of the reference's include/geometry.h:
//
//   feh::gpu::SamplePointCloudFromMesh   geometry.h:29-64
//   feh::gpu::ComputeErrorMetric         geometry.h:85-101 (host)
//   feh::gpu::MeasureSurfaceError        geometry.h:117-141
//   feh::gpu::FindPlaneNormal            geometry.h:18-26;  RotationBetweenVectors  core/utils.h:229-233
//   feh::gpu::AnnotateObjects            the per-object loop of AnnotationTool, src/annotation.cpp:103-168
//
// Header-only over the C ABI in visma_icp.h; works with either Eigen storage
// order (VISMA compiles with -DEIGEN_DEFAULT_TO_ROW_MAJOR, CMakeLists.txt:11-12).
// They live in feh::gpu so that the reference's own geometry.h can stay on the
// include path; `using namespace feh::gpu;` (or s/feh::/feh::gpu::/ at the three
// call sites, src/evaluation.cpp:252,320 and src/annotation.cpp:126) switches over.
//
// Differences to know about:
//  * the reference seeds std::knuth_b from the wall clock; here the stream is a
//    counter-based Philox4x32-10 keyed by `seed` (reproducible);
//  * SamplingMode::Reference reproduces the reference's mapping from uniforms to
//    points (one face late, parallelogram instead of triangle -- about half of
//    the points lie off the surface -- and no point when r < cdf[0]);
//    SamplingMode::Surface (default) samples the surface itself, which is what
//    the function's doc comment promises;
//  * errors throw std::runtime_error; there is no CPU fallback.
#pragma once
#include <Eigen/Core>

#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "visma_icp.h"
#include "visma_io.h"
#include "visma_icp_open3d.hpp"

namespace feh {
namespace gpu {

enum class SamplingMode { Surface = 0, Reference = 1 };

// feh::LoadMesh for .obj / .ply (core/utils.cpp:125-135: igl::readOBJ / igl::readPLY, first
// three columns kept).  Host code.  Returns false on failure like the reference.
template <typename DerivedV, typename DerivedF>
inline bool LoadMesh(const std::string &file, Eigen::PlainObjectBase<DerivedV> &V, Eigen::PlainObjectBase<DerivedF> &F)
{
    if (file.find(".obj") != std::string::npos) {
        double *v = nullptr; int32_t *f = nullptr; int64_t nv = 0, nf = 0; int fs = 0;
        if (visma_io_read_obj(file.c_str(), &v, &nv, &f, &nf, &fs) != VISMA_IO_OK) return false;
        V.resize(nv, 3);
        F.resize(nf, fs < 3 ? fs : 3);
        for (int64_t i = 0; i < nv; i++) for (int a = 0; a < 3; a++) V(i, a) = (typename DerivedV::Scalar)v[3 * i + a];
        for (int64_t i = 0; i < nf; i++) for (int a = 0; a < F.cols(); a++) F(i, a) = (typename DerivedF::Scalar)f[(size_t)i * fs + a];
        visma_io_free(v); visma_io_free(f);
        return true;
    }
    if (file.find(".ply") != std::string::npos) {
        visma_io_cloud c;
        if (visma_io_read_ply(file.c_str(), &c) != VISMA_IO_OK) return false;
        V.resize(c.n, 3);
        F.resize(c.n_faces, 3);
        for (int64_t i = 0; i < c.n; i++) for (int a = 0; a < 3; a++) V(i, a) = (typename DerivedV::Scalar)c.xyz[3 * i + a];
        for (int64_t i = 0; i < c.n_faces; i++) for (int a = 0; a < 3; a++) F(i, a) = (typename DerivedF::Scalar)c.faces[3 * i + a];
        visma_io_free_cloud(&c);
        return true;
    }
    return false;
}

template <typename T>
struct GenericErrorMetric {
    T mean_, std_, median_, min_, max_;
};

namespace detail {
template <typename M>
inline std::vector<double> rows3(const M &m)
{
    std::vector<double> o((size_t)m.rows() * 3);
    for (Eigen::Index i = 0; i < m.rows(); ++i)
        for (int c = 0; c < 3; ++c) o[(size_t)i * 3 + c] = (double)m(i, c);
    return o;
}
template <typename M>
inline std::vector<int32_t> faces3(const M &m)
{
    std::vector<int32_t> o((size_t)m.rows() * 3);
    for (Eigen::Index i = 0; i < m.rows(); ++i)
        for (int c = 0; c < 3; ++c) o[(size_t)i * 3 + c] = (int32_t)m(i, c);
    return o;
}
inline visma_icp_ctx *ctx() { return open3d::cicp::detail::ThreadContext::instance().get(); }
}  // namespace detail

template <typename T>
std::vector<Eigen::Matrix<T, 3, 1>> SamplePointCloudFromMesh(const Eigen::Matrix<T, Eigen::Dynamic, 3> &V,
                                                             const Eigen::Matrix<int, Eigen::Dynamic, 3> &F,
                                                             int max_num_pts = 1000,
                                                             SamplingMode mode = SamplingMode::Surface,
                                                             uint64_t seed = 0)
{
    std::vector<Eigen::Matrix<T, 3, 1>> out;
    if (max_num_pts <= 0 || F.rows() == 0) return out;
    const std::vector<double> v = detail::rows3(V);
    const std::vector<int32_t> f = detail::faces3(F);
    std::vector<double> p((size_t)max_num_pts * 3);
    int64_t m = 0;
    visma_icp_ctx *c = detail::ctx();
    open3d::cicp::detail::check(c, visma_icp_sample_mesh(c, v.data(), V.rows(), f.data(), F.rows(), max_num_pts,
                                                         mode == SamplingMode::Reference, seed, nullptr, p.data(), &m),
                                "visma_icp_sample_mesh");
    out.resize((size_t)m);
    for (int64_t i = 0; i < m; ++i)
        out[(size_t)i] = Eigen::Matrix<T, 3, 1>((T)p[3 * i], (T)p[3 * i + 1], (T)p[3 * i + 2]);
    return out;
}

template <typename T>
GenericErrorMetric<T> ComputeErrorMetric(std::vector<T> errors)
{
    std::vector<double> e(errors.begin(), errors.end());
    double o[5];
    if (visma_icp_error_metric(e.data(), (int64_t)e.size(), o) != VISMA_ICP_OK)
        throw std::runtime_error("visma_icp_error_metric: invalid arguments");
    return GenericErrorMetric<T>{(T)o[0], (T)o[1], (T)o[2], (T)o[3], (T)o[4]};
}

template <typename T>
GenericErrorMetric<T> MeasureSurfaceError(const Eigen::Matrix<T, Eigen::Dynamic, 3> &Vs,
                                          const Eigen::Matrix<int, Eigen::Dynamic, 3> &Fs,
                                          const Eigen::Matrix<T, Eigen::Dynamic, 3> &Vt,
                                          const Eigen::Matrix<int, Eigen::Dynamic, 3> &Ft, int num_samples,
                                          SamplingMode mode = SamplingMode::Surface, uint64_t seed = 0)
{
    const std::vector<double> vs = detail::rows3(Vs), vt = detail::rows3(Vt);
    const std::vector<int32_t> fs = detail::faces3(Fs), ft = detail::faces3(Ft);
    double o[5];
    visma_icp_ctx *c = detail::ctx();
    open3d::cicp::detail::check(
        c, visma_icp_measure_surface_error(c, vs.data(), Vs.rows(), fs.data(), Fs.rows(), vt.data(), Vt.rows(),
                                           ft.data(), Ft.rows(), num_samples, mode == SamplingMode::Reference, seed, o),
        "visma_icp_measure_surface_error");
    return GenericErrorMetric<T>{(T)o[0], (T)o[1], (T)o[2], (T)o[3], (T)o[4]};
}

// The reference's own signature: any options object with options["num_samples"].asInt()
// (Json::Value in the reference, geometry.h:122-124).
template <typename T, typename Options>
GenericErrorMetric<T> MeasureSurfaceError(const Eigen::Matrix<T, Eigen::Dynamic, 3> &Vs,
                                          const Eigen::Matrix<int, Eigen::Dynamic, 3> &Fs,
                                          const Eigen::Matrix<T, Eigen::Dynamic, 3> &Vt,
                                          const Eigen::Matrix<int, Eigen::Dynamic, 3> &Ft, const Options &options)
{
    return MeasureSurfaceError<T>(Vs, Fs, Vt, Ft, (int)options["num_samples"].asInt());
}

// feh::ICPRefinement (src/evaluation.cpp:248-271) with BOTH clouds made on the device: every model's mesh is
// sampled (SamplePointCloudFromMesh, samples_per_model draws), moved by its model_to_scene (PointCloud::Transform)
// and appended to the estimated scene there; the scan is voxel-down-sampled there (voxel_size <= 0: used as it is);
// then the registration runs.  The sampled points never cross PCIe (visma_icp_set_clouds_meshes_f64).
// `models`: anything iterable whose elements have V_, F_ (geometry.h's matrices) and model_to_scene_ -- the
// reference's feh::Model (include/evaluation.h), or MeshModel below; an unordered_map's values work through
// ICPRefinementMap.  Point-to-point estimator (options["use_point_to_plane"] = false, the reference's default).
// result.correspondence_set_ indexes the concatenation in model order / the voxels in ascending order;
// *scene_est_out (optional) receives the sampled cloud.
struct MeshModel {
    Eigen::Matrix<double, Eigen::Dynamic, 3> V_;
    Eigen::Matrix<int, Eigen::Dynamic, 3> F_;
    Eigen::Matrix4d model_to_scene_ = Eigen::Matrix4d::Identity();
};

template <typename ModelRange>
open3d::RegistrationResult ICPRefinement(const open3d::PointCloud &scene, const ModelRange &models,
                                         const Eigen::Matrix4d &T_scene_src, int samples_per_model, double voxel_size,
                                         double max_distance, SamplingMode mode = SamplingMode::Surface, uint64_t seed = 0,
                                         open3d::PointCloud *scene_est_out = nullptr)
{
    std::vector<std::vector<double>> vs, ts;
    std::vector<std::vector<int32_t>> fs;
    std::vector<visma_icp_mesh_source> ms;
    for (const auto &m : models) {
        vs.push_back(detail::rows3(m.V_));
        fs.push_back(detail::faces3(m.F_));
        ts.emplace_back(16);
        open3d::cicp::detail::to_rowmajor(m.model_to_scene_, ts.back().data());
    }
    for (size_t k = 0; k < vs.size(); k++)
        ms.push_back(visma_icp_mesh_source{vs[k].data(), (int64_t)(vs[k].size() / 3), fs[k].data(), (int64_t)(fs[k].size() / 3),
                                           (int64_t)samples_per_model, ts[k].data()});
    visma_icp_ctx *c = detail::ctx();
    int64_t ns = 0, nt = 0;
    open3d::cicp::detail::check(
        c, visma_icp_set_clouds_meshes_f64(c, ms.data(), (int)ms.size(), mode == SamplingMode::Reference, seed,
                                           open3d::cicp::detail::xyz(scene.points_), (int64_t)scene.points_.size(), 3,
                                           voxel_size > 0.0 ? voxel_size : 0.0, &ns, &nt),
        "visma_icp_set_clouds_meshes_f64");
    if (scene_est_out) {
        std::vector<double> p((size_t)ns * 3);
        open3d::cicp::detail::check(c, visma_icp_get_mesh_source(c, p.data(), ns), "visma_icp_get_mesh_source");
        scene_est_out->points_.resize((size_t)ns);
        for (int64_t i = 0; i < ns; i++) scene_est_out->points_[(size_t)i] = Eigen::Vector3d(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    }
    const open3d::ICPConvergenceCriteria crit;
    double T[16];
    open3d::cicp::detail::to_rowmajor(T_scene_src, T);
    visma_icp_result r;
    open3d::cicp::detail::check(c, visma_icp_run(c, T, max_distance, crit.max_iteration_, crit.relative_fitness_,
                                                 crit.relative_rmse_, VISMA_ICP_SOLVER_KABSCH, 0, &r), "visma_icp_run");
    open3d::RegistrationResult result(T_scene_src);
    open3d::cicp::detail::fill_result(c, r, (size_t)ns, result);
    return result;
}

// The reference's container: std::unordered_map<int, Model> (iterated in ITS order, as `for (const auto &kv : src)` does)
template <typename Map>
open3d::RegistrationResult ICPRefinementMap(const open3d::PointCloud &scene, const Map &src, const Eigen::Matrix4d &T_scene_src,
                                            int samples_per_model, double voxel_size, double max_distance,
                                            SamplingMode mode = SamplingMode::Surface, uint64_t seed = 0)
{
    std::vector<MeshModel> models;
    for (const auto &kv : src) {
        MeshModel m;
        m.V_ = kv.second.V_; m.F_ = kv.second.F_; m.model_to_scene_ = kv.second.model_to_scene_;
        models.push_back(m);
    }
    return ICPRefinement(scene, models, T_scene_src, samples_per_model, voxel_size, max_distance, mode, seed);
}

// ---- feh::AnnotationTool's steps around RegisterModelToScene (src/annotation.cpp:71-168) --------------------------
// feh::FindPlaneNormal (include/geometry.h:18-26): with the sign Eigen's JacobiSVD gives the singular vector
template <typename T>
inline Eigen::Matrix<T, 3, 1> FindPlaneNormal(const Eigen::Matrix<T, Eigen::Dynamic, 3> &pts)
{
    const std::vector<double> p = detail::rows3(pts);
    double n[3];
    if (visma_geom_find_plane_normal(p.data(), (int64_t)pts.rows(), n) != VISMA_ICP_OK) throw std::runtime_error("visma_geom_find_plane_normal");
    return Eigen::Matrix<T, 3, 1>((T)n[0], (T)n[1], (T)n[2]);
}
inline Eigen::Vector3d FindPlaneNormal(const std::vector<Eigen::Vector3d> &pts)
{
    double n[3];
    if (visma_geom_find_plane_normal(open3d::cicp::detail::xyz(pts), (int64_t)pts.size(), n) != VISMA_ICP_OK) throw std::runtime_error("visma_geom_find_plane_normal");
    return Eigen::Vector3d(n[0], n[1], n[2]);
}
// feh::RotationBetweenVectors (core/utils.h:229-233)
inline Eigen::Matrix3d RotationBetweenVectors(const Eigen::Vector3d &u, const Eigen::Vector3d &v)
{
    const double a[3] = {u(0), u(1), u(2)}, b[3] = {v(0), v(1), v(2)};
    double R[9];
    if (visma_geom_rotation_between_vectors(a, b, R) != VISMA_ICP_OK) throw std::runtime_error("visma_geom_rotation_between_vectors: zero vector");
    Eigen::Matrix3d M;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = R[3 * i + j];
    return M;
}
// T0 of src/annotation.cpp:82-89: the rotation that turns the floor's normal onto +Y (translation zero: the reference
// leaves T0's fourth column uninitialised; zero is the reading under which its output means something)
inline Eigen::Matrix4d GravityAlignment(const open3d::PointCloud &floor)
{
    Eigen::Matrix4d T0 = Eigen::Matrix4d::Identity();
    T0.block<3, 3>(0, 0) = RotationBetweenVectors(FindPlaneNormal(floor.points_), Eigen::Vector3d(0.0, 1.0, 0.0));
    return T0;
}

struct AnnotationObject {
    std::string name;                                        // the key in alignment.json (the scan's name)
    std::shared_ptr<open3d::PointCloud> scan;                // the object's fragment, raw (sensor frame)
    Eigen::Matrix<double, Eigen::Dynamic, 3> V;              // its CAD model
    Eigen::Matrix<int, Eigen::Dynamic, 3> F;
};
struct AnnotationPose {
    Eigen::Matrix4d T1, T2, T3, Ttot;                        // src/annotation.cpp:114-153
    int n_scan = 0, n_model = 0;                             // |down-sampled scan|, model samples (2 x that)
};

// The loop of AnnotationTool (src/annotation.cpp:103-168) over MANY objects: per object the scan is voxel-down-sampled
// on the device (:112), turned upright (T0) and centred on the floor (T1), the model sampled on the device with
// 2 x |scan| points (:126) and centred (T2); all (model, scan) pairs are then registered TOGETHER by the native work
// queue (RegisterModelsToScenes: rotation_level yaw starts per pair in flight), and Ttot = (T1 T0)^-1 T3 T2 comes back
// per object.  `alignment_json` (optional): written as the tool writes it (:156, 170-186).
inline std::vector<AnnotationPose> AnnotateObjects(const std::vector<AnnotationObject> &objects, const Eigen::Matrix4d &T0,
                                                   double voxel_size, int rotation_level, double distance_threshold,
                                                   const std::string &alignment_json = std::string(),
                                                   SamplingMode mode = SamplingMode::Surface, uint64_t seed = 0,
                                                   const std::vector<int> &devices = std::vector<int>())
{
    const size_t n = objects.size();
    std::vector<AnnotationPose> out(n);
    std::vector<std::pair<std::shared_ptr<open3d::PointCloud>, std::shared_ptr<open3d::PointCloud>>> pairs(n);
    auto shift = [](open3d::PointCloud &pc, const Eigen::Matrix4d &T) {          // PointCloud::Transform, rows 0..2
        for (auto &p : pc.points_) {
            const Eigen::Vector4d q = T * Eigen::Vector4d(p(0), p(1), p(2), 1.0);
            p = q.head<3>();
        }
    };
    for (size_t k = 0; k < n; k++) {
        const AnnotationObject &o = objects[k];
        auto scan = voxel_size > 0.0 ? open3d::cicp::VoxelDownSample(*o.scan, voxel_size) : std::make_shared<open3d::PointCloud>(*o.scan);
        shift(*scan, T0);
        double t[3];
        out[k].T1.setIdentity();
        if (visma_geom_centre_on_floor(open3d::cicp::detail::xyz(scan->points_), (int64_t)scan->points_.size(), t) != VISMA_ICP_OK)
            throw std::runtime_error("AnnotateObjects: empty scan '" + o.name + "'");
        out[k].T1.block<3, 1>(0, 3) = Eigen::Vector3d(t[0], t[1], t[2]);
        shift(*scan, out[k].T1);
        auto model = std::make_shared<open3d::PointCloud>();
        model->points_ = SamplePointCloudFromMesh(o.V, o.F, (int)(scan->points_.size() << 1), mode, seed + k);
        out[k].T2.setIdentity();
        if (visma_geom_centre_on_floor(open3d::cicp::detail::xyz(model->points_), (int64_t)model->points_.size(), t) != VISMA_ICP_OK)
            throw std::runtime_error("AnnotateObjects: no samples from the model of '" + o.name + "'");
        out[k].T2.block<3, 1>(0, 3) = Eigen::Vector3d(t[0], t[1], t[2]);
        shift(*model, out[k].T2);
        out[k].n_scan = (int)scan->points_.size();
        out[k].n_model = (int)model->points_.size();
        pairs[k] = std::make_pair(model, scan);
    }
    const std::vector<Eigen::Matrix4d> T3 = open3d::cicp::RegisterModelsToScenes(pairs, rotation_level, distance_threshold, devices);
    std::vector<visma_io_pose> poses(n);
    for (size_t k = 0; k < n; k++) {
        out[k].T3 = T3[k];
        double a[4][16], tot[16];
        open3d::cicp::detail::to_rowmajor(T0, a[0]);
        open3d::cicp::detail::to_rowmajor(out[k].T1, a[1]);
        open3d::cicp::detail::to_rowmajor(out[k].T2, a[2]);
        open3d::cicp::detail::to_rowmajor(out[k].T3, a[3]);
        visma_annot_total_pose(a[0], a[1], a[2], a[3], tot);
        out[k].Ttot = open3d::cicp::detail::from_rowmajor(tot);
        std::snprintf(poses[k].name, sizeof(poses[k].name), "%s", objects[k].name.c_str());
        for (int i = 0; i < 12; i++) poses[k].T[i] = tot[i];
    }
    if (!alignment_json.empty() && visma_io_write_alignment_json(alignment_json.c_str(), poses.data(), (int64_t)n) != 0)
        throw std::runtime_error(std::string("visma_io_write_alignment_json: ") + visma_io_last_error());
    return out;
}

}  // namespace gpu
}  // namespace feh
