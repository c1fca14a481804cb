// ref_igl.cpp -- C-ABI harness around the reference's vendored libigl
// (thirdparty/libigl, header-only): igl::AABB::squared_distance exactly as
// feh::MeasureSurfaceError uses it (include/geometry.h:123-136).
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  include/geometry.h
// itself is not compiled: it pulls in utils.h (OpenCV, jsoncpp, abseil, glog),
// none of which exist in this image.
#include <igl/AABB.h>
#include <igl/readOBJ.h>

#include <cstdlib>

#include <cstdint>

extern "C" void ref_igl_point_mesh_sqdist(const double *P, int64_t np, const double *V, int64_t nv,
                                          const int32_t *F, int64_t nf, double *d2, int32_t *face,
                                          double *closest)
{
    Eigen::MatrixXd Vm(nv, 3), Pm(np, 3);
    Eigen::MatrixXi Fm(nf, 3);
    for (int64_t i = 0; i < nv; i++) for (int a = 0; a < 3; a++) Vm(i, a) = V[3 * i + a];
    for (int64_t i = 0; i < np; i++) for (int a = 0; a < 3; a++) Pm(i, a) = P[3 * i + a];
    for (int64_t i = 0; i < nf; i++) for (int a = 0; a < 3; a++) Fm(i, a) = F[3 * i + a];
    igl::AABB<Eigen::MatrixXd, 3> tree;
    tree.init(Vm, Fm);
    Eigen::VectorXd D2;
    Eigen::VectorXi I;
    Eigen::MatrixXd C;
    tree.squared_distance(Vm, Fm, Pm, D2, I, C);
    for (int64_t i = 0; i < np; i++) {
        d2[i] = D2(i);
        if (face) face[i] = I(i);
        if (closest) for (int a = 0; a < 3; a++) closest[3 * i + a] = C(i, a);
    }
}

// igl::readOBJ(path, V, F) as src/evaluation.cpp:140 / src/annotation.cpp:125 call it.
// Returns 1 on success; *V is nv x vcols (row-major), *F nf x fcols; free with std::free.
extern "C" int ref_igl_read_obj(const char *path, double **V, int64_t *nv, int *vcols, int32_t **F, int64_t *nf,
                                int *fcols)
{
    Eigen::MatrixXd Vm;
    Eigen::MatrixXi Fm;
    if (!igl::readOBJ(path, Vm, Fm)) return 0;
    *nv = Vm.rows(); *vcols = (int)Vm.cols(); *nf = Fm.rows(); *fcols = (int)Fm.cols();
    *V = (double *)std::malloc(sizeof(double) * (size_t)(Vm.size() ? Vm.size() : 1));
    *F = (int32_t *)std::malloc(sizeof(int32_t) * (size_t)(Fm.size() ? Fm.size() : 1));
    for (int64_t i = 0; i < Vm.rows(); i++) for (int a = 0; a < Vm.cols(); a++) (*V)[i * Vm.cols() + a] = Vm(i, a);
    for (int64_t i = 0; i < Fm.rows(); i++) for (int a = 0; a < Fm.cols(); a++) (*F)[i * Fm.cols() + a] = Fm(i, a);
    return 1;
}

// ---- the gravity-alignment helpers of feh::AnnotationTool (src/annotation.cpp:82-91) ------------------------------
// include/geometry.h:18-26 (FindPlaneNormal) and core/utils.h:229-233 (RotationBetweenVectors) cannot be compiled
// here (utils.h: OpenCV, jsoncpp, abseil).  Both are three Eigen expressions; what decides their output -- the sign of
// the normal included -- is Eigen's JacobiSVD / Quaternion, and THOSE are the vendored library itself below: the
// expressions restated on the real Eigen 3.3.2.
#include <Eigen/SVD>
#include <Eigen/Geometry>

extern "C" void ref_find_plane_normal(const double *xyz, int64_t n, double out[3])
{
    Eigen::Matrix<double, Eigen::Dynamic, 3> pts(n, 3);
    for (int64_t i = 0; i < n; i++) for (int a = 0; a < 3; a++) pts(i, a) = xyz[3 * i + a];
    // geometry.h:20-25
    auto pts_n = pts.rowwise() - pts.colwise().mean();
    auto P = pts_n.transpose() * pts_n / pts.rows();
    // (geometry.h:22 also passes ComputeThinU: on a fixed-size 3 x 3 matrix that trips an eigen_assert in a build
    //  without NDEBUG -- JacobiSVD.h:632 -- and in the reference's release build only decides whether U is formed;
    //  V, the only factor used, is the same)
    Eigen::JacobiSVD<Eigen::Matrix<double, 3, 3>> svd(P, Eigen::ComputeFullV);
    Eigen::Matrix<double, 3, 1> nrm = svd.matrixV().col(2);
    nrm.normalize();
    for (int a = 0; a < 3; a++) out[a] = nrm(a);
}

// Eigen::JacobiSVD of a 3x3 matrix (row-major in / out), full U and V: the decomposition FindPlaneNormal rests on
extern "C" void ref_jacobi_svd3(const double A[9], double U[9], double S[3], double V[9])
{
    Eigen::Matrix<double, 3, 3> M;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = A[3 * i + j];
    Eigen::JacobiSVD<Eigen::Matrix<double, 3, 3>> svd(M, Eigen::ComputeFullU | Eigen::ComputeFullV);
    for (int i = 0; i < 3; i++) {
        S[i] = svd.singularValues()(i);
        for (int j = 0; j < 3; j++) { U[3 * i + j] = svd.matrixU()(i, j); V[3 * i + j] = svd.matrixV()(i, j); }
    }
}

// core/utils.h:231-232
extern "C" void ref_rotation_between_vectors(const double u[3], const double v[3], double R[9])
{
    const Eigen::Matrix<double, 3, 3> M =
        Eigen::Quaternion<double>::FromTwoVectors(Eigen::Vector3d(u[0], u[1], u[2]), Eigen::Vector3d(v[0], v[1], v[2])).toRotationMatrix();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = M(i, j);
}
