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
