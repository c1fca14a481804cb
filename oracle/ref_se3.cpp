// ref_se3.cpp -- C-ABI harness around the reference's header-only SE(3) / SO(3) classes (core/se3.h:10-169).
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  The header is used where it lies and AS IT IS: it does not
// build with g++ 11 (se3.h:142 lacks the `template` disambiguator inside SE3Type::cast, a member nobody on this path
// instantiates), so this one translation unit is compiled with the image's clang (ROCm's llvm) under
// -fdelayed-template-parsing, which parses a member template's body only when it is instantiated -- no stand-in, no
// edit (oracle/Makefile).  SE3Type::from_RT (se3.h:154-163: a constructor call that cannot deduce) is likewise never
// instantiated.  Compiled WITH -DEIGEN_DEFAULT_TO_ROW_MAJOR like ref_rodrigues.cpp (VISMA's CMakeLists.txt:11-12);
// everything crossing this boundary is a raw row-major double array.
#include "se3.h"

namespace {
typedef feh::SE3Type<double> SE3;
typedef feh::SO3Type<double> SO3;
SE3 load(const double g[12])
{
    Eigen::Matrix<double, 3, 4> m;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) m(i, j) = g[i * 4 + j];
    return SE3::from_matrix3x4(m);                            // se3.h:146-152
}
void store(const SE3 &g, double out[12])
{
    const Eigen::Matrix<double, 3, 4> m = g.matrix3x4();      // se3.h:128-131
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) out[i * 4 + j] = m(i, j);
}
}  // namespace

extern "C" {

// SE3Type::operator*(SE3Type) (se3.h:96-100)
void ref_se3_compose(const double a[12], const double b[12], double out[12]) { store(load(a) * load(b), out); }

// SE3Type::operator*(point) (se3.h:102-106)
void ref_se3_act(const double g[12], const double v[3], double out[3])
{
    const Eigen::Matrix<double, 3, 1> p = load(g) * Eigen::Matrix<double, 3, 1>(v[0], v[1], v[2]);
    for (int i = 0; i < 3; i++) out[i] = p(i);
}

// SE3Type::inv (se3.h:108-110)
void ref_se3_inv(const double g[12], double out[12]) { store(load(g).inv(), out); }

// SE3Type::matrix (se3.h:133-138): the 4 x 4 form
void ref_se3_matrix(const double g[12], double out[16])
{
    const Eigen::Matrix<double, 4, 4> m = load(g).matrix();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i * 4 + j] = m(i, j);
}

// (SO3Type::fitToSO3, se3.h:57-60, is NOT bound: it asks a fixed-size JacobiSVD for thin U / V, which Eigen 3.3.2 refuses
//  with an assertion (JacobiSVD.h:633) -- VISMA's own flags, CMakeLists.txt:5-6, do not define NDEBUG, so the reference's
//  build aborts there too.  projectSO3 of core/rodrigues.h:229-237 is the working form and is pinned by ref_rodrigues.cpp.)

// SO3Type::exp (se3.h:53-55), ::log (:44-46), the (axis, angle) constructor (:23-24)
void ref_so3_exp(const double w[3], double R[9])
{
    const Eigen::Matrix<double, 3, 3> r = SO3::exp(Eigen::Matrix<double, 3, 1>(w[0], w[1], w[2])).matrix();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = r(i, j);
}
void ref_so3_log(const double R[9], double w[3])
{
    Eigen::Matrix<double, 3, 3> m;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) m(i, j) = R[i * 3 + j];
    const Eigen::Matrix<double, 3, 1> v = SO3(m).log();
    for (int i = 0; i < 3; i++) w[i] = v(i);
}
void ref_so3_axis_angle(const double axis[3], double angle, double R[9])
{
    const Eigen::Matrix<double, 3, 3> r = SO3(Eigen::Matrix<double, 3, 1>(axis[0], axis[1], axis[2]), angle).matrix();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = r(i, j);
}

}  // extern "C"
