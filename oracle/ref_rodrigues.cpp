// ref_rodrigues.cpp -- C-ABI harness around the reference's header-only
// SO(3) math (core/rodrigues.h).
//
// core/se3.h does not build with g++ 11 (se3.h:142 lacks the `template`
// disambiguator): its harness is ref_se3.cpp, compiled with the image's clang
// (round 6) -- SE3Type compose/act/inv (se3.h:96-110) are pinned by ITS outputs
// (tests/golden/se3.npz) since then, not by the restatement alone.
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  Compiled as its own
// translation unit WITH -DEIGEN_DEFAULT_TO_ROW_MAJOR, as VISMA's
// CMakeLists.txt:11-12 does: rodrigues.h:173-174 maps Matrix3::data() into a
// 9-vector, so its Jacobians are only self-consistent with row-major storage.
// Everything crossing this boundary is a raw double array (row-major), so the
// storage-order macro cannot leak into the Open3D translation units.
#include "rodrigues.h"

extern "C" {

void ref_rodrigues(const double w[3], double R[9], double *dR_dw /*9x3*/)
{
    Eigen::Matrix<double, 3, 1> wv(w[0], w[1], w[2]);
    Eigen::Matrix<double, 9, 3> D;
    Eigen::Matrix<double, 3, 3> Rm = feh::rodrigues(wv, dR_dw ? &D : nullptr);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = Rm(i, j);
    if (dR_dw)
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 3; j++) dR_dw[i * 3 + j] = D(i, j);
}

void ref_invrodrigues(const double R[9], double w[3], double *dw_dR /*3x9*/)
{
    Eigen::Matrix<double, 3, 3> Rm;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rm(i, j) = R[i * 3 + j];
    Eigen::Matrix<double, 3, 9> D;
    Eigen::Matrix<double, 3, 1> wv = feh::invrodrigues(Rm, dw_dR ? &D : nullptr);
    for (int i = 0; i < 3; i++) w[i] = wv(i);
    if (dw_dR)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 9; j++) dw_dR[i * 9 + j] = D(i, j);
}

void ref_hat(const double u[3], double M[9])
{
    Eigen::Matrix<double, 3, 1> uv(u[0], u[1], u[2]);
    Eigen::Matrix<double, 3, 3> H = feh::hat(uv);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[i * 3 + j] = H(i, j);
}

}  // extern "C"
