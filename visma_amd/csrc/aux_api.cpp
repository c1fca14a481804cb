// aux_api.cpp -- C ABI of the steps either side of the ICP loop: VoxelDownSample, EstimateNormals, mesh sampling
// and point-to-mesh distance, the error metric, and the SO(3) / SE(3) functions (host + device self-tests).
#include "driver_ctx.hpp"
#include "plane_math.hpp"

extern "C" {

int visma_icp_voxel_down_sample(visma_icp_ctx *ctx, const double *xyz, int64_t n, const double *normals,
                                const double *colors, double voxel_size, double *out_xyz,
                                double *out_normals, double *out_colors, int64_t *n_out)
{
    CTX_CHECK();
    if (!n_out || n < 0 || (n > 0 && (!xyz || !out_xyz)) || (normals && !out_normals) || (colors && !out_colors))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad voxel_down_sample arguments");
    if (n > 0x7fffffff) return ctx->fail(VISMA_ICP_ERR_INVALID, "too many points for 32-bit indices");
    if (!ctx->eng->supports_device_loop())   // only the HIP engine owns a GPU
        return ctx->fail(VISMA_ICP_ERR_STATE, "voxel_down_sample needs the HIP engine");
    int too_fine = 0;
    if (int rc = ctx->eng->bind_device()) return ctx->eng_fail(rc);
    hipError_t e = voxel_down_sample_device(xyz, normals, colors, n, voxel_size, out_xyz, out_normals,
                                            out_colors, n_out, &too_fine, ctx->eng->aux_stream());
    if (e != hipSuccess) return ctx->fail(VISMA_ICP_ERR_HIP, std::string("voxel_down_sample: ") + hipGetErrorString(e));
    if (too_fine) return ctx->fail(VISMA_ICP_ERR_INVALID, "voxel grid too fine to key in 62 bits");
    return VISMA_ICP_OK;
}

int visma_icp_estimate_normals(visma_icp_ctx *ctx, const double *xyz, int64_t n, const double *normals_in,
                               int search_type, int knn, double radius, double *normals_out)
{
    CTX_CHECK();
    if (n < 0 || (n > 0 && (!xyz || !normals_out)) || search_type < 0 || search_type > 2)
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad estimate_normals arguments");
    if (n > 0x7fffffff) return ctx->fail(VISMA_ICP_ERR_INVALID, "too many points for 32-bit indices");
    if (!ctx->eng->supports_device_loop()) return ctx->fail(VISMA_ICP_ERR_STATE, "estimate_normals needs the HIP engine");
    if (int rc = ctx->eng->bind_device()) return ctx->eng_fail(rc);
    hipError_t e = estimate_normals_device(xyz, n, normals_in, search_type, knn, radius, normals_out,
                                           ctx->eng->aux_stream());
    if (e != hipSuccess) return ctx->fail(e == hipErrorInvalidValue ? VISMA_ICP_ERR_INVALID : VISMA_ICP_ERR_HIP,
                                          std::string("estimate_normals: ") + hipGetErrorString(e));
    return VISMA_ICP_OK;
}

int visma_icp_point_mesh_distance(visma_icp_ctx *ctx, const double *P, int64_t np, const double *V,
                                  int64_t nv, const int32_t *F, int64_t nf, double *d2, int32_t *face,
                                  double *closest)
{
    CTX_CHECK();
    if (np < 0 || nv < 0 || nf < 0 || (np > 0 && (!P || !d2)) || (nf > 0 && (!V || !F)))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad point_mesh_distance arguments");
    if (!ctx->eng->supports_device_loop()) return ctx->fail(VISMA_ICP_ERR_STATE, "needs the HIP engine");
    float ms = 0.f, bms = 0.f;
    if (int rc = ctx->eng->bind_device()) return ctx->eng_fail(rc);
    hipError_t e = point_mesh_distance_device(P, np, V, nv, F, nf, ctx->mesh_method, d2, face, closest, &ms, &bms,
                                              ctx->eng->aux_stream());
    if (e != hipSuccess) return ctx->fail(e == hipErrorInvalidValue ? VISMA_ICP_ERR_INVALID : VISMA_ICP_ERR_HIP,
                                          std::string("point_mesh_distance: ") + hipGetErrorString(e));
    ctx->last_aux_kernel_ms = ms;
    ctx->last_aux_build_ms = bms;
    return VISMA_ICP_OK;
}

int visma_icp_last_mesh_kernel_ms(visma_icp_ctx *ctx, double *query_ms, double *build_ms)
{
    CTX_CHECK();
    if (query_ms) *query_ms = ctx->last_aux_kernel_ms;
    if (build_ms) *build_ms = ctx->last_aux_build_ms;
    return VISMA_ICP_OK;
}

int visma_icp_set_mesh_search(visma_icp_ctx *ctx, int method)
{
    CTX_CHECK();
    if (method < 0 || method > 2) return ctx->fail(VISMA_ICP_ERR_INVALID, "mesh search method must be 0, 1 or 2");
    ctx->mesh_method = method;
    return VISMA_ICP_OK;
}

int visma_icp_sample_mesh(visma_icp_ctx *ctx, const double *V, int64_t nv, const int32_t *F, int64_t nf,
                          int64_t n, int reference_quirks, uint64_t seed, const double *uniforms,
                          double *out_xyz, int64_t *n_out)
{
    CTX_CHECK();
    if (!n_out || n < 0 || nv < 0 || nf < 0 || (n > 0 && !out_xyz) || (nf > 0 && (!V || !F)))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad sample_mesh arguments");
    if (n > 0x7fffffff) return ctx->fail(VISMA_ICP_ERR_INVALID, "too many samples for 32-bit indices");
    if (!ctx->eng->supports_device_loop()) return ctx->fail(VISMA_ICP_ERR_STATE, "needs the HIP engine");
    if (int rc = ctx->eng->bind_device()) return ctx->eng_fail(rc);
    hipError_t e = sample_mesh_device(V, nv, F, nf, n, reference_quirks, (unsigned long long)seed, uniforms,
                                      out_xyz, n_out, ctx->eng->aux_stream());
    if (e != hipSuccess) return ctx->fail(e == hipErrorInvalidValue ? VISMA_ICP_ERR_INVALID : VISMA_ICP_ERR_HIP,
                                          std::string("sample_mesh: ") + hipGetErrorString(e));
    return VISMA_ICP_OK;
}

int visma_icp_error_metric(const double *errors, int64_t n, double out[5])
{
    if (!out || n <= 0 || !errors) return VISMA_ICP_ERR_INVALID;   // the reference indexes errors[n >> 1]
    // feh::ComputeErrorMetric (include/geometry.h:85-101), same accumulation order
    double mean = 0.0, sq = 0.0, mn = std::numeric_limits<double>::max(), mx = std::numeric_limits<double>::lowest();
    for (int64_t i = 0; i < n; i++) {
        mean += errors[i];
        sq += errors[i] * errors[i];
        mn = std::min(mn, errors[i]);
        mx = std::max(mx, errors[i]);
    }
    mean /= (double)n;
    // sorted[n >> 1] (geometry.h:96-97) without the full sort
    std::vector<double> s(errors, errors + n);
    if (n > 0) std::nth_element(s.begin(), s.begin() + (n >> 1), s.end());
    out[0] = mean;
    out[1] = std::sqrt(sq / (double)n - mean * mean);
    out[2] = n > 0 ? s[(size_t)n >> 1] : 0.0;
    out[3] = mn;
    out[4] = mx;
    return VISMA_ICP_OK;
}

int visma_icp_measure_surface_error(visma_icp_ctx *ctx, const double *Vs, int64_t nvs, const int32_t *Fs,
                                    int64_t nfs, const double *Vt, int64_t nvt, const int32_t *Ft,
                                    int64_t nft, int64_t num_samples, int reference_quirks, uint64_t seed,
                                    double out[5])
{
    CTX_CHECK();
    if (!out || num_samples <= 0) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad measure_surface_error arguments");
    if (nvs < 0 || nfs < 0 || nvt < 0 || nft < 0 || (nfs > 0 && (!Vs || !Fs)) || (nft > 0 && (!Vt || !Ft)))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad measure_surface_error arguments");
    if (!ctx->eng->supports_device_loop()) return ctx->fail(VISMA_ICP_ERR_STATE, "needs the HIP engine");
    std::vector<double> dist((size_t)num_samples);
    int64_t m = 0;
    float ms = 0.f, bms = 0.f;
    if (int rc = ctx->eng->bind_device()) return ctx->eng_fail(rc);
    hipError_t e = surface_distances_device(Vs, nvs, Fs, nfs, Vt, nvt, Ft, nft, num_samples, reference_quirks,
                                            (unsigned long long)seed, ctx->mesh_method, dist.data(), &m, &ms, &bms,
                                            ctx->eng->aux_stream());
    if (e != hipSuccess) return ctx->fail(e == hipErrorInvalidValue ? VISMA_ICP_ERR_INVALID : VISMA_ICP_ERR_HIP,
                                          std::string("measure_surface_error: ") + hipGetErrorString(e));
    ctx->last_aux_kernel_ms = ms;
    ctx->last_aux_build_ms = bms;
    if (m == 0) return ctx->fail(VISMA_ICP_ERR_INVALID, "no sample could be drawn from the source mesh");
    return visma_icp_error_metric(dist.data(), m, out);
}

int visma_icp_selftest_so3(const double *w, double *R, double *w_back, int n)
{
    if (!w || !R || !w_back || n <= 0) return VISMA_ICP_ERR_INVALID;
    double *dw = nullptr, *dR = nullptr, *dw2 = nullptr;
    int rc = VISMA_ICP_ERR_HIP;
    if (hipMalloc(&dw, sizeof(double) * 3 * n) == hipSuccess &&
        hipMalloc(&dR, sizeof(double) * 9 * n) == hipSuccess &&
        hipMalloc(&dw2, sizeof(double) * 3 * n) == hipSuccess &&
        hipMemcpy(dw, w, sizeof(double) * 3 * n, hipMemcpyHostToDevice) == hipSuccess &&
        launch_so3_selftest(dw, dR, dw2, n, nullptr) == hipSuccess &&
        hipMemcpy(R, dR, sizeof(double) * 9 * n, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(w_back, dw2, sizeof(double) * 3 * n, hipMemcpyDeviceToHost) == hipSuccess)
        rc = VISMA_ICP_OK;
    if (rc != VISMA_ICP_OK) g_create_error = "so3 selftest: HIP call failed (no GPU?)";
    (void)hipFree(dw); (void)hipFree(dR); (void)hipFree(dw2);
    return rc;
}

int visma_se3_compose(const double a[12], const double b[12], double out[12])
{
    if (!a || !b || !out) return VISMA_ICP_ERR_INVALID;
    se3_compose(a, b, out);
    return VISMA_ICP_OK;
}

int visma_se3_act(const double g[12], const double v[3], double out[3])
{
    if (!g || !v || !out) return VISMA_ICP_ERR_INVALID;
    se3_act(g, v, out);
    return VISMA_ICP_OK;
}

int visma_se3_inv(const double g[12], double out[12])
{
    if (!g || !out) return VISMA_ICP_ERR_INVALID;
    se3_inv(g, out);
    return VISMA_ICP_OK;
}

int visma_icp_selftest_se3(const double *g, const double *h, const double *v, int n, double *gh, double *gv, double *gi)
{
    if (n < 0 || (n > 0 && (!g || !h || !v || !gh || !gv || !gi))) return VISMA_ICP_ERR_INVALID;
    if (n == 0) return VISMA_ICP_OK;
    double *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sz[6] = {12, 12, 3, 12, 3, 12};
    const double *in[3] = {g, h, v};
    double *outp[3] = {gh, gv, gi};
    bool ok = true;
    for (int k = 0; k < 6 && ok; k++) ok = hipMalloc((void **)&d[k], sizeof(double) * sz[k] * (size_t)n) == hipSuccess;
    for (int k = 0; k < 3 && ok; k++) ok = hipMemcpy(d[k], in[k], sizeof(double) * sz[k] * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && launch_se3_selftest(d[0], d[1], d[2], n, d[3], d[4], d[5], nullptr) == hipSuccess;
    for (int k = 0; k < 3 && ok; k++) ok = hipMemcpy(outp[k], d[3 + k], sizeof(double) * sz[3 + k] * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
    for (int k = 0; k < 6; k++) if (d[k]) (void)hipFree(d[k]);
    if (!ok) { (void)hipGetLastError(); g_create_error = "se3 selftest: HIP call failed (no GPU?)"; return VISMA_ICP_ERR_HIP; }
    return VISMA_ICP_OK;
}

int visma_so3_rodrigues(const double w[3], double R[9], double dR_dw[27])
{
    if (!w || !R) return VISMA_ICP_ERR_INVALID;
    double D[27];
    rodrigues_jac(w, R, D);
    if (dR_dw) std::memcpy(dR_dw, D, sizeof(D));
    return VISMA_ICP_OK;
}

int visma_so3_invrodrigues(const double R[9], double w[3], double dw_dR[27])
{
    if (!w || !R) return VISMA_ICP_ERR_INVALID;
    double D[27];
    invrodrigues_jac(R, w, D);
    if (dw_dR) std::memcpy(dw_dR, D, sizeof(D));
    return VISMA_ICP_OK;
}

int visma_so3_project(const double A[9], double R[9])
{
    if (!A || !R) return VISMA_ICP_ERR_INVALID;
    project_so3(A, R);
    return VISMA_ICP_OK;
}

int visma_so3_matrix_derivatives(const double A[9], const double B[9], double dAB_dA_out[81], double dAB_dB_out[81],
                                 double dAt_dA_out[81], double dhat_out[27], double dvee_out[27])
{
    if (dAB_dA_out) { if (!B) return VISMA_ICP_ERR_INVALID; dAB_dA(B, dAB_dA_out); }
    if (dAB_dB_out) { if (!A) return VISMA_ICP_ERR_INVALID; dAB_dB(A, dAB_dB_out); }
    if (dAt_dA_out) dAt_dA(dAt_dA_out);
    if (dhat_out) dhat(dhat_out);
    if (dvee_out) dvee(dvee_out);
    return VISMA_ICP_OK;
}

int visma_icp_selftest_so3_jac(const double *w, int n, double *R, double *dR_dw, double *w_back, double *dw_dR,
                               double *proj)
{
    if (!w || !R || !dR_dw || !w_back || !dw_dR || !proj || n <= 0) return VISMA_ICP_ERR_INVALID;
    double *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sz[6] = {3, 9, 27, 3, 27, 9};
    int rc = VISMA_ICP_ERR_HIP;
    bool ok = true;
    for (int k = 0; k < 6 && ok; k++) ok = hipMalloc(&d[k], sizeof(double) * sz[k] * n) == hipSuccess;
    ok = ok && hipMemcpy(d[0], w, sizeof(double) * 3 * n, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && launch_so3_selftest_jac(d[0], n, d[1], d[2], d[3], d[4], d[5], nullptr) == hipSuccess;
    double *out[6] = {nullptr, R, dR_dw, w_back, dw_dR, proj};
    for (int k = 1; k < 6 && ok; k++) ok = hipMemcpy(out[k], d[k], sizeof(double) * sz[k] * n, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok) rc = VISMA_ICP_OK;
    else g_create_error = "so3 selftest: HIP call failed (no GPU?)";
    for (int k = 0; k < 6; k++) (void)hipFree(d[k]);
    return rc;
}

// ---- the gravity alignment and pose composition of feh::AnnotationTool (src/annotation.cpp:82-91, 111-153): host
// code (plane_math.hpp); the clouds it looks at are the ones about to be uploaded
int visma_geom_find_plane_normal(const double *xyz, int64_t n, double normal_out[3])
{
    if (!normal_out || n < 0 || (n > 0 && !xyz)) return VISMA_ICP_ERR_INVALID;
    visma::plane::find_plane_normal(xyz, n, normal_out);
    return VISMA_ICP_OK;
}

int visma_geom_jacobi_svd3(const double A[9], double U[9], double S[3], double V[9])
{
    if (!A || !U || !S || !V) return VISMA_ICP_ERR_INVALID;
    visma::plane::jacobi_svd3(A, U, S, V);
    return VISMA_ICP_OK;
}

int visma_geom_rotation_between_vectors(const double u[3], const double v[3], double R[9])
{
    if (!u || !v || !R) return VISMA_ICP_ERR_INVALID;
    const double lu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2], lv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (!(lu > 0.0) || !(lv > 0.0)) return VISMA_ICP_ERR_INVALID;
    visma::plane::rotation_between_vectors(u, v, R);
    return VISMA_ICP_OK;
}

int visma_geom_centre_on_floor(const double *xyz, int64_t n, double t_out[3])
{
    if (!t_out || n <= 0 || !xyz) return VISMA_ICP_ERR_INVALID;
    visma::plane::centre_on_floor(xyz, n, t_out);
    return VISMA_ICP_OK;
}

// Ttot = (T1 T0)^-1 T3 T2 with the reference's rigid inverse (rotation transposed, t <- -R^T t), row-major 4x4
int visma_annot_total_pose(const double T0[16], const double T1[16], const double T2[16], const double T3[16], double Ttot[16])
{
    if (!T0 || !T1 || !T2 || !T3 || !Ttot) return VISMA_ICP_ERR_INVALID;
    auto mul = [](const double *a, const double *b, double *c) {
        double r[16];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double v = 0.0;
                for (int k = 0; k < 4; k++) v += a[4 * i + k] * b[4 * k + j];
                r[4 * i + j] = v;
            }
        for (int i = 0; i < 16; i++) c[i] = r[i];
    };
    double A[16], Ai[16];
    mul(T1, T0, A);
    for (int i = 0; i < 16; i++) Ai[i] = A[i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ai[4 * i + j] = A[4 * j + i];      // block<3,3>.transposeInPlace()
    for (int i = 0; i < 3; i++) Ai[4 * i + 3] = -(Ai[4 * i] * A[3] + Ai[4 * i + 1] * A[7] + Ai[4 * i + 2] * A[11]);
    mul(Ai, T3, A);
    mul(A, T2, Ttot);
    return VISMA_ICP_OK;
}

}  // extern "C"
