// interpose_driver.cpp -- a caller that knows NOTHING of this repository: only Open3D's own headers, the calls of
// src/evaluation.cpp:260-271 as they stand.  Linked with libvisma_open3d_interpose.so ahead of the library that holds
// Open3D's own open3d::RegistrationICP (here: the compiled reference, oracle/_ref), its ICP runs on the GPU.
// Usage: interpose_driver <mode> <in.bin> <out.bin>   (file formats of shim_driver.cpp)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include <Core/Geometry/PointCloud.h>
#include <Core/Registration/Registration.h>

extern "C" int visma_open3d_interpose_calls();

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
    auto scene_est = std::make_shared<open3d::PointCloud>();
    auto scene = std::make_shared<open3d::PointCloud>();
    read_cloud(f, scene_est->points_, ns);
    read_cloud(f, scene->points_, nt);
    if (mode == "plane") { read_cloud(f, scene->normals_, nt); read_cloud(f, scene_est->normals_, ns); }
    std::fclose(f);
    Eigen::Matrix4d T_scene_src;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T_scene_src(i, j) = init_rm[i * 4 + j];

    open3d::RegistrationResult result;
    try {
        if (mode == "plane") {              // src/evaluation.cpp:261-265, verbatim but for the options lookup
            result = open3d::RegistrationICP(*scene_est,
                                            *scene,
                                            radius,
                                            T_scene_src,
                                            open3d::TransformationEstimationPointToPlane());
        } else if (mode == "default") {     // src/evaluation.cpp:267-270
            result = open3d::RegistrationICP(*scene_est,
                                            *scene,
                                            radius,
                                            T_scene_src);
        } else if (mode == "criteria") {    // src/annotation.cpp:51-56 with the stock estimator
            result = open3d::RegistrationICP(*scene_est, *scene, radius, T_scene_src,
                                             open3d::TransformationEstimationPointToPoint(),
                                             open3d::ICPConvergenceCriteria(0.0, 0.0, iters));
        } else if (mode == "evaluate") {
            result = open3d::EvaluateRegistration(*scene_est, *scene, radius, T_scene_src);
        } else {
            return 2;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "interpose_driver: %s\n", e.what());
        return 3;
    }
    FILE *o = std::fopen(argv[3], "wb");
    double out[20];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = result.transformation_(i, j);
    out[16] = result.fitness_; out[17] = result.inlier_rmse_;
    out[18] = (double)result.correspondence_set_.size();
    out[19] = (double)visma_open3d_interpose_calls();        // > 0: the substituted entry point ran
    std::fwrite(out, sizeof(double), 20, o);
    std::fclose(o);
    return 0;
}
