// annotate_driver.cpp -- feh::AnnotationTool's flow (src/annotation.cpp:71-168) through the C++ shim:
// GravityAlignment(floor) -> AnnotateObjects(objects, T0, voxel, level, threshold, alignment.json).
// Usage: annotate_driver <in.bin> <out.bin> <alignment.json>
//   in : int64 n_floor, n_scan, nv, nf; double voxel, threshold; int32 level, copies; floor, scan, V (doubles), F (int32)
//   out: T0[16], then per object: T1[16] T2[16] T3[16] Ttot[16] n_scan n_model
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "constrained_ICP.h"
#include "visma_geometry.hpp"

static void read_pts(FILE *f, std::vector<Eigen::Vector3d> &v, int64_t n)
{
    v.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        double p[3];
        if (fread(p, 8, 3, f) != 3) std::exit(2);
        v[(size_t)i] = Eigen::Vector3d(p[0], p[1], p[2]);
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t nfl, ns, nv, nf; double voxel, thr; int32_t level, copies;
    if (fread(&nfl, 8, 1, f) != 1 || fread(&ns, 8, 1, f) != 1 || fread(&nv, 8, 1, f) != 1 || fread(&nf, 8, 1, f) != 1 ||
        fread(&voxel, 8, 1, f) != 1 || fread(&thr, 8, 1, f) != 1 || fread(&level, 4, 1, f) != 1 || fread(&copies, 4, 1, f) != 1)
        return 2;
    open3d::PointCloud floor;
    auto scan = std::make_shared<open3d::PointCloud>();
    std::vector<Eigen::Vector3d> verts;
    read_pts(f, floor.points_, nfl);
    read_pts(f, scan->points_, ns);
    read_pts(f, verts, nv);
    Eigen::Matrix<double, Eigen::Dynamic, 3> V(nv, 3);
    for (int64_t i = 0; i < nv; i++) for (int c = 0; c < 3; c++) V(i, c) = verts[(size_t)i](c);
    Eigen::Matrix<int, Eigen::Dynamic, 3> F(nf, 3);
    for (int64_t i = 0; i < nf; i++) {
        int32_t t[3];
        if (fread(t, 4, 3, f) != 3) return 2;
        for (int c = 0; c < 3; c++) F(i, c) = t[c];
    }
    std::fclose(f);
    try {
        const Eigen::Matrix4d T0 = feh::gpu::GravityAlignment(floor);
        std::vector<feh::gpu::AnnotationObject> objs;
        for (int k = 0; k < copies; k++) objs.push_back(feh::gpu::AnnotationObject{"hermanmiller_aeron_" + std::to_string(k), scan, V, F});
        const auto poses = feh::gpu::AnnotateObjects(objs, T0, voxel, level, thr, argv[3]);
        FILE *o = std::fopen(argv[2], "wb");
        auto put = [&](const Eigen::Matrix4d &M) {
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { const double v = M(i, j); std::fwrite(&v, 8, 1, o); }
        };
        put(T0);
        for (const auto &p : poses) {
            put(p.T1); put(p.T2); put(p.T3); put(p.Ttot);
            const double c[2] = {(double)p.n_scan, (double)p.n_model};
            std::fwrite(c, 8, 2, o);
        }
        std::fclose(o);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "annotate_driver: %s\n", e.what());
        return 3;
    }
    return 0;
}
