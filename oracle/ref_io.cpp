// ref_io.cpp -- C-ABI harness around the reference's file readers.
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  Compiled together with
// O3D/IO/FileFormat/FilePLY.cpp and O3D/3rdparty/rply/rply.c where they lie
// (oracle/Makefile, target `ref`):
//   open3d::ReadPointCloudFromPLY   O3D/IO/FileFormat/FilePLY.cpp:206-264
//   open3d::ReadTriangleMeshFromPLY O3D/IO/FileFormat/FilePLY.cpp:336-397
// and with O3D/IO/FileFormat/FilePCD.cpp + O3D/3rdparty/liblzf/lzf_{c,d}.c:
//   open3d::ReadPointCloudFromPCD   O3D/IO/FileFormat/FilePCD.cpp:727-760
//   open3d::WritePointCloudToPCD    O3D/IO/FileFormat/FilePCD.cpp:762-788  (writes the binary_compressed fixtures)
#include <Core/Geometry/PointCloud.h>
#include <Core/Geometry/TriangleMesh.h>
#include <IO/ClassIO/PointCloudIO.h>
#include <IO/ClassIO/TriangleMeshIO.h>

#include <cstdint>
#include <cstdlib>
#include <string>

namespace {
double *dump(const std::vector<Eigen::Vector3d> &v)
{
    double *o = (double *)std::malloc(sizeof(double) * 3 * (v.size() ? v.size() : 1));
    for (size_t i = 0; i < v.size(); i++) for (int a = 0; a < 3; a++) o[3 * i + a] = v[i](a);
    return o;
}
}  // namespace

extern "C" {

// returns 1 on success; arrays are malloc'ed (free with ref_io_free)
int ref_read_ply_cloud(const char *path, double **xyz, int64_t *n, double **nrm, int64_t *nn, double **col, int64_t *nc)
{
    open3d::PointCloud pc;
    if (!open3d::ReadPointCloudFromPLY(path, pc)) return 0;
    *n = (int64_t)pc.points_.size(); *nn = (int64_t)pc.normals_.size(); *nc = (int64_t)pc.colors_.size();
    *xyz = dump(pc.points_); *nrm = dump(pc.normals_); *col = dump(pc.colors_);
    return 1;
}

int ref_read_ply_mesh(const char *path, double **xyz, int64_t *n, int32_t **tri, int64_t *nt)
{
    open3d::TriangleMesh m;
    if (!open3d::ReadTriangleMeshFromPLY(path, m)) return 0;
    *n = (int64_t)m.vertices_.size(); *nt = (int64_t)m.triangles_.size();
    *xyz = dump(m.vertices_);
    *tri = (int32_t *)std::malloc(sizeof(int32_t) * 3 * (m.triangles_.size() ? m.triangles_.size() : 1));
    for (size_t i = 0; i < m.triangles_.size(); i++) for (int a = 0; a < 3; a++) (*tri)[3 * i + a] = m.triangles_[i](a);
    return 1;
}

int ref_read_pcd_cloud(const char *path, double **xyz, int64_t *n, double **nrm, int64_t *nn, double **col, int64_t *nc)
{
    open3d::PointCloud pc;
    if (!open3d::ReadPointCloudFromPCD(path, pc)) return 0;
    *n = (int64_t)pc.points_.size(); *nn = (int64_t)pc.normals_.size(); *nc = (int64_t)pc.colors_.size();
    *xyz = dump(pc.points_); *nrm = dump(pc.normals_); *col = dump(pc.colors_);
    return 1;
}

int ref_write_pcd(const char *path, const double *xyz, int64_t n, const double *nrm, const double *col, int ascii,
                  int compressed)
{
    open3d::PointCloud pc;
    for (int64_t i = 0; i < n; i++) {
        pc.points_.push_back(Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        if (nrm) pc.normals_.push_back(Eigen::Vector3d(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
        if (col) pc.colors_.push_back(Eigen::Vector3d(col[3 * i], col[3 * i + 1], col[3 * i + 2]));
    }
    return open3d::WritePointCloudToPCD(path, pc, ascii != 0, compressed != 0) ? 1 : 0;
}

void ref_io_free(void *p) { std::free(p); }

}  // extern "C"
