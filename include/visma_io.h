/* visma_io.h -- readers for the on-disk formats either side of the ICP path
 * (SURVEY.md 8f row 4): the clouds and meshes VISMA's callers load before they
 * call RegistrationICP.  Host code (C++), part of libvisma_icp.so; plain C ABI.
 *
 *   visma_io_read_ply   open3d::ReadPointCloudFromPLY / ReadTriangleMeshFromPLY
 *                       (O3D/IO/FileFormat/FilePLY.cpp:206-264, :336-397; rply):
 *                       scene and scan clouds, src/evaluation.cpp:124,211,
 *                       src/annotation.cpp:76,80,111,157
 *   visma_io_read_pcd   open3d::ReadPointCloudFromPCD (O3D/IO/FileFormat/FilePCD.cpp:727-760)
 *   visma_io_read_alignment_json / _write_ / visma_io_read_result_json   the pose files
 *                       (core/utils.h:305-339, src/evaluation.cpp:126-181, src/annotation.cpp:147-153)
 *   visma_io_read_obj   igl::readOBJ(path, V, F) (libigl readOBJ.cpp:20-236): the
 *                       CAD models, src/evaluation.cpp:140,183, src/annotation.cpp:125,159,
 *                       core/utils.cpp:125-135 (LoadMesh keeps the first 3 columns)
 *
 * Same values as the reference readers, bit for bit (tests/test_io.py pins them
 * against outputs of the compiled reference): every PLY scalar type is widened to
 * double exactly, ASCII numbers go through strtod like rply, colours are
 * value / 255.0, OBJ indices are shifted like igl (1-based, negative = relative).
 * Binary vertex blocks are decoded on several host threads.
 *
 * Arrays are malloc'ed by the library; release them with visma_io_free_cloud /
 * visma_io_free.  Status 0 = OK; visma_io_last_error() (thread-local) says why not.
 */
#ifndef VISMA_IO_H
#define VISMA_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define VISMA_IO_API
#else
#define VISMA_IO_API __attribute__((visibility("default")))
#endif

enum { VISMA_IO_OK = 0, VISMA_IO_ERR_INVALID = 1, VISMA_IO_ERR_OPEN = 2, VISMA_IO_ERR_FORMAT = 3 };

typedef struct visma_io_cloud {
    int64_t n;            /* vertices ("element vertex") */
    double *xyz;          /* n x 3 */
    int64_t n_normals;    /* n if the file has nx, ny, nz; else 0 (FilePLY.cpp:228-231) */
    double *normals;      /* n_normals x 3 or NULL */
    int64_t n_colors;     /* n if the file has red, green, blue; else 0 */
    double *colors;       /* n_colors x 3, value / 255.0 (FilePLY.cpp:104-105) or NULL */
    int64_t n_faces;      /* "element face" with a list vertex_indices / vertex_index; else 0 */
    int32_t *faces;       /* n_faces x 3: the first three indices of every face (FilePLY.cpp:181-190) */
} visma_io_cloud;

VISMA_IO_API int visma_io_read_ply(const char *path, visma_io_cloud *out);
VISMA_IO_API void visma_io_free_cloud(visma_io_cloud *c);

/* V: nv x 3 (the first three numbers of every "v" line, as LoadMesh keeps them);
 * F: nf x face_size vertex indices, 0-based; every face must have the same number of
 * corners (igl's matrix overload fails otherwise -- so does this).  vt / vn and the
 * /vt/vn parts of face corners are parsed and dropped, like readOBJ(path, V, F). */
VISMA_IO_API int visma_io_read_obj(const char *path, double **V, int64_t *nv, int32_t **F,
                                   int64_t *nf, int *face_size);
VISMA_IO_API void visma_io_free(void *p);
VISMA_IO_API const char *visma_io_last_error(void);

/* open3d::ReadPointCloudFromPCD (O3D/IO/FileFormat/FilePCD.cpp:727-760): the format of the
 * reference's real-scan fixtures (fragment.pcd, cloud_bin_*.pcd).  DATA ascii / binary /
 * binary_compressed (LZF); fields x y z [normal_x normal_y normal_z] [rgb | rgba]; numeric types
 * I / U of 1, 2, 4 bytes and F of 4 bytes (anything else reads as 0, like the reference); colours
 * are the bytes of the 4-byte field as B, G, R, each / 255.0; rows whose x or y is NaN are
 * removed (the reference tests x twice and never z: a NaN z stays).  n_faces = 0. */
VISMA_IO_API int visma_io_read_pcd(const char *path, visma_io_cloud *out);

/* The pose files of the two callers.  A pose is the 3x4 matrix [R | t], 12 doubles row by row
 * (GetMatrixFromJson<double,3,4> / WriteMatrixToJson, core/utils.h:305-339).
 *   alignment.json   { "<model>_<k>": [12 numbers], ... }  read by src/evaluation.cpp:126-137, written by
 *                    src/annotation.cpp:147-153; poses come back in key order (jsoncpp iterates a
 *                    std::map), id = -1
 *   result.json      [ packet, ... ], packet = [ {"id", "status", "model_name", "model_pose": [12]}, ... ];
 *                    src/evaluation.cpp:166-181 evaluates the LAST packet: packet = -1
 * Arrays are calloc'ed: release with visma_io_free. */
typedef struct visma_io_pose {
    char name[256];
    int id, status;
    double T[12];
} visma_io_pose;
VISMA_IO_API int visma_io_read_alignment_json(const char *path, visma_io_pose **poses, int64_t *n);
VISMA_IO_API int visma_io_write_alignment_json(const char *path, const visma_io_pose *poses, int64_t n);
VISMA_IO_API int visma_io_read_result_json(const char *path, int64_t packet, visma_io_pose **poses, int64_t *n);

#ifdef __cplusplus
}
#endif
#endif /* VISMA_IO_H */
