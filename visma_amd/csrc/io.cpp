// io.cpp -- host readers for the on-disk formats either side of the ICP path
// (C ABI in include/visma_io.h).
//
//   PLY  open3d::ReadPointCloudFromPLY / ReadTriangleMeshFromPLY semantics
//        (O3D/IO/FileFormat/FilePLY.cpp:206-264, :336-397, on rply): element
//        "vertex" -> x,y,z [nx,ny,nz] [red,green,blue -> /255.0]; element "face" ->
//        the first three entries of the list vertex_indices / vertex_index.
//   OBJ  igl::readOBJ(path, V, F) semantics (libigl readOBJ.cpp:51-236).
//
// The whole file is read into memory once; a binary vertex block with a fixed
// record size is decoded on several host threads.
#include "../../include/visma_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_io_err;

int fail(int code, const std::string &msg)
{
    g_io_err = msg;
    return code;
}

// The file, read-only in memory, followed by one '\0' (the text parsers rely on it): mapped
// when its size leaves room for the terminator inside the last page, else read into a buffer.
struct FileBytes {
    const char *data = nullptr;
    size_t size = 0;                           // without the terminator
    void *map = nullptr;
    size_t map_len = 0;
    std::vector<char> buf;
    ~FileBytes() { if (map) munmap(map, map_len); }
    bool open(const char *path)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
        size = (size_t)st.st_size;
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        if (size > 0 && size % page != 0) {    // bytes after EOF in the last page read as 0
            void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { map = m; map_len = size; data = (const char *)m; ::close(fd); return true; }
        }
        buf.resize(size + 1);
        size_t got = 0;
        while (got < size) {
            const ssize_t r = ::read(fd, buf.data() + got, size - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        ::close(fd);
        if (got != size) return false;
        buf[size] = '\0';
        data = buf.data();
        return true;
    }
};

template <typename F>
void parallel_chunks(int64_t n, int64_t chunk, F fn)
{
    const int64_t nch = (n + chunk - 1) / chunk;
    int64_t nt = (int64_t)std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt > nch) nt = nch;
    if (nt <= 1) { for (int64_t c = 0; c < nch; c++) fn(c * chunk, std::min(n, (c + 1) * chunk)); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int64_t t = 0; t < nt; t++)
        th.emplace_back([&]() {
            for (;;) {
                const int64_t c = next.fetch_add(1);
                if (c >= nch) break;
                fn(c * chunk, std::min(n, (c + 1) * chunk));
            }
        });
    for (auto &t : th) t.join();
}

// ---------------------------------------------------------------- PLY
enum PlyType { T_NONE = -1, T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64 };
const int kTypeSize[8] = {1, 1, 2, 2, 4, 4, 4, 8};

PlyType ply_type(const std::string &s)
{
    static const struct { const char *n; PlyType t; } names[] = {
        {"char", T_I8}, {"int8", T_I8}, {"uchar", T_U8}, {"uint8", T_U8}, {"short", T_I16}, {"int16", T_I16},
        {"ushort", T_U16}, {"uint16", T_U16}, {"int", T_I32}, {"int32", T_I32}, {"uint", T_U32},
        {"uint32", T_U32}, {"float", T_F32}, {"float32", T_F32}, {"double", T_F64}, {"float64", T_F64}};
    for (const auto &e : names)
        if (s == e.n) return e.t;
    return T_NONE;
}

struct PlyProp {
    std::string name;
    bool list = false;
    PlyType type = T_NONE, count_type = T_NONE;
    int offset = 0;                            // within a fixed-size record
};
struct PlyElem {
    std::string name;
    int64_t count = 0;
    std::vector<PlyProp> props;
    bool fixed() const { for (const auto &p : props) if (p.list) return false; return true; }
    int record() const { int s = 0; for (const auto &p : props) s += kTypeSize[p.type]; return s; }
    int find(const char *n) const
    {
        for (size_t i = 0; i < props.size(); i++) if (props[i].name == n) return (int)i;
        return -1;
    }
};

inline double load_bin(const unsigned char *p, PlyType t, bool swap)
{
    unsigned char b[8];
    const int n = kTypeSize[t];
    if (swap) for (int i = 0; i < n; i++) b[i] = p[n - 1 - i];
    else std::memcpy(b, p, (size_t)n);
    switch (t) {
    case T_I8: { int8_t v; std::memcpy(&v, b, 1); return (double)v; }
    case T_U8: { uint8_t v; std::memcpy(&v, b, 1); return (double)v; }
    case T_I16: { int16_t v; std::memcpy(&v, b, 2); return (double)v; }
    case T_U16: { uint16_t v; std::memcpy(&v, b, 2); return (double)v; }
    case T_I32: { int32_t v; std::memcpy(&v, b, 4); return (double)v; }
    case T_U32: { uint32_t v; std::memcpy(&v, b, 4); return (double)v; }
    case T_F32: { float v; std::memcpy(&v, b, 4); return (double)v; }
    case T_F64: { double v; std::memcpy(&v, b, 8); return v; }
    default: return 0.0;
    }
}

// one ASCII value, the way rply reads it: strtod for the floating types, strtol for the
// integer ones (the text of a "float" property keeps its full double value)
inline bool load_ascii(const char *&p, const char *end, PlyType t, double &out)
{
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    if (p >= end) return false;
    char *q = nullptr;
    if (t == T_F32 || t == T_F64) out = std::strtod(p, &q);
    else out = (double)std::strtol(p, &q, 10);
    if (q == p) return false;
    p = q;
    return true;
}

template <typename T>
T *alloc_arr(int64_t n) { return (T *)std::malloc(sizeof(T) * (size_t)(n > 0 ? n : 1)); }

}  // namespace

extern "C" {

const char *visma_io_last_error(void) { return g_io_err.c_str(); }

void visma_io_free(void *p) { std::free(p); }

void visma_io_free_cloud(visma_io_cloud *c)
{
    if (!c) return;
    std::free(c->xyz); std::free(c->normals); std::free(c->colors); std::free(c->faces);
    std::memset(c, 0, sizeof(*c));
}

int visma_io_read_ply(const char *path, visma_io_cloud *out)
{
    if (!path || !out) return fail(VISMA_IO_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    FileBytes file;
    if (!file.open(path)) return fail(VISMA_IO_ERR_OPEN, std::string("unable to open file: ") + path);
    const char *p = file.data, *end = file.data + file.size;

    // ---- header
    auto next_line = [&](std::string &line) {
        if (p >= end) return false;
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *stop = nl ? nl : end;
        line.assign(p, stop);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        p = nl ? nl + 1 : end;
        return true;
    };
    std::string line;
    if (!next_line(line) || line != "ply") return fail(VISMA_IO_ERR_FORMAT, "not a PLY file");
    int format = -1;                           // 0 ascii, 1 little endian, 2 big endian
    std::vector<PlyElem> elems;
    bool done = false;
    while (!done && next_line(line)) {
        std::vector<std::string> tok;
        size_t i = 0;
        while (i < line.size()) {
            while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) i++;
            size_t j = i;
            while (j < line.size() && line[j] != ' ' && line[j] != '\t') j++;
            if (j > i) tok.push_back(line.substr(i, j - i));
            i = j;
        }
        if (tok.empty()) continue;
        if (tok[0] == "format" && tok.size() >= 2) {
            format = tok[1] == "ascii" ? 0 : tok[1] == "binary_little_endian" ? 1 : tok[1] == "binary_big_endian" ? 2 : -1;
        } else if (tok[0] == "element" && tok.size() >= 3) {
            PlyElem e;
            e.name = tok[1];
            e.count = std::strtoll(tok[2].c_str(), nullptr, 10);
            if (e.count < 0) return fail(VISMA_IO_ERR_FORMAT, "negative element count");
            elems.push_back(e);
        } else if (tok[0] == "property" && !elems.empty()) {
            PlyProp pr;
            if (tok.size() >= 5 && tok[1] == "list") {
                pr.list = true;
                pr.count_type = ply_type(tok[2]);
                pr.type = ply_type(tok[3]);
                pr.name = tok[4];
                if (pr.count_type == T_NONE) return fail(VISMA_IO_ERR_FORMAT, "unknown list count type");
            } else if (tok.size() >= 3) {
                pr.type = ply_type(tok[1]);
                pr.name = tok[2];
            }
            if (pr.type == T_NONE) return fail(VISMA_IO_ERR_FORMAT, "unknown property type in the header");
            elems.back().props.push_back(pr);
        } else if (tok[0] == "end_header") {
            done = true;
        }                                     // comment / obj_info: skipped
    }
    if (!done || format < 0) return fail(VISMA_IO_ERR_FORMAT, "unable to parse header");

    const PlyElem *ve = nullptr;
    for (const auto &e : elems) if (e.name == "vertex") { ve = &e; break; }
    const int ix = ve ? ve->find("x") : -1;
    if (!ve || ix < 0 || ve->count <= 0) return fail(VISMA_IO_ERR_FORMAT, "number of vertex <= 0");   // FilePLY.cpp:238-242
    const bool swap = format == 2;             // this code runs on little-endian hosts

    // ---- body: the elements in file order
    visma_io_cloud c;
    std::memset(&c, 0, sizeof(c));
    auto bail = [&](const std::string &m) { visma_io_free_cloud(&c); return fail(VISMA_IO_ERR_FORMAT, m); };
    bool have_faces = false;
    for (const auto &e : elems) {
        const bool is_vertex = &e == ve;
        const bool is_face = e.name == "face";
        // A declared count is not trusted: it must fit 32-bit indices and the bytes that are left
        // (a record, or an ASCII value, takes at least one byte).  A wrapped 3 * count would
        // otherwise size a tiny buffer that the decode below overruns.
        if (e.count < 0 || e.count > 0x7fffffffll) return bail("element count out of range");
        {
            const int64_t left = (int64_t)(end - p);
            const int64_t per = (format != 0 && e.fixed()) ? std::max(1, e.record()) : 1;
            if (!e.props.empty() && e.count > left / per) return bail("unable to read file: truncated element data");
        }
        int fprop = is_face ? e.find("vertex_indices") : -1;
        if (is_face && fprop < 0) fprop = e.find("vertex_index");
        if (is_face && fprop >= 0 && !e.props[fprop].list) fprop = -1;
        int want[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
        if (is_vertex) {
            static const char *names[9] = {"x", "y", "z", "nx", "ny", "nz", "red", "green", "blue"};
            for (int k = 0; k < 9; k++) {
                want[k] = e.find(names[k]);
                if (want[k] >= 0 && e.props[want[k]].list) want[k] = -1;
            }
            // FilePLY.cpp:228-236,77-79: the normal / colour arrays are sized by `nx` / `red` alone, while `ny`, `nz`,
            // `green`, `blue` get their callbacks regardless: a file with one of those but not the sizing property
            // makes the callback return 0 on the first vertex and ply_read fail ("unable to read file").
            if (e.count > 0 && ((want[3] < 0 && (want[4] >= 0 || want[5] >= 0)) || (want[6] < 0 && (want[7] >= 0 || want[8] >= 0))))
                return bail("unable to read file: normal / colour components without nx / red");
            c.n = e.count;
            c.n_normals = want[3] >= 0 ? e.count : 0;
            c.n_colors = want[6] >= 0 ? e.count : 0;
            c.xyz = alloc_arr<double>(3 * c.n);
            if (c.n_normals) c.normals = alloc_arr<double>(3 * c.n);
            if (c.n_colors) c.colors = alloc_arr<double>(3 * c.n);
            if (!c.xyz || (c.n_normals && !c.normals) || (c.n_colors && !c.colors)) return bail("out of memory");
            // a property of a triple that the file lacks stays 0 (complete triples are written
            // in full by the decode below: no zero fill, the decoding threads touch the pages)
            if (want[0] < 0 || want[1] < 0 || want[2] < 0) std::memset(c.xyz, 0, sizeof(double) * 3 * (size_t)c.n);
            if (c.normals && (want[4] < 0 || want[5] < 0)) std::memset(c.normals, 0, sizeof(double) * 3 * (size_t)c.n);
            if (c.colors && (want[7] < 0 || want[8] < 0)) std::memset(c.colors, 0, sizeof(double) * 3 * (size_t)c.n);
        }
        // (rply hands an element's values to the callbacks of the FIRST element of that name only:
        //  a repeated "face" element is walked, not stored)
        if (is_face && have_faces) fprop = -1;
        if (is_face && fprop >= 0) {
            have_faces = true;
            c.n_faces = e.count;
            c.faces = alloc_arr<int32_t>(3 * c.n_faces);
            if (!c.faces) return bail("out of memory");
            std::memset(c.faces, 0, sizeof(int32_t) * 3 * (size_t)c.n_faces);
        }
        auto store_vertex = [&](int64_t i, int k, double v) {
            if (k < 3) c.xyz[3 * i + k] = v;
            else if (k < 6) c.normals[3 * i + (k - 3)] = v;
            else c.colors[3 * i + (k - 6)] = v / 255.0;
        };
        if (format != 0 && e.fixed()) {
            // fixed-size binary records
            const int rec = e.record();
            if ((int64_t)(end - p) < (int64_t)rec * e.count) return bail("unable to read file: truncated element data");
            if (is_vertex) {
                std::vector<int> off(e.props.size());
                int o = 0;
                for (size_t k = 0; k < e.props.size(); k++) { off[k] = o; o += kTypeSize[e.props[k].type]; }
                const unsigned char *base = (const unsigned char *)p;
                parallel_chunks(e.count, 65536, [&](int64_t lo, int64_t hi) {
                    for (int64_t i = lo; i < hi; i++) {
                        const unsigned char *r = base + (size_t)i * rec;
                        for (int k = 0; k < 9; k++)
                            if (want[k] >= 0) store_vertex(i, k, load_bin(r + off[want[k]], e.props[want[k]].type, swap));
                    }
                });
            }
            p += (size_t)rec * (size_t)e.count;
        } else {
            // ASCII, or binary records with lists: walk value by value
            for (int64_t i = 0; i < e.count; i++) {
                for (size_t k = 0; k < e.props.size(); k++) {
                    const PlyProp &pr = e.props[k];
                    auto read_one = [&](PlyType t, double &v) {
                        if (format == 0) return load_ascii(p, end, t, v);
                        if (end - p < kTypeSize[t]) return false;
                        v = load_bin((const unsigned char *)p, t, swap);
                        p += kTypeSize[t];
                        return true;
                    };
                    double v = 0.0;
                    if (!pr.list) {
                        if (!read_one(pr.type, v)) return bail("unable to read file: truncated element data");
                        if (is_vertex)
                            for (int w = 0; w < 9; w++)
                                if (want[w] == (int)k) store_vertex(i, w, v);
                    } else {
                        double cnt = 0.0;
                        if (!read_one(pr.count_type, cnt) || cnt < 0) return bail("unable to read file: bad list length");
                        const int64_t len = (int64_t)cnt;
                        for (int64_t j = 0; j < len; j++) {
                            if (!read_one(pr.type, v)) return bail("unable to read file: truncated list");
                            if (is_face && (int)k == fprop && j < 3) c.faces[3 * i + j] = (int32_t)v;   // FilePLY.cpp:181-186
                        }
                    }
                }
            }
        }
    }
    *out = c;
    return VISMA_IO_OK;
}

int visma_io_read_obj(const char *path, double **V, int64_t *nv, int32_t **F, int64_t *nf, int *face_size)
{
    if (!path || !V || !nv || !F || !nf || !face_size) return fail(VISMA_IO_ERR_INVALID, "null argument");
    *V = nullptr; *F = nullptr; *nv = *nf = 0; *face_size = 0;
    FileBytes file;
    if (!file.open(path)) return fail(VISMA_IO_ERR_OPEN, std::string(path) + " could not be opened");
    std::vector<double> v;                     // 3 per vertex
    std::vector<int32_t> f;
    int64_t n_v = 0, n_vt = 0, n_vn = 0, n_f = 0;
    int vcols = -1, fcols = -1;
    int line_no = 0;
    const char *p = file.data, *end = file.data + file.size;
    while (p < end) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        ++line_no;
        // first word = record type (sscanf "%s" skips leading blanks)
        const char *w = p;
        while (w < le && (*w == ' ' || *w == '\t' || *w == '\r')) ++w;
        const char *we = w;
        while (we < le && *we != ' ' && *we != '\t' && *we != '\r') ++we;
        const size_t wl = (size_t)(we - w);
        if (wl == 1 && *w == 'v') {
            // igl parses &line[1] as a run of numbers (readOBJ.cpp:84-85)
            const char *q = p + 1;
            double vals[3] = {0, 0, 0};
            int cnt = 0;
            for (;;) {
                while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
                if (q >= le) break;
                char *e2 = nullptr;
                const double x = std::strtod(q, &e2);
                if (e2 == q || e2 > le) break;
                if (cnt < 3) vals[cnt] = x;
                ++cnt;
                q = e2;
            }
            if (cnt < 3) return fail(VISMA_IO_ERR_FORMAT, "vertex on line " + std::to_string(line_no) + " should have at least 3 coordinates");
            if (vcols < 0) vcols = cnt;
            else if (vcols != cnt) return fail(VISMA_IO_ERR_FORMAT, "vertices with different numbers of coordinates (line " + std::to_string(line_no) + ")");
            v.insert(v.end(), vals, vals + 3);
            ++n_v;
        } else if (wl == 2 && w[0] == 'v' && w[1] == 'n') {
            ++n_vn;
        } else if (wl == 2 && w[0] == 'v' && w[1] == 't') {
            ++n_vt;
        } else if (wl == 1 && *w == 'f') {
            const char *q = we;
            int corners = 0;
            for (;;) {
                while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
                if (q >= le) break;
                char *e2 = nullptr;
                const long i = std::strtol(q, &e2, 10);
                if (e2 == q) return fail(VISMA_IO_ERR_FORMAT, "face on line " + std::to_string(line_no) + " has invalid element format");
                f.push_back((int32_t)(i < 0 ? i + n_v : i - 1));        // readOBJ.cpp:135-138
                ++corners;
                q = e2;
                while (q < le && *q != ' ' && *q != '\t' && *q != '\r') ++q;   // the /vt/vn part of the corner
            }
            if (corners > 0) {
                if (fcols < 0) fcols = corners;
                else if (fcols != corners) return fail(VISMA_IO_ERR_FORMAT, "faces with different numbers of corners (line " + std::to_string(line_no) + ")");
                ++n_f;
            }
        }
        p = nl ? nl + 1 : end;
    }
    (void)n_vt; (void)n_vn;
    *V = alloc_arr<double>(3 * n_v);
    *F = alloc_arr<int32_t>((int64_t)f.size());
    if (!*V || !*F) { std::free(*V); std::free(*F); *V = nullptr; *F = nullptr; return fail(VISMA_IO_ERR_INVALID, "out of memory"); }
    if (n_v) std::memcpy(*V, v.data(), sizeof(double) * v.size());
    if (!f.empty()) std::memcpy(*F, f.data(), sizeof(int32_t) * f.size());
    *nv = n_v;
    *nf = n_f;
    *face_size = fcols < 0 ? 0 : fcols;
    return VISMA_IO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- PCD
namespace {

struct PcdField {
    std::string name;
    int size = 4;
    char type = 'F';
    int count = 1;
    int count_offset = 0, offset = 0;
};

void split_ws(const std::string &line, std::vector<std::string> &out)
{
    out.clear();
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && std::strchr("\t\r\n ", line[i])) i++;
        size_t j = i;
        while (j < line.size() && !std::strchr("\t\r\n ", line[j])) j++;
        if (j > i) out.push_back(line.substr(i, j - i));
        i = j;
    }
}

// `sstream >> int` of the reference: the value stays what it was when the token is not a number
int stream_int(const std::vector<std::string> &st, size_t k, int keep)
{
    if (k >= st.size()) return keep;
    char *end = nullptr;
    const long v = std::strtol(st[k].c_str(), &end, 10);
    if (end == st[k].c_str()) return keep;
    return (int)v;
}

double pcd_unpack_bin(const unsigned char *p, char type, int size)   // FilePCD.cpp:234-282
{
    if (type == 'I') {
        if (size == 1) { int8_t v; std::memcpy(&v, p, 1); return (double)v; }
        if (size == 2) { int16_t v; std::memcpy(&v, p, 2); return (double)v; }
        if (size == 4) { int32_t v; std::memcpy(&v, p, 4); return (double)v; }
        return 0.0;
    }
    if (type == 'U') {
        if (size == 1) { uint8_t v; std::memcpy(&v, p, 1); return (double)v; }
        if (size == 2) { uint16_t v; std::memcpy(&v, p, 2); return (double)v; }
        if (size == 4) { uint32_t v; std::memcpy(&v, p, 4); return (double)v; }
        return 0.0;
    }
    if (type == 'F' && size == 4) { float v; std::memcpy(&v, p, 4); return (double)v; }
    return 0.0;                                                 // (8-byte floats read as 0, like the reference)
}

void pcd_color_from_bytes(const unsigned char d[4], double *rgb)    // packed BGR, FilePCD.cpp:284-297
{
    rgb[0] = (double)d[2] / 255.0;
    rgb[1] = (double)d[1] / 255.0;
    rgb[2] = (double)d[0] / 255.0;
}

double pcd_unpack_ascii(const char *s, char type)              // FilePCD.cpp:299-312
{
    char *end;
    if (type == 'I') return (double)std::strtol(s, &end, 0);
    if (type == 'U') return (double)std::strtoul(s, &end, 0);
    if (type == 'F') return std::strtod(s, &end);
    return 0.0;
}

void pcd_color_ascii(const char *s, char type, int size, double *rgb)   // FilePCD.cpp:314-336
{
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    if (size != 4) return;
    unsigned char d[4] = {0, 0, 0, 0};
    char *end;
    if (type == 'I') { const int32_t v = (int32_t)std::strtol(s, &end, 0); std::memcpy(d, &v, 4); }
    else if (type == 'U') { const uint32_t v = (uint32_t)std::strtoul(s, &end, 0); std::memcpy(d, &v, 4); }
    else if (type == 'F') { const float v = std::strtof(s, &end); std::memcpy(d, &v, 4); }
    pcd_color_from_bytes(d, rgb);
}

// LZF (Marc Lehmann's format, the one PCL writes): a control byte c < 32 starts a literal run of c + 1
// bytes; otherwise a back reference of length (c >> 5) + 2 (a length field of 7 takes one more byte)
// at distance ((c & 31) << 8 | next byte) + 1.  Returns the bytes produced, 0 on malformed input.
size_t lzf_decode(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len)
{
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned c = in[ip++];
        if (c < 32) {
            const size_t run = (size_t)c + 1;
            if (ip + run > in_len || op + run > out_len) return 0;
            std::memcpy(out + op, in + ip, run);
            ip += run;
            op += run;
        } else {
            size_t len = c >> 5;
            if (len == 7) {
                if (ip >= in_len) return 0;
                len += in[ip++];
            }
            if (ip >= in_len) return 0;
            const size_t dist = (((size_t)c & 31u) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_len) return 0;
            for (size_t k = 0; k < len; k++, op++) out[op] = out[op - dist];   // may overlap: byte by byte
        }
    }
    return op;
}

}  // namespace

extern "C" int visma_io_read_pcd(const char *path, visma_io_cloud *out)
{
    if (!path || !out) return fail(VISMA_IO_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    FileBytes f;
    if (!f.open(path)) return fail(VISMA_IO_ERR_OPEN, std::string("Read PCD failed: unable to open file: ") + path);
    const char *p = f.data, *end = f.data + f.size;

    // ---- header (FilePCD.cpp:132-232): line by line up to the DATA line
    std::vector<PcdField> fields;
    int width = 0, height = 0, points = 0, elementnum = 0, pointsize = 0, datatype = 0;
    std::vector<std::string> st;
    bool bad = false;
    while (p < end) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl + 1 : end;
        const std::string line(p, le);
        p = le;
        split_ws(line, st);
        if (st.empty()) continue;
        const std::string &t = st[0];
        const size_t nch = fields.size();
        if (t[0] == '#') {
        } else if (t.compare(0, 7, "VERSION") == 0) {
        } else if (t.compare(0, 6, "FIELDS") == 0 || t.compare(0, 7, "COLUMNS") == 0) {
            if (st.size() < 2) { bad = true; break; }
            fields.assign(st.size() - 1, PcdField());
            for (size_t i = 0; i < fields.size(); i++) {
                fields[i].name = st[i + 1];
                fields[i].count_offset = (int)i;
                fields[i].offset = (int)(4 * i);
            }
            elementnum = (int)fields.size();
            pointsize = (int)(4 * fields.size());
        } else if (t.compare(0, 4, "SIZE") == 0) {
            if (nch != st.size() - 1) { bad = true; break; }
            long long offset = 0;
            int v = 0;
            for (size_t i = 0; i < nch; i++) {
                v = stream_int(st, i + 1, v);
                if (v < 0 || v > 1024 || offset > (1ll << 30)) { bad = true; break; }
                fields[i].size = v;
                fields[i].offset = (int)offset;
                offset += v;
            }
            if (bad || offset > (1ll << 30)) { bad = true; break; }
            pointsize = (int)offset;
        } else if (t.compare(0, 4, "TYPE") == 0) {
            if (nch != st.size() - 1) { bad = true; break; }
            for (size_t i = 0; i < nch; i++) fields[i].type = st[i + 1][0];
        } else if (t.compare(0, 5, "COUNT") == 0) {
            if (nch != st.size() - 1) { bad = true; break; }
            long long co = 0, offset = 0;
            int v = 0;
            for (size_t i = 0; i < nch; i++) {
                v = stream_int(st, i + 1, v);
                if (v < 0 || v > (1 << 20) || co > (1ll << 30) || offset > (1ll << 30)) { bad = true; break; }
                fields[i].count = v;
                fields[i].count_offset = (int)co;
                fields[i].offset = (int)offset;
                co += v;
                offset += (long long)v * fields[i].size;
            }
            if (bad || co > (1ll << 30) || offset > (1ll << 30)) { bad = true; break; }
            elementnum = (int)co;
            pointsize = (int)offset;
        } else if (t.compare(0, 5, "WIDTH") == 0) {
            width = stream_int(st, 1, width);
        } else if (t.compare(0, 6, "HEIGHT") == 0) {
            height = stream_int(st, 1, height);
            {
                const long long wh = (long long)width * height;
                if (width < 0 || height < 0 || wh > 0x7fffffffll) { bad = true; break; }
                points = (int)wh;
            }
        } else if (t.compare(0, 9, "VIEWPOINT") == 0) {
        } else if (t.compare(0, 6, "POINTS") == 0) {
            points = stream_int(st, 1, points);
        } else if (t.compare(0, 4, "DATA") == 0) {
            datatype = 0;
            if (st.size() >= 2) {
                if (st[1].compare(0, 17, "binary_compressed") == 0) datatype = 2;
                else if (st[1].compare(0, 6, "binary") == 0) datatype = 1;
            }
            break;
        }
    }
    // CheckHeader (FilePCD.cpp:77-130)
    int fx = -1, fy = -1, fz = -1, fnx = -1, fny = -1, fnz = -1, fc = -1;
    for (size_t i = 0; i < fields.size(); i++) {
        const std::string &n = fields[i].name;
        if (n == "x") fx = (int)i; else if (n == "y") fy = (int)i; else if (n == "z") fz = (int)i;
        else if (n == "normal_x") fnx = (int)i; else if (n == "normal_y") fny = (int)i;
        else if (n == "normal_z") fnz = (int)i; else if (n == "rgb" || n == "rgba") fc = (int)i;
    }
    if (bad || points <= 0 || pointsize <= 0 || fields.empty() || fx < 0 || fy < 0 || fz < 0)
        return fail(VISMA_IO_ERR_FORMAT, "Read PCD failed: unable to parse header.");
    for (const auto &fd : fields)
        if (fd.size < 0 || fd.count < 0 || fd.offset < 0) return fail(VISMA_IO_ERR_FORMAT, "Read PCD failed: unable to parse header.");
    const bool has_n = fnx >= 0 && fny >= 0 && fnz >= 0, has_c = fc >= 0;
    const int64_t n = points;
    // A field that is READ must hold at least one element (COUNT 0 would make the ascii branch index one token
    // past the line, the binary ones read a neighbour's bytes) ...
    {
        const int read[7] = {fx, fy, fz, has_n ? fnx : -1, has_n ? fny : -1, has_n ? fnz : -1, fc};
        for (int k = 0; k < 7; k++)
            if (read[k] >= 0 && (fields[(size_t)read[k]].count < 1 || fields[(size_t)read[k]].size < 1))
                return fail(VISMA_IO_ERR_FORMAT, "Read PCD failed: unable to parse header.");
    }
    // ... and the header's point count is checked against what the file can hold BEFORE anything is allocated
    // (a 100-byte file announcing 2^31 points must not cost 3 x 48 GB of zeroed memory).
    {
        const uint64_t left = (uint64_t)(end - p);
        bool fits = true;
        if (datatype == 0) fits = (uint64_t)n <= left / 2 + 1;                   // >= 2 bytes per ascii record
        else if (datatype == 1) fits = (uint64_t)n <= left / (uint64_t)pointsize;
        else {
            uint32_t usz = 0;
            if (left >= 8) std::memcpy(&usz, p + 4, 4);
            fits = left >= 8 && (uint64_t)n * (uint64_t)pointsize <= (uint64_t)usz && (uint64_t)usz <= (1ull << 32) - 2;
        }
        if (!fits) return fail(VISMA_IO_ERR_FORMAT, "Read PCD failed: unable to read data.");
    }

    visma_io_cloud c;
    std::memset(&c, 0, sizeof(c));
    auto bail = [&](const std::string &m) { visma_io_free_cloud(&c); return fail(VISMA_IO_ERR_FORMAT, m); };
    c.xyz = alloc_arr<double>(3 * n);
    if (has_n) c.normals = alloc_arr<double>(3 * n);
    if (has_c) c.colors = alloc_arr<double>(3 * n);
    if (!c.xyz || (has_n && !c.normals) || (has_c && !c.colors)) return bail("out of memory");
    std::memset(c.xyz, 0, sizeof(double) * 3 * (size_t)n);
    if (has_n) std::memset(c.normals, 0, sizeof(double) * 3 * (size_t)n);
    if (has_c) std::memset(c.colors, 0, sizeof(double) * 3 * (size_t)n);
    // which destination a field feeds: 0..2 xyz, 3..5 normal, 6 colour, -1 none.  (A normal
    // component of a file without all three normals has nowhere to go.)
    std::vector<int> dest(fields.size(), -1);
    for (size_t i = 0; i < fields.size(); i++) {
        const std::string &nm = fields[i].name;
        if (nm == "x") dest[i] = 0; else if (nm == "y") dest[i] = 1; else if (nm == "z") dest[i] = 2;
        else if (has_n && nm == "normal_x") dest[i] = 3; else if (has_n && nm == "normal_y") dest[i] = 4;
        else if (has_n && nm == "normal_z") dest[i] = 5; else if (nm == "rgb" || nm == "rgba") dest[i] = 6;
    }
    if (datatype == 0) {
        // ascii (FilePCD.cpp:352-397): a line with fewer tokens than elements is skipped
        int64_t idx = 0;
        while (p < end && idx < n) {
            const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl + 1 : end;
            const std::string line(p, le);
            p = le;
            split_ws(line, st);
            if ((int)st.size() < elementnum) continue;
            for (size_t i = 0; i < fields.size(); i++) {
                if (dest[i] < 0) continue;
                const char *tok = st[(size_t)fields[i].count_offset].c_str();
                if (dest[i] < 3) c.xyz[3 * idx + dest[i]] = pcd_unpack_ascii(tok, fields[i].type);
                else if (dest[i] < 6) c.normals[3 * idx + dest[i] - 3] = pcd_unpack_ascii(tok, fields[i].type);
                else pcd_color_ascii(tok, fields[i].type, fields[i].size, c.colors + 3 * idx);
            }
            idx++;
        }
    } else if (datatype == 1) {
        // binary records (FilePCD.cpp:398-438)
        if ((int64_t)(end - p) / pointsize < n) return bail("Read PCD failed: unable to read data.");
        for (const auto &fd : fields)
            if ((int64_t)fd.offset + fd.size > pointsize) return bail("Read PCD failed: unable to parse header.");
        const unsigned char *base = (const unsigned char *)p;
        parallel_chunks(n, 65536, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                const unsigned char *r = base + (size_t)i * pointsize;
                for (size_t k = 0; k < fields.size(); k++) {
                    if (dest[k] < 0) continue;
                    const PcdField &fd = fields[k];
                    if (dest[k] < 3) c.xyz[3 * i + dest[k]] = pcd_unpack_bin(r + fd.offset, fd.type, fd.size);
                    else if (dest[k] < 6) c.normals[3 * i + dest[k] - 3] = pcd_unpack_bin(r + fd.offset, fd.type, fd.size);
                    else if (fd.size == 4) pcd_color_from_bytes(r + fd.offset, c.colors + 3 * i);
                }
            }
        });
    } else {
        // binary_compressed (FilePCD.cpp:439-509): two u32 sizes, one LZF block, fields stored one
        // after another (structure of arrays)
        if (end - p < 8) return bail("Read PCD failed: unable to read data.");
        uint32_t csize, usize;
        std::memcpy(&csize, p, 4);
        std::memcpy(&usize, p + 4, 4);
        p += 8;
        if ((uint64_t)(end - p) < csize) return bail("Read PCD failed: unable to read data.");
        std::vector<unsigned char> buf((size_t)usize + 1);
        if (lzf_decode((const unsigned char *)p, csize, buf.data(), usize) != usize)
            return bail("Read PCD failed: unable to read data.");
        for (size_t k = 0; k < fields.size(); k++) {
            if (dest[k] < 0) continue;
            const PcdField &fd = fields[k];
            const uint64_t first = (uint64_t)fd.offset * (uint64_t)n, stride = (uint64_t)fd.size * (uint64_t)fd.count;
            // (the reference does not check this; a short block would be read past its end there)
            if (first + stride * (uint64_t)(n - 1) + (uint64_t)fd.size > usize) return bail("Read PCD failed: unable to read data.");
            const unsigned char *b = buf.data() + first;
            for (int64_t i = 0; i < n; i++) {
                const unsigned char *e = b + (uint64_t)i * stride;
                if (dest[k] < 3) c.xyz[3 * i + dest[k]] = pcd_unpack_bin(e, fd.type, fd.size);
                else if (dest[k] < 6) c.normals[3 * i + dest[k] - 3] = pcd_unpack_bin(e, fd.type, fd.size);
                else if (fd.size == 4) pcd_color_from_bytes(e, c.colors + 3 * i);
            }
        }
    }
    // RemoveNanData (FilePCD.cpp:514-534): the reference tests x, y and x again -- a NaN z stays
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        if (std::isnan(c.xyz[3 * i]) || std::isnan(c.xyz[3 * i + 1])) continue;
        if (k != i) {
            std::memcpy(c.xyz + 3 * k, c.xyz + 3 * i, 3 * sizeof(double));
            if (has_n) std::memcpy(c.normals + 3 * k, c.normals + 3 * i, 3 * sizeof(double));
            if (has_c) std::memcpy(c.colors + 3 * k, c.colors + 3 * i, 3 * sizeof(double));
        }
        k++;
    }
    c.n = k;
    c.n_normals = has_n ? k : 0;
    c.n_colors = has_c ? k : 0;
    *out = c;
    return VISMA_IO_OK;
}

// ---------------------------------------------------------------- pose files (JSON)
namespace {

// A JSON reader for what the pose files hold: objects, arrays, strings, numbers, true / false / null.
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;      // file order
    const JVal *get(const std::string &k) const
    {
        for (const auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    int depth = 0;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool string(std::string &out)
    {
        if (p >= end || *p != '"') return fail("expected a string");
        p++;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': {
                    if (end - p < 5) return fail("bad \\u escape");
                    unsigned v = 0;
                    for (int i = 1; i <= 4; i++) {
                        const char ch = p[i];
                        v = v * 16 + (unsigned)(ch >= '0' && ch <= '9' ? ch - '0' : (ch | 32) >= 'a' && (ch | 32) <= 'f' ? (ch | 32) - 'a' + 10 : 0);
                    }
                    p += 4;
                    if (v < 0x80) out += (char)v;
                    else if (v < 0x800) { out += (char)(0xC0 | (v >> 6)); out += (char)(0x80 | (v & 63)); }
                    else { out += (char)(0xE0 | (v >> 12)); out += (char)(0x80 | ((v >> 6) & 63)); out += (char)(0x80 | (v & 63)); }
                    break;
                }
                default: out += *p;
                }
                p++;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool value(JVal &v)
    {
        if (++depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of file");
        bool ok = true;
        if (*p == '{') {
            v.kind = JVal::OBJ;
            p++;
            ws();
            if (p < end && *p == '}') { p++; }
            else for (;;) {
                ws();
                std::string k;
                JVal c;
                if (!string(k)) { ok = false; break; }
                ws();
                if (p >= end || *p != ':') { ok = fail("expected ':'"); break; }
                p++;
                if (!value(c)) { ok = false; break; }
                v.obj.emplace_back(std::move(k), std::move(c));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; break; }
                ok = fail("expected ',' or '}'");
                break;
            }
        } else if (*p == '[') {
            v.kind = JVal::ARR;
            p++;
            ws();
            if (p < end && *p == ']') { p++; }
            else for (;;) {
                JVal c;
                if (!value(c)) { ok = false; break; }
                v.arr.push_back(std::move(c));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; break; }
                ok = fail("expected ',' or ']'");
                break;
            }
        } else if (*p == '"') {
            v.kind = JVal::STR;
            ok = string(v.str);
        } else if (end - p >= 4 && std::strncmp(p, "true", 4) == 0) { v.kind = JVal::BOOL; v.b = true; p += 4; }
        else if (end - p >= 5 && std::strncmp(p, "false", 5) == 0) { v.kind = JVal::BOOL; v.b = false; p += 5; }
        else if (end - p >= 4 && std::strncmp(p, "null", 4) == 0) { v.kind = JVal::NUL; p += 4; }
        else {
            char *e = nullptr;
            const double d = std::strtod(p, &e);             // (the buffer is NUL-terminated)
            if (e == p) ok = fail("unexpected character");
            else { v.kind = JVal::NUM; v.num = d; p = e; }
        }
        --depth;
        return ok;
    }
};

bool json_load(const char *path, JVal &root, std::string &err)
{
    FileBytes f;
    if (!f.open(path)) { err = std::string("unable to open ") + path; return false; }
    JParser ps{f.data, f.data + f.size, "", 0};
    if (!ps.value(root)) { err = "JSON: " + ps.err; return false; }
    ps.ws();
    if (ps.p != ps.end) { err = "JSON: trailing characters"; return false; }
    return true;
}

// GetMatrixFromJson<double, 3, 4>(v, key): entry (i, j) = v[key][4 i + j].asDouble() (core/utils.h:305-322)
bool pose_from(const JVal *a, double T[12])
{
    if (!a || a->kind != JVal::ARR || a->arr.size() < 12) return false;
    for (int i = 0; i < 12; i++) {
        const JVal &e = a->arr[(size_t)i];
        if (e.kind == JVal::NUM) T[i] = e.num;
        else if (e.kind == JVal::BOOL) T[i] = e.b ? 1.0 : 0.0;       // jsoncpp's asDouble() of a bool
        else if (e.kind == JVal::NUL) T[i] = 0.0;                   // ... and of null
        else return false;
    }
    return true;
}

}  // namespace

extern "C" int visma_io_read_alignment_json(const char *path, visma_io_pose **poses, int64_t *n)
{
    if (!path || !poses || !n) return fail(VISMA_IO_ERR_INVALID, "null argument");
    *poses = nullptr;
    *n = 0;
    JVal root;
    std::string err;
    if (!json_load(path, root, err)) return fail(VISMA_IO_ERR_OPEN, err);
    if (root.kind != JVal::OBJ) return fail(VISMA_IO_ERR_FORMAT, "alignment file: the top level is not an object");
    // jsoncpp iterates the members of an object in key order (std::map): so do the callers' loops
    std::vector<const std::pair<std::string, JVal> *> items;
    for (const auto &kv : root.obj) items.push_back(&kv);
    std::stable_sort(items.begin(), items.end(), [](const std::pair<std::string, JVal> *a, const std::pair<std::string, JVal> *b) { return a->first < b->first; });
    visma_io_pose *o = (visma_io_pose *)std::calloc(items.size() ? items.size() : 1, sizeof(visma_io_pose));
    if (!o) return fail(VISMA_IO_ERR_FORMAT, "out of memory");
    for (size_t i = 0; i < items.size(); i++) {
        if (!pose_from(&items[i]->second, o[i].T)) { std::free(o); return fail(VISMA_IO_ERR_FORMAT, "alignment file: \"" + items[i]->first + "\" is not an array of 12 numbers"); }
        std::snprintf(o[i].name, sizeof(o[i].name), "%s", items[i]->first.c_str());
        o[i].id = -1;
        o[i].status = 0;
    }
    *poses = o;
    *n = (int64_t)items.size();
    return VISMA_IO_OK;
}

extern "C" int visma_io_write_alignment_json(const char *path, const visma_io_pose *poses, int64_t n)
{
    if (!path || n < 0 || (n > 0 && !poses)) return fail(VISMA_IO_ERR_INVALID, "bad arguments");
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(VISMA_IO_ERR_OPEN, std::string("unable to open ") + path);
    std::fputs("{\n", f);
    for (int64_t i = 0; i < n; i++) {
        std::fputs("  \"", f);
        for (const char *c = poses[i].name; *c; c++) {
            if (*c == '"' || *c == '\\') std::fputc('\\', f);
            std::fputc(*c, f);
        }
        std::fputs("\": [", f);
        for (int k = 0; k < 12; k++) std::fprintf(f, "%s%.17g", k ? ", " : "", poses[i].T[k]);   // WriteMatrixToJson: row by row
        std::fprintf(f, "]%s\n", i + 1 < n ? "," : "");
    }
    std::fputs("}\n", f);
    const bool ok = std::fclose(f) == 0;
    return ok ? VISMA_IO_OK : fail(VISMA_IO_ERR_OPEN, "write failed");
}

extern "C" int visma_io_read_result_json(const char *path, int64_t packet, visma_io_pose **poses, int64_t *n)
{
    if (!path || !poses || !n) return fail(VISMA_IO_ERR_INVALID, "null argument");
    *poses = nullptr;
    *n = 0;
    JVal root;
    std::string err;
    if (!json_load(path, root, err)) return fail(VISMA_IO_ERR_OPEN, err);
    if (root.kind != JVal::ARR || root.arr.empty()) return fail(VISMA_IO_ERR_FORMAT, "result file: the top level is not a non-empty array");
    const int64_t np = (int64_t)root.arr.size();
    if (packet < 0) packet = np + packet;                      // -1: the last packet (src/evaluation.cpp:169)
    if (packet < 0 || packet >= np) return fail(VISMA_IO_ERR_INVALID, "result file: no such packet");
    const JVal &pk = root.arr[(size_t)packet];
    if (pk.kind != JVal::ARR) return fail(VISMA_IO_ERR_FORMAT, "result file: a packet is not an array");
    visma_io_pose *o = (visma_io_pose *)std::calloc(pk.arr.size() ? pk.arr.size() : 1, sizeof(visma_io_pose));
    if (!o) return fail(VISMA_IO_ERR_FORMAT, "out of memory");
    for (size_t i = 0; i < pk.arr.size(); i++) {
        const JVal &ob = pk.arr[i];
        const JVal *nm = ob.kind == JVal::OBJ ? ob.get("model_name") : nullptr;
        if (ob.kind != JVal::OBJ || !pose_from(ob.get("model_pose"), o[i].T)) { std::free(o); return fail(VISMA_IO_ERR_FORMAT, "result file: object without a 12-number model_pose"); }
        std::snprintf(o[i].name, sizeof(o[i].name), "%s", nm && nm->kind == JVal::STR ? nm->str.c_str() : "");
        const JVal *id = ob.get("id"), *stt = ob.get("status");
        o[i].id = id && id->kind == JVal::NUM ? (int)id->num : 0;
        o[i].status = stt && stt->kind == JVal::NUM ? (int)stt->num : 0;
    }
    *poses = o;
    *n = (int64_t)pk.arr.size();
    return VISMA_IO_OK;
}
