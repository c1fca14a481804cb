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
