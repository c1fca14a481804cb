// driver.cpp -- the host driver's C ABI: contexts, cloud upload, passes, the RegistrationICP loop and its
// batched forms, options, multi-GPU bring-up.  (Engine / HipEngine: engine.hpp, hip_engine.cpp; the steps either
// side of ICP: aux_api.cpp; the corpus work queue: corpus.cpp.)
#include "driver_ctx.hpp"
#ifdef VISMA_TEST_SEAMS
#include "../../include/visma_icp_testing.h"
#endif

namespace {

#ifdef VISMA_TEST_SEAMS
class HookEngine : public Engine {
public:
    HookEngine(const visma_icp_engine &vt, void *user) : vt_(vt), user_(user) {}
    int set_source(const float *p, int64_t n) override { ns_ = n; return wrap(vt_.set_source(user_, p, n)); }
    int set_target(const float *p, int64_t n) override { nt_ = n; has_normals_ = false; return wrap(vt_.set_target(user_, p, n)); }
    int set_target_normals(const float *p, int64_t n) override
    {
        if (!vt_.set_target_normals) { err_ = "engine has no normals support"; return VISMA_ICP_ERR_ENGINE; }
        has_normals_ = true;
        return wrap(vt_.set_target_normals(user_, p, n));
    }
    int nn_pass(const Mat4 &Tc, double r) override { return wrap(vt_.nn_pass(user_, Tc.m, r)); }
    int reduce(const Mat4 &Tc, bool plane, const double offset[3], double *st) override
    {
        const double zero[3] = {0, 0, 0};
        return wrap(vt_.reduce(user_, Tc.m, offset ? offset : zero, plane ? 1 : 0, st));
    }
    int get_correspondences(int32_t *idx, float *d2) override
    {
        std::vector<float> tmp;
        if (!d2) { tmp.resize((size_t)(ns_ > 0 ? ns_ : 1)); d2 = tmp.data(); }
        return wrap(vt_.get_correspondences(user_, idx, d2));
    }

private:
    int wrap(int rc)
    {
        if (rc == 0) return VISMA_ICP_OK;
        err_ = "engine callback returned " + std::to_string(rc);
        return VISMA_ICP_ERR_ENGINE;
    }
    visma_icp_engine vt_;
    void *user_;
};
#endif  // VISMA_TEST_SEAMS


}  // namespace

namespace {

// (x - c) as fp32 (x,y,z,0) rows; `par`: spread over host threads
void pack_f64_to(const double *xyz, int64_t n, int stride, const double c[3], float *out, bool par)
{
    auto body = [&](int64_t ch) {
        const int64_t lo = ch * kHostChunk, hi = std::min(n, lo + kHostChunk);
        for (int64_t i = lo; i < hi; i++) {
            const double *p = xyz + (size_t)i * stride;
            out[4 * i + 0] = (float)(p[0] - c[0]);
            out[4 * i + 1] = (float)(p[1] - c[1]);
            out[4 * i + 2] = (float)(p[2] - c[2]);
            out[4 * i + 3] = 0.f;
        }
    };
    const int64_t nch = (n + kHostChunk - 1) / kHostChunk;
    if (par) parallel_for(nch, 1, body);
    else for (int64_t ch = 0; ch < nch; ch++) body(ch);
}

int pack_f64(const double *xyz, int64_t n, int stride, const double c[3], std::vector<float> &out)
{
    out.resize((size_t)(n > 0 ? n : 1) * 4);
    pack_f64_to(xyz, n, stride, c, out.data(), false);
    return 0;
}

void pack_f32_to(const float *xyz, int64_t n, int stride, float *out)
{
    for (int64_t i = 0; i < n; i++) {
        const float *p = xyz + (size_t)i * stride;
        out[4 * i + 0] = p[0];
        out[4 * i + 1] = p[1];
        out[4 * i + 2] = p[2];
        out[4 * i + 3] = 0.f;
    }
}

// Centroid in f64.  Fixed chunks summed in index order and combined in chunk order: the
// value does not depend on the number of threads (every rank of a multi-GPU run, and every
// repeat, gets the same centre).
void centroid_f64(const double *xyz, int64_t n, int stride, double c[3], bool par)
{
    c[0] = c[1] = c[2] = 0.0;
    if (n <= 0) return;
    const int64_t nch = (n + kHostChunk - 1) / kHostChunk;
    std::vector<double> part((size_t)nch * 3, 0.0);
    auto body = [&](int64_t ch) {
        const int64_t lo = ch * kHostChunk, hi = std::min(n, lo + kHostChunk);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int64_t i = lo; i < hi; i++) {
            const double *p = xyz + (size_t)i * stride;
            s0 += p[0]; s1 += p[1]; s2 += p[2];
        }
        part[3 * ch] = s0; part[3 * ch + 1] = s1; part[3 * ch + 2] = s2;
    };
    if (par) parallel_for(nch, 1, body);
    else for (int64_t ch = 0; ch < nch; ch++) body(ch);
    for (int64_t ch = 0; ch < nch; ch++)
        for (int a = 0; a < 3; a++) c[a] += part[3 * ch + a];
    for (int a = 0; a < 3; a++) c[a] /= (double)n;
}

// Spatial (Morton / Z-order) permutation of packed xyzw points.  Neighbouring
// lanes then query neighbouring grid cells, so a wave's candidate runs overlap
// in L1/L2.  Returns order[pos] = original index and permutes `pts` in place.
inline uint32_t spread10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

void morton_order_ptr(float *pts, int64_t n, std::vector<int32_t> &order, bool par)
{
    order.resize((size_t)n);
    if (n <= 0) return;
    float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float v = pts[4 * i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    float ext = 0.f;
    for (int a = 0; a < 3; a++) ext = std::max(ext, mx[a] - mn[a]);
    const float scale = (ext > 0.f && std::isfinite(ext)) ? 1023.0f / ext : 0.f;
    std::vector<uint32_t> key((size_t)n), key2((size_t)n);
    std::vector<int32_t> idx2((size_t)n);
    const int64_t nch = (n + kHostChunk - 1) / kHostChunk;
    auto keys_body = [&](int64_t ch) {
        const int64_t lo = ch * kHostChunk, hi = std::min(n, lo + kHostChunk);
        for (int64_t i = lo; i < hi; i++) {
            uint32_t q[3];
            for (int a = 0; a < 3; a++) {
                float u = (pts[4 * i + a] - mn[a]) * scale;
                if (!(u >= 0.f)) u = 0.f;
                if (u > 1023.f) u = 1023.f;
                q[a] = (uint32_t)u;
            }
            key[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
            order[i] = (int32_t)i;
        }
    };
    if (par) parallel_for(nch, 1, keys_body);
    else for (int64_t ch = 0; ch < nch; ch++) keys_body(ch);
    // stable LSD radix sort, 3 passes of 10 bits: ties keep the original index order
    uint32_t *ka = key.data(), *kb = key2.data();
    int32_t *ia = order.data(), *ib = idx2.data();
    for (int pass = 0; pass < 3; pass++) {
        const int sh = 10 * pass;
        uint32_t hist[1025];
        std::memset(hist, 0, sizeof(hist));
        for (int64_t i = 0; i < n; i++) hist[((ka[i] >> sh) & 1023u) + 1]++;
        for (int b = 0; b < 1024; b++) hist[b + 1] += hist[b];
        for (int64_t i = 0; i < n; i++) {
            const uint32_t d = (ka[i] >> sh) & 1023u;
            const uint32_t pos = hist[d]++;
            kb[pos] = ka[i];
            ib[pos] = ia[i];
        }
        std::swap(ka, kb);
        std::swap(ia, ib);
    }
    if (ia != order.data()) std::memcpy(order.data(), ia, (size_t)n * sizeof(int32_t));
    std::vector<float> out((size_t)n * 4);
    auto gather = [&](int64_t ch) {
        const int64_t lo = ch * kHostChunk, hi = std::min(n, lo + kHostChunk);
        for (int64_t pos = lo; pos < hi; pos++)
            std::memcpy(&out[4 * pos], &pts[4 * (size_t)order[pos]], 4 * sizeof(float));
    };
    if (par) parallel_for(nch, 1, gather);
    else for (int64_t ch = 0; ch < nch; ch++) gather(ch);
    std::memcpy(pts, out.data(), out.size() * sizeof(float));
}

void morton_order(std::vector<float> &pts, int64_t n, std::vector<int32_t> &order)
{
    morton_order_ptr(pts.data(), n, order, false);
}

}  // namespace

extern "C" {

const char *visma_icp_version(void) { return "visma-icp-mi355x 0.1 (gfx950)"; }

int visma_icp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int visma_icp_create(visma_icp_ctx **out, int device)
{
    if (!out) { g_create_error = "out is NULL"; return VISMA_ICP_ERR_INVALID; }
    *out = nullptr;
    int rc = VISMA_ICP_OK;
    std::string emsg;
    std::unique_ptr<Engine> e(new_hip_engine(device, &rc, &emsg));
    if (!e) { g_create_error = emsg; return rc ? rc : VISMA_ICP_ERR_NO_DEVICE; }
    visma_icp_ctx *c = new visma_icp_ctx();
    c->eng = std::move(e);
    if (const char *p = std::getenv("VISMA_ICP_SEARCH_PRECISION")) {     // initial value of the option
        const int v = std::atoi(p);
        if (v >= 0 && v <= 2) c->search_precision = v;
    }
    c->eng->set_exact(c->search_precision == 1);
    *out = c;
    return VISMA_ICP_OK;
}

#ifdef VISMA_TEST_SEAMS
int visma_icp_create_with_engine(visma_icp_ctx **out, const visma_icp_engine *engine, void *user)
{
    if (!out || !engine || !engine->set_source || !engine->set_target || !engine->nn_pass ||
        !engine->reduce || !engine->get_correspondences) {
        g_create_error = "incomplete engine table";
        return VISMA_ICP_ERR_INVALID;
    }
    visma_icp_ctx *c = new visma_icp_ctx();
    c->eng.reset(new HookEngine(*engine, user));
    *out = c;
    return VISMA_ICP_OK;
}
#endif  // VISMA_TEST_SEAMS

int visma_icp_destroy(visma_icp_ctx *ctx)
{
    delete ctx;
    return VISMA_ICP_OK;
}

const char *visma_icp_last_error(const visma_icp_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}


struct MeshSourceSpec {                  // the source comes from meshes sampled on the device (src == NULL then)
    const Engine::MeshSource *meshes;
    int n;
    int quirks;
    unsigned long long seed;
    int64_t *ns_out;
};

static int set_clouds_f64_impl(visma_icp_ctx *ctx, const double *src, int64_t ns, int sstride, const double *tgt,
                               int64_t nt, int tstride, double voxel, int64_t *nt_out, const MeshSourceSpec *ms = nullptr);

int visma_icp_set_clouds_f64(visma_icp_ctx *ctx, const double *src, int64_t ns, int sstride,
                             const double *tgt, int64_t nt, int tstride)
{
    return set_clouds_f64_impl(ctx, src, ns, sstride, tgt, nt, tstride, 0.0, nullptr);
}

int visma_icp_set_clouds_f64_voxel_target(visma_icp_ctx *ctx, const double *src, int64_t ns, int sstride,
                                          const double *scene, int64_t n_scene, int tstride, double voxel_size,
                                          int64_t *nt_out)
{
    if (ctx && (!nt_out || !(voxel_size > 0.0))) return ctx->fail(VISMA_ICP_ERR_INVALID, "voxel_size must be positive, nt_out not NULL");
    return set_clouds_f64_impl(ctx, src, ns, sstride, scene, n_scene, tstride, voxel_size, nt_out);
}

int visma_icp_set_clouds_meshes_f64(visma_icp_ctx *ctx, const visma_icp_mesh_source *meshes, int n_meshes,
                                    int reference_quirks, uint64_t seed, const double *scene_xyz, int64_t n_scene,
                                    int scene_stride, double voxel_size, int64_t *ns_out, int64_t *nt_out)
{
    CTX_CHECK();
    if (n_meshes < 0 || (n_meshes > 0 && !meshes) || !ns_out || !nt_out || !(voxel_size >= 0.0))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad mesh-source arguments");
    *ns_out = 0;
    *nt_out = 0;
    std::vector<Engine::MeshSource> ms((size_t)n_meshes);
    for (int k = 0; k < n_meshes; k++) {
        ms[(size_t)k].V = meshes[k].V; ms[(size_t)k].nv = meshes[k].nv;
        ms[(size_t)k].F = meshes[k].F; ms[(size_t)k].nf = meshes[k].nf;
        ms[(size_t)k].samples = meshes[k].samples;
        ms[(size_t)k].has_transform = meshes[k].model_to_scene != nullptr;
        if (meshes[k].model_to_scene) std::memcpy(ms[(size_t)k].T, meshes[k].model_to_scene, sizeof(double) * 16);
    }
    MeshSourceSpec spec{ms.data(), n_meshes, reference_quirks, (unsigned long long)seed, ns_out};
    int64_t nt = n_scene;
    int rc = set_clouds_f64_impl(ctx, nullptr, 0, 3, scene_xyz, n_scene, scene_stride, voxel_size, &nt, &spec);
    if (rc) return rc;
    *nt_out = voxel_size > 0.0 ? nt : n_scene;
    return VISMA_ICP_OK;
}

int visma_icp_set_radius_hint(visma_icp_ctx *ctx, double max_correspondence_distance)
{
    CTX_CHECK();
    if (!(max_correspondence_distance >= 0.0)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad radius");
    ctx->radius_hint = max_correspondence_distance;
    return VISMA_ICP_OK;
}

int visma_icp_get_mesh_source(visma_icp_ctx *ctx, double *xyz_out, int64_t ns)
{
    CTX_CHECK();
    if (ns < 0 || (ns > 0 && !xyz_out)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad arguments");
    int rc = ctx->eng->get_mesh_source(xyz_out, ns);
    if (rc) return ctx->eng_fail(rc);
    return VISMA_ICP_OK;
}

int visma_icp_get_voxel_target(visma_icp_ctx *ctx, double *xyz_out, int64_t nt)
{
    CTX_CHECK();
    if (nt < 0 || (nt > 0 && !xyz_out)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad arguments");
    int rc = ctx->eng->get_voxel_target(xyz_out, nt);
    if (rc) return ctx->eng_fail(rc);
    return VISMA_ICP_OK;
}

// voxel > 0: the target is VoxelDownSample(tgt, voxel), made and installed on the device (*nt_out = its size)
static int set_clouds_f64_impl(visma_icp_ctx *ctx, const double *src, int64_t ns, int sstride, const double *tgt,
                               int64_t nt, int tstride, double voxel, int64_t *nt_out, const MeshSourceSpec *ms)
{
    CTX_CHECK();
    if (ns < 0 || nt < 0 || sstride < 3 || tstride < 3 || (ns > 0 && !src) || (nt > 0 && !tgt))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad cloud arguments");
    // centre on the target centroid: sequential f64 sum in index order
    // centre on the target centroid (fixed chunked f64 sum: thread-count independent)
    auto t_now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    static const bool trace = std::getenv("VISMA_ICP_UPLOAD_TRACE") != nullptr;
    double tm[8]; int ti = 0; tm[ti++] = t_now();
    double c[3] = {0, 0, 0};
    // (the HIP engine sums the centroid while it stages the target: one pass over the caller's memory instead
    //  of two; engines without a raw upload get it from centroid_f64 -- the same chunk sums, the same value)
    if (ctx->fixed_centre) std::memcpy(c, ctx->centre, sizeof(c));
    tm[ti++] = t_now();   // (centroid: inside the target upload)
    const bool want64 = ctx->search_precision != 0;     // (target-sharded ranks too: they compare shards in f64)
    // the target goes up as the caller's f64 values and is expanded on the device (HIP engine); otherwise it
    // is packed on a few host threads straight into the engine's (pinned) staging memory
    static const int64_t raw_min = []() {                 // VISMA_ICP_RAW_UPLOAD_MIN: smallest target sent up raw
        const char *e = std::getenv("VISMA_ICP_RAW_UPLOAD_MIN");
        return e ? (int64_t)std::atoll(e) : (int64_t)0;
    }();
    int rc;
    if (voxel > 0.0) {
        rc = ctx->eng->set_target_voxel_f64(tgt, nt, tstride, voxel, c, !ctx->fixed_centre, want64 && ctx->eng->supports_device_loop(), nt_out);
        if (rc == VISMA_ICP_ERR_STATE) return ctx->fail(VISMA_ICP_ERR_STATE, "the voxel-grid target needs the HIP engine");
        if (rc) return ctx->eng_fail(rc);
        nt = *nt_out;                                            // (what follows only needs the count)
    } else {
        rc = nt >= raw_min ? ctx->eng->set_target_f64(tgt, nt, tstride, c, !ctx->fixed_centre, want64 && ctx->eng->supports_device_loop())
                           : (int)VISMA_ICP_ERR_STATE;
    }
    const bool raw_target = rc == VISMA_ICP_OK;
    if (!raw_target) {
        if (rc != VISMA_ICP_ERR_STATE) return ctx->eng_fail(rc);
        if (!ctx->fixed_centre) centroid_f64(tgt, nt, tstride, c, true);
        float *tb = ctx->eng->staging(0, (size_t)std::max<int64_t>(nt, 1) * 4);
        pack_f64_to(tgt, nt, tstride, c, tb, true);
        rc = ctx->eng->set_target(tb, nt);
        if (rc) return ctx->eng_fail(rc);
    }
    tm[ti++] = t_now();   // target
    // the source too goes up raw and is Morton-ordered on the device (HIP engine with a raw target)
    bool raw_source = false;
    if (ms) {
        if (!raw_target) return ctx->fail(VISMA_ICP_ERR_STATE, "a mesh-sampled source needs the HIP engine");
        rc = ctx->eng->set_source_meshes_f64(ms->meshes, ms->n, ms->quirks, ms->seed, c, want64 && ctx->eng->supports_device_loop(),
                                             ctx->src_order, ms->ns_out);
        if (rc == VISMA_ICP_ERR_STATE) return ctx->fail(VISMA_ICP_ERR_STATE, "a mesh-sampled source needs the HIP engine");
        if (rc) return ctx->eng_fail(rc);
        raw_source = true;
        ns = *ms->ns_out;
    } else if (raw_target) {
        // the radius of the coming registration, if known (visma_icp_set_radius_hint, or the last one used): the grid
        // is built on the stream while the source is staged
        // (the hint speaks about the NEXT registration only: consumed here, so that a context that once got one and is
        //  later driven with another radius falls back to its previous radius instead of building a wasted grid per upload)
        const double hint = ctx->radius_hint > 0.0 ? ctx->radius_hint : ctx->last_radius;
        ctx->radius_hint = 0.0;
        if (hint > 0.0 && ctx->search_precision == 1) {
            ctx->eng->set_exact(true);
            rc = ctx->eng->prepare_search(ns, want64 && ctx->eng->supports_device_loop(), hint);
            if (rc) return ctx->eng_fail(rc);
        }
        rc = ctx->eng->set_source_f64(src, ns, sstride, c, want64 && ctx->eng->supports_device_loop(), ctx->src_order);
        raw_source = rc == VISMA_ICP_OK;
        if (!raw_source && rc != VISMA_ICP_ERR_STATE) return ctx->eng_fail(rc);
    }
    float *sb = nullptr;
    if (!raw_source) {
        sb = ctx->eng->staging(1, (size_t)std::max<int64_t>(ns, 1) * 4);
        pack_f64_to(src, ns, sstride, c, sb, true);
    }
    tm[ti++] = t_now();   // source pack
    if (!raw_source) morton_order_ptr(sb, ns, ctx->src_order, true);
    tm[ti++] = t_now();   // morton
    if (!raw_source) {
        rc = ctx->eng->set_source(sb, ns);
        if (rc) return ctx->eng_fail(rc);
    }
    // double-precision search: the caller's own f64 coordinates (centred in f64) go along
    // (the size-keyed policy of round 1 is gone: the exact search costs the same as the fp32 one)
    ctx->eng->set_exact(ctx->search_precision == 1);
    if (raw_source) {
        if (!(want64 && ctx->eng->supports_device_loop())) {
            rc = ctx->eng->set_clouds64(nullptr, nullptr);
            if (rc) return ctx->eng_fail(rc);
        }
    } else if (want64 && ctx->eng->supports_device_loop()) {
        // (Pt64 = 8 floats of staging; pinned on the HIP engine)
        Pt64 *t8 = raw_target ? nullptr : reinterpret_cast<Pt64 *>(ctx->eng->staging(2, (size_t)std::max<int64_t>(nt, 1) * 8));
        Pt64 *s8 = reinterpret_cast<Pt64 *>(ctx->eng->staging(3, (size_t)std::max<int64_t>(ns, 1) * 8));
        if (!raw_target) parallel_for((nt + kHostChunk - 1) / kHostChunk, 1, [&](int64_t ch) {
            const int64_t lo = ch * kHostChunk, hi = std::min(nt, lo + kHostChunk);
            for (int64_t j = lo; j < hi; j++) {
                const double *q = tgt + (size_t)j * tstride;
                t8[(size_t)j] = Pt64{q[0] - c[0], q[1] - c[1], q[2] - c[2], (unsigned long long)j};
            }
        });
        parallel_for((ns + kHostChunk - 1) / kHostChunk, 1, [&](int64_t ch) {
            const int64_t lo = ch * kHostChunk, hi = std::min(ns, lo + kHostChunk);
            for (int64_t pos = lo; pos < hi; pos++) {
                const double *q = src + (size_t)ctx->src_order[(size_t)pos] * sstride;    // Morton order, like sb
                s8[(size_t)pos] = Pt64{q[0] - c[0], q[1] - c[1], q[2] - c[2], (unsigned long long)ctx->src_order[(size_t)pos]};
            }
        });
        rc = raw_target ? ctx->eng->set_source64(s8) : ctx->eng->set_clouds64(s8, t8);
        if (rc) return ctx->eng_fail(rc);
    } else if (ctx->eng->supports_device_loop()) {
        rc = ctx->eng->set_clouds64(nullptr, nullptr);
        if (rc) return ctx->eng_fail(rc);
    }
    tm[ti++] = t_now();   // source upload + f64
    if (trace) std::fprintf(stderr, "upload trace: centroid %.2f target %.2f src-pack %.2f morton %.2f src-upload+f64 %.2f ms\n",
                            tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[5] - tm[4]);
    std::memcpy(ctx->centre, c, sizeof(c));
    ctx->have_src = ctx->have_tgt = true;
    ctx->centred_upload = true;
    return VISMA_ICP_OK;
}

int visma_icp_set_target(visma_icp_ctx *ctx, const float *xyz, int64_t nt, int stride)
{
    CTX_CHECK();
    ctx->eng->set_exact(ctx->search_precision == 1);     // (the search precision applies from an upload on)
    if (nt < 0 || stride < 3 || (nt > 0 && !xyz)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad target arguments");
    float *buf = ctx->eng->staging(0, (size_t)std::max<int64_t>(nt, 1) * 4);
    pack_f32_to(xyz, nt, stride, buf);
    int rc = ctx->eng->set_target(buf, nt);
    if (rc) return ctx->eng_fail(rc);
    ctx->enter_uncentred_frame(false);
    ctx->have_tgt = true;
    return VISMA_ICP_OK;
}

int visma_icp_set_source(visma_icp_ctx *ctx, const float *xyz, int64_t ns, int stride)
{
    CTX_CHECK();
    ctx->eng->set_exact(ctx->search_precision == 1);     // (the search precision applies from an upload on)
    if (ns < 0 || stride < 3 || (ns > 0 && !xyz)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad source arguments");
    float *buf = ctx->eng->staging(1, (size_t)std::max<int64_t>(ns, 1) * 4);
    pack_f32_to(xyz, ns, stride, buf);
    morton_order_ptr(buf, ns, ctx->src_order, true);
    int rc = ctx->eng->set_source(buf, ns);
    if (rc) return ctx->eng_fail(rc);
    ctx->enter_uncentred_frame(true);
    ctx->have_src = true;
    return VISMA_ICP_OK;
}

int visma_icp_set_target_device(visma_icp_ctx *ctx, const void *d, int64_t nt)
{
    CTX_CHECK();
    ctx->eng->set_exact(ctx->search_precision == 1);     // (the search precision applies from an upload on)
    if (nt < 0 || (nt > 0 && !d)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad target arguments");
    int rc = ctx->eng->set_target_device(d, nt);
    if (rc) return ctx->eng_fail(rc);
    ctx->enter_uncentred_frame(false);
    ctx->have_tgt = true;
    return VISMA_ICP_OK;
}

int visma_icp_set_source_device(visma_icp_ctx *ctx, const void *d, int64_t ns)
{
    CTX_CHECK();
    ctx->eng->set_exact(ctx->search_precision == 1);     // (the search precision applies from an upload on)
    if (ns < 0 || (ns > 0 && !d)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad source arguments");
    int rc = ctx->eng->set_source_device(d, ns);
    if (rc) return ctx->eng_fail(rc);
    ctx->src_order.clear();   // device-resident source is used in the caller's order
    ctx->enter_uncentred_frame(true);
    ctx->have_src = true;
    return VISMA_ICP_OK;
}

int visma_icp_set_target_normals_f64(visma_icp_ctx *ctx, const double *n, int64_t nt, int stride)
{
    CTX_CHECK();
    if (!ctx->have_tgt) return ctx->fail(VISMA_ICP_ERR_STATE, "set the target first");
    if (nt != ctx->eng->nt() || stride < 3 || (nt > 0 && !n))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad normals arguments");
    const double zero[3] = {0, 0, 0};
    float *buf = ctx->eng->staging(0, (size_t)std::max<int64_t>(nt, 1) * 4);
    pack_f64_to(n, nt, stride, zero, buf, true);
    int rc = ctx->eng->set_target_normals(buf, nt);
    if (rc) return ctx->eng_fail(rc);
    if (ctx->eng->supports_device_loop()) {                 // the f64 search takes the normals in f64 as well
        std::vector<Pt64> n8((size_t)std::max<int64_t>(nt, 1));
        for (int64_t j = 0; j < nt; j++)
            n8[(size_t)j] = Pt64{n[(size_t)j * stride], n[(size_t)j * stride + 1], n[(size_t)j * stride + 2], 0ull};
        rc = ctx->eng->set_target_normals64(n8.data());
        if (rc) return ctx->eng_fail(rc);
    }
    return VISMA_ICP_OK;
}

int visma_icp_nn_pass(visma_icp_ctx *ctx, const double T[16], double max_dist)
{
    CTX_CHECK();
    if (!T) return ctx->fail(VISMA_ICP_ERR_INVALID, "T is NULL");
    if (!ctx->have_src || !ctx->have_tgt) return ctx->fail(VISMA_ICP_ERR_STATE, "clouds not set");
    if (!(max_dist > 0.0)) return ctx->fail(VISMA_ICP_ERR_INVALID, "max_dist must be > 0");
    ctx->last_Tc = to_centred(Mat4::from(T), ctx->centre);
    int rc = ctx->eng->nn_pass(ctx->last_Tc, max_dist);
    if (rc) return ctx->eng_fail(rc);
    return VISMA_ICP_OK;
}

int visma_icp_reduce(visma_icp_ctx *ctx, double out_stats[VISMA_ICP_NSTATS])
{
    CTX_CHECK();
    if (!out_stats) return ctx->fail(VISMA_ICP_ERR_INVALID, "out_stats is NULL");
    const double zero[3] = {0, 0, 0};
    int rc = ctx->eng->reduce(ctx->last_Tc, false, zero, out_stats);
    if (rc) return ctx->eng_fail(rc);
    if (ctx->host_allreduce && !ctx->eng->has_device_allreduce())
        if (ctx->host_allreduce(ctx->host_allreduce_user, out_stats, VISMA_ICP_NSTATS) != 0)
            return ctx->fail(VISMA_ICP_ERR_ENGINE, "host all-reduce callback failed");
    return VISMA_ICP_OK;
}

int visma_icp_get_correspondences(visma_icp_ctx *ctx, int32_t *src_idx, int32_t *tgt_idx, float *d2,
                                  int64_t *k)
{
    CTX_CHECK();
    if (!src_idx || !tgt_idx || !k) return ctx->fail(VISMA_ICP_ERR_INVALID, "NULL output buffer");
    const int64_t ns = ctx->eng->ns();
    std::vector<int32_t> idx((size_t)(ns > 0 ? ns : 1));
    std::vector<float> dd((size_t)(ns > 0 ? ns : 1));
    int rc = ctx->eng->get_correspondences(idx.data(), dd.data());
    if (rc) return ctx->eng_fail(rc);
    if ((int64_t)ctx->src_order.size() == ns && ns > 0) {
        // the engine holds the source in Morton order: back to the caller's order
        std::vector<int32_t> idx2((size_t)ns);
        std::vector<float> dd2((size_t)ns);
        for (int64_t pos = 0; pos < ns; pos++) {
            idx2[ctx->src_order[pos]] = idx[pos];
            dd2[ctx->src_order[pos]] = dd[pos];
        }
        idx.swap(idx2);
        dd.swap(dd2);
    }
    int64_t c = 0;
    for (int64_t i = 0; i < ns; i++)
        if (idx[i] >= 0) {
            src_idx[c] = (int32_t)i;
            tgt_idx[c] = idx[i];
            if (d2) d2[c] = dd[i];
            ++c;
        }
    *k = c;
    return VISMA_ICP_OK;
}

int visma_icp_solve_from_stats(const double stats[VISMA_ICP_NSTATS], int solver, int with_scaling,
                               double T_update[16])
{
    if (!stats || !T_update) return VISMA_ICP_ERR_INVALID;
    Mat4 T;
    bool ok = true;
    switch (solver) {
    case VISMA_ICP_SOLVER_KABSCH: T = kabsch_from_stats(stats, with_scaling != 0); break;
    case VISMA_ICP_SOLVER_GN_EULER: T = gn_from_stats(stats, false, &ok); break;
    case VISMA_ICP_SOLVER_GN_EXPMAP: T = gn_from_stats(stats, true, &ok); break;
    default: return VISMA_ICP_ERR_INVALID;
    }
    std::memcpy(T_update, T.m, sizeof(T.m));
    return VISMA_ICP_OK;
}

int visma_icp_run(visma_icp_ctx *ctx, const double init[16], double max_dist, int max_iter,
                  double rel_fitness, double rel_rmse, int solver, int with_scaling,
                  visma_icp_result *out)
{
    CTX_CHECK();
    if (!init || !out || max_iter < 0) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad run arguments");
    if (solver < 0 || solver > VISMA_ICP_SOLVER_GN_EXPMAP) return ctx->fail(VISMA_ICP_ERR_INVALID, "unknown solver");
    return ctx->run(init, max_dist, max_iter, rel_fitness, rel_rmse, solver, with_scaling != 0, false, out);
}

int visma_icp_iterate(visma_icp_ctx *ctx, double T_inout[16], double max_dist, int steps, int solver,
                      int with_scaling, visma_icp_result *out)
{
    CTX_CHECK();
    if (!T_inout || steps < 0 || !(max_dist > 0.0)) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad iterate arguments");
    if (solver < 0 || solver > VISMA_ICP_SOLVER_GN_EXPMAP) return ctx->fail(VISMA_ICP_ERR_INVALID, "unknown solver");
    if (!ctx->have_src || !ctx->have_tgt) return ctx->fail(VISMA_ICP_ERR_STATE, "clouds not set");
    Mat4 Tc = to_centred(Mat4::from(T_inout), ctx->centre);
    double stats[VISMA_ICP_NSTATS], fit = 0, rmse = 0;
    int64_t k = 0;
    const bool world = visma_icp_ctx::wants_world_frame(solver, false);
    if (ctx->use_device_loop() && steps > 0) {
        Engine::LoopParams lp;
        lp.Tc0 = Tc;
        std::memcpy(lp.centre, ctx->centre, sizeof(ctx->centre));
        lp.max_dist = max_dist; lp.rel_fit = 0.0; lp.rel_rmse = 0.0;
        lp.max_iter = steps; lp.solver = solver; lp.passes = steps;
        lp.scaling = with_scaling != 0; lp.plane = false; lp.world = world; lp.check_stop = false;
        lp.ns_total = ctx->ns_total > 0 ? ctx->ns_total : ctx->eng->ns();
        Engine::LoopResult r;
        int rc = ctx->eng->run_loop(lp, nullptr, 1, &r);
        if (rc) return ctx->eng_fail(rc);
        ctx->last_Tc = r.Tc;
        const Mat4 T = from_centred(r.Tc, ctx->centre);
        std::memcpy(T_inout, T.m, sizeof(T.m));
        if (out) {
            std::memset(out, 0, sizeof(*out));
            std::memcpy(out->transformation, T.m, sizeof(T.m));
            out->fitness = r.fit;
            out->inlier_rmse = r.rmse;
            out->num_correspondences = r.k;
            out->iterations = r.iters;
            out->nn_passes = r.passes;
        }
        return VISMA_ICP_OK;
    }
    {
        Engine::LoopScope scope(ctx->eng.get(), ctx->solo() ? steps : 0);
        for (int i = 0; i < steps; i++) {
            int rc = ctx->pass(Tc, max_dist, false, world, stats, &fit, &rmse, &k);
            if (rc) return rc;
            Tc = ctx->apply_update(ctx->solve(stats, solver, with_scaling != 0, false), Tc, world);
        }
    }
    const Mat4 T = from_centred(Tc, ctx->centre);
    std::memcpy(T_inout, T.m, sizeof(T.m));
    if (out) {
        std::memset(out, 0, sizeof(*out));
        std::memcpy(out->transformation, T.m, sizeof(T.m));
        out->fitness = fit;
        out->inlier_rmse = rmse;
        out->num_correspondences = k;
        out->iterations = steps;
        out->nn_passes = steps;
    }
    return VISMA_ICP_OK;
}

int visma_icp_run_point_to_plane(visma_icp_ctx *ctx, const double init[16], double max_dist,
                                 int max_iter, double rel_fitness, double rel_rmse,
                                 visma_icp_result *out)
{
    CTX_CHECK();
    if (!init || !out || max_iter < 0) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad run arguments");
    return ctx->run(init, max_dist, max_iter, rel_fitness, rel_rmse, VISMA_ICP_SOLVER_GN_EULER, false, true, out);
}

static int yaw_sweep(visma_icp_ctx *ctx, int level, double max_dist, int max_iter, double rel_fitness,
                     double rel_rmse, int solver, bool plane, visma_icp_result *best, int *best_level,
                     visma_icp_result *per_level);

int visma_icp_run_yaw_sweep(visma_icp_ctx *ctx, int level, double max_dist, int max_iter,
                            double rel_fitness, double rel_rmse, int solver, visma_icp_result *best,
                            int *best_level, visma_icp_result *per_level)
{
    CTX_CHECK();
    if (solver < 0 || solver > VISMA_ICP_SOLVER_GN_EXPMAP) return ctx->fail(VISMA_ICP_ERR_INVALID, "unknown solver");
    return yaw_sweep(ctx, level, max_dist, max_iter, rel_fitness, rel_rmse, solver, false, best, best_level, per_level);
}

int visma_icp_run_yaw_sweep_point_to_plane(visma_icp_ctx *ctx, int level, double max_dist, int max_iter,
                                           double rel_fitness, double rel_rmse, visma_icp_result *best,
                                           int *best_level, visma_icp_result *per_level)
{
    CTX_CHECK();
    return yaw_sweep(ctx, level, max_dist, max_iter, rel_fitness, rel_rmse, VISMA_ICP_SOLVER_GN_EULER, true, best,
                     best_level, per_level);
}

static int yaw_sweep(visma_icp_ctx *ctx, int level, double max_dist, int max_iter, double rel_fitness,
                     double rel_rmse, int solver, bool plane, visma_icp_result *best, int *best_level,
                     visma_icp_result *per_level)
{
    if (level <= 0 || !best || max_iter < 0) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad sweep arguments");
    if (plane && ctx->have_tgt && !ctx->eng->has_normals() && max_dist > 0.0) {
        // Registration.cpp:152-157: every start returns RegistrationResult(init); none has correspondences
        const double interval0 = 2.0 * M_PI / (double)level;
        std::memset(best, 0, sizeof(*best));
        const Mat4 I = Mat4::identity();
        std::memcpy(best->transformation, I.m, sizeof(I.m));
        if (best_level) *best_level = -1;
        for (int i = 0; per_level && i < level; i++) {
            const double a = interval0 * i, c = std::cos(a), s = std::sin(a);
            Mat4 init = Mat4::identity();
            init(0, 0) = c; init(0, 2) = s; init(2, 0) = -s; init(2, 2) = c;
            std::memset(&per_level[i], 0, sizeof(per_level[i]));
            std::memcpy(per_level[i].transformation, init.m, sizeof(init.m));
        }
        return VISMA_ICP_OK;
    }
    // src/annotation.cpp:35-61
    const double interval = 2.0 * M_PI / (double)level;
    if (ctx->use_device_loop_batched() && max_dist > 0.0 && ctx->have_src && ctx->have_tgt) {
        // all `level` ICPs in flight together: one launch per iteration covers every
        // (yaw, source point) pair over the shared grid, `level` solves run in parallel
        std::vector<Mat4> inits((size_t)level), Tc0((size_t)level);
        for (int i = 0; i < level; i++) {
            const double a = interval * i, c = std::cos(a), s = std::sin(a);
            Mat4 init = Mat4::identity();
            init(0, 0) = c; init(0, 2) = s; init(2, 0) = -s; init(2, 2) = c;
            inits[i] = init;
            Tc0[i] = to_centred(init, ctx->centre);
        }
        Engine::LoopParams lp;
        lp.Tc0 = Tc0[0];
        std::memcpy(lp.centre, ctx->centre, sizeof(ctx->centre));
        lp.max_dist = max_dist; lp.rel_fit = rel_fitness; lp.rel_rmse = rel_rmse;
        lp.max_iter = max_iter; lp.solver = solver; lp.passes = max_iter + 1;
        lp.scaling = false; lp.plane = plane;
        lp.world = visma_icp_ctx::wants_world_frame(solver, plane);
        lp.check_stop = true;
        lp.ns_total = ctx->ns_total > 0 ? ctx->ns_total : ctx->eng->ns();
        std::vector<Engine::LoopResult> rs((size_t)level);
        int rc = ctx->eng->run_loop(lp, Tc0.data(), level, rs.data());
        if (rc == VISMA_ICP_OK) {
            visma_icp_result bb;
            std::memset(&bb, 0, sizeof(bb));
            const Mat4 I = Mat4::identity();
            std::memcpy(bb.transformation, I.m, sizeof(I.m));
            int bl = -1;
            for (int i = 0; i < level; i++) {
                visma_icp_result r;
                std::memset(&r, 0, sizeof(r));
                const Mat4 T = from_centred(rs[i].Tc, ctx->centre);
                std::memcpy(r.transformation, T.m, sizeof(T.m));
                r.fitness = rs[i].fit; r.inlier_rmse = rs[i].rmse;
                r.num_correspondences = rs[i].k;
                r.iterations = rs[i].iters; r.nn_passes = rs[i].passes;
                if (per_level) per_level[i] = r;
                if (r.num_correspondences > bb.num_correspondences) { bb = r; bl = i; }   // strict >
            }
            ctx->eng->select_problem(bl >= 0 ? bl : 0);
            if (bl >= 0) ctx->last_Tc = rs[bl].Tc;
            *best = bb;
            if (best_level) *best_level = bl;
            return VISMA_ICP_OK;
        }
        if (rc != VISMA_ICP_ERR_STATE) return ctx->eng_fail(rc);
        // (brute-force search selected, or RCCL attached): fall through to the sequential sweep
    }
    visma_icp_result b;
    std::memset(&b, 0, sizeof(b));
    const Mat4 I = Mat4::identity();
    std::memcpy(b.transformation, I.m, sizeof(I.m));
    int bl = -1;
    for (int i = 0; i < level; i++) {
        const double a = interval * i, c = std::cos(a), s = std::sin(a);
        Mat4 init = Mat4::identity();
        init(0, 0) = c; init(0, 2) = s; init(2, 0) = -s; init(2, 2) = c;
        visma_icp_result r;
        int rc = ctx->run(init.m, max_dist, max_iter, rel_fitness, rel_rmse, solver, false, plane, &r);
        if (rc) return rc;
        if (per_level) per_level[i] = r;
        if (r.num_correspondences > b.num_correspondences) { b = r; bl = i; }
    }
    *best = b;
    if (best_level) *best_level = bl;
    return VISMA_ICP_OK;
}

// normals == NULL: the point-to-point estimator with `solver`; otherwise normals[i] are the target
// normals of problem i (AoS stride 3; NULL = that cloud has none) and the estimator is point-to-plane
static int run_batch_impl(visma_icp_ctx *ctx, const visma_icp_problem *probs, const double *const *normals, int n,
                          int max_iter, double rel_fitness, double rel_rmse, int solver, visma_icp_result *out)
{
    ctx->eng->set_exact(ctx->search_precision == 1);     // (a batch uploads its own clouds)
    const bool plane = normals != nullptr;
    bool all_normals = plane;
    for (int i = 0; plane && i < n; i++)
        if (probs[i].nt > 0 && !normals[i]) all_normals = false;
    // (a problem without normals returns its initial transform, Registration.cpp:152-157: sequential path;
    //  point-to-plane batches run the exact or the fp32 search)
    const bool batch_ok = !plane || (all_normals && ctx->search_precision != 2);
    if (ctx->use_device_loop_batched() && n > 1 && batch_ok) {
        // every problem in flight together: concatenated clouds, one grid per problem,
        // one NN launch + one fold/solve launch per pass for the whole batch
        StageTrace tr("batch/driver");
        std::vector<std::vector<float>> sbuf((size_t)n), tbuf((size_t)n);
        std::vector<Engine::BatchProblem> pb((size_t)n);
        bool ok = true;
        for (int i = 0; i < n; i++) {
            const visma_icp_problem &q = probs[i];
            if (q.ns < 0 || q.nt < 0 || (q.ns > 0 && !q.src_xyz) || (q.nt > 0 && !q.tgt_xyz))
                return ctx->fail(VISMA_ICP_ERR_INVALID, "bad batch problem");
            if (!(q.max_dist > 0.0)) ok = false;   // rare: handled by the sequential path
        }
        // Problems often share clouds: the 24 yaw starts of one model (src/annotation.cpp:35-61),
        // every model against the same scene.  Equal (pointer, count) pairs are packed, uploaded
        // and gridded ONCE.
        std::vector<int> tshare((size_t)n, -1), sshare((size_t)n, -1), gshare((size_t)n, -1);
        for (int i = 0; ok && i < n; i++)
            for (int j = 0; j < i; j++) {
                if (tshare[j] >= 0) continue;                      // only first occurrences are referenced
                if (probs[j].tgt_xyz == probs[i].tgt_xyz && probs[j].nt == probs[i].nt) { tshare[i] = j; break; }
            }
        for (int i = 0; ok && i < n; i++) {
            const int t = tshare[i] >= 0 ? tshare[i] : i;
            for (int j = 0; j < i; j++) {
                const int tj = tshare[j] >= 0 ? tshare[j] : j;
                if (tj != t) continue;
                if (gshare[i] < 0 && gshare[j] < 0 && probs[j].max_dist == probs[i].max_dist) gshare[i] = j;
                if (sshare[i] < 0 && sshare[j] < 0 && probs[j].src_xyz == probs[i].src_xyz && probs[j].ns == probs[i].ns)
                    sshare[i] = j;
            }
        }
        tr.mark("sharing found");
        std::vector<std::array<double, 3>> cen((size_t)n);
        std::vector<std::array<float, 6>> tbox((size_t)n);
        // double-precision search for the whole batch when every problem qualifies
        const bool want64 = ctx->search_precision != 0;
        std::vector<std::vector<Pt64>> s8((size_t)n), t8((size_t)n), n8((size_t)n);
        std::vector<std::vector<float>> nbuf((size_t)n);
        if (plane)
            for (int i = 0; ok && i < n; i++)                     // clouds passed twice carry the same normals
                if (tshare[i] >= 0 && normals[tshare[i]] != normals[i]) ok = false;
        if (ok)
            parallel_for(n, 1, [&](int64_t i) {                    // targets: first occurrences only
                if (tshare[i] >= 0) return;
                if (plane && want64) {
                    n8[i].resize((size_t)std::max<int64_t>(probs[i].nt, 1));
                    for (int64_t j = 0; j < probs[i].nt; j++)
                        n8[i][(size_t)j] = Pt64{normals[i][3 * j], normals[i][3 * j + 1], normals[i][3 * j + 2], 0ull};
                } else if (plane) {
                    const double zero[3] = {0, 0, 0};
                    pack_f64(normals[i], probs[i].nt, 3, zero, nbuf[i]);
                }
                const visma_icp_problem &q = probs[i];
                double c[3];
                centroid_f64(q.tgt_xyz, q.nt, 3, c, false);       // the same value as set_clouds_f64 computes
                for (int a = 0; a < 3; a++) cen[i][a] = c[a];
                // the target itself is uploaded as it is and expanded on the device; the host only needs the
                // bounding box of its fp32 copy: the rounding (float)(x - c) is monotone, so the box of the
                // f64 values, rounded the same way, is the box of the rounded values
                double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                for (int64_t j = 0; j < q.nt; j++)
                    for (int a = 0; a < 3; a++) {
                        const double v = q.tgt_xyz[3 * j + a];
                        if (j == 0 || v < lo[a]) lo[a] = v;
                        if (j == 0 || v > hi[a]) hi[a] = v;
                    }
                for (int a = 0; a < 3; a++) { tbox[i][a] = (float)(lo[a] - c[a]); tbox[i][3 + a] = (float)(hi[a] - c[a]); }
            });
        tr.mark("targets: centroid, box");
        if (ok)
            parallel_for(n, 1, [&](int64_t i) {
                const visma_icp_problem &q = probs[i];
                const int t = tshare[i] >= 0 ? tshare[i] : (int)i;
                const double *c = cen[t].data();
                if (sshare[i] < 0) {
                    pack_f64(q.src_xyz, q.ns, 3, c, sbuf[i]);
                    std::vector<int32_t> order;
                    morton_order(sbuf[i], q.ns, order);
                    if (want64) {
                        s8[i].resize((size_t)std::max<int64_t>(q.ns, 1));
                        for (int64_t pos = 0; pos < q.ns; pos++) {
                            const double *sp = q.src_xyz + 3 * (size_t)order[(size_t)pos];
                            s8[i][(size_t)pos] = Pt64{sp[0] - c[0], sp[1] - c[1], sp[2] - c[2], (unsigned long long)order[(size_t)pos]};
                        }
                    }
                }
                Engine::BatchProblem &b = pb[i];
                b.src64 = (want64 && sshare[i] < 0) ? s8[i].data() : nullptr;
                b.tgt64 = nullptr;
                b.tgt_raw = probs[t].tgt_xyz;
                b.tgt_f64 = want64;
                b.nrm64 = (plane && want64) ? n8[t].data() : nullptr;
                b.nrm_xyzw = (plane && !want64) ? nbuf[t].data() : nullptr;
                b.src_xyzw = sshare[i] < 0 ? sbuf[i].data() : nullptr; b.ns = q.ns;
                b.tgt_xyzw = nullptr; b.nt = q.nt;
                b.src_share = sshare[i];
                b.grid_share = gshare[i];
                b.Tc0 = to_centred(Mat4::from(q.init), c);
                std::memcpy(b.centre, c, 3 * sizeof(double));
                b.max_dist = q.max_dist;
                for (int a = 0; a < 3; a++) { b.bb_min[a] = 0.f; b.bb_max[a] = 0.f; }
                if (gshare[i] >= 0) return;                        // the grid (and its box) is reused
                if (q.nt > 0)
                    for (int a = 0; a < 3; a++) { b.bb_min[a] = tbox[t][a]; b.bb_max[a] = tbox[t][3 + a]; }
            });
        tr.mark("sources: pack, order");
        if (ok) {
            Engine::LoopParams lp;
            lp.Tc0 = Mat4::identity();
            lp.centre[0] = lp.centre[1] = lp.centre[2] = 0.0;
            lp.max_dist = 0.0; lp.rel_fit = rel_fitness; lp.rel_rmse = rel_rmse;
            lp.max_iter = max_iter; lp.solver = solver; lp.passes = max_iter + 1;
            lp.scaling = false; lp.plane = plane;
            lp.world = visma_icp_ctx::wants_world_frame(solver, plane);
            lp.check_stop = true; lp.ns_total = 0;
            std::vector<Engine::LoopResult> rs((size_t)n);
            int rc = ctx->eng->run_loop_batch(lp, pb, rs.data());
            tr.mark("engine batch loop");
            if (rc == VISMA_ICP_OK) {
                for (int i = 0; i < n; i++) {
                    std::memset(&out[i], 0, sizeof(out[i]));
                    const Mat4 T = from_centred(rs[i].Tc, pb[i].centre);
                    std::memcpy(out[i].transformation, T.m, sizeof(T.m));
                    out[i].fitness = rs[i].fit; out[i].inlier_rmse = rs[i].rmse;
                    out[i].num_correspondences = rs[i].k;
                    out[i].iterations = rs[i].iters; out[i].nn_passes = rs[i].passes;
                }
                return VISMA_ICP_OK;
            }
            if (rc != VISMA_ICP_ERR_STATE) return ctx->eng_fail(rc);
        }
    }
    for (int i = 0; i < n; i++) {
        int rc = visma_icp_set_clouds_f64(ctx, probs[i].src_xyz, probs[i].ns, 3, probs[i].tgt_xyz, probs[i].nt, 3);
        if (rc) return rc;
        if (plane && normals[i]) {
            rc = visma_icp_set_target_normals_f64(ctx, normals[i], probs[i].nt, 3);
            if (rc) return rc;
        }
        rc = ctx->run(probs[i].init, probs[i].max_dist, max_iter, rel_fitness, rel_rmse, solver, false, plane, &out[i]);
        if (rc) return rc;
    }
    return VISMA_ICP_OK;
}

int visma_icp_run_batch(visma_icp_ctx *ctx, const visma_icp_problem *probs, int n, int max_iter,
                        double rel_fitness, double rel_rmse, int solver, visma_icp_result *out)
{
    CTX_CHECK();
    if (n < 0 || (n > 0 && (!probs || !out)) || max_iter < 0) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad batch arguments");
    if (solver < 0 || solver > VISMA_ICP_SOLVER_GN_EXPMAP) return ctx->fail(VISMA_ICP_ERR_INVALID, "unknown solver");
    return run_batch_impl(ctx, probs, nullptr, n, max_iter, rel_fitness, rel_rmse, solver, out);
}

int visma_icp_run_batch_point_to_plane(visma_icp_ctx *ctx, const visma_icp_problem *probs,
                                       const double *const *tgt_normals, int n, int max_iter, double rel_fitness,
                                       double rel_rmse, visma_icp_result *out)
{
    CTX_CHECK();
    if (n < 0 || (n > 0 && (!probs || !out || !tgt_normals)) || max_iter < 0)
        return ctx->fail(VISMA_ICP_ERR_INVALID, "bad batch arguments");
    static const double *const none[1] = {nullptr};
    return run_batch_impl(ctx, probs, n > 0 ? tgt_normals : none, n, max_iter, rel_fitness, rel_rmse,
                          VISMA_ICP_SOLVER_GN_EULER, out);
}

int visma_icp_set_nn_mode(visma_icp_ctx *ctx, int nn_mode)
{
    CTX_CHECK();
    int rc = ctx->eng->set_nn_mode(nn_mode);
    if (rc) return ctx->eng_fail(rc);
    return VISMA_ICP_OK;
}

int visma_icp_get_nn_mode_used(visma_icp_ctx *ctx, int *nn_mode)
{
    CTX_CHECK();
    if (!nn_mode) return ctx->fail(VISMA_ICP_ERR_INVALID, "nn_mode is NULL");
    *nn_mode = ctx->eng->nn_mode_used();
    return VISMA_ICP_OK;
}

int visma_icp_get_search_kernel_used(visma_icp_ctx *ctx, int *kernel)
{
    CTX_CHECK();
    if (!kernel) return ctx->fail(VISMA_ICP_ERR_INVALID, "kernel is NULL");
    *kernel = ctx->eng->search_kernel_used();
    return VISMA_ICP_OK;
}

int visma_icp_forget_winners(visma_icp_ctx *ctx)
{
    CTX_CHECK();
    int rc = ctx->eng->forget_winners();
    if (rc) return ctx->eng_fail(rc);
    return VISMA_ICP_OK;
}

int visma_icp_set_device_loop(visma_icp_ctx *ctx, int enabled)
{
    CTX_CHECK();
    ctx->loop_mode = enabled < 0 ? 2 : (enabled != 0 ? 1 : 0);
    return VISMA_ICP_OK;
}

int visma_icp_set_persistent(visma_icp_ctx *ctx, int enabled, double timeout_ms)
{
    CTX_CHECK();
    ctx->eng->set_persistent(enabled, timeout_ms);
    return VISMA_ICP_OK;
}

int visma_icp_set_persistent_cu_share(double share)
{
    if (!(share > 0.0) || share > 1.0) return VISMA_ICP_ERR_INVALID;
    persist_cu_share_ref().store(share);
    return VISMA_ICP_OK;
}

int visma_icp_get_persistent_info(visma_icp_ctx *ctx, visma_icp_persistent_info *out)
{
    CTX_CHECK();
    if (!out || out->struct_size < (int)sizeof(visma_icp_persistent_info))
        return ctx->fail(VISMA_ICP_ERR_INVALID, "visma_icp_persistent_info: set struct_size = sizeof(visma_icp_persistent_info)");
    const int sz = out->struct_size;
    std::memset(out, 0, sizeof(*out));
    out->struct_size = sz;
    out->cu_share = 1.0;
    ctx->eng->get_persistent_info(out);
    return VISMA_ICP_OK;
}

int visma_icp_set_ring_search(visma_icp_ctx *ctx, int mode)
{
    CTX_CHECK();
    ctx->eng->set_ring_search(mode);
    return VISMA_ICP_OK;
}

int visma_icp_get_ring_search(visma_icp_ctx *ctx, int *rings, double *cell, double *occupancy)
{
    CTX_CHECK();
    ctx->eng->get_ring_search(rings, cell, occupancy);
    return VISMA_ICP_OK;
}

int visma_icp_get_sweep_info(visma_icp_ctx *ctx, double *launches, double *aborts)
{
    CTX_CHECK();
    ctx->eng->get_sweep_info(launches, aborts);
    return VISMA_ICP_OK;
}

int visma_icp_test_stall_command(visma_icp_ctx *ctx, int nth, double ms)
{
    CTX_CHECK();
    if (nth < 0 || !(ms >= 0.0) || ms > 10000.0) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad stall arguments");
    ctx->eng->stall_command(nth, ms);
    return VISMA_ICP_OK;
}

int visma_icp_set_profiling(visma_icp_ctx *ctx, int enabled)
{
    CTX_CHECK();
    ctx->eng->set_profiling(enabled);
    return VISMA_ICP_OK;
}

int visma_icp_get_timing(visma_icp_ctx *ctx, visma_icp_timing *out, int reset)
{
    CTX_CHECK();
    if (!out) return ctx->fail(VISMA_ICP_ERR_INVALID, "out is NULL");
    ctx->eng->get_timing(out, reset != 0);
    return VISMA_ICP_OK;
}

int visma_icp_get_timing_sized(visma_icp_ctx *ctx, void *out, size_t struct_size, int reset)
{
    CTX_CHECK();
    if (!out) return ctx->fail(VISMA_ICP_ERR_INVALID, "out is NULL");
    visma_icp_timing t;
    ctx->eng->get_timing(&t, reset != 0);
    std::memcpy(out, &t, struct_size < sizeof(t) ? struct_size : sizeof(t));
    return VISMA_ICP_OK;
}

int visma_icp_plan_ring_grid(const float mn[3], const float mx[3], double max_dist, double cell, int dims[3], double *cell_out, int *rings)
{
    if (!mn || !mx) return VISMA_ICP_ERR_INVALID;
    const visma::GridParams g = visma::grid_plan_ring(mn, mx, max_dist, cell, visma::kGridMaxCells);
    if (dims) for (int a = 0; a < 3; a++) dims[a] = g.dim[a];
    if (cell_out) *cell_out = (double)g.h;
    if (rings) *rings = g.ring;
    return VISMA_ICP_OK;
}

int visma_icp_ring_visiting_order(int rings, int capacity, short *dy, short *dz, float *base, int *nrows)
{
    const std::vector<visma::RingRow> rows = visma::ring_visiting_order(rings);
    if (nrows) *nrows = (int)rows.size();
    if (rows.empty()) return VISMA_ICP_ERR_INVALID;
    for (int k = 0; k < (int)rows.size() && k < capacity; k++) {
        if (dy) dy[k] = rows[k].dy;
        if (dz) dz[k] = rows[k].dz;
        if (base) base[k] = rows[k].base;
    }
    return VISMA_ICP_OK;
}

int visma_icp_get_tile_config(int *s_tile, int *t_chunk, int *block)
{
    if (s_tile) *s_tile = kSTile;
    if (t_chunk) *t_chunk = kTChunk;
    if (block) *block = kBlock;
    return VISMA_ICP_OK;
}

int visma_icp_get_launch_config(visma_icp_ctx *ctx, int *src_tiles, int *tgt_splits)
{
    CTX_CHECK();
    int a = 0, b = 0;
    ctx->eng->launch_config(&a, &b);
    if (src_tiles) *src_tiles = a;
    if (tgt_splits) *tgt_splits = b;
    return VISMA_ICP_OK;
}

int visma_icp_comm_unique_id(void *out_id)
{
    if (!out_id) return VISMA_ICP_ERR_INVALID;
    if (!g_rccl.load()) { g_create_error = g_rccl.error; return VISMA_ICP_ERR_RCCL; }
    NcclId id;
    int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) { g_create_error = "ncclGetUniqueId failed"; return VISMA_ICP_ERR_RCCL; }
    std::memcpy(out_id, &id, sizeof(id));
    return VISMA_ICP_OK;
}

int visma_icp_comm_init(visma_icp_ctx *ctx, int rank, int nranks, const void *unique_id)
{
    CTX_CHECK();
    if (!unique_id || nranks < 1 || rank < 0 || rank >= nranks) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad comm arguments");
    int rc = ctx->eng->comm_init(rank, nranks, unique_id);
    if (rc) return ctx->eng_fail(rc);
    ctx->rank = rank;
    ctx->nranks = nranks;
    return VISMA_ICP_OK;
}

int visma_icp_comm_ipc_export(visma_icp_ctx *ctx, void *out_handle)
{
    CTX_CHECK();
    if (!out_handle) return ctx->fail(VISMA_ICP_ERR_INVALID, "out_handle is NULL");
    int rc = ctx->eng->ipc_export(out_handle);
    if (rc) return ctx->eng_fail(rc);
    return VISMA_ICP_OK;
}

int visma_icp_comm_ipc_init(visma_icp_ctx *ctx, int rank, int nranks, const void *handles)
{
    CTX_CHECK();
    if (!handles || nranks < 1 || rank < 0 || rank >= nranks) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad comm arguments");
    int rc = ctx->eng->ipc_init(rank, nranks, handles);
    if (rc) return ctx->eng_fail(rc);
    ctx->rank = rank;
    ctx->nranks = nranks;
    return VISMA_ICP_OK;
}

int visma_icp_set_allreduce(visma_icp_ctx *ctx, visma_icp_allreduce_fn fn, void *user, int rank, int nranks)
{
    CTX_CHECK();
    if (nranks < 1 || rank < 0 || rank >= nranks) return ctx->fail(VISMA_ICP_ERR_INVALID, "bad rank arguments");
    ctx->host_allreduce = fn;
    ctx->host_allreduce_user = user;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return VISMA_ICP_OK;
}

int visma_icp_set_search_precision(visma_icp_ctx *ctx, int mode)
{
    CTX_CHECK();
    if (mode < 0 || mode > 2) return ctx->fail(VISMA_ICP_ERR_INVALID, "search precision must be 0, 1 or 2");
    // takes effect at the next cloud upload (the header's contract): the clouds resident now were laid out
    // for the mode they were uploaded under (fp32 only / fp32 + f64 copies), and switching the flag alone would
    // run a third kernel (mode 0 over f64 copies = the all-f64 search), not the one asked for
    ctx->search_precision = mode;
    return VISMA_ICP_OK;
}

int visma_icp_get_search_precision_used(visma_icp_ctx *ctx, int *is_f64)
{
    CTX_CHECK();
    if (!is_f64) return ctx->fail(VISMA_ICP_ERR_INVALID, "null output");
    *is_f64 = ctx->eng->search_is_f64() ? 2 : (ctx->eng->search_is_exact() ? 1 : 0);
    return VISMA_ICP_OK;
}

int visma_icp_set_target_shard(visma_icp_ctx *ctx, int64_t global_offset, int64_t global_nt, const double centre[3])
{
    CTX_CHECK();
    int rc = ctx->eng->set_target_shard(global_offset, global_nt);
    if (rc) return ctx->eng_fail(rc);
    ctx->target_sharded = global_nt > 0;
    ctx->fixed_centre = centre != nullptr && global_nt > 0;
    if (ctx->fixed_centre) std::memcpy(ctx->centre, centre, 3 * sizeof(double));
    return VISMA_ICP_OK;
}

int visma_icp_set_minreduce(visma_icp_ctx *ctx, visma_icp_minreduce_fn fn, void *user)
{
    CTX_CHECK();
    ctx->eng->set_minreduce(fn, user);
    return VISMA_ICP_OK;
}

int visma_icp_set_global_source_count(visma_icp_ctx *ctx, int64_t ns_total)
{
    CTX_CHECK();
    if (ns_total < 0) return ctx->fail(VISMA_ICP_ERR_INVALID, "negative count");
    ctx->ns_total = ns_total;
    return VISMA_ICP_OK;
}

}  // extern "C"

