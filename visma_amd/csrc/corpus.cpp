// corpus.cpp -- native work queue over a corpus of (scene, CAD candidate) registrations (visma_icp_run_corpus).
//
// Replaces the per-object loop of feh::AnnotationTool (src/annotation.cpp:103-168) + RegisterModelToScene
// (src/annotation.cpp:29-64) when there are many objects: one host thread per context (= per GPU) pulls
// chunks of items from an atomic counter and runs each chunk's yaw starts as ONE batch on its GPU.  Built on
// the public C ABI only (visma_icp_run_batch); nothing here touches HIP.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/visma_icp.h"

namespace {

struct Shared {
    visma_icp_ctx *const *ctxs;
    const visma_icp_corpus_item *items;
    int64_t n_items;
    visma_icp_corpus_params p;
    int64_t *counter;
    visma_icp_corpus_result *results;
    std::mutex mu;
    int rc = VISMA_ICP_OK;
    std::string err;
    std::atomic<bool> stop{false};
};

void fail(Shared &s, int rc, const std::string &msg)
{
    std::lock_guard<std::mutex> g(s.mu);
    if (s.rc == VISMA_ICP_OK) { s.rc = rc; s.err = msg; }
    s.stop.store(true);
}

void worker(Shared &s, int w)
{
    visma_icp_ctx *ctx = s.ctxs[w];
    const int level = s.p.level, chunk = s.p.chunk;
    std::vector<visma_icp_problem> probs((size_t)chunk * level);
    std::vector<visma_icp_result> out((size_t)chunk * level);
    // the yaw starts (src/annotation.cpp:35-39): Eigen::AngleAxis(interval * i, UnitY)
    std::vector<double> inits((size_t)level * 16, 0.0);
    const double interval = 2.0 * M_PI / (double)level;
    for (int k = 0; k < level; k++) {
        double *T = &inits[(size_t)k * 16];
        const double a = interval * k, c = std::cos(a), sn = std::sin(a);
        T[0] = c; T[2] = sn; T[5] = 1.0; T[8] = -sn; T[10] = c; T[15] = 1.0;
    }
    while (!s.stop.load()) {
        const int64_t i0 = __atomic_fetch_add(s.counter, (int64_t)chunk, __ATOMIC_RELAXED);
        if (i0 >= s.n_items) break;
        const int n = (int)std::min<int64_t>(chunk, s.n_items - i0);
        for (int j = 0; j < n; j++) {
            const visma_icp_corpus_item &it = s.items[i0 + j];
            for (int k = 0; k < level; k++) {
                visma_icp_problem &q = probs[(size_t)j * level + k];
                q.src_xyz = it.model_xyz; q.ns = it.n_model;
                q.tgt_xyz = it.scene_xyz; q.nt = it.n_scene;
                std::memcpy(q.init, &inits[(size_t)k * 16], sizeof(q.init));
                q.max_dist = s.p.max_dist;
            }
        }
        const int rc = visma_icp_run_batch(ctx, probs.data(), n * level, s.p.max_iter, s.p.rel_fitness, s.p.rel_rmse,
                                           s.p.solver, out.data());
        if (rc != VISMA_ICP_OK) {
            const char *m = visma_icp_last_error(ctx);
            fail(s, rc, std::string("context ") + std::to_string(w) + ", items " + std::to_string(i0) + "..: " + (m ? m : "error"));
            return;
        }
        for (int j = 0; j < n; j++) {
            visma_icp_corpus_result &r = s.results[i0 + j];
            // src/annotation.cpp:36,59-61: best_result starts empty; a start replaces it only with MORE correspondences
            std::memset(&r.best, 0, sizeof(r.best));
            r.best.transformation[0] = r.best.transformation[5] = r.best.transformation[10] = r.best.transformation[15] = 1.0;
            r.best_level = -1;
            r.iterations_all_starts = 0;
            for (int k = 0; k < level; k++) {
                const visma_icp_result &o = out[(size_t)j * level + k];
                r.iterations_all_starts += o.iterations;
                if (o.num_correspondences > r.best.num_correspondences) { r.best = o; r.best_level = k; }
            }
            r.device = w;
        }
    }
}

}  // namespace

extern "C" int visma_icp_run_corpus(visma_icp_ctx *const *ctxs, int n_ctx, const visma_icp_corpus_item *items,
                                    int64_t n_items, const visma_icp_corpus_params *params, int64_t *counter,
                                    visma_icp_corpus_result *results, char *errbuf, size_t errbuf_len)
{
    auto bad = [&](const char *m) {
        if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", m);
        return (int)VISMA_ICP_ERR_INVALID;
    };
    if (errbuf && errbuf_len) errbuf[0] = 0;
    if (!ctxs || n_ctx < 1 || n_items < 0 || !params || (n_items > 0 && (!items || !results))) return bad("bad corpus arguments");
    if (params->level < 1 || params->level > 4096 || params->max_iter < 0 || !(params->max_dist > 0.0)) return bad("bad corpus parameters");
    for (int w = 0; w < n_ctx; w++)
        if (!ctxs[w]) return bad("NULL context");
    for (int64_t i = 0; i < n_items; i++) {
        const visma_icp_corpus_item &it = items[i];
        if (it.n_model < 0 || it.n_scene < 0 || (it.n_model > 0 && !it.model_xyz) || (it.n_scene > 0 && !it.scene_xyz))
            return bad("bad corpus item");
        results[i].device = -1;
    }
    Shared s;
    s.ctxs = ctxs; s.items = items; s.n_items = n_items; s.p = *params; s.results = results;
    if (s.p.chunk <= 0) s.p.chunk = 8;
    int64_t own = 0;
    s.counter = counter ? counter : &own;
    std::vector<std::thread> th;
    for (int w = 1; w < n_ctx; w++) th.emplace_back(worker, std::ref(s), w);
    worker(s, 0);                                   // the calling thread drives context 0
    for (auto &t : th) t.join();
    if (s.rc != VISMA_ICP_OK && errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", s.err.c_str());
    return s.rc;
}
