// corpus.cpp -- native work queue over a corpus of (scene, CAD candidate) registrations (visma_icp_run_corpus).
//
// Replaces the per-object loop of feh::AnnotationTool (src/annotation.cpp:103-168) + RegisterModelToScene
// (src/annotation.cpp:29-64) when there are many objects: one host thread per context (= per GPU) pulls
// chunks of items from an atomic counter and runs each chunk's yaw starts as ONE batch on its GPU.  Built on
// the public C ABI only (visma_icp_run_batch); nothing here touches HIP.
#include <algorithm>
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
    // a worker allocates chunk * level problems per pull and hands that count to visma_icp_run_batch as an int
    if ((int64_t)s.p.chunk > n_items) s.p.chunk = (int)std::max<int64_t>(n_items, 1);
    if ((int64_t)s.p.chunk * (int64_t)s.p.level > (int64_t)(1 << 20)) return bad("corpus chunk x level exceeds 1048576 registrations per batch");
    int64_t own = 0;
    s.counter = counter ? counter : &own;
    std::vector<std::thread> th;
    for (int w = 1; w < n_ctx; w++) th.emplace_back(worker, std::ref(s), w);
    worker(s, 0);                                   // the calling thread drives context 0
    for (auto &t : th) t.join();
    if (s.rc != VISMA_ICP_OK && errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", s.err.c_str());
    return s.rc;
}

// ---- one batch of problems over several worker contexts (BASELINE config 3: all objects of one scene) -----------
// Problems that pass the same target cloud stay together (they share one upload and one grid inside
// visma_icp_run_batch); the groups are dealt to the contexts largest first, each to the least loaded one; every
// context runs its share as ONE visma_icp_run_batch on its own stream and host thread, the shares side by side.
// A problem's result does not depend on what else is in its batch, so out[] equals the single-context call's.
extern "C" int visma_icp_run_batch_multi(visma_icp_ctx *const *ctxs, int n_ctx, const visma_icp_problem *probs, int n,
                                         int max_iter, double rel_fitness, double rel_rmse, int solver,
                                         visma_icp_result *out, char *errbuf, size_t errbuf_len)
{
    auto bad = [&](const char *m) {
        if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", m);
        return (int)VISMA_ICP_ERR_INVALID;
    };
    if (errbuf && errbuf_len) errbuf[0] = 0;
    if (!ctxs || n_ctx < 1 || n < 0 || (n > 0 && (!probs || !out))) return bad("bad batch arguments");
    for (int w = 0; w < n_ctx; w++)
        if (!ctxs[w]) return bad("NULL context");
    if (n == 0) return VISMA_ICP_OK;
    // groups of problems over the same target
    struct Group { const double *tgt; int64_t nt; double work; std::vector<int> members; };
    std::vector<Group> groups;
    for (int i = 0; i < n; i++) {
        size_t g = 0;
        for (; g < groups.size(); g++)
            if (groups[g].tgt == probs[i].tgt_xyz && groups[g].nt == probs[i].nt) break;
        if (g == groups.size()) groups.push_back(Group{probs[i].tgt_xyz, probs[i].nt, 0.0, {}});
        groups[g].members.push_back(i);
        groups[g].work += (double)probs[i].ns;
    }
    std::vector<size_t> order(groups.size());
    for (size_t g = 0; g < order.size(); g++) order[g] = g;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return groups[a].work > groups[b].work; });
    const int W = (int)std::min<size_t>((size_t)n_ctx, groups.size());
    std::vector<std::vector<int>> share((size_t)W);
    std::vector<double> load((size_t)W, 0.0);
    for (size_t g : order) {
        int best = 0;
        for (int w = 1; w < W; w++)
            if (load[(size_t)w] < load[(size_t)best]) best = w;
        share[(size_t)best].insert(share[(size_t)best].end(), groups[g].members.begin(), groups[g].members.end());
        load[(size_t)best] += groups[g].work;
    }
    std::mutex mu;
    int rc_all = VISMA_ICP_OK;
    std::string err;
    auto run_share = [&](int w) {
        std::vector<int> &idx = share[(size_t)w];
        std::sort(idx.begin(), idx.end());                       // (the caller's order inside a share)
        std::vector<visma_icp_problem> p(idx.size());
        std::vector<visma_icp_result> r(idx.size());
        for (size_t k = 0; k < idx.size(); k++) p[k] = probs[idx[k]];
        const int rc = visma_icp_run_batch(ctxs[w], p.data(), (int)p.size(), max_iter, rel_fitness, rel_rmse, solver, r.data());
        if (rc != VISMA_ICP_OK) {
            std::lock_guard<std::mutex> g(mu);
            if (rc_all == VISMA_ICP_OK) { rc_all = rc; err = visma_icp_last_error(ctxs[w]); }
            return;
        }
        for (size_t k = 0; k < idx.size(); k++) out[idx[k]] = r[k];
    };
    std::vector<std::thread> th;
    for (int w = 1; w < W; w++) th.emplace_back(run_share, w);
    run_share(0);                                               // the calling thread drives context 0
    for (auto &t : th) t.join();
    if (rc_all != VISMA_ICP_OK && errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", err.c_str());
    return rc_all;
}
