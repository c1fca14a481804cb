// engine.hpp -- what the host driver and its engines share (internal to libvisma_icp.so).
//
//   Engine        owns the clouds and produces the per-iteration statistics
//     HipEngine   the product engine: gfx950 kernels on one HIP stream            (hip_engine.cpp)
//     HookEngine  engine injected through visma_icp_create_with_engine (tests)     (driver.cpp, side build only)
//   visma_icp_ctx the driver: centring, the RegistrationICP loop
//                 (O3D/Core/Registration/Registration.cpp:141-186), the tiny f64
//                 solves, yaw sweep, batching, multi-GPU reduction               (driver.cpp)
//   aux_api.cpp   C ABI of the steps either side of ICP (voxel grid, normals, mesh sampling / distance, SO(3) / SE(3))
// No CPU fallback exists: without a GPU visma_icp_create fails.
#pragma once

#include "../../include/visma_icp.h"
#include "../../include/visma_icp_testing.h"   // (the knobs it declares are exported by the product; the engine seam
                                                // -- visma_icp_create_with_engine -- is DEFINED under VISMA_TEST_SEAMS only)

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "host_math.hpp"
#include "kernels.h"

using namespace visma;

namespace visma {
namespace drv {

inline thread_local std::string g_create_error;   // message of a failure that has no context to carry it

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            err_ = std::string(#expr) + ": " + hipGetErrorString(e__);             \
            return VISMA_ICP_ERR_HIP;                                              \
        }                                                                          \
    } while (0)

// ---- RCCL, loaded at run time ------------------------------------------------
struct NcclId { char internal[VISMA_ICP_UNIQUE_ID_BYTES]; };
typedef void *NcclComm;
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
    bool load()
    {
        if (handle) return true;
        // The system's RCCL by its full path first, bound to ITSELF (RTLD_LOCAL | RTLD_DEEPBIND): a host program
        // that has imported PyTorch already carries PyTorch's bundled copy under the same soname, and neither
        // copy's symbols may resolve into the other.  VISMA_ICP_RCCL_PATH overrides.
        const char *env = getenv("VISMA_ICP_RCCL_PATH");
        const char *names[] = {env ? env : "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            if (handle) break;
        }
        if (!handle) { error = std::string("dlopen(librccl): ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
        AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) {
            error = "librccl: missing symbols";
            return false;
        }
        return true;
    }
};
inline Rccl g_rccl;
constexpr int kNcclFloat64 = 8;  // ncclFloat64 / ncclDouble (rccl.h)
constexpr int kNcclSum = 0;      // ncclSum
constexpr int kNcclMin = 3;      // ncclMin
constexpr int kNcclUint64 = 5;   // ncclUint64

// ---- engines --------------------------------------------------------------------
// Host threads for the passes over the caller's clouds (staging, packing, ordering).  Starting a std::thread
// costs ~30 us and a C4 upload runs five such passes: the workers are started ONCE per process and parked on a
// condition variable.  One pass at a time uses them; a caller that finds them busy (another context's upload on
// another host thread -- the corpus workers) starts threads of its own, as every pass did before.
class HostPool {
public:
    // a few pools (never destroyed: their threads are parked): concurrent callers -- several contexts working a
    // corpus or a scene's batches side by side -- each find one, instead of the second caller starting 16 threads of its
    // own for every pass (pool 0: up to 31 helpers; the others 15)
    static constexpr int kPools = 4;
    static HostPool &instance(int k)
    {
        static HostPool *p[kPools] = {(owner_pid(), new HostPool(32)), new HostPool(16), new HostPool(16), new HostPool(16)};
        return *p[k];
    }
    static pid_t owner_pid()
    {
        static const pid_t pid = getpid();                         // (first call: the process that starts the workers)
        return pid;
    }
    int size() const { return (int)workers_; }
    // the pools' threads exist only in the process that started them: a child made by fork() (Python multiprocessing's
    // default start method) inherits the bookkeeping but no workers and would wait for them forever
    static bool usable() { return owner_pid() == getpid(); }
    // fn(i) for i in [0, n) on up to `threads` threads including the caller's; false: busy, nothing was run
    template <typename F>
    bool run(int64_t n, int threads, F &fn)
    {
        std::unique_lock<std::mutex> busy(run_mu_, std::try_to_lock);
        if (!busy.owns_lock()) return false;
        struct Job { F *fn; std::atomic<int64_t> next; int64_t n; } job{&fn, {0}, n};
        auto body = [](void *j) {
            Job *jb = static_cast<Job *>(j);
            for (;;) {
                const int64_t i = jb->next.fetch_add(1, std::memory_order_relaxed);
                if (i >= jb->n) break;
                (*jb->fn)(i);
            }
        };
        const int helpers = std::min(threads - 1, (int)workers_);
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = &job; body_ = body; wanted_ = helpers; pending_ = helpers; gen_++;
        }
        for (int w = 0; w < helpers; w++) cv_work_[w].notify_one();       // (only the workers this pass uses wake up)
        body(&job);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
        return true;
    }

private:
    explicit HostPool(int max_threads)
    {
        int hc = (int)std::thread::hardware_concurrency();
        workers_ = (size_t)std::max(0, std::min(hc, max_threads) - 1);
        cv_work_.reset(new std::condition_variable[workers_ ? workers_ : 1]);
        for (size_t w = 0; w < workers_; w++) std::thread([this, w] { loop((int)w); }).detach();
    }
    void loop(int w)
    {
        unsigned long long seen = 0;
        for (;;) {
            void *job; void (*body)(void *);
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_[w].wait(lk, [&] { return gen_ != seen && w < wanted_; });
                seen = gen_;
                job = job_; body = body_;
            }
            body(job);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) cv_done_.notify_one();
            }
        }
    }
    std::mutex run_mu_, mu_;
    std::unique_ptr<std::condition_variable[]> cv_work_;       // one per worker
    std::condition_variable cv_done_;
    void *job_ = nullptr;
    void (*body_)(void *) = nullptr;
    int wanted_ = 0, pending_ = 0;
    unsigned long long gen_ = 0;
    size_t workers_ = 0;
};

// the process-wide cap on the part of a device's workgroup slots a persistent launch may hold
// (visma_icp_set_persistent_cu_share; VISMA_ICP_PERSIST_CU_SHARE)
inline std::atomic<double> &persist_cu_share_ref()
{
    static std::atomic<double> share([] {
        const char *e = std::getenv("VISMA_ICP_PERSIST_CU_SHARE");
        const double v = e ? std::atof(e) : 1.0;
        return (v > 0.0 && v <= 1.0) ? v : 1.0;
    }());
    return share;
}

// (one flag for every instantiation of parallel_for below: a pass started from inside a pass with ANOTHER functor type
//  must be seen as nested too -- a flag inside the template is per functor type)
inline bool &parallel_for_inside()
{
    static thread_local bool inside = false;
    return inside;
}

// Run fn(i) for i in [0, n) on a few host threads (packing / ordering of clouds).
template <typename F>
void parallel_for(int64_t n, int64_t min_per_thread, F fn)
{
    int64_t nt = (int64_t)std::thread::hardware_concurrency();
    if (nt > 32) nt = 32;
    if (nt < 1) nt = 1;
    if (n / (min_per_thread > 0 ? min_per_thread : 1) < nt) nt = std::max<int64_t>(1, n / (min_per_thread > 0 ? min_per_thread : 1));
    // (a pass started from inside a pass -- fn itself calling parallel_for -- runs on the calling thread: the pools'
    //  run mutex is not recursive)
    bool &inside = parallel_for_inside();
    if (nt <= 1 || inside) { for (int64_t i = 0; i < n; i++) fn(i); return; }
    struct Guard { bool &f; explicit Guard(bool &x) : f(x) { f = true; } ~Guard() { f = false; } } guard(inside);
    HostPool::instance(0);                                         // (starts the pools in the first process that asks)
    if (HostPool::usable())
        for (int k = 0; k < HostPool::kPools; k++)
            if (HostPool::instance(k).run(n, (int)nt, fn)) return;
    if (nt > 16) nt = 16;
    std::vector<std::thread> th;
    std::atomic<int64_t> next(0);
    for (int64_t t = 0; t < nt; t++)
        th.emplace_back([&]() {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n) break;
                fn(i);
            }
        });
    for (auto &t : th) t.join();
}

constexpr int64_t kHostChunk = 16384;   // points per work item of the host passes below


// VISMA_ICP_BATCH_TRACE=1: wall-clock marks of the batch path's host stages on stderr (measurement aid)
struct StageTrace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    const char *who;
    explicit StageTrace(const char *w) : on(std::getenv("VISMA_ICP_BATCH_TRACE") != nullptr), who(w)
    {
        if (on) t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[%s] %-28s +%8.1f us  (%8.1f)\n", who, what,
                     std::chrono::duration<double, std::micro>(now - last).count(),
                     std::chrono::duration<double, std::micro>(now - t0).count());
        last = now;
    }
};

class Engine {
public:
    virtual ~Engine() {}
    virtual int set_source(const float *xyzw, int64_t ns) = 0;
    virtual int set_target(const float *xyzw, int64_t nt) = 0;
    virtual int set_target_normals(const float *nxyzw, int64_t nt) = 0;
    virtual int set_source_device(const void *, int64_t) { err_ = "not supported by this engine"; return VISMA_ICP_ERR_STATE; }
    virtual int set_target_device(const void *, int64_t) { err_ = "not supported by this engine"; return VISMA_ICP_ERR_STATE; }
    virtual int nn_pass(const Mat4 &Tc, double max_dist) = 0;
    // `offset` shifts the frame the statistics are expressed in (p+offset,
    // q+offset): zero = centred frame, the cloud centre = the caller's frame.
    virtual int reduce(const Mat4 &Tc, bool plane, const double offset[3], double *stats) = 0;
    virtual int get_correspondences(int32_t *idx, float *d2) = 0;
    // The host loop of ONE registration announces itself: between loop_begin(n) and loop_end() the caller runs at most
    // n passes (nn_pass + reduce, nothing else) -- an engine may then keep ONE launch alive across them (HipEngine:
    // the persistent certificate kernel).  loop_end() must follow on every path; LoopScope does that.
    virtual void set_persistent(int /*enabled*/, double /*timeout_ms*/) {}
    virtual void set_ring_search(int /*mode*/) {}
    virtual void get_ring_search(int *rings, double *cell, double *occupancy) const
    {
        if (rings) *rings = 0;
        if (cell) *cell = 0.0;
        if (occupancy) *occupancy = 0.0;
    }
    virtual void get_persistent_info(visma_icp_persistent_info *out) const { (void)out; }
    virtual void get_sweep_info(double *launches, double *aborts) const { if (launches) *launches = 0.0; if (aborts) *aborts = 0.0; }
    virtual void stall_command(int /*nth*/, double /*ms*/) {}
    virtual bool loop_across_ranks_ok() const { return false; }   // source-sharded ranks may keep a launch alive across passes too
    virtual void loop_begin(int /*max_passes*/) {}
    virtual int loop_end() { return VISMA_ICP_OK; }
    struct LoopScope {
        Engine *e;
        LoopScope(Engine *eng, int max_passes) : e(eng) { e->loop_begin(max_passes); }
        ~LoopScope() { (void)e->loop_end(); }
        LoopScope(const LoopScope &) = delete;
        LoopScope &operator=(const LoopScope &) = delete;
    };
    virtual int comm_init(int, int, const void *) { err_ = "RCCL needs the HIP engine"; return VISMA_ICP_ERR_STATE; }
    // The whole loop on the device (no per-iteration host round trip).
    struct LoopParams {
        Mat4 Tc0;
        double centre[3];
        double max_dist, rel_fit, rel_rmse;
        int max_iter, solver, passes;   // passes = NN passes to enqueue at most
        bool scaling, plane, world, check_stop;
        int64_t ns_total;
    };
    struct LoopResult {
        Mat4 Tc;
        double fit, rmse;
        int64_t k;
        int iters, passes;
    };
    virtual bool supports_device_loop() const { return false; }
    // nprob problems over the SAME clouds (own initial transform each; lp.Tc0 is
    // ignored when Tc0s is given).  select_problem() picks whose correspondences
    // get_correspondences() returns afterwards.
    virtual int run_loop(const LoopParams &, const Mat4 *, int, LoopResult *) { err_ = "no device loop"; return VISMA_ICP_ERR_STATE; }
    virtual void select_problem(int) {}
    // A batch of problems with their OWN clouds, advanced together on the device.
    struct BatchProblem {
        const float *src_xyzw; int64_t ns;
        const float *tgt_xyzw; int64_t nt;
        Mat4 Tc0;
        double centre[3];
        double max_dist;
        float bb_min[3], bb_max[3];   // bounding box of the (centred) target
        // problems that share clouds (24 yaw starts of one model, every model against the
        // same scene): index of an EARLIER problem whose uploaded source / built grid is reused
        int src_share = -1, grid_share = -1;
        // f64 copies for the double-precision search (all problems of a batch or none)
        const Pt64 *src64 = nullptr, *tgt64 = nullptr;
        // the target as the caller's f64 array (stride 3) instead of tgt_xyzw / tgt64: uploaded as it is and
        // expanded on the device around `centre` (tgt_f64: with the f64 copy)
        const double *tgt_raw = nullptr;
        bool tgt_f64 = false;
        // target normals (point-to-plane batches: every problem or none), indexed like the target
        const float *nrm_xyzw = nullptr;
        const Pt64 *nrm64 = nullptr;
    };
    virtual int run_loop_batch(const LoopParams &, const std::vector<BatchProblem> &, LoopResult *)
    {
        err_ = "no batched device loop";
        return VISMA_ICP_ERR_STATE;
    }
    virtual int set_nn_mode(int mode) { return mode == VISMA_ICP_NN_AUTO ? VISMA_ICP_OK : VISMA_ICP_ERR_STATE; }
    virtual int nn_mode_used() const { return VISMA_ICP_NN_AUTO; }
    virtual int search_kernel_used() const { return 0; }
    virtual int forget_winners() { return VISMA_ICP_OK; }
    virtual void set_profiling(int) {}
    virtual void get_timing(visma_icp_timing *t, bool) { std::memset(t, 0, sizeof(*t)); }
    virtual void launch_config(int *tiles, int *splits) { *tiles = 0; *splits = 0; }
    virtual bool has_device_allreduce() const { return false; }
    virtual bool shard_loop_on_device() const { return false; }
    // aux entry points (voxel / mesh steps) run on THIS context's device and stream
    virtual int bind_device() { return VISMA_ICP_OK; }
    virtual hipStream_t aux_stream() { return nullptr; }
    virtual int ipc_export(void *) { err_ = "the peer-to-peer all-reduce needs the HIP engine"; return VISMA_ICP_ERR_STATE; }
    virtual int ipc_init(int, int, const void *) { err_ = "the peer-to-peer all-reduce needs the HIP engine"; return VISMA_ICP_ERR_STATE; }
    // f64 copies of the clouds for the double-precision search (after set_source / set_target;
    // same order as those: source in Morton order).  nullptr pair = drop them.
    virtual int set_clouds64(const Pt64 *, const Pt64 *) { err_ = "double-precision search needs the HIP engine"; return VISMA_ICP_ERR_STATE; }
    // The target straight from the caller's f64 array (stride doubles per point): uploaded as it is (24 bytes
    // per point instead of 16 + 32) and expanded on the device into the fp32 copy (float)(x - c) and, with
    // want64, the f64 copy {x - c, index}.  set_source64 then completes the pair of f64 clouds.
    virtual int set_target_f64(const double *, int64_t, int, double * /* centre: in, or out when computed */, bool /* compute */, bool) { return VISMA_ICP_ERR_STATE; }
    // the target = VoxelDownSample(scene, voxel), down-sampled and installed without leaving the device
    virtual int set_target_voxel_f64(const double *, int64_t, int, double /* voxel */, double * /* centre */, bool, bool, int64_t * /* nt */) { return VISMA_ICP_ERR_STATE; }
    virtual int get_voxel_target(double *, int64_t) { return VISMA_ICP_ERR_STATE; }
    // the source sampled from meshes on the device (HIP engine)
    struct MeshSource {
        const double *V; int64_t nv; const int32_t *F; int64_t nf; int64_t samples;
        int has_transform; double T[16];
    };
    virtual int set_source_meshes_f64(const MeshSource *, int, int /* quirks */, unsigned long long /* seed */, const double * /* centre */,
                                      bool, std::vector<int32_t> &, int64_t * /* ns */) { return VISMA_ICP_ERR_STATE; }
    virtual int get_mesh_source(double *, int64_t) { return VISMA_ICP_ERR_STATE; }
    // between the target upload and set_source_f64(ns points, want64): build the search structure for `max_dist` NOW,
    // on the stream, so that it runs while the host stages the source (HIP engine; others: nothing to do)
    virtual int prepare_search(int64_t /* ns */, bool /* want64 */, double /* max_dist */) { return VISMA_ICP_OK; }
    virtual int set_source64(const Pt64 *) { return VISMA_ICP_ERR_STATE; }
    // The source likewise: raw f64 up, then expanded, Morton-ordered and gathered on the device (order.hip);
    // `order` receives the original index of the point at every position.
    virtual int set_source_f64(const double *, int64_t, int, const double *, bool, std::vector<int32_t> &) { return VISMA_ICP_ERR_STATE; }
    virtual bool search_is_f64() const { return false; }
    virtual bool search_is_exact() const { return false; }
    virtual void set_exact(bool) {}
    virtual int set_target_normals64(const Pt64 *) { return VISMA_ICP_OK; }     // f64 normals for the f64 search
    // Host staging for the packed (x,y,z,0) fp32 clouds handed to set_source / set_target.
    // The HIP engine returns pinned memory (grow-only), so the upload runs at link speed.
    virtual float *staging(int slot, size_t nfloats)
    {
        std::vector<float> &v = stage_[slot & 3];
        if (v.size() < nfloats) v.resize(nfloats);
        return v.data();
    }
    std::vector<float> stage_[4];     // slots 0/1: fp32 target / source; 2/3: their f64 copies
    // target-sharded ranks: this engine holds targets [offset, offset + nt) of the global cloud
    virtual int set_target_shard(int64_t, int64_t) { err_ = "target sharding needs the HIP engine"; return VISMA_ICP_ERR_STATE; }
    virtual void set_minreduce(visma_icp_minreduce_fn, void *) {}
    const std::string &error() const { return err_; }
    int64_t ns() const { return ns_; }
    int64_t nt() const { return nt_; }
    bool has_normals() const { return has_normals_; }

protected:
    std::string err_;
    int64_t ns_ = 0, nt_ = 0;
    bool has_normals_ = false;
};


// the product engine (hip_engine.cpp); NULL + *rc / *err on failure (no gfx950 device, HIP error)
Engine *new_hip_engine(int device, int *rc, std::string *err);

}  // namespace drv
}  // namespace visma
