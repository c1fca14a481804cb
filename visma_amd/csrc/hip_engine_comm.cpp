// hip_engine_comm.cpp -- HipEngine: the transports of the source- and target-sharded modes: IPC mailboxes, RCCL, the target-shard exchanges.
#include "hip_engine.hpp"

namespace visma {
namespace drv {

int HipEngine::set_target_shard(int64_t offset, int64_t global_nt)
{
    if (offset < 0 || global_nt < 0 || global_nt > 0x7fffffffll) {
        err_ = "bad target shard (global indices must fit 31 bits)";
        return VISMA_ICP_ERR_INVALID;
    }
    tshard_ = global_nt > 0;
    tgt_offset_ = offset;
    tgt_global_ = global_nt;
    return VISMA_ICP_OK;
}

int HipEngine::shard_exchange(const Xform64 &T64, bool plane, const double offset[3], double *pub, unsigned long long seq)
{
    if (tgt_offset_ + nt_ > tgt_global_) { err_ = "target shard exceeds the global target"; return VISMA_ICP_ERR_INVALID; }
    if (!comm_ && !minreduce_) { err_ = "target-sharded mode needs visma_icp_comm_init or visma_icp_set_minreduce"; return VISMA_ICP_ERR_STATE; }
    if (ns_ > gkeys_cap_) {
        free_dev(d_gkeys_); free_dev(d_claim_);
        HIP_TRY(hipMalloc(&d_gkeys_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
        HIP_TRY(hipMalloc(&d_claim_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
        gkeys_cap_ = ns_;
    }
    // MIN over the ranks of `keys` (RCCL on the stream, or the host-supplied exchange)
    auto min_reduce = [&](void *keys) -> int {
        if (comm_) {
            int rc = g_rccl.AllReduce(keys, keys, (size_t)ns_, kNcclUint64, kNcclMin, comm_, stream_);
            if (rc != 0) {
                err_ = std::string("ncclAllReduce(min): ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
                return VISMA_ICP_ERR_RCCL;
            }
            return VISMA_ICP_OK;
        }
        h_gkeys_.resize((size_t)ns_);
        HIP_TRY(hipMemcpyAsync(h_gkeys_.data(), keys, sizeof(unsigned long long) * ns_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        if (ns_ > 0 && minreduce_(minreduce_user_, (uint64_t *)h_gkeys_.data(), ns_) != 0) {
            err_ = "min-reduce callback failed";
            return VISMA_ICP_ERR_ENGINE;
        }
        HIP_TRY(hipMemcpyAsync(keys, h_gkeys_.data(), sizeof(unsigned long long) * ns_, hipMemcpyHostToDevice, stream_));
        return VISMA_ICP_OK;
    };
    if (shard_d64()) {
        // Every shard ran the EXACT search: compare the shards in f64.  (1) MIN of the f64 d2 bits =
        // the global nearest distance; (2) MIN of the global index over the shards that hold it =
        // lowest index on exact ties, like one GPU; the owner accumulates from the f64 coordinates.
        HIP_TRY(launch_shard_keys64((const int32_t *)d_idx_, (const double *)d_d64_, ns_, (unsigned long long *)d_gkeys_, stream_));
        int rc = min_reduce(d_gkeys_);
        if (rc) return rc;
        HIP_TRY(launch_shard_claim64((const int32_t *)d_idx_, (const double *)d_d64_, (const unsigned long long *)d_gkeys_,
                                     ns_, (unsigned)tgt_offset_, (unsigned long long *)d_claim_, stream_));
        rc = min_reduce(d_claim_);
        if (rc) return rc;
        int nb = 1;
        HIP_TRY(launch_shard_accumulate64((const Pt64 *)d_src64_, ns_, (const unsigned long long *)d_gkeys_,
                                          (const unsigned long long *)d_claim_, (const Pt64 *)d_tgt64_, nt_,
                                          (unsigned)tgt_offset_, (const float4 *)d_nrm_, (const Pt64 *)d_nrm64_, T64,
                                          offset, r2d_, plane ? 1 : 0, (int32_t *)d_idx_, (float *)d_d2_,
                                          (double *)d_partials_, reduce_max_blocks(), &nb, stream_));
        HIP_TRY(launch_finalize((const double *)d_partials_, nb, plane ? 1 : 0, (double *)d_stats_, stream_, pub, seq));
        return VISMA_ICP_OK;
    }
    HIP_TRY(launch_shard_keys((const int32_t *)d_idx_, (const float *)d_d2_, ns_, (unsigned)tgt_offset_,
                              (unsigned long long *)d_gkeys_, stream_));
    {
        int rc = min_reduce(d_gkeys_);
        if (rc) return rc;
    }
    int nblocks = 1;
    HIP_TRY(launch_shard_accumulate((const float4 *)d_src_, ns_, (const unsigned long long *)d_gkeys_,
                                    (const float4 *)d_tgt_, nt_, (unsigned)tgt_offset_, (const float4 *)d_nrm_,
                                    T64, offset, r2f_, plane ? 1 : 0, (int32_t *)d_idx_, (float *)d_d2_,
                                    (double *)d_partials_, reduce_max_blocks(), &nblocks, stream_));
    HIP_TRY(launch_finalize((const double *)d_partials_, nblocks, plane ? 1 : 0, (double *)d_stats_, stream_, pub, seq));
    return VISMA_ICP_OK;
}

int HipEngine::shard_exchange_on_stream(const DevIcpState *st, int plane, int *nblocks)
{
    if (tgt_offset_ + nt_ > tgt_global_) { err_ = "target shard exceeds the global target"; return VISMA_ICP_ERR_INVALID; }
    if (ns_ > gkeys_cap_) {
        free_dev(d_gkeys_); free_dev(d_claim_);
        HIP_TRY(hipMalloc(&d_gkeys_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
        HIP_TRY(hipMalloc(&d_claim_, sizeof(unsigned long long) * std::max<int64_t>(ns_, 1)));
        gkeys_cap_ = ns_;
    }
    auto min_reduce = [&](void *keys) -> int {
        int rc = g_rccl.AllReduce(keys, keys, (size_t)ns_, kNcclUint64, kNcclMin, comm_, stream_);
        if (rc != 0) {
            err_ = std::string("ncclAllReduce(min): ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
            return VISMA_ICP_ERR_RCCL;
        }
        return VISMA_ICP_OK;
    };
    HIP_TRY(launch_shard_keys64((const int32_t *)d_idx_, (const double *)d_d64_, ns_, (unsigned long long *)d_gkeys_, stream_));
    int rc = min_reduce(d_gkeys_);
    if (rc) return rc;
    HIP_TRY(launch_shard_claim64((const int32_t *)d_idx_, (const double *)d_d64_, (const unsigned long long *)d_gkeys_,
                                 ns_, (unsigned)tgt_offset_, (unsigned long long *)d_claim_, stream_));
    rc = min_reduce(d_claim_);
    if (rc) return rc;
    const Xform64 T64{};
    HIP_TRY(launch_shard_accumulate64((const Pt64 *)d_src64_, ns_, (const unsigned long long *)d_gkeys_,
                                      (const unsigned long long *)d_claim_, (const Pt64 *)d_tgt64_, nt_,
                                      (unsigned)tgt_offset_, (const float4 *)d_nrm_, (const Pt64 *)d_nrm64_, T64,
                                      nullptr, r2d_, plane, (int32_t *)d_idx_, (float *)d_d2_,
                                      (double *)d_partials_, reduce_max_blocks(), nblocks, stream_, st));
    return VISMA_ICP_OK;
}

int HipEngine::comm_init(int rank, int nranks, const void *id)
{
    HIP_TRY(hipSetDevice(device_));
    if (!g_rccl.load()) { err_ = g_rccl.error; return VISMA_ICP_ERR_RCCL; }
    NcclId nid;
    std::memcpy(&nid, id, sizeof(nid));
    int rc = g_rccl.CommInitRank(&comm_, nranks, nid, rank);
    if (rc != 0) {
        err_ = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
        comm_ = nullptr;
        return VISMA_ICP_ERR_RCCL;
    }
    return VISMA_ICP_OK;
}

// ---- one-shot all-reduce through IPC-mapped mailboxes (kernels.hip: ipc_allreduce_kernel) ----
int HipEngine::ensure_mailbox()
{
    if (d_mbox_) return VISMA_ICP_OK;
    const size_t bytes = sizeof(double) * 2 * kNStats * kIpcMaxRanks * 2;   // two halves, see ipc_allreduce_kernel
    // uncached device memory: remote stores and local polls both go to memory
    if (hipExtMallocWithFlags(&d_mbox_, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipExtMallocWithFlags(&d_mbox_, bytes, hipDeviceMallocFinegrained));
    }
    HIP_TRY(hipMemset(d_mbox_, 0, bytes));
    HIP_TRY(hipMalloc(&d_ipc_flag_, 16));                    // {int timeout flag, pad, u64 exchange counter}
    HIP_TRY(hipMemset(d_ipc_flag_, 0, 16));
    return VISMA_ICP_OK;
}

int HipEngine::ipc_export(void *out)
{
    HIP_TRY(hipSetDevice(device_));
    int rc = ensure_mailbox();
    if (rc) return rc;
    // Exporting starts a NEW session: peers only learn the handle after this call, so nothing can be on its
    // way into the mailbox yet -- drop the mappings of an earlier session, clear the granules it left (a
    // retry after a failed handshake must not read them as this session's) and restart the count.
    HIP_TRY(hipStreamSynchronize(stream_));
    for (int r = 0; r < kIpcMaxRanks; r++) {
        if (r != ipc_rank_ && peers_.box[r]) (void)hipIpcCloseMemHandle(peers_.box[r]);
        peers_.box[r] = nullptr;
    }
    ipc_n_ = 0;
    HIP_TRY(hipMemset(d_mbox_, 0, sizeof(double) * 2 * kNStats * kIpcMaxRanks * 2));
    HIP_TRY(hipMemset(d_ipc_flag_, 0, 16));
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, d_mbox_));
    static_assert(sizeof(h) <= VISMA_ICP_IPC_HANDLE_BYTES, "handle size");
    std::memset(out, 0, VISMA_ICP_IPC_HANDLE_BYTES);
    std::memcpy(out, &h, sizeof(h));
    return VISMA_ICP_OK;
}

int HipEngine::ipc_init(int rank, int nranks, const void *handles)
{
    HIP_TRY(hipSetDevice(device_));
    if (nranks < 1 || nranks > kIpcMaxRanks || rank < 0 || rank >= nranks) { err_ = "bad rank arguments"; return VISMA_ICP_ERR_INVALID; }
    int rc = ensure_mailbox();
    if (rc) return rc;
    for (int r = 0; r < nranks; r++) {
        if (r == rank) { peers_.box[r] = d_mbox_; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char *)handles + (size_t)r * VISMA_ICP_IPC_HANDLE_BYTES, sizeof(h));
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < r; q++)
                if (q != rank && peers_.box[q]) { (void)hipIpcCloseMemHandle(peers_.box[q]); peers_.box[q] = nullptr; }
            err_ = std::string("hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e);
            return VISMA_ICP_ERR_HIP;
        }
        peers_.box[r] = p;
    }
    ipc_rank_ = rank;
    ipc_n_ = nranks;
    // Handshake (the call is collective): one all-reduce of known values proves that every peer's
    // stores arrive in this rank's mailbox and the other way round -- a mapping that opens but does
    // not carry traffic (no peer access between two devices) must fail HERE, not in the first iteration.
    if (nranks > 1) {
        double *d_hs = nullptr;
        HIP_TRY(hipMalloc((void **)&d_hs, sizeof(double) * kNStats));
        std::vector<double> hs((size_t)kNStats);
        for (int a = 0; a < kNStats; a++) hs[(size_t)a] = (double)((rank + 1) * (a + 1));
        hipError_t e = hipMemcpyAsync(d_hs, hs.data(), sizeof(double) * kNStats, hipMemcpyHostToDevice, stream_);
        if (e == hipSuccess) e = hipMemsetAsync(d_ipc_flag_, 0, sizeof(int), stream_);
        if (e == hipSuccess)
            e = launch_ipc_allreduce(d_hs, d_hs, peers_, ipc_rank_, ipc_n_, ipc_seq_dev(), nullptr, 0, (int *)d_ipc_flag_,
                                     stream_, kIpcHandshakeSpins);
        int flag = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(hs.data(), d_hs, sizeof(double) * kNStats, hipMemcpyDeviceToHost, stream_);
        if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_ipc_flag_, sizeof(int), hipMemcpyDeviceToHost, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        (void)hipFree(d_hs);
        bool good = e == hipSuccess && flag == 0;
        const double tri = 0.5 * (double)nranks * (double)(nranks + 1);
        for (int a = 0; good && a < kNStats; a++) good = hs[(size_t)a] == tri * (double)(a + 1);
        if (!good) {
            (void)hipGetLastError();
            (void)hipMemset(d_ipc_flag_, 0, sizeof(int));
            for (int q = 0; q < nranks; q++)
                if (q != rank && peers_.box[q]) { (void)hipIpcCloseMemHandle(peers_.box[q]); peers_.box[q] = nullptr; }
            ipc_n_ = 0;
            err_ = e != hipSuccess ? std::string("peer-to-peer handshake: ") + hipGetErrorString(e)
                 : flag ? "peer-to-peer handshake: rank " + std::to_string(flag - 1) + " did not answer"
                        : std::string("peer-to-peer handshake: wrong sum");
            return VISMA_ICP_ERR_HIP;
        }
        // Who sits where: a second all-reduce in which every rank contributes the PCI address of its device in its own
        // slot.  Ranks that share a device (tests run two on one GPU) must not keep persistent launches alive side by
        // side; one rank per GPU may (hip_engine.hpp: persist_ranks_ok).
        {
            int dom = 0, bus = 0, dev = 0;
            (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device_);
            (void)hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device_);
            (void)hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device_);
            (void)hipGetLastError();
            const double me = 1.0 + (double)(((long long)dom << 16) | ((long long)bus << 8) | (long long)dev);
            for (int a = 0; a < kNStats; a++) hs[(size_t)a] = a == rank ? me : 0.0;
            peers_share_device_ = true;
            // (the exchange runs in the statistics buffer every context owns: no allocation that could fail on one rank
            //  only -- a rank that skipped this collective would leave the mailbox call count, and with it the half in use,
            //  out of step with its peers for every later pass)
            double *d_id = (double *)d_stats_;
            {
                hipError_t e2 = hipMemcpyAsync(d_id, hs.data(), sizeof(double) * kNStats, hipMemcpyHostToDevice, stream_);
                if (e2 == hipSuccess)
                    e2 = launch_ipc_allreduce(d_id, d_id, peers_, ipc_rank_, ipc_n_, ipc_seq_dev(), nullptr, 0, (int *)d_ipc_flag_,
                                              stream_, kIpcHandshakeSpins);
                if (e2 == hipSuccess) e2 = hipMemcpyAsync(hs.data(), d_id, sizeof(double) * kNStats, hipMemcpyDeviceToHost, stream_);
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(stream_);
                if (e2 == hipSuccess) {
                    // anything but a proper address in a slot (a late peer leaves NaN, a silent one 0) counts as shared:
                    // persistent launches across ranks only where every peer is KNOWN to sit elsewhere
                    bool shared = false;
                    for (int r = 0; r < nranks; r++) shared = shared || !(hs[(size_t)r] >= 1.0) || (r != rank && hs[(size_t)r] == me);
                    peers_share_device_ = shared;
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        // the mailboxes as a table in device memory (what the persistent kernel's fold indexes)
        free_dev(d_peer_table_);
        if (hipMalloc(&d_peer_table_, sizeof(void *) * kIpcMaxRanks) == hipSuccess) {
            if (hipMemcpy(d_peer_table_, peers_.box, sizeof(void *) * kIpcMaxRanks, hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipGetLastError();
                free_dev(d_peer_table_);
            }
        } else {
            (void)hipGetLastError();
            d_peer_table_ = nullptr;
        }
    }
    return VISMA_ICP_OK;
}

}  // namespace drv
}  // namespace visma
