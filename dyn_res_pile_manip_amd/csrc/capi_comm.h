// capi_comm.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: RCCL entry points (drp_comm_*).

// ---- RCCL -------------------------------------------------------------------------------------
namespace {
RcclApi* need_rccl(drp_ctx* c) {
    RcclApi* R = rccl_api();
    if (!R) (void)fail(c, DRP_ECOMM, "RCCL is not available: %s", g_rccl.error.c_str());
    return R;
}
// ids this process has already built a communicator from: a ncclUniqueId serves ONE ncclCommInitRank per rank --
// a second one with the same id never completes
std::mutex g_used_ids_mu;
std::vector<std::string> g_used_ids;
}  // namespace

int drp_comm_unique_id(char* id128) {
    if (!id128) return DRP_EINVAL;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    RcclApi* R = need_rccl(nullptr);
    if (!R) return DRP_ECOMM;
    ncclUniqueId id;
    ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, DRP_ECOMM, "ncclGetUniqueId: %s", R->GetErrorString(r));
    memcpy(id128, &id, 128);
    return DRP_OK;
}

int drp_comm_init(drp_ctx* c, const char* id128, int rank, int n_ranks) {
    if (!c || !id128 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(c, DRP_EINVAL, "bad comm args");
    RcclApi* R = need_rccl(c);
    if (!R) return DRP_ECOMM;
    HIPCHK(c, hipSetDevice(c->device));
    // an id serves ONE ncclCommInitRank per rank (a second one with the same id never returns): keyed on (id, rank) -- the
    // ranks of one process, a context per GPU, share their id --, looked up first, recorded when the call is about to go out
    const std::string key = std::string(id128, 128) + ":" + std::to_string(rank);
    {
        std::lock_guard<std::mutex> lk(g_used_ids_mu);
        for (const std::string& u : g_used_ids)
            if (u == key) return fail(c, DRP_ECOMM, "this ncclUniqueId has already been used for rank %d's communicator in this process: "
                                      "every communicator needs a fresh id from rank 0 (drp_comm_unique_id)", rank);
    }
    if (c->comm) { CHK(guarded_wait(c, nullptr)); if (c->comm) (void)R->CommDestroy(c->comm); c->comm = nullptr; }
    c->comm_failed = false;
    c->n_ranks = 1;
    c->rank = 0;
    // every rank has to arrive: the call runs on a helper thread so that a missing peer costs a deadline
    // (DRP_COMM_INIT_TIMEOUT_S), not the process; a helper nobody waits for any more aborts what it finally gets
    struct InitState { std::atomic<int> state{0} /* 0 waiting, 1 finished, 2 given up */; ncclComm_t comm = nullptr; ncclResult_t res = ncclSuccess; };
    auto stt = std::make_shared<InitState>();
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const int dev = c->device;
    {
        std::lock_guard<std::mutex> lk(g_used_ids_mu);
        g_used_ids.push_back(key);
    }
    auto hs = helper_register(c);
    std::thread([stt, R, id, rank, n_ranks, dev, hs] {
        (void)hipSetDevice(dev);
        stt->res = R->CommInitRank(&stt->comm, n_ranks, id, rank);
        int waiting = 0;
        if (!stt->state.compare_exchange_strong(waiting, 1, std::memory_order_acq_rel) && stt->res == ncclSuccess && stt->comm)
            (void)R->CommAbort(stt->comm);            // the caller has given up: nobody will ever own this communicator
        hs->done.store(1, std::memory_order_release);
    }).detach();
    const double t0 = now_s();
    while (stt->state.load(std::memory_order_acquire) == 0) {
        if (now_s() - t0 > c->comm_init_timeout_s) {
            int waiting = 0;
            if (!stt->state.compare_exchange_strong(waiting, 2, std::memory_order_acq_rel)) break;    // it arrived just now
            return fail(c, DRP_ECOMM, "ncclCommInitRank: rank %d waited %.0f s for the other %d rank(s) (DRP_COMM_INIT_TIMEOUT_S)",
                        rank, c->comm_init_timeout_s, n_ranks - 1);
        }
        usleep(200);
    }
    if (stt->res != ncclSuccess) return fail(c, DRP_ECOMM, "ncclCommInitRank: %s", R->GetErrorString(stt->res));
    c->comm = stt->comm;
    c->rank = rank;
    c->n_ranks = n_ranks;
    return DRP_OK;
}

int drp_comm_info(drp_ctx* c, int* n_ranks, int* rank, int* version, char* path, size_t path_len) {
    if (!c) return DRP_EINVAL;
    if (n_ranks) *n_ranks = 0;
    if (rank) *rank = -1;
    if (version) *version = 0;
    if (path && path_len) path[0] = 0;
    RcclApi* R = need_rccl(c);
    if (!R) return DRP_ECOMM;
    if (version) *version = R->version;
    if (path && path_len) snprintf(path, path_len, "%s", R->path.c_str());
    if (c->comm) {
        int v = 0;
        ncclResult_t r = R->CommCount(c->comm, &v);
        if (r != ncclSuccess) return fail(c, DRP_ECOMM, "ncclCommCount: %s", R->GetErrorString(r));
        if (n_ranks) *n_ranks = v;
        r = R->CommUserRank(c->comm, &v);
        if (r != ncclSuccess) return fail(c, DRP_ECOMM, "ncclCommUserRank: %s", R->GetErrorString(r));
        if (rank) *rank = v;
    }
    return DRP_OK;
}

int drp_comm_allgather(drp_ctx* c, const void* send, size_t bytes, void* recv) {
    if (!c || !send || !recv || bytes == 0) return fail(c, DRP_EINVAL, "bad all-gather arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm_failed) return comm_failed_error(c);
    if (!comm_live(c)) {
        memcpy(recv, send, bytes);
        return DRP_OK;
    }
    RcclApi* R = rccl_api();
    CHK(ensure(c, c->xchg, bytes * (size_t)(c->n_ranks + 1)));
    char* dsend = static_cast<char*>(c->xchg.p);
    char* drecv = dsend + bytes;
    HIPCHK(c, hipMemcpyAsync(dsend, send, bytes, hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = R->AllGather(dsend, drecv, bytes, ncclChar, c->comm, c->stream);
    if (r != ncclSuccess) return fail(c, DRP_ECOMM, "ncclAllGather: %s", R->GetErrorString(r));
    CHK(d2h(c, recv, drecv, bytes * (size_t)c->n_ranks));
    return drp_sync(c);
}

int drp_comm_destroy(drp_ctx* c) {
    if (!c) return DRP_EINVAL;
    helpers_wait(5.0, c);                             // an abort of this context still draining the device
    int rc = DRP_OK;
    if (c->comm) {
        RcclApi* R = rccl_api();
        rc = guarded_wait(c, nullptr);                // aborts the communicator itself when the wait gives up (and reports it)
        if (c->comm && R) (void)R->CommDestroy(c->comm);
    }
    c->comm = nullptr;
    c->n_ranks = 1;
    c->rank = 0;
    c->comm_failed = false;                           // the caller has seen the failure and goes on alone -- AFTER the wait above,
    return rc;                                        // whose own give-up would have raised the flag again
}
