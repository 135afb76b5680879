// capi_pipeline.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: internal helpers and pipelines: guarded waits, probes, the graph build, one predict_one_step on every engine (run_step), reward, rollouts (run_rollout), weight packing, deferred weight gradients, the range check.

namespace {

int fail(drp_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail((c), DRP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

#define CHK(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != DRP_OK) return rc_; \
    } while (0)

int guarded_wait(drp_ctx* c, hipEvent_t ev);
int ensure(drp_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return DRP_OK;
    // hipFree waits for the device: behind a collective that cannot finish it would never return
    if (b.p && c && c->comm != nullptr && (c->n_ranks > 1 || c->comm_always)) CHK(guarded_wait(c, nullptr));
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) return fail(c, DRP_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    b.cap = bytes;
    return DRP_OK;
}

template <typename T>
T* ptr(const DevBuf& b) { return static_cast<T*>(b.p); }

int h2d(drp_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    CHK(ensure(c, b, bytes));
    HIPCHK(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return DRP_OK;
}

int d2h(drp_ctx* c, void* dst, const void* src, size_t bytes) {
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return DRP_OK;
}

// ---- waits that cannot hang on a dead peer --------------------------------------------------------------------
// With a communicator attached, the stream may hold an ncclAllGather that never completes (a rank died, a rank took
// another branch).  Every host wait of the context then polls instead of blocking: the stream / event, the
// communicator's asynchronous error, and a deadline (DRP_COMM_TIMEOUT_S, default 60 s).  On error or timeout the
// communicator is ABORTED (ncclCommAbort ends the collective's kernel on this rank), the context falls back to one
// rank and the call returns DRP_ECOMM: the process can report and exit instead of sitting in hipStreamSynchronize.
bool comm_live(const drp_ctx* c) { return c->comm != nullptr && (c->n_ranks > 1 || c->comm_always); }

// Helper threads (ncclCommAbort behind a dead collective, ncclCommInitRank waiting for its peers) are tracked: drp_destroy,
// drp_comm_destroy and process exit give them a bounded time to finish, so that none is still inside RCCL when the stream,
// the context or the HIP / RCCL libraries' own statics go away.
struct HelperState { std::atomic<int> done{0}; const void* owner = nullptr; };   // owner: the context the thread works for
std::mutex g_helpers_mu;
std::vector<std::shared_ptr<HelperState>> g_helpers;
std::shared_ptr<HelperState> helper_register(const void* owner) {
    auto h = std::make_shared<HelperState>();
    h->owner = owner;
    std::lock_guard<std::mutex> lk(g_helpers_mu);
    static bool at_exit = false;
    if (!at_exit) {
        at_exit = true;
        atexit([] {
            const double t0 = now_s();
            for (;;) {
                bool busy = false;
                { std::lock_guard<std::mutex> lk2(g_helpers_mu); for (auto& q : g_helpers) busy = busy || !q->done.load(std::memory_order_acquire); }
                if (!busy || now_s() - t0 > 5.0) return;
                usleep(1000);
            }
        });
    }
    g_helpers.erase(std::remove_if(g_helpers.begin(), g_helpers.end(), [](const std::shared_ptr<HelperState>& q) { return q->done.load() != 0; }), g_helpers.end());
    g_helpers.push_back(h);
    return h;
}
// the helper threads of ONE context (another context's communicator still waiting for its peers is not this one's business)
void helpers_wait(double seconds, const void* owner) {
    const double t0 = now_s();
    for (;;) {
        bool busy = false;
        { std::lock_guard<std::mutex> lk(g_helpers_mu); for (auto& q : g_helpers) busy = busy || (q->owner == owner && !q->done.load(std::memory_order_acquire)); }
        if (!busy || now_s() - t0 > seconds) return;
        usleep(500);
    }
}

void comm_abort(drp_ctx* c) {
    RcclApi* R = rccl_api();
    // ncclCommAbort raises the communicator's abort flag (a collective's kernel spinning on a peer sees it and ends) and
    // then waits for the device to drain: on a helper thread, so that the caller gets its error code NOW
    if (c->comm && R) {
        ncclComm_t comm = c->comm;
        const int dev = c->device;
        auto h = helper_register(c);
        std::thread([R, comm, dev, h] { (void)hipSetDevice(dev); (void)R->CommAbort(comm); h->done.store(1, std::memory_order_release); }).detach();
    }
    // the failure is STICKY: the ranks' shards are no longer combined, so nothing that would have used the communicator may
    // quietly carry on with this rank's data alone
    c->comm_failed = true;
    c->comm_failed_ranks = c->n_ranks;
    c->comm = nullptr;
    c->n_ranks = 1;
    c->rank = 0;
}
int comm_failed_error(drp_ctx* c) {
    return fail(c, DRP_ECOMM, "the communicator of %d ranks was aborted after a failed wait or an RCCL error: call drp_comm_destroy "
                "(continue alone) or drp_comm_init with a fresh id before the next collective step", c->comm_failed_ranks);
}

int guarded_wait(drp_ctx* c, hipEvent_t ev) {
    if (!comm_live(c)) {
        const hipError_t e = ev ? hipEventSynchronize(ev) : hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, DRP_EHIP, "%s failed: %s", ev ? "hipEventSynchronize" : "hipStreamSynchronize", hipGetErrorString(e));
        return DRP_OK;
    }
    RcclApi* R = rccl_api();
    const double t0 = now_s();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(c->stream);
        if (e == hipSuccess) return DRP_OK;
        if (e != hipErrorNotReady) return fail(c, DRP_EHIP, "%s failed: %s", ev ? "hipEventQuery" : "hipStreamQuery", hipGetErrorString(e));
        if ((spin & 63) == 63) {
            ncclResult_t ae = ncclSuccess;
            if (R && R->CommGetAsyncError(c->comm, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) {
                comm_abort(c);
                return fail(c, DRP_ECOMM, "RCCL reported an asynchronous error (%s); communicator aborted", R->GetErrorString(ae));
            }
            const double dt = now_s() - t0;
            if (dt > c->comm_timeout_s) {
                const int nr = c->n_ranks, rk = c->rank;
                comm_abort(c);
                return fail(c, DRP_ECOMM, "rank %d of %d waited %.1f s behind a collective (DRP_COMM_TIMEOUT_S=%g): a peer is gone "
                            "or took another path; communicator aborted", rk, nr, dt, c->comm_timeout_s);
            }
            if (dt > 2e-3) usleep(50);            // past the length of any iteration's tail: stop burning the core
            else sched_yield();
        }
    }
}

// RAII-less probe bracket
struct ProbeScope {
    drp_ctx* c;
    bool on;
    ProbeScope(drp_ctx* ctx, int cls) : c(ctx), on(ctx->probe_cls == cls) {
        if (on) rec();
    }
    ~ProbeScope() {
        if (on) rec();
    }
    void rec() {
        if (c->probe_used == c->probe_ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
            c->probe_ev.push_back(e);
        }
        (void)hipEventRecord(c->probe_ev[c->probe_used++], c->stream);
    }
};

int ensure_step_ws(drp_ctx* c, int B, int N, int engine = -1) {
    if (engine < 0) engine = c->engine;
    const size_t bn = (size_t)B * N;
    CHK(ensure(c, c->s_delta, bn * 3 * sizeof(float)));
    CHK(ensure(c, c->nbr_idx, bn * DRP_K * sizeof(int16_t)));
    CHK(ensure(c, c->nbr_cnt, bn));
    CHK(ensure(c, c->eff, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->c_node, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->agg, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->proj, bn * 128 * sizeof(float)));
    CHK(ensure(c, c->proj2, bn * 128 * sizeof(float)));
    // edge constants [B,N,10,64] for the engines that materialise them; the fused engine only parks the graph build's
    // sorted positions and strip starts there (launch_graph)
    const size_t graph_scratch = (size_t)B * (((size_t)N + 3) & ~(size_t)3) * 16 + (size_t)B * (GC_MAX_BANDS * GC_XS + 1) * sizeof(int);
    CHK(ensure(c, c->c_edge, engine == DRP_ENGINE_FUSED ? graph_scratch : std::max(graph_scratch, bn * DRP_K * 64 * sizeof(float))));
    c->lastB = B;
    c->lastN = N;
    return DRP_OK;
}

// Every few launches whose pairing depends on it (prop_pair), the mean in-degree of the lists just built goes to host
// memory behind the launch: the next launches of this shape read it there, without waiting for anything.
static void note_degrees(drp_ctx* c, long spw, long N, long B) {
    const long rows = spw * N;
    if (rows > c->prop_pair_rows || rows <= c->prop_pair_always) return;
    if ((c->deg_tick++ & 7u) != 0) return;
    if (!c->deg_stat) {
        if (hipHostMalloc(reinterpret_cast<void**>(&c->deg_stat), sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) {
            c->deg_stat = nullptr;
            (void)hipGetLastError();
            return;
        }
        *c->deg_stat = 0ull;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->deg_stat_dev), c->deg_stat, 0) != hipSuccess) {
            (void)hipHostFree(c->deg_stat);
            c->deg_stat = nullptr;
            (void)hipGetLastError();
            return;
        }
    }
    hipLaunchKernelGGL(k_deg_stat, dim3(1), dim3(1024), 0, c->stream, ptr<uint8_t>(c->nbr_cnt),
                       (int)std::min(B * N, (long)DEG_STAT_MAX_ROWS), (int)N, c->deg_stat_dev);
}

struct StepArgs {
    const float* s_prev; int prev_mod; size_t prev_stride;   // state read by sample b: row b % prev_mod
    const float* attr; int attr_mod;
    const float* dens; int dens_mod;
    const float* actions; size_t act_stride;                  // null: s_delta already in workspace
    bool build_graph;                                         // false: nbr lists already in workspace
    float* s_out; size_t out_stride;
    int B, N;
    // tape for the backward pass (fused engine only, km_prop<., TAPE>):
    float* eff_hist = nullptr;      // [4][B*N*64]: effect after the encoder and after every propagation step
    unsigned* mask_hist = nullptr;  // [3][B*N*10][2]: ReLU masks of the relation effects of every propagation step
    float* agg_hist = nullptr;      // [3][B*N*64]: aggregated edge effects of every propagation step (training), nullable
    const float* cself = nullptr;   // [B,64] self-edge constant + per-sample validity (fused engine, k_cself)
    const uint8_t* cself_ok = nullptr;
    bool padded = false;            // training batches: zero-padded (coincident) particles -> plain k_graph
    mutable bool encoded = false;   // run_step: the particle encoder already ran, in the neighbour lists' launch (km_graph_q4_encode)
    int* rev_off = nullptr;         // the GD planner's forward, samples of one graph chunk: the reversed lists in the lists' own launch
    int* rev = nullptr;             //   (k_graph_rev); run_step says in rev_built whether it did
    bool* rev_built = nullptr;
};

// km_prop3 / kmb_step_bwd (a workgroup owns whole samples and runs all propagation steps in one launch) or the
// per-step kernels (the tiles of all samples dealt over the chip)?  Whole samples whenever (nearly) every CU gets one --
// and for ANY batch of samples of up to 256 particles (one round of tiles per step for the workgroup's eight waves):
// a small batch is latency, and one launch per rollout step instead of five is what counts (B = 32 ... 255 at 50 / 100
// particles: 1.7 - 2.0 -> 1.0 - 1.2 ms per MPPI iteration; 300 particles: 2 - 7 % slower below 200 samples, 27 % faster
// at 255).
bool whole_samples(const drp_ctx* c, long B, int N) {
    if (c->prop3_min_b > 0) return B >= c->prop3_min_b;
    return B >= c->n_cu - c->n_cu / 5 || N <= 256;
}
int graph_chunks(int N) { return (N + GRAPH_THREADS - 1) / GRAPH_THREADS; }
// neighbour lists: x-strip variant for samples of at least two workgroups (below that a wave's range is the whole
// sample anyway), plain sweep otherwise and for zero-padded batches (coincident particles tie at the cut)
void launch_graph(drp_ctx* c, hipStream_t st, const float* s_prev, int prev_mod, size_t prev_stride, const float* actions,
                  size_t act_stride, float* s_delta, int B, int N, int16_t* nbr_idx, uint8_t* nbr_cnt, int self_first,
                  bool padded);
size_t graph_lds(int N) { return (size_t)4 * N * sizeof(float); }

// which build launch_graph picks, in its order: cells, x strips, then k_graph_q4 for a handful of samples
bool graph_takes_q4(const drp_ctx* c, int B, int N, bool padded) {
    if (c->graph_cells && c->graph_strips && !padded && N >= c->graph_cells_min_n) return false;
    if (c->graph_strips && !padded && N > GRAPH_THREADS) return false;
    return c->graph_q4 != 0 && N >= 64 && (c->graph_q4 == 2 || (long)B * ((N + 127) / 128) * 2 <= c->n_cu);
}
// km_prop3 with the particle encoder as its first phase (run_step_mfma): no encoder launch to share
bool step_has_phase_e(const drp_ctx* c, int B, int N) {
    const int spw = (int)((B + c->n_cu - 1) / c->n_cu);
    const bool prop3 = c->engine == DRP_ENGINE_FUSED && c->prop3 && whole_samples(c, B, N) && ((long)spw * N + 31) / 32 >= c->prop3_min_tiles;
    return prop3 && c->prop3e;
}
void launch_graph(drp_ctx* c, hipStream_t st, const float* s_prev, int prev_mod, size_t prev_stride, const float* actions,
                  size_t act_stride, float* s_delta, int B, int N, int16_t* nbr_idx, uint8_t* nbr_cnt, int self_first,
                  bool padded) {
    if (c->graph_cells && c->graph_strips && !padded && N >= c->graph_cells_min_n) {
        // two-dimensional cells: y bands of height hb ~ sqrt(16 / density) (a 16-receiver block of a band is then about
        // as wide as the band is high; the density of a pile spread over the 0.4 x 0.4 workspace -- any positive hb
        // gives the same lists), 1-cm x strips inside a band
        const size_t Np = ((size_t)N + 3) & ~(size_t)3;
        float4* sorted = reinterpret_cast<float4*>(c->c_edge.p);
        int* starts = reinterpret_cast<int*>(sorted + (size_t)B * Np);
        float hb = sqrtf(16.0f * 0.16f / (float)N);
        if (c->graph_cells_hb > 0.0f) hb = c->graph_cells_hb;
        int gy = (int)ceilf(0.64f / hb);
        if (gy > GC_MAX_BANDS) gy = GC_MAX_BANDS;
        if (gy < 1) gy = 1;
        const float inv_hb = (float)gy / 0.64f;
        const int ncell = gy * GC_XS;
        c->dv(DV_GRAPH_CELLS);
        hipLaunchKernelGGL(k_graph_sort2, dim3(B), dim3(GRAPH_SORT_THREADS), 0, st, s_prev, prev_mod, prev_stride, actions,
                           act_stride, s_delta, N, c->cam, gy, inv_hb, sorted, starts);
        const float halo = c->graph_cells_halo > 0.0f ? c->graph_cells_halo
                           // expected distance of the 10th neighbour in a pile of this density, with a third to spare
                           : 1.3f * sqrtf(10.0f * 0.16f / (3.14159265f * (float)N));
        // receivers are dealt to quarter waves band by band: at most N / 16 + gy quarters, 16 per workgroup
        const int chunks = ((N + 15) / 16 + gy + GC_THREADS / 16 - 1) / (GC_THREADS / 16);
        hipLaunchKernelGGL(k_graph_cells, dim3(SPREAD_GRID(B * chunks)), dim3(GC_THREADS), GRAPH_CELLS_LDS(ncell), st,
                           (const float4*)sorted, (const int*)starts, N, gy, inv_hb, nbr_idx, nbr_cnt, c->thr, chunks,
                           B * chunks, self_first, halo);
    }
    else if (c->graph_strips && !padded && N > GRAPH_THREADS) {
        // sorted positions and strip starts live in the edge-constant buffer: whatever uses it runs after the lists exist
        const size_t Np = ((size_t)N + 3) & ~(size_t)3;
        float4* sorted = reinterpret_cast<float4*>(c->c_edge.p);
        int* starts = reinterpret_cast<int*>(sorted + (size_t)B * Np);
        hipLaunchKernelGGL(k_graph_sort, dim3(B), dim3(GRAPH_SORT_THREADS), 0, st, s_prev, prev_mod, prev_stride, actions,
                           act_stride, s_delta, N, c->cam, sorted, starts);
        c->dv(N >= 800 ? DV_GRAPH_STRIPS256 : DV_GRAPH_STRIPS);
        if (N >= 800) {
            const int chunks = (N + 255) / 256;
            hipLaunchKernelGGL(k_graph_strips_q<256>, dim3(SPREAD_GRID(B * chunks)), dim3(256), GRAPH_STRIPS_LDS(N, 256), st,
                               (const float4*)sorted, (const int*)starts, N, nbr_idx, nbr_cnt, c->thr, chunks, B * chunks, self_first);
        } else {
            hipLaunchKernelGGL(k_graph_strips_q<GRAPH_THREADS>, dim3(SPREAD_GRID(B * graph_chunks(N))), dim3(GRAPH_THREADS), GRAPH_STRIPS_LDS(N, GRAPH_THREADS), st,
                               (const float4*)sorted, (const int*)starts, N, nbr_idx, nbr_cnt, c->thr, graph_chunks(N), B * graph_chunks(N), self_first);
        }
    }
    else if (graph_takes_q4(c, B, N, padded)) {
        // a handful of samples (training batches): four threads per receiver, each over a quarter of the senders
        const int chunks = (N + 127) / 128;
        c->dv(DV_GRAPH_Q4);
        hipLaunchKernelGGL(k_graph_q4, dim3((unsigned)(B * chunks)), dim3(GRAPH_Q4_THREADS), GRAPH_Q4_LDS(N), st, s_prev, prev_mod,
                           prev_stride, actions, act_stride, s_delta, N, nbr_idx, nbr_cnt, c->cam, c->thr, chunks, self_first);
    }
    else {
        c->dv(DV_GRAPH_PLAIN);
        hipLaunchKernelGGL(k_graph, dim3(SPREAD_GRID(B * graph_chunks(N))), dim3(GRAPH_THREADS), graph_lds(N), st, s_prev,
                           prev_mod, prev_stride, actions, act_stride, s_delta, N, nbr_idx, nbr_cnt, c->cam, c->thr,
                           graph_chunks(N), B * graph_chunks(N), self_first);
    }
}

void launch_aggregate(drp_ctx* c, int B, int N) {
    ProbeScope ps(c, KC_AGGREGATE);
    // a handful of samples (training batches): several workgroups per sample on the global variant
    int chunks = 1;
    if (B < c->n_cu / 2) {
        chunks = (N + 15) / 16;
        if (chunks > 2048 / B) chunks = 2048 / B;
        if (chunks < 1) chunks = 1;
    }
    c->dv((N <= K_AGG_LDS_MAX_N && !c->agg_global_only && chunks == 1) ? DV_AGGREGATE_LDS : DV_AGGREGATE);
    if (N <= K_AGG_LDS_MAX_N && !c->agg_global_only && chunks == 1)
        hipLaunchKernelGGL(k_aggregate_lds, dim3(B), dim3(512), (size_t)N * 256, c->stream,
                           ptr<float>(c->c_edge), ptr<float>(c->proj), ptr<int16_t>(c->nbr_idx),
                           ptr<uint8_t>(c->nbr_cnt), N, ptr<float>(c->agg));
    else
        hipLaunchKernelGGL(k_aggregate, dim3(B * chunks), dim3(256), 0, c->stream, ptr<float>(c->c_edge),
                           ptr<float>(c->proj), ptr<int16_t>(c->nbr_idx), ptr<uint8_t>(c->nbr_cnt), N,
                           ptr<float>(c->agg), chunks);
}

// kernels whose tile loop is workgroup-cyclic first (tile = block + grid x (wave + 8 round)): one workgroup per tile up to the chip
int mfma_grid_spread(drp_ctx* c, long ntiles) {
    const long cap = (long)c->n_cu;
    return (int)(ntiles < cap ? (ntiles > 0 ? ntiles : 1) : cap);
}
int mfma_grid(drp_ctx* c, long ntiles) {
    long blocks = (ntiles + MFMA_WAVES - 1) / MFMA_WAVES;
    long cap = (long)c->n_cu;
    return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

// MLP stages of one step on the fp32 MFMA kernels (graph already built, s_delta in workspace)
int run_step_mfma(drp_ctx* c, const StepArgs& a) {
    const int B = a.B, N = a.N;
    hipStream_t st = c->stream;
    const float* mw = ptr<float>(c->w_mfma);
    const dim3 blk(64 * MFMA_WAVES);
    const long node_tiles = (long)B * ((N + 31) / 32);
    const long edge_tiles = (long)B * ((N * DRP_K + 31) / 32);
    const size_t bn64 = (size_t)B * N * 64;
    const bool tape = a.eff_hist != nullptr;
    // the tape of the reverse-mode kernels: km_prop<., TAPE> on the fused engine; on the fp32 matrix engine (what the
    // gradient-descent planner and the trainer fall back to when the split-fp16 relation encoder refuses the weights or the
    // inputs) the stage kernels run as always and the tape is copied / written beside them (tape_mfma below)
    if (tape && c->engine != DRP_ENGINE_FUSED && c->engine != DRP_ENGINE_MFMA)
        return fail(c, DRP_ESTATE, "the backward tape is written by the fused or the fp32 matrix engine");
    const bool tape_mfma = tape && c->engine == DRP_ENGINE_MFMA;
    float* eff0 = (tape && !tape_mfma) ? a.eff_hist : ptr<float>(c->eff);
    // chip-filling batches on the fused engine: the three propagation steps are one launch (km_prop3), and the
    // particle encoder is its first phase unless switched off
    const int tps3 = (N + 31) / 32;
    const int spw = (int)((B + c->n_cu - 1) / c->n_cu);
    const bool prop3 = c->engine == DRP_ENGINE_FUSED && c->prop3 && whole_samples(c, B, N) && ((long)spw * N + 31) / 32 >= c->prop3_min_tiles;
    const bool phase_e = prop3 && c->prop3e;
    if (!phase_e && !a.encoded) {
        ProbeScope ps(c, KC_NODE_ENCODE);
        c->dv(c->engine == DRP_ENGINE_FUSED ? DV_NODE_ENCODE_SPLIT : DV_NODE_ENCODE);
        if (c->engine == DRP_ENGINE_FUSED)
            hipLaunchKernelGGL(km_node_encode_split, dim3(mfma_grid_spread(c, node_tiles)), blk, KM_NODE_SPLIT_LDS, st,
                               ptr<uint16_t>(c->w_split6), mw, ptr<float>(c->s_delta), a.attr, a.attr_mod, a.dens,
                               a.dens_mod, N, B, eff0, ptr<float>(c->c_node), ptr<float>(c->proj));
        else
            hipLaunchKernelGGL(km_node_encode, dim3(mfma_grid(c, node_tiles)), blk, KM_NODE_LDS, st, mw,
                               ptr<float>(c->s_delta), a.attr, a.attr_mod, a.dens, a.dens_mod, N, B,
                               ptr<float>(c->eff), ptr<float>(c->c_node), ptr<float>(c->proj));
    }
    // split engine, small enough samples: the relation encoder is recomputed inside the
    // aggregate of every propagation step and c_edge is never materialised
    const bool fused = (c->engine == DRP_ENGINE_FUSED);
    const bool split = fused || c->engine == DRP_ENGINE_SPLIT || c->engine == DRP_ENGINE_FUSED;
    if (!fused) {
        ProbeScope ps(c, KC_EDGE_ENCODE);
        c->dv(split ? DV_EDGE_ENCODE_SPLIT : DV_EDGE_ENCODE);
        if (split)
            hipLaunchKernelGGL(km_edge_encode_split, dim3(mfma_grid(c, edge_tiles)), blk, KM_EDGE_SPLIT_LDS, st,
                               ptr<uint16_t>(c->w_split), mw, a.s_prev, a.prev_mod, a.prev_stride, a.attr,
                               a.attr_mod, a.dens, a.dens_mod, ptr<int16_t>(c->nbr_idx),
                               ptr<uint8_t>(c->nbr_cnt), N, B, ptr<float>(c->c_edge), c->re_scale, c->re_inv);
        else
            hipLaunchKernelGGL(km_edge_encode, dim3(mfma_grid(c, edge_tiles)), blk, KM_EDGE_LDS, st, mw,
                               a.s_prev, a.prev_mod, a.prev_stride, a.attr, a.attr_mod, a.dens, a.dens_mod,
                               ptr<int16_t>(c->nbr_idx), ptr<uint8_t>(c->nbr_cnt), N, B, ptr<float>(c->c_edge));
    }
    if (fused) {
        // graph -> node_encode -> the three propagation steps: one launch (km_prop3: a workgroup owns whole
        // samples and barriers locally between steps) when every CU gets a sample and a workgroup at least
        // PROP_WAVES tiles per step; otherwise one launch per step with the tiles of all samples dealt over the chip
        float* pa = ptr<float>(c->proj);
        float* pb = ptr<float>(c->proj2);
        if (prop3) {
            ProbeScope ps(c, KC_PROP);
            const dim3 pblk(64 * PROP_WAVES);
            // cached or recomputing: by the pile size alone (drp_ctx::ec_shape).  A cached batch too large for one launch of
            // at most ec_rows_cap rows per workgroup goes out as several launches over consecutive blocks of samples, the same
            // cache buffer under each -- the tape's launches too: the history buffers are laid out for the whole batch, a block
            // starts `ro` rows into every slot and the kernel takes the slots' stride as an argument (hist_rows)
            bool ec = c->ec_shape(N, tape);
            long chunk = B;
            if (ec) {
                long unit = 1;
                bool ok = true;
                for (int mod : {a.prev_mod, a.attr_mod, a.dens_mod})
                    if (mod < B) { if (unit % mod != 0 && mod % unit != 0) ok = false; else unit = std::max(unit, (long)mod); }
                const long cap = c->ec_chunk(N, unit);
                if (ok && cap > 0 && cap < B) chunk = cap;
            }
            if (ec) {
                // a launch that cannot be split (the tape's; batch columns no block is a multiple of) takes a cache over the whole
                // batch, 2.5 KB per row: beyond ecache_hard_max_mb it recomputes instead of failing for memory
                const long B0 = std::min((long)B, chunk), spw0 = (B0 + c->n_cu - 1) / c->n_cu;
                const size_t need = (size_t)((B0 + spw0 - 1) / spw0) * drp_ctx::ecache_stride(spw0 * N, false) * 16;
                if (need > ((size_t)c->ecache_hard_max_mb << 20)) { ec = false; chunk = B; }
                else CHK(ensure(c, c->ecache, need));
            }
            note_degrees(c, spw, N, B);
            unsigned long long* const wk = c->work_ptr();    // not null: the counting instantiations (drp_probe_begin("prop+work"))
            for (long b_off = 0; b_off < B; b_off += chunk) {
                const int Bc = (int)std::min(chunk, (long)B - b_off);
                const int spw_c = (Bc + c->n_cu - 1) / c->n_cu;
                const dim3 grid((unsigned)((Bc + spw_c - 1) / spw_c));
                // the block's view of every per-sample buffer: inputs replicated over the batch columns (row b reads column
                // b % mod) keep their base -- a block starts at a multiple of mod --, everything indexed by the row moves on
                const size_t ro = (size_t)b_off * N;
                const float* s_prev_c = a.prev_mod >= B ? a.s_prev + (size_t)b_off * a.prev_stride : a.s_prev;
                const int prev_mod_c = a.prev_mod >= B ? Bc : a.prev_mod;
                const float* attr_c = a.attr_mod >= B ? a.attr + ro : a.attr;
                const int attr_mod_c = a.attr_mod >= B ? Bc : a.attr_mod;
                const float* dens_c = a.dens_mod >= B ? a.dens + b_off : a.dens;
                const int dens_mod_c = a.dens_mod >= B ? Bc : a.dens_mod;
                float* eff_base = (tape ? a.eff_hist : ptr<float>(c->eff)) + ro * 64;
                unsigned* mask_hist = tape ? a.mask_hist + ro * DRP_K * 2 : nullptr;
                float* agg_hist = (tape && a.agg_hist) ? a.agg_hist + ro * 64 : nullptr;
                const size_t hist_rows = tape ? (size_t)B * N : 0;
                const float* sd_c = phase_e ? (const float*)(ptr<float>(c->s_delta) + ro * 3) : (const float*)nullptr;
                const float* cself_c = a.cself ? a.cself + (size_t)b_off * 64 : nullptr;
                const uint8_t* cself_ok_c = a.cself_ok ? a.cself_ok + b_off : nullptr;
#define PROP3_ARGS ptr<uint16_t>(c->w_split), ptr<uint16_t>(c->w_split6), mw, s_prev_c, prev_mod_c, a.prev_stride, \
                   attr_c, attr_mod_c, dens_c, dens_mod_c, ptr<int16_t>(c->nbr_idx) + ro * DRP_K, ptr<uint8_t>(c->nbr_cnt) + ro, pa + ro * 128, pb + ro * 128, \
                   ptr<float>(c->c_node) + ro * 64, eff_base, N, Bc, spw_c, sd_c, \
                   a.s_out + (size_t)b_off * a.out_stride, a.out_stride, cself_c, cself_ok_c, mask_hist, agg_hist, c->re_scale, c->re_inv, (c->prop3_order ? 1 : 0)
                const bool pair = c->prop_pair(spw_c, N, B);
                const size_t ec_stride = drp_ctx::ecache_stride((long)spw_c * N, pair);
                // ONE: no more tiles than waves in a workgroup -- the cached kernel then hands a tile's own rows from one propagation
                // step to the next in registers
                const bool one = ec && (pair ? ((long)spw_c * N + 15) / 16 : ((long)spw_c * N + 31) / 32) <= PROP_WAVES;
                if (ec && (size_t)grid.x * ec_stride * 16 > c->ecache.cap) CHK(ensure(c, c->ecache, (size_t)grid.x * ec_stride * 16));
#define PROP3_LAUNCH_W(TAPE_, PAIR_, EC_, ONE_) do { \
                    if (wk) hipLaunchKernelGGL((km_prop3<TAPE_, PAIR_, EC_, true, ONE_>), grid, pblk, KM_PROP3_LDS, st, PROP3_ARGS, ptr<float4>(c->ecache), ec_stride, wk, hist_rows); \
                    else hipLaunchKernelGGL((km_prop3<TAPE_, PAIR_, EC_, false, ONE_>), grid, pblk, KM_PROP3_LDS, st, PROP3_ARGS, ptr<float4>(c->ecache), ec_stride, wk, hist_rows); } while (0)
#define PROP3_LAUNCH(TAPE_, PAIR_) do { \
                    if (one) PROP3_LAUNCH_W(TAPE_, PAIR_, true, true); else if (ec) PROP3_LAUNCH_W(TAPE_, PAIR_, true, false); \
                    else PROP3_LAUNCH_W(TAPE_, PAIR_, false, false); } while (0)
                c->dv(DV_PROP3 + 12 * (tape ? 1 : 0) + 6 * (pair ? 1 : 0) + 2 * (one ? 2 : ec ? 1 : 0) + (wk ? 1 : 0));
                if (!tape && !pair) PROP3_LAUNCH(false, false);
                else if (!tape) PROP3_LAUNCH(false, true);
                else if (!pair) PROP3_LAUNCH(true, false);
                else PROP3_LAUNCH(true, true);
#undef PROP3_LAUNCH_W
#undef PROP3_LAUNCH
#undef PROP3_ARGS
            }
        }
        for (int p = 0; p < DRP_PSTEP && !prop3; ++p) {
            const bool last = (p + 1 == DRP_PSTEP);
            ProbeScope ps(c, KC_PROP);
            long pb_ = (node_tiles + PROP_WAVES - 1) / PROP_WAVES;
            // few tiles (up to four per CU): one per workgroup first, so that a tile has its SIMD to itself
            const int spread = (c->prop_spread && node_tiles <= 4L * c->n_cu) ? 1 : 0;
            // fewer still (up to two per CU): tiles of 16 receivers x two slots, half the slot iterations each
            const long tiles16 = (long)B * ((N + 15) / 16);
            const bool pair = spread && c->prop_pair_rows > 0 && node_tiles <= 2L * c->n_cu;
            if (spread) pb_ = pair ? tiles16 : node_tiles;
            const dim3 grid((unsigned)(pb_ < c->n_cu ? pb_ : c->n_cu)), pblk(64 * PROP_WAVES);
            const float* eff_in = tape ? a.eff_hist + (size_t)p * bn64 : ptr<float>(c->eff);
            float* eff_out = tape ? a.eff_hist + (size_t)(p + 1) * bn64 : ptr<float>(c->eff);
            unsigned* mask_out = tape ? a.mask_hist + (size_t)p * B * N * DRP_K * 2 : nullptr;
            float* agg_out = (tape && a.agg_hist) ? a.agg_hist + (size_t)p * bn64 : nullptr;
#define PROP_ARGS ptr<uint16_t>(c->w_split), ptr<uint16_t>(c->w_split6), mw, a.s_prev, a.prev_mod, a.prev_stride, \
                  a.attr, a.attr_mod, a.dens, a.dens_mod, ptr<int16_t>(c->nbr_idx), ptr<uint8_t>(c->nbr_cnt), pa, \
                  ptr<float>(c->c_node), eff_in, eff_out, N, B, pb, a.s_out, a.out_stride, a.cself, a.cself_ok, mask_out, agg_out, \
                  c->re_scale, c->re_inv, spread, c->work_ptr()
#define PROP_LAUNCH(PAIR_, WORK_) do { \
                if (!tape) { \
                    if (!last) hipLaunchKernelGGL((km_prop<false, false, PAIR_, WORK_>), grid, pblk, KM_PROP_LDS(false), st, PROP_ARGS); \
                    else hipLaunchKernelGGL((km_prop<true, false, PAIR_, WORK_>), grid, pblk, KM_PROP_LDS(true), st, PROP_ARGS); \
                } else { \
                    if (!last) hipLaunchKernelGGL((km_prop<false, true, PAIR_, WORK_>), grid, pblk, KM_PROP_LDS(false), st, PROP_ARGS); \
                    else hipLaunchKernelGGL((km_prop<true, true, PAIR_, WORK_>), grid, pblk, KM_PROP_LDS(true), st, PROP_ARGS); \
                } } while (0)
            c->dv(DV_PROP + 8 * (last ? 1 : 0) + 4 * (tape ? 1 : 0) + 2 * (pair ? 1 : 0) + (c->work_ptr() ? 1 : 0));
            if (c->work_ptr()) { if (pair) PROP_LAUNCH(true, true); else PROP_LAUNCH(false, true); }
            else if (pair) PROP_LAUNCH(true, false);
            else PROP_LAUNCH(false, false);
#undef PROP_LAUNCH
#undef PROP_ARGS
            float* tmp = pa; pa = pb; pb = tmp;
        }
        return DRP_OK;
    }
    if (tape_mfma) HIPCHK(c, hipMemcpyAsync(a.eff_hist, c->eff.p, bn64 * sizeof(float), hipMemcpyDeviceToDevice, st));
    for (int p = 0; p < DRP_PSTEP; ++p) {
        if (tape_mfma) {
            // the aggregate that also leaves the edges' ReLU bits; the aggregated rows and the effects are copied into the tape
            ProbeScope pa(c, KC_AGGREGATE);
            int chunks = 1;
            if (B < c->n_cu / 2) chunks = std::max(1, std::min((N + 15) / 16, 2048 / B));
            c->dv(DV_AGGREGATE_TAPE);
            hipLaunchKernelGGL(k_aggregate_tape, dim3(B * chunks), dim3(256), 0, st, ptr<float>(c->c_edge), ptr<float>(c->proj),
                               ptr<int16_t>(c->nbr_idx), ptr<uint8_t>(c->nbr_cnt), N, ptr<float>(c->agg), chunks,
                               a.mask_hist + (size_t)p * B * N * DRP_K * 2);
            if (a.agg_hist)
                HIPCHK(c, hipMemcpyAsync(a.agg_hist + (size_t)p * bn64, c->agg.p, bn64 * sizeof(float), hipMemcpyDeviceToDevice, st));
        } else {
            launch_aggregate(c, B, N);
        }
        {
        ProbeScope ps(c, p + 1 < DRP_PSTEP ? KC_UPDATE : KC_PREDICT);
        c->dv(DV_UPDATE);
        if (p + 1 < DRP_PSTEP)
            hipLaunchKernelGGL(km_update<false>, dim3(mfma_grid(c, node_tiles)), blk, KM_UPD_LDS, st, mw,
                               ptr<float>(c->agg), ptr<float>(c->c_node), ptr<float>(c->eff), N, B,
                               ptr<float>(c->proj), a.s_prev, a.prev_mod, a.prev_stride, a.s_out, a.out_stride);
        else
            hipLaunchKernelGGL(km_update<true>, dim3(mfma_grid(c, node_tiles)), blk, KM_UPD_LDS, st, mw,
                               ptr<float>(c->agg), ptr<float>(c->c_node), ptr<float>(c->eff), N, B,
                               ptr<float>(c->proj), a.s_prev, a.prev_mod, a.prev_stride, a.s_out, a.out_stride);
        }
        // the step's effect is the next tape entry (km_update keeps it in place, the last step's too)
        if (tape_mfma)
            HIPCHK(c, hipMemcpyAsync(a.eff_hist + (size_t)(p + 1) * bn64, c->eff.p, bn64 * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return DRP_OK;
}

// One predict_one_step (model/gnn_dyn.py:209-254) [+ gen_s_delta, planners.py:346] for B samples.
int run_step(drp_ctx* c, const StepArgs& a) {
    const int B = a.B, N = a.N;
    hipStream_t st = c->stream;
    float* s_delta = ptr<float>(c->s_delta);
    int16_t* nbr_idx = ptr<int16_t>(c->nbr_idx);
    uint8_t* nbr_cnt = ptr<uint8_t>(c->nbr_cnt);
    const float* vw = ptr<float>(c->w_valu);
    a.encoded = false;
    if (a.build_graph) {
        ProbeScope ps(c, KC_GRAPH);
        const int self_first = (c->engine == DRP_ENGINE_FUSED && a.cself != nullptr) ? 1 : 0;
        if (a.rev_off != nullptr && N <= GRAPH_THREADS && c->graph_rev) {
            c->dv(DV_GRAPH_REV);
            hipLaunchKernelGGL(k_graph_rev, dim3(SPREAD_GRID(B)), dim3(GRAPH_THREADS), (size_t)12 * N * sizeof(int), st, a.s_prev,
                               a.prev_mod, a.prev_stride, a.actions, a.act_stride, s_delta, N, nbr_idx, nbr_cnt, c->cam, c->thr,
                               B, self_first, a.rev_off, a.rev);
            if (a.rev_built) *a.rev_built = true;
        } else if (graph_takes_q4(c, B, N, a.padded) && a.actions == nullptr && c->engine == DRP_ENGINE_FUSED && c->graph_encode &&
                   !step_has_phase_e(c, B, N)) {
            // a handful of samples whose impulses are data (the trainer's forward pass): the lists and the particle encoder
            // read nothing of one another -- one launch (k_rollout.h)
            const int chunks = (N + 127) / 128, n_graph = B * chunks;
            const long node_tiles = (long)B * ((N + 31) / 32);
            const bool tape = a.eff_hist != nullptr;
            c->dv(DV_GRAPH_Q4_ENCODE);
            hipLaunchKernelGGL(km_graph_q4_encode, dim3((unsigned)(n_graph + mfma_grid_spread(c, node_tiles))), dim3(GRAPH_Q4_THREADS),
                               KM_GRAPH_Q4_ENCODE_LDS(N), st, a.s_prev, a.prev_mod, a.prev_stride, s_delta, N, B, nbr_idx, nbr_cnt, c->cam,
                               c->thr, chunks, self_first, n_graph, ptr<uint16_t>(c->w_split6), ptr<float>(c->w_mfma), a.attr, a.attr_mod,
                               a.dens, a.dens_mod, tape ? a.eff_hist : ptr<float>(c->eff), ptr<float>(c->c_node), ptr<float>(c->proj));
            a.encoded = true;
        } else {
            launch_graph(c, st, a.s_prev, a.prev_mod, a.prev_stride, a.actions, a.act_stride, s_delta, B, N, nbr_idx, nbr_cnt,
                         self_first, a.padded);
        }
    }
    if (c->engine != DRP_ENGINE_VALU) {
        int rc = run_step_mfma(c, a);
        if (rc != DRP_OK) return rc;
        HIPCHK(c, hipGetLastError());
        return DRP_OK;
    }
    c->dv(DV_VALU_STEP);
    {
        ProbeScope ps(c, KC_NODE_ENCODE);
        hipLaunchKernelGGL(k_node_encode<8>, dim3(B), dim3(256), 0, st, vw, s_delta, a.attr,
                           a.attr_mod, a.dens, a.dens_mod, N, ptr<float>(c->eff), ptr<float>(c->c_node));
    }
    {
        ProbeScope ps(c, KC_EDGE_ENCODE);
        hipLaunchKernelGGL(k_edge_encode, dim3(B), dim3(256), (6 * 64 + 3 * 4096) * sizeof(float), st,
                           vw, a.s_prev, a.prev_mod, a.prev_stride, a.attr, a.attr_mod, a.dens,
                           a.dens_mod, nbr_idx, nbr_cnt, N, ptr<float>(c->c_edge));
    }
    for (int p = 0; p < DRP_PSTEP; ++p) {
        {
            ProbeScope ps(c, KC_PROJECT);
            hipLaunchKernelGGL(k_project<8>, dim3(B), dim3(256), 0, st, vw, ptr<float>(c->eff), N,
                               ptr<float>(c->proj));
        }
        launch_aggregate(c, B, N);
        {
            ProbeScope ps(c, KC_UPDATE);
            hipLaunchKernelGGL(k_update<8>, dim3(B), dim3(256), 0, st, vw, ptr<float>(c->agg),
                               ptr<float>(c->c_node), N, ptr<float>(c->eff));
        }
    }
    {
        ProbeScope ps(c, KC_PREDICT);
        hipLaunchKernelGGL(k_predict<8>, dim3(B), dim3(256), 0, st, vw, ptr<float>(c->eff), a.s_prev,
                           a.prev_mod, a.prev_stride, N, a.s_out, a.out_stride);
    }
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int run_reward(drp_ctx* c, const float* state, size_t row_stride, int rows, int N, int normalize,
               float* out) {
    ProbeScope ps(c, KC_REWARD);
    c->dv(DV_REWARD);
    hipLaunchKernelGGL(k_reward, dim3(rows), dim3(256), (2 * ((N + 3) & ~3) + 8) * sizeof(float), c->stream, state,
                       row_stride, N, ptr<float>(c->goal_field), c->goal_h, c->goal_w,
                       ptr<float>(c->goal_coor), c->goal_m, c->cam, normalize, out);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

// H-step rollout over device-resident s0/attr/dens (in s_in/attr/dens, nb rows) and actions.
// Self-edge constant of the fused engine (k_cself): one vector per sample, constant over a whole
// rollout (it depends on the attributes and the density only).  Null pointers when it does not apply.
int prepare_cself(drp_ctx* c, int attr_mod, int N, int B, const float** cself, const uint8_t** cself_ok) {
    *cself = nullptr;
    *cself_ok = nullptr;
    if (c->engine == DRP_ENGINE_FUSED && c->self_const) {
        CHK(ensure(c, c->cself, (size_t)B * 64 * sizeof(float) + (size_t)B));
        float* cs = ptr<float>(c->cself);
        uint8_t* ok = reinterpret_cast<uint8_t*>(cs + (size_t)B * 64);
        hipLaunchKernelGGL(k_cself, dim3(B), dim3(64), 0, c->stream, ptr<float>(c->w_valu), ptr<float>(c->attr), attr_mod,
                           ptr<float>(c->dens), attr_mod, N, cs, ok);
        *cself = cs;
        *cself_ok = ok;
        ++c->cself_tag;
    }
    return DRP_OK;
}

int run_rollout(drp_ctx* c, int nb, int N, int B, int H, bool reward_all, bool reward_last, bool session = false) {
    CHK(ensure_step_ws(c, B, N));
    CHK(ensure(c, c->states, (size_t)B * H * N * 3 * sizeof(float)));
    CHK(ensure(c, c->rewards, (size_t)B * H * sizeof(float)));
    c->lastH = H;
    float* states = ptr<float>(c->states);
    const size_t hstride = (size_t)H * N * 3;
    const float* cself = nullptr;
    const uint8_t* cself_ok = nullptr;
    // the self-edge constants depend on attributes and densities only: an MPC session computes them once (its first
    // rollout) and keeps them while nobody else has refilled the buffer
    if (session && c->mpc_cself_tag != 0 && c->mpc_cself_tag == c->cself_tag) {
        cself = c->mpc_cself;
        cself_ok = c->mpc_cself_ok;
    } else {
        CHK(prepare_cself(c, nb, N, B, &cself, &cself_ok));
        if (session) { c->mpc_cself_tag = c->cself_tag; c->mpc_cself = cself; c->mpc_cself_ok = cself_ok; }
    }
    // small piles on the fused engine: the whole rollout is ONE launch (km_rollout, k_rollout.h) -- a workgroup owns its
    // samples from the first step to the last, builds their neighbour lists itself and keeps the node matrices in LDS
    // A cached shape (drp_ctx::ec_shape: by the pile size alone) gives a workgroup at most ec_rows_cap rows; a batch that needs
    // more goes out as several launches over consecutive blocks of samples (whole multiples of the batch columns)
    const bool ec = c->engine == DRP_ENGINE_FUSED && c->ec_shape(N);
    long chunk_r = B;
    if (ec) { const long cap = c->ec_chunk(N, nb); if (cap > 0 && cap < B) chunk_r = cap; }
    const int spw_r = (int)((std::min((long)B, chunk_r) + c->n_cu - 1) / c->n_cu);
    // up to rollout_max_n particles whatever the batch; up to rollout_mid_n while a workgroup holds no more than rollout_mid_rows
    const bool roll_size = N <= c->rollout_max_n || (N <= c->rollout_mid_n && (long)spw_r * N <= c->rollout_mid_rows && (N <= 200 || B >= c->n_cu / 2));
    const bool one_launch = c->engine == DRP_ENGINE_FUSED && c->rollout_fused && c->prop3 && c->prop3e && roll_size &&
                            whole_samples(c, B, N) && ((long)spw_r * N + 31) / 32 >= c->prop3_min_tiles &&
                            (long)spw_r * N <= KM_ROLLOUT_MAX_ROWS && (long)spw_r * N <= c->rollout_max_rows;
    if (one_launch) {
        const int n_chunks = (int)((B + chunk_r - 1) / chunk_r);
        std::vector<RolloutArgs> blocks((size_t)n_chunks);
        std::vector<char> pairs((size_t)n_chunks);
        if (ec) CHK(ensure(c, c->ecache, (size_t)((std::min((long)B, chunk_r) + spw_r - 1) / spw_r) * drp_ctx::ecache_stride((long)spw_r * N, false) * 16));
        for (int q = 0; q < n_chunks; ++q) {
            const long b_off = (long)q * chunk_r;
            const int Bc = (int)std::min(chunk_r, (long)B - b_off);
            const size_t ro = (size_t)b_off * N;
            RolloutArgs& ra = blocks[(size_t)q];
            ra = RolloutArgs{};
            ra.sw = ptr<uint16_t>(c->w_split); ra.sw6 = ptr<uint16_t>(c->w_split6); ra.mw = ptr<float>(c->w_mfma);
            // the first state, the attributes and the densities are replicated over the batch columns (row b reads column b % nb;
            // a block starts at a multiple of nb): same base for every block; everything indexed by the row moves on
            ra.s_in = ptr<float>(c->s_in); ra.attr = ptr<float>(c->attr); ra.dens = ptr<float>(c->dens);
            ra.states = states + ro * 3 * H;
            ra.actions = ptr<float>(c->actions) + (size_t)b_off * H * 4;
            ra.s_delta = ptr<float>(c->s_delta) + ro * 3; ra.nbr_idx = ptr<int16_t>(c->nbr_idx) + ro * DRP_K;
            ra.nbr_cnt = ptr<uint8_t>(c->nbr_cnt) + ro; ra.proj_a = ptr<float>(c->proj) + ro * 128; ra.proj_b = ptr<float>(c->proj2) + ro * 128;
            ra.c_node = ptr<float>(c->c_node) + ro * 64; ra.eff = ptr<float>(c->eff) + ro * 64;
            ra.cself = cself ? cself + (size_t)b_off * 64 : nullptr; ra.cself_ok = cself_ok ? cself_ok + b_off : nullptr;
            ra.N = N; ra.B = Bc; ra.spw = (Bc + c->n_cu - 1) / c->n_cu; ra.nb = nb; ra.H = H; ra.order_rows = (c->prop3_order ? 1 : 0);
            ra.thr = c->thr; ra.re_scale = c->re_scale; ra.re_inv = c->re_inv; ra.cam = c->cam;
            const bool pair_q = c->prop_pair(ra.spw, N, B);
            pairs[(size_t)q] = pair_q ? 1 : 0;
            ra.ec_stride = drp_ctx::ecache_stride((long)ra.spw * N, pair_q);
            ra.ecache = ec ? ptr<float4>(c->ecache) : nullptr;
            ra.work = c->work_ptr();
        }
        // the argument blocks sit in device memory; they are uploaded when they change (every iteration of an MPC session
        // passes the same ones), behind whatever still runs on the stream
        if (!c->roll_args_valid || c->roll_args_host.size() != blocks.size() ||
            memcmp(blocks.data(), c->roll_args_host.data(), blocks.size() * sizeof(RolloutArgs)) != 0) {
            c->roll_args_host = blocks;
            c->roll_args_valid = false;
            CHK(h2d(c, c->roll_args, c->roll_args_host.data(), blocks.size() * sizeof(RolloutArgs)));
            c->roll_args_valid = true;
        }
        ProbeScope ps(c, KC_PROP);
        c->dv(DV_GRAPH_IN_ROLLOUT);
        for (int q = 0; q < n_chunks; ++q) {
            const RolloutArgs& ra = blocks[(size_t)q];
            const bool pair_r = pairs[(size_t)q] != 0;
            const unsigned grid_r = (unsigned)((ra.B + ra.spw - 1) / ra.spw);
            // ONE: no more tiles than waves in a workgroup -- the cached kernel then hands a tile's own rows (P_r, its own P_s, its
            // effect) from one propagation step to the next in registers
            const long tiles_r = pair_r ? ((long)ra.spw * N + 15) / 16 : ((long)ra.spw * N + 31) / 32;
            const bool one = ec && tiles_r <= PROP_WAVES;
#define ROLLOUT_LAUNCH_W(PAIR_, EC_, WORK_, ONE_) hipLaunchKernelGGL((km_rollout<PAIR_, EC_, WORK_, ONE_>), dim3(grid_r), dim3(64 * PROP_WAVES), KM_ROLLOUT_LDS, \
                                                                     c->stream, ptr<RolloutArgs>(c->roll_args) + q)
#define ROLLOUT_LAUNCH(PAIR_, EC_, ONE_) do { if (ra.work) ROLLOUT_LAUNCH_W(PAIR_, EC_, true, ONE_); else ROLLOUT_LAUNCH_W(PAIR_, EC_, false, ONE_); } while (0)
            c->dv(DV_ROLLOUT + 6 * (pair_r ? 1 : 0) + 2 * (one ? 2 : ec ? 1 : 0) + (ra.work ? 1 : 0));
            if (pair_r) { if (one) ROLLOUT_LAUNCH(true, true, true); else if (ec) ROLLOUT_LAUNCH(true, true, false); else ROLLOUT_LAUNCH(true, false, false); }
            else { if (one) ROLLOUT_LAUNCH(false, true, true); else if (ec) ROLLOUT_LAUNCH(false, true, false); else ROLLOUT_LAUNCH(false, false, false); }
#undef ROLLOUT_LAUNCH_W
#undef ROLLOUT_LAUNCH
        }
        HIPCHK(c, hipGetLastError());
        note_degrees(c, spw_r, N, B);           // the last step's lists
    }
    for (int t = 0; t < H && !one_launch; ++t) {
        StepArgs a{};
        a.cself = cself; a.cself_ok = cself_ok;
        if (t == 0) {
            a.s_prev = ptr<float>(c->s_in); a.prev_mod = nb; a.prev_stride = (size_t)N * 3;
        } else {
            a.s_prev = states + (size_t)(t - 1) * N * 3; a.prev_mod = B; a.prev_stride = hstride;
        }
        a.attr = ptr<float>(c->attr); a.attr_mod = nb;
        a.dens = ptr<float>(c->dens); a.dens_mod = nb;
        a.actions = ptr<float>(c->actions) + (size_t)t * 4; a.act_stride = (size_t)H * 4;
        a.build_graph = true;
        a.s_out = states + (size_t)t * N * 3; a.out_stride = hstride;
        a.B = B; a.N = N;
        CHK(run_step(c, a));
    }
    if (reward_all) {
        // rows = B*H consecutive [N,3] blocks
        CHK(run_reward(c, states, (size_t)N * 3, B * H, N, 1, ptr<float>(c->rewards)));
    } else if (reward_last) {
        // only the last step's state of every sample; written at rewards[b*H + H-1]
        CHK(ensure(c, c->scratch, (size_t)B * sizeof(float)));
        CHK(run_reward(c, states + (size_t)(H - 1) * N * 3, hstride, B, N, 1, ptr<float>(c->scratch)));
        HIPCHK(c, hipMemcpy2DAsync(ptr<float>(c->rewards) + (H - 1), H * sizeof(float), c->scratch.p,
                                   sizeof(float), sizeof(float), B, hipMemcpyDeviceToDevice, c->stream));
    }
    return DRP_OK;
}

void pack_valu(const float* w, std::vector<float>& v) {
    v.assign(V_TOTAL, 0.0f);
    auto T = [&](int dst, int src, int out, int in, int ld, int col0) {
        // dst[k][o] = w[src + o*ld + col0 + k]
        for (int o = 0; o < out; ++o)
            for (int k = 0; k < in; ++k) v[dst + k * 64 + o] = w[src + o * ld + col0 + k];
    };
    auto C = [&](int dst, int src, int n) { for (int i = 0; i < n; ++i) v[dst + i] = w[src + i]; };
    T(V_PE0_T, W_PE0_W, 64, 5, 5, 0);   C(V_PE0_B, W_PE0_B, 64);
    T(V_PE2_T, W_PE2_W, 64, 64, 64, 0); C(V_PE2_B, W_PE2_B, 64);
    T(V_PPE_T, W_PP_W, 64, 64, 129, 0);
    for (int o = 0; o < 64; ++o) v[V_PP_WD + o] = w[W_PP_W + o * 129 + 128];
    C(V_PP_B, W_PP_B, 64);
    T(V_AGG_T, W_PP_W, 64, 64, 129, 64);
    T(V_RE0_T, W_RE0_W, 64, 6, 6, 0);   C(V_RE0_B, W_RE0_B, 64);
    T(V_RE2_T, W_RE2_W, 64, 64, 64, 0); C(V_RE2_B, W_RE2_B, 64);
    T(V_RE4_T, W_RE4_W, 64, 64, 64, 0); C(V_RE4_B, W_RE4_B, 64);
    T(V_RPE_T, W_RP_W, 64, 64, 193, 0);
    for (int o = 0; o < 64; ++o) v[V_RP_WD + o] = w[W_RP_W + o * 193 + 192];
    C(V_RP_B, W_RP_B, 64);
    T(V_RPR_T, W_RP_W, 64, 64, 193, 64);
    T(V_RPS_T, W_RP_W, 64, 64, 193, 128);
    T(V_PR0_T, W_PR0_W, 64, 64, 64, 0); C(V_PR0_B, W_PR0_B, 64);
    C(V_PR1_W, W_PR1_W, 192);
    C(V_PR1_B, W_PR1_B, 3);
}

int need(drp_ctx* c, bool weights, bool cam, bool goal) {
    if (!c) return DRP_EINVAL;
    if (weights && !c->have_weights) return fail(c, DRP_ESTATE, "weights not loaded (drp_load_weights)");
    if (cam && !c->have_cam) return fail(c, DRP_ESTATE, "camera not set (drp_set_camera)");
    if (goal && !c->have_goal) return fail(c, DRP_ESTATE, "goal not set (drp_set_goal)");
    return DRP_OK;
}

// Weight-gradient jobs are queued and go out together (flush_wgrad): one pair of launches for all the jobs whose
// inputs exist at that point of the stream.  flush_wgrad must run before a kernel overwrites a queued job's g or x.
void flush_wgrad(drp_ctx* c) {
    const int n = (int)c->wg_jobs.size();
    if (n == 0 || c->wg_defer_now) return;
    WgradJobs J{};
    int max_blocks = 1;
    for (int q = 0; q < n; ++q) {
        J.j[q] = c->wg_jobs[q];
        J.j[q].part = static_cast<float*>(c->tr_part.p) + (size_t)q * KT_WGRAD_MAX_BLOCKS * 66 * 64;
        if (J.j[q].blocks > max_blocks) max_blocks = J.j[q].blocks;
    }
    c->dv(c->wgrad_mfma ? DV_WGRAD_MFMA : DV_WGRAD_VALU);
    if (c->wgrad_mfma)
        hipLaunchKernelGGL(kt_wgrad_mfma_multi, dim3((unsigned)max_blocks, (unsigned)n), dim3(256), KT_WGRAD_MULTI_LDS, c->stream, J);
    else
        hipLaunchKernelGGL(kt_wgrad_multi, dim3((unsigned)max_blocks, (unsigned)n), dim3(256), KT_WGRAD_MULTI_LDS, c->stream, J);
    hipLaunchKernelGGL(kt_wgrad_reduce_multi, dim3(66, (unsigned)n), dim3(256), 0, c->stream, J);
    c->wg_jobs.clear();
}

// The deferred jobs of a whole backward pass.  Jobs of one size go through one launch (blockIdx.y walks that size's
// slice of `order`); then ONE reduction launch in which a block owns a target dW and adds its jobs' sums in queue order
// -- what the in-between flushes did launch after launch, so the gradients keep their bits.
int flush_wgrad_all(drp_ctx* c) {
    const int n = (int)c->wg_jobs.size();
    c->wg_defer_now = false;
    if (n == 0) return DRP_OK;
    // partial sums: one slab per job
    size_t part_floats = 0;
    std::vector<size_t> part_off(n);
    for (int q = 0; q < n; ++q) { part_off[q] = part_floats; part_floats += (size_t)c->wg_jobs[q].blocks * 66 * 64; }
    CHK(ensure(c, c->tr_part, std::max(part_floats, (size_t)KT_WGRAD_MAX_JOBS * KT_WGRAD_MAX_BLOCKS * 66 * 64) * sizeof(float)));
    for (int q = 0; q < n; ++q) c->wg_jobs[q].part = static_cast<float*>(c->tr_part.p) + part_off[q];
    // launch order: by size; reduction lists: by target, in queue order
    std::vector<int> order(n);
    for (int q = 0; q < n; ++q) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return c->wg_jobs[a].blocks > c->wg_jobs[b].blocks; });
    std::vector<float*> targets;
    std::vector<std::vector<int>> lists;
    for (int q = 0; q < n; ++q) {
        size_t k = 0;
        while (k < targets.size() && targets[k] != c->wg_jobs[q].dW) ++k;
        if (k == targets.size()) { targets.push_back(c->wg_jobs[q].dW); lists.emplace_back(); }
        lists[k].push_back(q);
    }
    const int nt = (int)targets.size();
    std::vector<int> idx;                      // order[n] | tgt_off[nt + 1] | tgt_jobs[n]
    idx.insert(idx.end(), order.begin(), order.end());
    int off = 0;
    for (int k = 0; k < nt; ++k) { idx.push_back(off); off += (int)lists[k].size(); }
    idx.push_back(off);
    for (int k = 0; k < nt; ++k) idx.insert(idx.end(), lists[k].begin(), lists[k].end());
    // upload when anything changed (the same shape queues the same jobs iteration after iteration)
    const size_t jb = (size_t)n * sizeof(WgradJob), ib = idx.size() * sizeof(int);
    std::vector<unsigned char> img(jb + ib);
    memcpy(img.data(), c->wg_jobs.data(), jb);
    memcpy(img.data() + jb, idx.data(), ib);
    if (img != c->wg_uploaded) {
        c->wg_uploaded.swap(img);               // the copies' source stays alive in the context
        CHK(h2d(c, c->wg_jobs_dev, c->wg_uploaded.data(), jb));
        CHK(h2d(c, c->wg_idx_dev, c->wg_uploaded.data() + jb, ib));
    }
    c->dv(DV_WGRAD_DEFERRED);
    c->dv(c->wgrad_mfma ? DV_WGRAD_MFMA : DV_WGRAD_VALU);
    const WgradJob* jd = static_cast<const WgradJob*>(c->wg_jobs_dev.p);
    const int* od = static_cast<const int*>(c->wg_idx_dev.p);
    for (int a = 0; a < n;) {
        int b = a;
        while (b < n && c->wg_jobs[order[b]].blocks == c->wg_jobs[order[a]].blocks) ++b;
        const dim3 grid((unsigned)c->wg_jobs[order[a]].blocks, (unsigned)(b - a));
        if (c->wgrad_mfma) hipLaunchKernelGGL(kt_wgrad_mfma_list, grid, dim3(256), KT_WGRAD_MULTI_LDS, c->stream, jd, od, a);
        else hipLaunchKernelGGL(kt_wgrad_list, grid, dim3(256), KT_WGRAD_MULTI_LDS, c->stream, jd, od, a);
        a = b;
    }
    hipLaunchKernelGGL(kt_wgrad_reduce_lists, dim3(66, (unsigned)nt), dim3(256), 0, c->stream, jd, od + n, od + n + nt + 1);
    c->wg_jobs.clear();
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

template <int IN>
void launch_wgrad(drp_ctx* c, const float* g, int ldg, const float* x, int ldx, long M, float* dW, int lane_stride,
                  int k_stride, float* db, float* dwd, const float* dens, int dens_mod, long rows_per_sample) {
    long blocks = (M + 63) / 64;
    if (blocks > KT_WGRAD_MAX_BLOCKS) blocks = KT_WGRAD_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    if ((int)c->wg_jobs.size() == KT_WGRAD_MAX_JOBS && !c->wg_defer_now) flush_wgrad(c);
    WgradJob q{};
    q.g = g; q.x = x; q.dW = dW; q.db = db; q.dwd = dwd; q.dens = dens; q.part = nullptr;
    q.M = M; q.rows_per_sample = rows_per_sample;
    q.ldg = ldg; q.ldx = ldx; q.lane_stride = lane_stride; q.k_stride = k_stride; q.dens_mod = dens_mod; q.in = IN;
    q.blocks = (int)blocks;
    c->wg_jobs.push_back(q);
}

// The one-shot entry points stage their inputs in the buffers the planner sessions keep their state in
// (s_in, attr, dens, actions, states): a session interrupted by one of them is over -- its next call returns
// DRP_ESTATE instead of results computed from overwritten inputs.
void end_sessions(drp_ctx* c) {
    c->mpc_on = false;
    c->gd_on = false;
    for (int q = 0; q < DRP_GD_SLOTS; ++q) c->gd_pending[q] = false;
    c->mpc_pending[0] = c->mpc_pending[1] = false;
}

// The split relation encoder's range shift was proven for an envelope of inputs (drp_load_weights); a call
// whose attributes, densities or impulses leave it is refused instead of risking a saturated fp16 piece.
float max_abs(const float* p, size_t n) {
    float m = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float v = fabsf(p[i]);
        if (v > m || v != v) m = (v != v) ? INFINITY : v;
    }
    return m;
}
// largest |s_delta| a push can cause (planners.py:238-254: the impulse is at most the push's own length in the
// camera frame): actions [n][4] = (sx, sy, ex, ey) in world units
float push_len_bound(const drp_ctx* c, const float* actions, size_t n) {
    // spectral norm of the world -> camera map's 3x3 part (1 for the rotation a camera is; the Frobenius norm used
    // until round 2 is sqrt(3) too large, which put the DEFAULT clip box outside the proven envelope): sqrt of the
    // largest eigenvalue of M^T M
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += (double)c->cam.m[k * 4 + i] * (double)c->cam.m[k * 4 + j];
            A[i][j] = v;
        }
    // largest eigenvalue of the symmetric 3x3 in closed form (the trigonometric solution of its cubic): an upper bound of
    // the impulse must not come from an iteration that converges from BELOW (a map with two close singular values)
    double lam;
    const double p1 = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double q = (A[0][0] + A[1][1] + A[2][2]) / 3.0;
    if (p1 == 0.0) {
        lam = fmax(A[0][0], fmax(A[1][1], A[2][2]));
    } else {
        const double p2 = (A[0][0] - q) * (A[0][0] - q) + (A[1][1] - q) * (A[1][1] - q) + (A[2][2] - q) * (A[2][2] - q) + 2.0 * p1;
        const double p = sqrt(p2 / 6.0);
        double Bm[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Bm[i][j] = (A[i][j] - (i == j ? q : 0.0)) / p;
        double r = 0.5 * (Bm[0][0] * (Bm[1][1] * Bm[2][2] - Bm[1][2] * Bm[2][1]) - Bm[0][1] * (Bm[1][0] * Bm[2][2] - Bm[1][2] * Bm[2][0]) +
                          Bm[0][2] * (Bm[1][0] * Bm[2][1] - Bm[1][1] * Bm[2][0]));
        r = fmin(1.0, fmax(-1.0, r));
        lam = q + 2.0 * p * cos(acos(r) / 3.0);
    }
    // rounding slack of the formula, never above the Frobenius norm (itself a bound)
    const double frob = sqrt(A[0][0] + A[1][1] + A[2][2]);
    const float fro = (float)fmin(frob, sqrt(fmax(lam, 0.0)) * (1.0 + 1e-6));
    float l2 = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float dx = actions[i * 4 + 2] - actions[i * 4 + 0], dy = actions[i * 4 + 3] - actions[i * 4 + 1];
        const float v = dx * dx + dy * dy;
        if (v > l2 || v != v) l2 = (v != v) ? INFINITY : v;
    }
    return fro * sqrtf(l2) / c->cam.gs;
}
// tape: the caller runs the fused engine whatever drp_set_engine chose (the gradient-descent planner's and the trainer's
// forward pass write their tape with it)
int range_check(drp_ctx* c, float max_attr, float max_dens, float max_sdelta, bool tape = false) {
    if (!tape && c->engine != DRP_ENGINE_FUSED && c->engine != DRP_ENGINE_SPLIT) return DRP_OK;
    const double A = max_attr, dm = max_dens / DRP_DENS_SCALE, D = (double)c->adj_thresh + 2.0 * max_sdelta;
    const SplitRange& r = c->re_range;
    if (!c->re_ok)
        return fail(c, DRP_ERANGE, "weights outside the range of the split-fp16 relation encoder (largest |w| %g, activation "
                    "bound %g, shift %d): use DRP_ENGINE_MFMA", (double)r.wmax, split_range_bound(r, r.env_attr, r.env_delta, r.env_dens), r.shift);
    if (A <= r.env_attr && dm <= r.env_dens && D <= r.env_delta) return DRP_OK;
    const double bound = split_range_bound(r, A, D, dm);
    if (ldexp(bound, r.shift) <= 65504.0) return DRP_OK;       // outside the envelope, still provably inside fp16
    return fail(c, DRP_ERANGE, "inputs beyond the range the split-fp16 relation encoder is scaled for (max |attr| %g, "
                "density %g, |s_delta| %g; activation bound %g x 2^%d): use DRP_ENGINE_MFMA for this call",
                A, (double)max_dens, (double)max_sdelta, bound, r.shift);
}

// Which engine writes the tape of the gradient-descent planner / the trainer: the fused one (km_prop<., TAPE>) unless the
// caller has selected an fp32 engine (drp_set_engine) or the split-fp16 relation encoder would refuse these weights or
// inputs -- then the fp32 matrix engine's stage kernels with k_aggregate_tape: several times slower, no range limit.  The
// live planner of the reference IS the gradient-descent one (env/flex_env.py:973-976): it must not stop on DRP_ERANGE.
int pick_tape_engine(drp_ctx* c, float max_attr, float max_dens, float max_sdelta, int* engine) {
    if (c->engine == DRP_ENGINE_MFMA || c->engine == DRP_ENGINE_VALU) { *engine = DRP_ENGINE_MFMA; return DRP_OK; }
    const int rc = range_check(c, max_attr, max_dens, max_sdelta, true);
    if (rc == DRP_ERANGE) { *engine = DRP_ENGINE_MFMA; c->err.clear(); return DRP_OK; }
    *engine = DRP_ENGINE_FUSED;
    return rc;
}

// range shift of the split relation encoder: proven for |attr| <= 2 (the reference's are 0), |s_r - s_s| <= 1.5
// per coordinate (radius 0.08 + two impulses; the default clip box's longest push is 8.5 sqrt(2) / 24 = 0.50
// camera-frame units, the whole workspace diagonal 0.59: 0.08 + 2 x 0.59 = 1.26), density <= 10 000 (training
// range: 15 .. 6 500); calls beyond are re-checked one by one (range_check)
void set_split_range(drp_ctx* c, const float* blob) {
    split_range_init(blob, c->re_range, SPLIT_ENV_ATTR, SPLIT_ENV_DELTA, SPLIT_ENV_DENS);
    if (c->re_shift_env != 0x7fffffff) c->re_range.shift = c->re_shift_env;
    // weights no shift can carry (a matrix entry beyond fp16, NaN): the split engines refuse every call
    // (range_check); the fp32 engines are unaffected
    c->re_ok = c->re_range.finite && c->re_range.wmax < 6.0e4f &&
               ldexp(split_range_bound(c->re_range, 2.0, 1.5, 2.0), c->re_range.shift) <= 65504.0;
    c->re_scale = ldexpf(1.0f, c->re_range.shift);
    c->re_inv = ldexpf(1.0f, -c->re_range.shift);
}

int check_bn(drp_ctx* c, int B, int N) {
    if (B <= 0 || N <= 0 || N > 4096) return fail(c, DRP_EINVAL, "bad shape B=%d N=%d (N <= 4096)", B, N);
    return DRP_OK;
}

}  // namespace
