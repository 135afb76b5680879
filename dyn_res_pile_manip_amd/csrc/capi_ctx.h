// capi_ctx.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: the kernel-variant names, the run-time RCCL binding and the context (struct drp_ctx): every workspace, session and switch.

namespace {

std::string g_create_error;

enum KClass { KC_GRAPH = 0, KC_NODE_ENCODE, KC_EDGE_ENCODE, KC_PROJECT, KC_AGGREGATE, KC_UPDATE,
              KC_PREDICT, KC_REWARD, KC_MPPI, KC_PROP, KC_TAPE_COPY, KC_BWD_REWARD, KC_BWD_LISTS, KC_BWD_NODE, KC_BWD_EDGE,
              KC_BWD_PUSH, KC_OPT, KC_COUNT };
const char* const kclass_names[KC_COUNT] = {"graph", "node_encode", "edge_encode", "project",
                                            "aggregate", "update", "predict", "reward", "mppi", "prop",
                                            "tape_copy", "bwd_reward", "bwd_lists", "bwd_node", "bwd_edge", "bwd_push", "opt"};

// ---- which kernel variant served a launch (drp_last_dispatch) ---------------------------------------------------
// Every place that chooses between kernels or template instantiations marks the variant it launched in the context; the
// host asks for the names (drp_last_dispatch) and for the whole list (drp_dispatch_variants).  tests/test_gpu_fuzz_oracle.py
// draws shapes under the default dispatch, checks each against the oracle and fails if a variant in the list was never hit:
// a threshold change that orphans an instantiation turns the suite red.
enum DispatchVariant {
    DV_GRAPH_PLAIN = 0, DV_GRAPH_Q4, DV_GRAPH_Q4_ENCODE, DV_GRAPH_STRIPS, DV_GRAPH_STRIPS256, DV_GRAPH_CELLS, DV_GRAPH_REV, DV_GRAPH_IN_ROLLOUT,
    DV_VALU_STEP, DV_NODE_ENCODE, DV_NODE_ENCODE_SPLIT, DV_EDGE_ENCODE, DV_EDGE_ENCODE_SPLIT, DV_AGGREGATE, DV_AGGREGATE_LDS,
    DV_AGGREGATE_TAPE, DV_UPDATE,
    DV_PROP,                        // + 8 LAST + 4 TAPE + 2 PAIR + WORK
    DV_PROP3 = DV_PROP + 16,        // + 12 TAPE + 6 PAIR + 2 cache (0 off, 1 on, 2 on with the rows kept in registers) + WORK
    DV_ROLLOUT = DV_PROP3 + 24,     // + 6 PAIR + 2 cache + WORK
    DV_REWARD = DV_ROLLOUT + 12, DV_BWD_REWARD, DV_REV_256, DV_REV_1024, DV_BWD_ROWS, DV_BWD_STEP, DV_BWD_STAGES_MFMA,
    DV_BWD_STAGES_VALU, DV_BWD_EDGE_MFMA, DV_BWD_EDGE_VALU, DV_TRAIN_NODE_FUSED, DV_TRAIN_NODE_FUSED_COOP, DV_TRAIN_NODE_MFMA, DV_TRAIN_NODE_VALU, DV_WGRAD_MFMA, DV_WGRAD_VALU,
    DV_WGRAD_DEFERRED, DV_MPPI_SOFTMAX, DV_ELITE_SORT, DV_ELITE_ROUNDS, DV_FPS_REG, DV_FPS_MEM, DV_DT_CV5, DV_DT_EXACT,
    DV_TRAIN_BARRIER_RETRY,
    DV_COUNT
};
// name of variant `id`; *by_default = reachable without an environment switch (DRP_NO_* / drp_probe_begin("prop+work"))
void dv_name(int id, char* buf, size_t n, bool* by_default) {
    bool dflt = true;
    static const char* const cache_names[3] = {"", ",cache", ",cache+rows"};
    if (id >= DV_PROP && id < DV_PROP3) {
        const int f = id - DV_PROP;
        snprintf(buf, n, "km_prop<%s%s%s%s>", (f & 8) ? "last" : "mid", (f & 4) ? ",tape" : "", (f & 2) ? ",pair" : "", (f & 1) ? ",work" : "");
        dflt = !(f & 1);
    } else if (id >= DV_PROP3 && id < DV_ROLLOUT) {
        const int f = id - DV_PROP3;
        snprintf(buf, n, "km_prop3<%s%s%s%s>", (f / 12) ? "tape" : "plain", ((f / 6) & 1) ? ",pair" : "", cache_names[(f % 6) / 2], (f & 1) ? ",work" : "");
        // paired tiles mean at most 128 rows per workgroup: the cache always fits and the rows stay in registers, unless
        // DRP_ECACHE_MAX_MB says otherwise
        dflt = !(f & 1) && !(((f / 6) & 1) && (f % 6) / 2 != 2);
    } else if (id >= DV_ROLLOUT && id < DV_REWARD) {
        const int f = id - DV_ROLLOUT;
        snprintf(buf, n, "km_rollout<%s%s%s>", (f / 6) ? "pair" : "tile32", cache_names[(f % 6) / 2], (f & 1) ? ",work" : "");
        dflt = !(f & 1) && !((f / 6) && (f % 6) / 2 != 2);
    } else {
        const char* s = "?";
        switch (id) {
        case DV_GRAPH_PLAIN: s = "graph:k_graph"; break;
        case DV_GRAPH_Q4: s = "graph:k_graph_q4"; break;
        case DV_GRAPH_Q4_ENCODE: s = "graph:km_graph_q4_encode (+ particle encoder)"; break;
        case DV_GRAPH_STRIPS: s = "graph:k_graph_strips_q<128>"; break;
        case DV_GRAPH_STRIPS256: s = "graph:k_graph_strips_q<256>"; dflt = false; break;   // from 800 particles, where the cells have taken over (DRP_NO_GRAPH_CELLS=1)
        case DV_GRAPH_CELLS: s = "graph:k_graph_cells"; break;
        case DV_GRAPH_REV: s = "graph:k_graph_rev"; break;
        case DV_GRAPH_IN_ROLLOUT: s = "graph:in km_rollout"; break;
        case DV_VALU_STEP: s = "valu:k_node_encode..k_predict"; break;
        case DV_NODE_ENCODE: s = "km_node_encode"; break;
        case DV_NODE_ENCODE_SPLIT: s = "km_node_encode_split"; break;
        case DV_EDGE_ENCODE: s = "km_edge_encode"; break;
        case DV_EDGE_ENCODE_SPLIT: s = "km_edge_encode_split"; break;
        case DV_AGGREGATE: s = "k_aggregate"; break;
        case DV_AGGREGATE_LDS: s = "k_aggregate_lds"; break;
        case DV_AGGREGATE_TAPE: s = "k_aggregate_tape"; break;
        case DV_UPDATE: s = "km_update"; break;
        case DV_REWARD: s = "k_reward"; break;
        case DV_BWD_REWARD: s = "kb_reward"; break;
        case DV_REV_256: s = "kb_reverse_lists<256>"; break;
        case DV_REV_1024: s = "kb_reverse_lists<1024>"; break;
        case DV_BWD_ROWS: s = "bwd:kmb_rows_bwd"; break;
        case DV_BWD_STEP: s = "bwd:kmb_step_bwd"; break;
        case DV_BWD_STAGES_MFMA: s = "bwd:stages kmb_*"; break;
        case DV_BWD_STAGES_VALU: s = "bwd:stages kb_*"; dflt = false; break;              // DRP_BWD_VALU_STAGES=1 (KMB_MIN_TILES is 1 since round 3)
        case DV_BWD_EDGE_MFMA: s = "bwd:kmb_edge_encode"; break;
        case DV_BWD_EDGE_VALU: s = "bwd:kb_edge_encode"; dflt = false; break;
        case DV_TRAIN_NODE_FUSED: s = "train:kmb_step_bwd<dump>"; dflt = false; break;    // DRP_TRAIN_COOP=0 (by default a workgroup of the one-launch pass has one tile)
        case DV_TRAIN_NODE_FUSED_COOP: s = "train:kmb_step_bwd<dump,coop>"; break;
        case DV_TRAIN_NODE_MFMA: s = "train:stages kmb_*"; break;
        case DV_TRAIN_NODE_VALU: s = "train:stages kb_*"; dflt = false; break;
        case DV_WGRAD_MFMA: s = "train:kt_wgrad_mfma"; break;
        case DV_WGRAD_VALU: s = "train:kt_wgrad"; dflt = false; break;
        case DV_WGRAD_DEFERRED: s = "train:deferred wgrad lists"; break;
        case DV_MPPI_SOFTMAX: s = "mppi:k_mppi_partials+update"; break;
        case DV_ELITE_SORT: s = "mppi:k_elite_local sort"; break;
        case DV_ELITE_ROUNDS: s = "mppi:k_elite_local rounds"; break;
        case DV_FPS_REG: s = "k_fps_reg"; break;
        case DV_FPS_MEM: s = "k_fps"; break;
        case DV_DT_CV5: s = "k_dt_cv5"; break;
        case DV_DT_EXACT: s = "k_edt"; break;
        case DV_TRAIN_BARRIER_RETRY: s = "train:barrier gave up, step re-run with one workgroup per group"; dflt = false; break;   // a shared / masked device
        default: break;
        }
        snprintf(buf, n, "%s", s);
    }
    if (by_default) *by_default = dflt;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// ---- RCCL, bound at run time -----------------------------------------------------------------------------------
// libdrp.so does not link librccl: a process must not end up with two copies of it (PyTorch ships its own
// librccl.so beside the one under /opt/rocm; which of two mapped copies answered a call used to depend on the import
// order).  The first call that needs RCCL takes, in this order: $DRP_RCCL_LIB, the librccl that sits NEXT TO THE HIP RUNTIME
// this library itself runs on (dladdr of hipGetDeviceCount), a librccl the process has already mapped (dl_iterate_phdr),
// /opt/rocm/lib/librccl.so.1 (include/drp.h says the same).
// Only entry points whose ABI has been stable since NCCL 2.4 are used (no ncclConfig_t crosses the boundary).
struct RcclApi {
    void* handle = nullptr;
    std::string path, error;
    int version = 0;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl_find_mapped(struct dl_phdr_info* info, size_t, void* data) {
    const char* name = info->dlpi_name;
    if (name && *name) {
        const char* base = strrchr(name, '/');
        base = base ? base + 1 : name;
        if (strncmp(base, "librccl.so", 10) == 0) {
            *static_cast<std::string*>(data) = name;
            return 1;
        }
    }
    return 0;
}

RcclApi g_rccl;
RcclApi* rccl_api() {
    RcclApi& api = g_rccl;
    static std::once_flag once;
    std::call_once(once, [&api] {
        std::vector<std::string> tries;
        if (const char* e = getenv("DRP_RCCL_LIB")) tries.push_back(e);
        // The RCCL that belongs to the HIP runtime THIS library runs on comes first: a process can hold two HIP runtimes
        // (PyTorch's wheel ships its own copy next to its librccl; imported after this library it does not replace the
        // system runtime this library is already bound to), and an RCCL talking to the other one finds no device
        // (ncclCommInitRank: "no ROCm-capable device is detected").
        {
            Dl_info hi;
            if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &hi) && hi.dli_fname) {
                std::string dir(hi.dli_fname);
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) {
                    dir.resize(slash + 1);
                    tries.push_back(dir + "librccl.so.1");
                    tries.push_back(dir + "librccl.so");
                }
            }
        }
        std::string mapped;
        dl_iterate_phdr(rccl_find_mapped, &mapped);
        if (!mapped.empty()) tries.push_back(mapped);
        tries.push_back("librccl.so.1");
        tries.push_back("/opt/rocm/lib/librccl.so.1");
        tries.push_back("librccl.so");
        for (const std::string& t : tries) {
            api.handle = dlopen(t.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            const char* de = dlerror();
            api.error += t + ": " + (de ? de : "?") + "; ";
        }
        if (!api.handle) return;
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(api.handle, n); if (!p) { ok = false; api.error += std::string(n) + " missing; "; } return p; };
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
        api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) { dlclose(api.handle); api.handle = nullptr; return; }
        Dl_info di;
        if (dladdr(reinterpret_cast<void*>(api.AllGather), &di) && di.dli_fname) api.path = di.dli_fname;
        (void)api.GetVersion(&api.version);
    });
    return api.handle ? &api : nullptr;
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

}  // namespace

struct drp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int engine = DRP_ENGINE_VALU;
    int n_cu = 256;
    bool agg_global_only = false;   // always gather sender rows from L2/HBM (timing builds)
    bool rev_global_only = false;   // DRP_REV_GLOBAL=1: reversed neighbour lists built in global memory (the N > 3072 path)
    bool self_const = true;         // DRP_NO_SELF_CONST=1: always run the encoder chain on the self slot too
    bool prop3 = true;              // DRP_NO_PROP3=1: one launch per propagation step even for chip-filling batches
    int prop3_min_b = 0;            // km_prop3 / kmb_step_bwd from this many samples (0: whole_samples() decides)
    int prop3_min_tiles = 1;        // km_prop3 from this many tiles per workgroup and step
    int bwd_fused_min_tiles = 1;    // the same for kmb_step_bwd
    bool graph_cells = true;        // DRP_NO_GRAPH_CELLS=1: x strips only (k_graph_strips) for large samples
    int graph_cells_min_n = 400;    // DRP_GRAPH_CELLS_MIN_N: two-dimensional cells from this many particles up (measured: slower at 300, 8 % faster at 450)
    float graph_cells_halo = 0.0f;  // DRP_GRAPH_CELLS_HALO: first-sweep halo in camera-frame units (default: from the particle count)
    float graph_cells_hb = 0.0f;    // DRP_GRAPH_CELLS_HB: band height in camera-frame units (default: from the particle count)
    bool graph_strips = true;       // DRP_NO_GRAPH_STRIPS=1: plain neighbour sweep for every shape
    bool comm_always = false;       // DRP_COMM_ALWAYS=1: a one-rank communicator still goes through ncclAllGather (bench.py --force-comm)
    bool bwd_fused = true;          // DRP_NO_BWD_FUSED=1: the GD planner's backward pass as one launch per stage
    bool graph_rev = true;          // DRP_NO_GRAPH_REV=1: the GD planner's reversed lists always in a launch of their own (kb_reverse_lists)
    bool graph_encode = true;       // DRP_NO_GRAPH_ENCODE=1: k_graph_q4 and km_node_encode_split as two launches where they could be one (km_graph_q4_encode)
    int train_fused = -1;           // DRP_TRAIN_FUSED=0/1: the trainer's node stages as one launch per rollout step (kmb_step_bwd<dump>) never / for any batch (-1: up to n_cu / 4 tiles)
    int train_coop = -1;            // DRP_TRAIN_COOP=0/1: the workgroup-wide gather of the edge terms off / on whatever the tile count (-1: by tiles per workgroup)
    double* tr_loss_host = nullptr; // drp_train_step: pinned host memory the loss kernel stores its terms to (null: c->tr_loss)
    bool train_copy_upload = false; // DRP_TRAIN_COPY_UPLOAD=1: the training batch goes up by a copy on the stream instead of inside kt_unpack_inputs
    bool debug_force_giveup = false; // DRP_DEBUG_FORCE_GIVEUP=1 (tests): drp_train_step's first pass ends as if kmb_step_bwd's barrier had timed out
    int train_parts = 0;            // DRP_TRAIN_PARTS=n: workgroups per group of samples in the trainer's kmb_step_bwd (0: as many as there are CUs for)
    bool bwd_valu_stages = false;   // DRP_BWD_VALU_STAGES=1: the reverse-mode node stages on the VALU row kernels (kb_predict ... kb_node_encode; cross-check)
    bool bwd_rows = true;           // DRP_NO_BWD_ROWS=1: piles of up to 256 particles through kmb_step_bwd (rows through memory) instead of kmb_rows_bwd
    bool prop3_order = true;        // false: km_prop3's tiles in the natural row order instead of by in-degree
    int prop_pair_rows = 128;       // DRP_PROP_PAIR_ROWS: a workgroup of the whole-sample kernels with up to so many rows runs tiles of
                                    // 16 receivers x two slots (0 = never)
    int prop_pair_always = 64;      // DRP_PROP_PAIR_ALWAYS: ... whatever the in-degrees up to so many rows (four tiles of 16: a SIMD each),
    int prop_pair_deg10 = 83;       // DRP_PROP_PAIR_DEG10: above that while the piles' mean in-degree (x 10) is at most this
    // the mean in-degree the last lists of this shape had (k_deg_stat, every few launches): sum | rows << 24 | N << 48 in
    // host memory the device writes; only ever a question of speed -- paired and unpaired tiles give the same bits
    unsigned long long* deg_stat = nullptr;
    unsigned long long* deg_stat_dev = nullptr;
    unsigned deg_tick = 0;
    bool prop_pair(long spw, long N, long B) const {
        const long rows = spw * N;
        if (rows > prop_pair_rows) return false;
        if (rows <= prop_pair_always || deg_stat == nullptr) return true;
        const unsigned long long v = *reinterpret_cast<volatile const unsigned long long*>(deg_stat);
        const long sum = (long)(v & 0xffffffull), st_rows = (long)((v >> 24) & 0xffffffull), st_n = (long)(v >> 48);
        if (st_n != N || st_rows != std::min(B * N, (long)DEG_STAT_MAX_ROWS) || st_rows == 0) return true;   // not known (yet)
        return sum * 10 <= st_rows * (long)prop_pair_deg10;
    }
    bool prop3e = true;             // false: the particle encoder stays its own launch in front of km_prop3
    bool rollout_fused = true;      // DRP_NO_ROLLOUT_FUSED=1: one graph + one km_prop3 launch per rollout step for small piles too
    int rollout_max_n = 64;         // DRP_ROLLOUT_MAX_N: km_rollout (the whole rollout in one launch) up to this many particles ...
    int rollout_mid_n = 256, rollout_mid_rows = 256;  // ... up to 256 particles for workgroups of up to 256 rows (small batches; the
                                    // kernels with the kept rows and the lists beside the encoder: 256 x 80 / 100 / 128 / 150 / 200 / 256
                                    // + 23 / + 14 / + 13 / + 13 / + 5 / + 6 %, 512 x 100 / 128 + 13 / + 16 %, 128 x 150 + 7 %, 341 x 96 + 16 %;
                                    // 64 x 256 - 5 %: above 200 particles only from half a chip of samples; 1024 x 80 / 100 (320 / 400 rows): - 1 %)
    int rollout_max_rows = 704;     // ... and this many rows (samples x particles) per workgroup.  Measured
                                    // against the step-by-step pipeline at 1024 samples: +18 % at 10 particles, +12 % at 20, +2 % at
                                    // 50, +5 % at 64, -1 % at 80, -10 % at 150 (the strip build wins); 50 particles x 4096 samples
                                    // (800 rows per workgroup) -5 %, 20 x 8192 (640 rows) +7 %

    // Edge-chain cache of the whole-sample kernels (prop_tiles, EC): the relation encoder's chain runs in the first propagation
    // step only and its output is read back in the other two, from a workgroup-private buffer of 80 KB per tile of 32 receivers
    // (2.5 KB per receiver).  The cached kernels differ from the recomputing ones in the last place of one sum, so WHICH of the
    // two serves a sample must not depend on how many samples travel with it (a 1 024-sample shard of an 8 192-sample job, a
    // rank's half of the planner's 1 500 rows: the sharded and the unsharded run must agree bit for bit): the choice is a function
    // of the PILE SIZE alone (ec_shape; DRP_ECACHE_MAX_MB=0: never) -- and the buffer stays small by construction instead: a
    // cached launch gives a workgroup at most ec_rows_cap(N) rows, and a batch that needs more than one such launch is run as
    // several, one after the other on the stream, over the same buffer (run_rollout, run_step_mfma; 256 workgroups x 9 tiles
    // x 80 KB = 189 MB, inside the 256 MB of last-level cache).
    // Which pile sizes: measured with the blocks in place (tools/ab_env_shapes.sh, DRP_ECACHE_MAX_N=64 against 256, one box):
    // 256 samples x 80 / 100 / 150 / 200 particles + 15 / + 31 / + 35 / + 19 %, 1 024 x 80 / 100 / 128 / 256 + 8 / + 7 / + 7 /
    // + 3 %, but 1 024 x 150 - 12 % and x 200 - 5 %: one sample of 129 ... 224 particles leaves three to one of a workgroup's
    // eight waves without a tile.  So: up to ecache_max_n = 128 particles (two samples of up to 128 fill the eight tiles), and
    // ecache_full_n = 225 ... 256 (one sample, eight tiles).  The TAPE's launches (gradient-descent planner, trainer) write one
    // history buffer over the whole batch and are not split: their cache covers the whole batch, which pays up to
    // ecache_tape_max_n = 40 particles at the planner's 1 500 rows (50 particles: 0.398 ms per iteration recomputing, 0.42 cached).
    int ecache_max_mb = 192;
    int ecache_hard_max_mb = 4096;  // a cached launch that cannot be split (the tape's: 1 500 x 40 rows are 150 MB) and would need more recomputes:
                                    // 1.6 million rows -- no caller of the reference comes near; the one place where the batch decides the kernel
    int ecache_max_n = 128, ecache_full_n = 225, ecache_tape_max_n = 40;
    DevBuf ecache;
    // how many float4 a workgroup of `rows` receivers needs
    static size_t ecache_stride(long rows, bool pair) {
        const long tiles = pair ? (rows + 15) / 16 : (rows + 31) / 32;
        return (size_t)tiles * (pair ? 5 : DRP_K) * EC_UNITS;
    }
    bool ec_shape(int N, bool tape = false) const {
        if (ecache_max_mb <= 0) return false;
        if (tape) return N <= ecache_tape_max_n;
        return N <= ecache_max_n || (N >= ecache_full_n && N <= 256);
    }
    // rows a workgroup of a cached launch may hold: nine tiles of 32 (up to 64 particles: the measured best at 1 024 x 64 is
    // four samples = eight tiles), eight -- one per wave, rows kept in registers -- above
    static long ec_rows_cap(int N) { return N <= 64 ? 288 : 256; }
    // samples per launch of a cached shape: every CU a workgroup of at most ec_rows_cap rows, in whole multiples of `unit`
    // (the batch columns: row b reads column b % unit of the replicated inputs)
    long ec_chunk(int N, long unit) const {
        const long spw = std::max(1L, ec_rows_cap(N) / N);
        long chunk = (long)n_cu * spw;
        if (unit > 1) chunk = chunk / unit * unit;
        return chunk;
    }

    // model constants
    bool have_weights = false, have_cam = false, have_goal = false;
    float adj_thresh = 0.08f, thr = 0.0064f;
    SplitRange re_range{};          // range shift 2^k of the split relation encoder and the bound it rests on
    float re_scale = 1.0f, re_inv = 1.0f;
    bool re_ok = true;
    int re_shift_env = 0x7fffffff;  // a fixed shift k instead of the one derived from the weights (experiments)
    DevBuf w_raw, w_valu, w_mfma, w_mfma_bwd, w_split, w_split6, w_split6_bwd;
    DrpCam cam{};
    DevBuf goal_field, goal_coor, cself;
    unsigned cself_tag = 0;         // bumped by every prepare_cself: who filled c->cself last
    int goal_h = 0, goal_w = 0, goal_m = 0;

    // workspaces
    DevBuf s_in, attr, dens, s_delta, nbr_idx, nbr_cnt, eff, c_node, agg, proj, c_edge, states,
        actions, rewards, s_out, scratch, proj2;

    // MPC state
    bool mpc_on = false;
    unsigned mpc_cself_tag = 0;     // the session's self-edge constants are in c->cself while this equals cself_tag
    const float* mpc_cself = nullptr;
    const uint8_t* mpc_cself_ok = nullptr;
    float sess_attr_max = 0.0f, sess_dens_max = 0.0f;   // of the running MPC session (range check of later uploads)
    drp_mpc_params mpc{};
    DevBuf nominal, noise, partials, gathered, stats, elite, elite_all, xchg;
    int n_ranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    bool comm_failed = false;            // a wait gave up or RCCL reported an error: the communicator is gone and every entry point
    int comm_failed_ranks = 0;           // that would use it answers DRP_ECOMM until drp_comm_destroy / a fresh drp_comm_init
    double comm_timeout_s = 60.0;        // DRP_COMM_TIMEOUT_S: a wait behind a collective gives up after this long (guarded_wait)
    double comm_init_timeout_s = 300.0;  // DRP_COMM_INIT_TIMEOUT_S: ncclCommInitRank (every rank must arrive)

    // gradient-descent planner state
    int gd_engine = DRP_ENGINE_FUSED, tr_engine = DRP_ENGINE_FUSED;   // which engine writes the tape (pick_tape_engine)
    bool gd_on = false;
    int gd_nb = 0, gd_N = 0, gd_B = 0, gd_H = 0, gd_iter = 0;
    float* gd_pin[DRP_GD_SLOTS] = {};        // drp_gd_step_async: pinned host copies [B rewards | B*H*4 pushes] of the iterations in flight,
    size_t gd_pin_floats = 0;                //   written by the iteration's own kernels (kb_reward, k_adam): no copy on the stream
    hipEvent_t gd_ev[DRP_GD_SLOTS] = {};
    float* gd_host_rewards = nullptr;        // where the iteration being enqueued writes them (null: device buffers only)
    float* gd_host_actions = nullptr;
    KbAdam gd_adam = KbAdam{};               // gd_iteration: the optimiser step rides on the last kb_sdelta launch (act == null: gradients only)
    bool gd_pending[DRP_GD_SLOTS] = {};
    float* mpc_pin[2] = {nullptr, nullptr};  // drp_mpc_fetch_async: [B*H*4 pushes | B final rewards] of two iterations in flight
    size_t mpc_pin_floats = 0;
    hipEvent_t mpc_ev[2] = {nullptr, nullptr};
    bool mpc_pending[2] = {false, false};
    unsigned gd_cself_tag = 0;      // the self-edge constants of this GD problem are in c->cself while the tags match
    const float* gd_cself = nullptr;
    const uint8_t* gd_cself_ok = nullptr;
    double gd_lr = 0.05;
    float gd_lo[4] = {0, 0, 0, 0}, gd_hi[4] = {0, 0, 0, 0};
    DevBuf eff_hist, g_eff, g_cnode, g_agg, g_proj, g_state, g_sdelta, g_act, adam_m, adam_v;
    DevBuf tape_sdelta, tape_idx, tape_cnt, tape_mask, g_agg_hist, rev_off, rev, gpos_edge;

    // particle extraction (row f2)
    DevBuf px_depth, px_mask, px_blk, px_bmin, px_bmax, px_grid, px_pcd, px_keys, px_cellcnt, px_cellfill,
        px_celloff, px_list, px_down, px_down32, px_init, px_dist, px_chosen, px_pts, px_r, px_rr, px_out;

    // training (row f4)
    bool tr_on = false;
    int tr_nroll = 0, tr_iter = 0;
    double tr_lr = 1e-3, tr_beta1 = 0.9;
    std::vector<float> w_host;
    std::vector<WgradJob> wg_jobs;  // weight-gradient jobs waiting for the next flush_wgrad
    // DEFERRED weight gradients (training, DRP_NO_WGRAD_DEFER=1 turns it off): every operand of an iteration's jobs keeps a
    // buffer of its own (per rollout step, per propagation step), the jobs queue up for the whole backward pass and go out
    // in a handful of launches at its end (flush_wgrad_all) instead of 25 pairs in between
    bool wgrad_defer = true, wg_defer_now = false;
    DevBuf wg_jobs_dev, wg_idx_dev;
    std::vector<unsigned char> wg_uploaded;     // what wg_jobs_dev / wg_idx_dev hold (re-uploaded when the iteration's jobs change)
    DevBuf tr_part, tr_grad, tr_m, tr_v, tr_loss, agg_hist, tr_hact, tr_gh, tr_gpe, tr_a1n,
        tr_gh1, tr_xn, ed_re, ed_a2, ed_a1, ed_x0, ed_gce, ed_g3, ed_g2, ed_g1;

    // goal pre-processing (row f3)
    DevBuf gl_goal, gl_seg, gl_tmp, gl_dist, gl_blk, gl_pix, gl_fps;

    // re-packing after an optimiser step on the device (k_train.h): gather maps of the plain packers, pinned copy of the blob
    DevBuf map_valu, map_mfma, map_mfma_bwd;
    bool repack_maps_ready = false;
    float* w_pin = nullptr;         // pinned: the blob after an optimiser step [W_TOTAL], then the device's range shift (one int)
    void* tr_pin = nullptr;         // pinned staging of a training batch (drp_train_step: one upload)
    size_t tr_pin_cap = 0;
    DevBuf tr_arena, re_shift_dev;  // the batch as uploaded; the shift kt_repack_all derived
    int graph_q4 = 1;               // DRP_GRAPH_Q4=0 / 1 / 2: four threads per receiver in the plain neighbour sweep -- never / for a handful
                                    // of samples (fewer workgroups than half the CUs) / whenever the plain sweep is chosen
    bool wgrad_mfma = true;         // DRP_NO_WGRAD_MFMA=1: the weight gradients' outer-product sums on the VALU kernel (kt_wgrad_multi)
    bool prop_spread = true;        // DRP_NO_PROP_SPREAD=1: km_prop's tiles eight to a workgroup whatever their number
    bool bwd_edge_mfma = true;      // DRP_NO_BWD_EDGE_MFMA=1: the relation encoder's backward on the VALU kernel (kb_edge_encode)
    bool repack_device = true;      // DRP_NO_REPACK_DEVICE=1: fetch the blob and run the host packers (the round-2 path)

    // km_rollout's argument block (device copy + what it holds)
    DevBuf roll_args;
    std::vector<RolloutArgs> roll_args_host;
    bool roll_args_valid = false;

    // last shapes (for debug fetch)
    int lastB = 0, lastN = 0, lastH = 0;

    // kernel variants launched since drp_dispatch_reset (DispatchVariant)
    unsigned char dv_hit[DV_COUNT] = {};
    void dv(int id) { dv_hit[id] = 1; }

    // probe
    DevBuf probe_work;              // PROP_WORK_* counters of the propagation kernels while their class is probed
    bool probe_count = false;       // drp_probe_begin("prop+work"): the kernels count what they execute (not for timed regions: the
                                    // counting costs the 300-particle launch 8 %)
    unsigned long long* work_ptr() const { return (probe_cls == KC_PROP && probe_count) ? static_cast<unsigned long long*>(probe_work.p) : nullptr; }
    int probe_cls = -1;
    std::vector<hipEvent_t> probe_ev;
    size_t probe_used = 0;
};

