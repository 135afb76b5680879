// C ABI of the engine (include/drp.h): context, device workspaces, kernel pipelines.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c (this file and csrc/inst_*.hip, in parallel), then -shared
// (__graft_entry__.build)
#include "../../include/drp.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enums only: the library itself is bound at run time (RcclApi below)

#include <dlfcn.h>
#include <link.h>
#include <sched.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "drp_common.h"
#include "k_aggregate.h"
#include "k_graph.h"
#include "k_mlp_valu.h"
#include "k_mppi.h"
#include "k_reward.h"
#include "k_backward.h"
#include "k_fps.h"
#include "k_particles.h"
#include "k_goal.h"
#include "k_train.h"
#include "k_mlp_mfma.h"
#include "k_mlp_split.h"
#include "k_backward_mfma.h"
#include "k_rollout.h"
#include "k_prop_inst.h"       // km_prop / km_prop3 / km_rollout: declared here, instantiated in inst_*.hip

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
    DV_GRAPH_PLAIN = 0, DV_GRAPH_Q4, DV_GRAPH_STRIPS, DV_GRAPH_STRIPS256, DV_GRAPH_CELLS, DV_GRAPH_REV, DV_GRAPH_IN_ROLLOUT,
    DV_VALU_STEP, DV_NODE_ENCODE, DV_NODE_ENCODE_SPLIT, DV_EDGE_ENCODE, DV_EDGE_ENCODE_SPLIT, DV_AGGREGATE, DV_AGGREGATE_LDS,
    DV_AGGREGATE_TAPE, DV_UPDATE,
    DV_PROP,                        // + 8 LAST + 4 TAPE + 2 PAIR + WORK
    DV_PROP3 = DV_PROP + 16,        // + 12 TAPE + 6 PAIR + 2 cache (0 off, 1 on, 2 on with the rows kept in registers) + WORK
    DV_ROLLOUT = DV_PROP3 + 24,     // + 6 PAIR + 2 cache + WORK
    DV_REWARD = DV_ROLLOUT + 12, DV_BWD_REWARD, DV_REV_256, DV_REV_1024, DV_BWD_ROWS, DV_BWD_STEP, DV_BWD_STAGES_MFMA,
    DV_BWD_STAGES_VALU, DV_BWD_EDGE_MFMA, DV_BWD_EDGE_VALU, DV_TRAIN_NODE_MFMA, DV_TRAIN_NODE_VALU, DV_WGRAD_MFMA, DV_WGRAD_VALU,
    DV_WGRAD_DEFERRED, DV_MPPI_SOFTMAX, DV_ELITE_SORT, DV_ELITE_ROUNDS, DV_FPS_REG, DV_FPS_MEM, DV_DT_CV5, DV_DT_EXACT,
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
    DevBuf tr_part, tr_states, tr_sdelta, tr_nums, tr_grad, tr_m, tr_v, tr_loss, agg_hist, tr_hact, tr_gh, tr_gpe, tr_a1n,
        tr_gh1, tr_xn, ed_re, ed_a2, ed_a1, ed_x0, ed_gce, ed_g3, ed_g2, ed_g1;

    // goal pre-processing (row f3)
    DevBuf gl_goal, gl_seg, gl_tmp, gl_dist, gl_blk, gl_pix, gl_fps;

    // re-packing after an optimiser step on the device (k_train.h): gather maps of the plain packers, pinned copy of the blob
    DevBuf map_valu, map_mfma, map_mfma_bwd;
    bool repack_maps_ready = false;
    float* w_pin = nullptr;
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
    else if (c->graph_q4 != 0 && N >= 64 && (c->graph_q4 == 2 || (long)B * ((N + 127) / 128) * 2 <= c->n_cu)) {
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
    if (!phase_e) {
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
            // cache buffer under each; the tape's launches (one history buffer over the whole batch) take a larger buffer instead
            const bool ec = c->ec_shape(N, tape);
            long chunk = B;
            if (ec && !tape) {
                long unit = 1;
                bool ok = true;
                for (int mod : {a.prev_mod, a.attr_mod, a.dens_mod})
                    if (mod < B) { if (unit % mod != 0 && mod % unit != 0) ok = false; else unit = std::max(unit, (long)mod); }
                const long cap = c->ec_chunk(N, unit);
                if (ok && cap > 0 && cap < B) chunk = cap;
            }
            {
                const long B0 = std::min((long)B, chunk), spw0 = (B0 + c->n_cu - 1) / c->n_cu;
                if (ec) CHK(ensure(c, c->ecache, (size_t)((B0 + spw0 - 1) / spw0) * drp_ctx::ecache_stride(spw0 * N, false) * 16));
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
                float* eff_base = (tape ? a.eff_hist : ptr<float>(c->eff)) + ro * 64;       // (the tape's launch is never split: ro = 0)
                unsigned* mask_hist = tape ? a.mask_hist : nullptr;
                float* agg_hist = tape ? a.agg_hist : nullptr;
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
                    if (wk) hipLaunchKernelGGL((km_prop3<TAPE_, PAIR_, EC_, true, ONE_>), grid, pblk, KM_PROP3_LDS, st, PROP3_ARGS, ptr<float4>(c->ecache), ec_stride, wk); \
                    else hipLaunchKernelGGL((km_prop3<TAPE_, PAIR_, EC_, false, ONE_>), grid, pblk, KM_PROP3_LDS, st, PROP3_ARGS, ptr<float4>(c->ecache), ec_stride, wk); } while (0)
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
    if (a.build_graph) {
        ProbeScope ps(c, KC_GRAPH);
        const int self_first = (c->engine == DRP_ENGINE_FUSED && a.cself != nullptr) ? 1 : 0;
        if (a.rev_off != nullptr && N <= GRAPH_THREADS && c->graph_rev) {
            c->dv(DV_GRAPH_REV);
            hipLaunchKernelGGL(k_graph_rev, dim3(SPREAD_GRID(B)), dim3(GRAPH_THREADS), (size_t)12 * N * sizeof(int), st, a.s_prev,
                               a.prev_mod, a.prev_stride, a.actions, a.act_stride, s_delta, N, nbr_idx, nbr_cnt, c->cam, c->thr,
                               B, self_first, a.rev_off, a.rev);
            if (a.rev_built) *a.rev_built = true;
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
    split_range_init(blob, c->re_range, 2.0, 1.5, 2.0);
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

extern "C" {

int drp_create(int device, drp_ctx** out) {
    if (!out) return fail(nullptr, DRP_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, DRP_EHIP, "no HIP device available: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, DRP_EINVAL, "device %d out of range (%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, DRP_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
    drp_ctx* c = new drp_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, DRP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->n_cu = prop.multiProcessorCount;
    c->self_const = getenv("DRP_NO_SELF_CONST") == nullptr;
    c->prop3 = getenv("DRP_NO_PROP3") == nullptr;
    c->graph_strips = getenv("DRP_NO_GRAPH_STRIPS") == nullptr;
    c->graph_cells = getenv("DRP_NO_GRAPH_CELLS") == nullptr;
    if (const char* e = getenv("DRP_GRAPH_CELLS_MIN_N")) c->graph_cells_min_n = atoi(e);
    if (const char* e = getenv("DRP_GRAPH_CELLS_HB")) c->graph_cells_hb = (float)atof(e);
    if (const char* e = getenv("DRP_GRAPH_CELLS_HALO")) c->graph_cells_halo = (float)atof(e);
    c->rollout_fused = getenv("DRP_NO_ROLLOUT_FUSED") == nullptr;
    c->repack_device = getenv("DRP_NO_REPACK_DEVICE") == nullptr;
    c->bwd_edge_mfma = getenv("DRP_NO_BWD_EDGE_MFMA") == nullptr;
    c->prop_spread = getenv("DRP_NO_PROP_SPREAD") == nullptr;
    c->wgrad_mfma = getenv("DRP_NO_WGRAD_MFMA") == nullptr;
    c->wgrad_defer = getenv("DRP_NO_WGRAD_DEFER") == nullptr;
    if (const char* e = getenv("DRP_GRAPH_Q4")) c->graph_q4 = atoi(e);
    if (const char* e = getenv("DRP_ROLLOUT_MAX_N")) { c->rollout_max_n = atoi(e); c->rollout_mid_n = 0; c->rollout_max_rows = KM_ROLLOUT_MAX_ROWS; }
    if (const char* e = getenv("DRP_PROP_PAIR_ROWS")) c->prop_pair_rows = std::min(256, std::max(0, atoi(e)));   // km_rollout<pair> keeps 16 B per row in the 4 KB behind the encoder's matrices
    if (const char* e = getenv("DRP_PROP_PAIR_ALWAYS")) c->prop_pair_always = std::max(0, atoi(e));
    if (const char* e = getenv("DRP_PROP_PAIR_DEG10")) c->prop_pair_deg10 = std::max(0, atoi(e));
    c->bwd_fused = getenv("DRP_NO_BWD_FUSED") == nullptr;
    c->bwd_rows = getenv("DRP_NO_BWD_ROWS") == nullptr;
    c->bwd_valu_stages = getenv("DRP_BWD_VALU_STAGES") != nullptr;
    c->graph_rev = getenv("DRP_NO_GRAPH_REV") == nullptr;
    c->comm_always = getenv("DRP_COMM_ALWAYS") != nullptr;
    if (const char* e = getenv("DRP_COMM_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) c->comm_timeout_s = v; }
    if (const char* e = getenv("DRP_COMM_INIT_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) c->comm_init_timeout_s = v; }
    c->rev_global_only = getenv("DRP_REV_GLOBAL") != nullptr;
    if (const char* e = getenv("DRP_ECACHE_MAX_MB")) c->ecache_max_mb = std::max(0, atoi(e));
    if (const char* e = getenv("DRP_ECACHE_MAX_N")) { c->ecache_max_n = std::max(0, atoi(e)); c->ecache_full_n = 257; }
    if (const char* e = getenv("DRP_ECACHE_TAPE_MAX_N")) c->ecache_tape_max_n = std::max(0, atoi(e));
    if (hipFuncSetAttribute((const void*)k_graph, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_graph_q4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_elite_local, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_elite_update, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_graph_strips_q<GRAPH_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_graph_strips_q<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_reward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_REWARD_LDS(4096)) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_edge_encode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_EDGE_ENCODE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_reverse_lists<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_REV_LDS(KB_REV_LDS_MAX_N, 1)) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_reverse_lists<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_REV_LDS(KB_REV_LDS_MAX_N, 1)) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_aggregate_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                            K_AGG_LDS_MAX_N * 256) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, DRP_EHIP, "hipFuncSetAttribute (dynamic LDS size of k_graph, kb_edge_encode, kb_reverse_lists or k_aggregate_lds) failed");
    }
    // the MFMA kernels keep packed weights + per-wave transposition tiles in LDS (> 64 KiB)
    if (hipFuncSetAttribute((const void*)km_edge_encode, hipFuncAttributeMaxDynamicSharedMemorySize, KM_EDGE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_node_encode, hipFuncAttributeMaxDynamicSharedMemorySize, KM_NODE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_edge_encode_split, hipFuncAttributeMaxDynamicSharedMemorySize, KM_EDGE_SPLIT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_node_encode_split, hipFuncAttributeMaxDynamicSharedMemorySize, KM_NODE_SPLIT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_step_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_FUSED_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_rows_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_ROWS_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_edge_encode, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_EDGE_ENCODE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kt_wgrad_multi, hipFuncAttributeMaxDynamicSharedMemorySize, KT_WGRAD_MULTI_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kt_wgrad_mfma_multi, hipFuncAttributeMaxDynamicSharedMemorySize, KT_WGRAD_MULTI_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_update<false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_UPD_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_update<true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_UPD_LDS) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, DRP_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    }
    c->engine = DRP_ENGINE_FUSED;
    *out = c;
    return DRP_OK;
}

void drp_destroy(drp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)guarded_wait(c, nullptr);           // a collective that cannot finish must not keep the destructor
    helpers_wait(5.0, c);                     // no helper thread of this context (an abort, an init) inside RCCL while its stream goes away
    if (c->comm) { RcclApi* R = rccl_api(); if (R) (void)R->CommDestroy(c->comm); c->comm = nullptr; }
    DevBuf* bufs[] = {&c->probe_work, &c->ecache, &c->tape_mask, &c->g_agg_hist, &c->rev_off, &c->rev, &c->gpos_edge, &c->tape_sdelta, &c->tape_idx, &c->tape_cnt, &c->eff_hist, &c->g_eff, &c->g_cnode, &c->g_agg, &c->g_proj, &c->g_state,
                      &c->g_sdelta, &c->g_act, &c->adam_m, &c->adam_v, &c->w_raw, &c->w_valu, &c->w_mfma, &c->w_mfma_bwd, &c->w_split, &c->w_split6, &c->w_split6_bwd, &c->proj2, &c->goal_field, &c->goal_coor, &c->s_in,
                      &c->attr, &c->dens, &c->s_delta, &c->nbr_idx, &c->nbr_cnt, &c->eff, &c->c_node,
                      &c->agg, &c->proj, &c->c_edge, &c->states, &c->actions, &c->rewards, &c->s_out,
                      &c->scratch, &c->nominal, &c->noise, &c->partials, &c->gathered, &c->stats, &c->elite, &c->elite_all, &c->xchg, &c->cself,
                      &c->px_depth, &c->px_mask, &c->px_blk, &c->px_bmin, &c->px_bmax, &c->px_grid, &c->px_pcd, &c->px_keys,
                      &c->px_cellcnt, &c->px_cellfill, &c->px_celloff, &c->px_list, &c->px_down, &c->px_down32, &c->px_init,
                      &c->px_dist, &c->px_chosen, &c->px_pts, &c->px_r, &c->px_rr, &c->px_out,
                      &c->gl_goal, &c->gl_seg, &c->gl_tmp, &c->gl_dist, &c->gl_blk, &c->gl_pix, &c->gl_fps,
                      &c->tr_part, &c->tr_states, &c->tr_sdelta, &c->tr_nums, &c->tr_grad, &c->tr_m, &c->tr_v, &c->tr_loss, &c->agg_hist,
                      &c->tr_hact, &c->tr_gh, &c->tr_gpe, &c->tr_a1n, &c->tr_gh1, &c->tr_xn, &c->ed_re, &c->ed_a2, &c->ed_a1,
                      &c->ed_x0, &c->ed_gce, &c->ed_g3, &c->ed_g2, &c->ed_g1, &c->roll_args, &c->map_valu, &c->map_mfma, &c->map_mfma_bwd,
                      &c->wg_jobs_dev, &c->wg_idx_dev};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t ev : c->probe_ev) (void)hipEventDestroy(ev);
    for (int q = 0; q < DRP_GD_SLOTS; ++q) {
        if (c->gd_pin[q]) (void)hipHostFree(c->gd_pin[q]);
        if (c->gd_ev[q]) (void)hipEventDestroy(c->gd_ev[q]);
    }
    for (int q = 0; q < 2; ++q) {
        if (c->mpc_pin[q]) (void)hipHostFree(c->mpc_pin[q]);
        if (c->mpc_ev[q]) (void)hipEventDestroy(c->mpc_ev[q]);
    }
    if (c->w_pin) (void)hipHostFree(c->w_pin);
    if (c->deg_stat) (void)hipHostFree(c->deg_stat);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* drp_last_error(const drp_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int drp_sync(drp_ctx* c) {
    if (!c) return DRP_EINVAL;
    return guarded_wait(c, nullptr);
}

int drp_set_engine(drp_ctx* c, int engine) {
    if (!c) return DRP_EINVAL;
    if (engine == DRP_ENGINE_VALU) { c->engine = engine; return DRP_OK; }
    if (engine == DRP_ENGINE_MFMA || engine == DRP_ENGINE_SPLIT || engine == DRP_ENGINE_FUSED) {
        c->engine = engine;
        return DRP_OK;
    }
    return fail(c, DRP_EINVAL, "engine %d not available in this build", engine);
}

int drp_device_info(drp_ctx* c, char* name, size_t name_len, int* n_cu, size_t* hbm_bytes) {
    if (!c) return DRP_EINVAL;
    hipDeviceProp_t p;
    HIPCHK(c, hipGetDeviceProperties(&p, c->device));
    if (name && name_len) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return DRP_OK;
}

int drp_load_weights(drp_ctx* c, const float* blob, size_t n_floats, float adj_thresh) {
    if (!c || !blob) return DRP_EINVAL;
    if (n_floats != (size_t)W_TOTAL)
        return fail(c, DRP_EINVAL, "weight blob has %zu floats, expected %d", n_floats, (int)W_TOTAL);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<float> v;
    pack_valu(blob, v);
    CHK(h2d(c, c->w_raw, blob, n_floats * sizeof(float)));
    CHK(h2d(c, c->w_valu, v.data(), v.size() * sizeof(float)));
    {
        std::vector<float> m;
        pack_mfma(blob, m);
        CHK(h2d(c, c->w_mfma, m.data(), m.size() * sizeof(float)));
        std::vector<float> mbv;
        pack_mfma_bwd(blob, mbv);
        CHK(h2d(c, c->w_mfma_bwd, mbv.data(), mbv.size() * sizeof(float)));
        CHK(guarded_wait(c, nullptr));     // mbv is about to go out of scope... kept alive until here
        set_split_range(c, blob);
        std::vector<uint16_t> sp;
        pack_split(blob, sp, c->re_range.shift);
        CHK(h2d(c, c->w_split, sp.data(), sp.size() * sizeof(uint16_t)));
        std::vector<uint16_t> sp6;
        pack_split6(blob, sp6);
        CHK(h2d(c, c->w_split6, sp6.data(), sp6.size() * sizeof(uint16_t)));
        // the transposed layers of the GD planner's backward pass in the same split: packed on the device from the raw blob
        CHK(ensure(c, c->w_split6_bwd, (size_t)SB6_TOTAL * 16));
        hipLaunchKernelGGL(kt_repack_split6_bwd, dim3(6 * 16), dim3(256), 0, c->stream, ptr<float>(c->w_raw), ptr<uint16_t>(c->w_split6_bwd));
        CHK(guarded_wait(c, nullptr));     // sp6 too
        CHK(guarded_wait(c, nullptr));     // m, sp are about to go out of scope
    }
    CHK(guarded_wait(c, nullptr));
    c->w_host.assign(blob, blob + n_floats);
    c->adj_thresh = adj_thresh;
    // threshold = adj_thresh * adj_thresh in Python doubles, then an fp32 scalar
    // (model/gnn_dyn.py:229,236)
    c->thr = (float)((double)adj_thresh * (double)adj_thresh);
    c->have_weights = true;
    return DRP_OK;
}

int drp_set_camera(drp_ctx* c, const float m34[12], float global_scale, const float intr[4]) {
    if (!c || !m34 || !intr) return DRP_EINVAL;
    memcpy(c->cam.m, m34, 12 * sizeof(float));
    c->cam.gs = global_scale;
    c->cam.fx = intr[0]; c->cam.fy = intr[1]; c->cam.cx = intr[2]; c->cam.cy = intr[3];
    c->have_cam = true;
    return DRP_OK;
}

int drp_set_goal(drp_ctx* c, const float* field, int h, int w, const float* goal_coor, int m) {
    if (!c || !field || !goal_coor || h <= 0 || w <= 0 || m <= 0) return fail(c, DRP_EINVAL, "bad goal");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->goal_field, field, (size_t)h * w * sizeof(float)));
    CHK(h2d(c, c->goal_coor, goal_coor, (size_t)m * 2 * sizeof(float)));
    CHK(guarded_wait(c, nullptr));
    c->goal_h = h; c->goal_w = w; c->goal_m = m;
    c->have_goal = true;
    return DRP_OK;
}

int drp_gen_s_delta(drp_ctx* c, const float* s_cur, const float* action, int B, int N, float* out) {
    CHK(need(c, false, true, false));
    CHK(check_bn(c, B, N));
    if (!s_cur || !action || !out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(h2d(c, c->s_in, s_cur, (size_t)B * N * 3 * sizeof(float)));
    CHK(h2d(c, c->actions, action, (size_t)B * 4 * sizeof(float)));
    CHK(ensure(c, c->s_delta, (size_t)B * N * 3 * sizeof(float)));
    hipLaunchKernelGGL(k_sdelta, dim3(B), dim3(256), 0, c->stream, ptr<float>(c->s_in),
                       ptr<float>(c->actions), N, ptr<float>(c->s_delta), c->cam);
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, out, c->s_delta.p, (size_t)B * N * 3 * sizeof(float)));
    return drp_sync(c);
}

int drp_build_graph(drp_ctx* c, const float* s_cur, const float* s_delta, int B, int N,
                    int16_t* nbr_idx_out, uint8_t* nbr_cnt_out) {
    CHK(need(c, true, false, false));
    CHK(check_bn(c, B, N));
    if (!s_cur || !s_delta || !nbr_idx_out || !nbr_cnt_out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(ensure_step_ws(c, B, N));
    CHK(h2d(c, c->s_in, s_cur, (size_t)B * N * 3 * sizeof(float)));
    CHK(h2d(c, c->s_delta, s_delta, (size_t)B * N * 3 * sizeof(float)));
    {
    ProbeScope ps(c, KC_GRAPH);
    launch_graph(c, c->stream, ptr<float>(c->s_in), B, (size_t)N * 3, (const float*)nullptr, (size_t)0,
                 ptr<float>(c->s_delta), B, N, ptr<int16_t>(c->nbr_idx), ptr<uint8_t>(c->nbr_cnt), 0, false);
    }
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, nbr_idx_out, c->nbr_idx.p, (size_t)B * N * DRP_K * sizeof(int16_t)));
    CHK(d2h(c, nbr_cnt_out, c->nbr_cnt.p, (size_t)B * N));
    return drp_sync(c);
}

static int step_common(drp_ctx* c, const float* a_cur, const float* s_cur, const float* s_delta,
                       const float* dens, const int16_t* nbr_idx, const uint8_t* nbr_cnt, int B, int N,
                       float* s_pred_out) {
    CHK(need(c, true, false, false));
    CHK(check_bn(c, B, N));
    if (!a_cur || !s_cur || !s_delta || !dens || !s_pred_out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(range_check(c, max_abs(a_cur, (size_t)B * N), max_abs(dens, (size_t)B), max_abs(s_delta, (size_t)B * N * 3)));
    CHK(ensure_step_ws(c, B, N));
    const size_t bn = (size_t)B * N;
    CHK(h2d(c, c->s_in, s_cur, bn * 3 * sizeof(float)));
    CHK(h2d(c, c->s_delta, s_delta, bn * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, a_cur, bn * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)B * sizeof(float)));
    CHK(ensure(c, c->s_out, bn * 3 * sizeof(float)));
    if (nbr_idx) {
        CHK(h2d(c, c->nbr_idx, nbr_idx, bn * DRP_K * sizeof(int16_t)));
        CHK(h2d(c, c->nbr_cnt, nbr_cnt, bn));
    }
    StepArgs a{};
    a.s_prev = ptr<float>(c->s_in); a.prev_mod = B; a.prev_stride = (size_t)N * 3;
    a.attr = ptr<float>(c->attr); a.attr_mod = B;
    a.dens = ptr<float>(c->dens); a.dens_mod = B;
    a.actions = nullptr; a.act_stride = 0;
    a.build_graph = (nbr_idx == nullptr);
    a.s_out = ptr<float>(c->s_out); a.out_stride = (size_t)N * 3;
    a.B = B; a.N = N;
    CHK(run_step(c, a));
    CHK(d2h(c, s_pred_out, c->s_out.p, bn * 3 * sizeof(float)));
    return drp_sync(c);
}

int drp_step(drp_ctx* c, const float* a_cur, const float* s_cur, const float* s_delta,
             const float* dens, int B, int N, float* s_pred_out) {
    return step_common(c, a_cur, s_cur, s_delta, dens, nullptr, nullptr, B, N, s_pred_out);
}

int drp_forward(drp_ctx* c, const float* a_cur, const float* s_cur, const float* s_delta,
                const float* dens, const int16_t* nbr_idx, const uint8_t* nbr_cnt, int B, int N,
                float* s_pred_out) {
    if (!nbr_idx || !nbr_cnt) return fail(c, DRP_EINVAL, "null neighbour lists");
    return step_common(c, a_cur, s_cur, s_delta, dens, nbr_idx, nbr_cnt, B, N, s_pred_out);
}

int drp_rollout(drp_ctx* c, const float* s0, const float* attr, const float* dens, int nb, int N,
                const float* actions, int B, int H, float* states_out, float* reward_out) {
    CHK(need(c, true, true, reward_out != nullptr));
    CHK(check_bn(c, B, N));
    if (!s0 || !attr || !dens || !actions) return fail(c, DRP_EINVAL, "null buffer");
    if (nb <= 0 || H <= 0 || B % nb != 0)
        return fail(c, DRP_EINVAL, "bad rollout shape nb=%d B=%d H=%d (B must be a multiple of nb)", nb, B, H);
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(range_check(c, max_abs(attr, (size_t)nb * N), max_abs(dens, (size_t)nb), push_len_bound(c, actions, (size_t)B * H)));
    CHK(h2d(c, c->s_in, s0, (size_t)nb * N * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, attr, (size_t)nb * N * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)nb * sizeof(float)));
    CHK(h2d(c, c->actions, actions, (size_t)B * H * 4 * sizeof(float)));
    CHK(run_rollout(c, nb, N, B, H, reward_out != nullptr, false));
    if (states_out) CHK(d2h(c, states_out, c->states.p, (size_t)B * H * N * 3 * sizeof(float)));
    if (reward_out) CHK(d2h(c, reward_out, c->rewards.p, (size_t)B * H * sizeof(float)));
    return drp_sync(c);
}

int drp_reward(drp_ctx* c, const float* state, int Bp, int N, int normalize, float* reward_out) {
    CHK(need(c, false, true, true));
    CHK(check_bn(c, Bp, N));
    if (!state || !reward_out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->s_out, state, (size_t)Bp * N * 3 * sizeof(float)));
    CHK(ensure(c, c->scratch, (size_t)Bp * sizeof(float)));
    CHK(run_reward(c, ptr<float>(c->s_out), (size_t)N * 3, Bp, N, normalize, ptr<float>(c->scratch)));
    CHK(d2h(c, reward_out, c->scratch.p, (size_t)Bp * sizeof(float)));
    return drp_sync(c);
}

// ---- sampling MPC -------------------------------------------------------------------------
int drp_mpc_begin(drp_ctx* c, const drp_mpc_params* p, const float* s0, const float* attr,
                  const float* dens, const double* nominal) {
    CHK(need(c, true, true, true));
    if (!p || !s0 || !attr || !dens || !nominal) return fail(c, DRP_EINVAL, "null argument");
    if (p->n_batch <= 0 || p->n_sample <= 0 || p->n_look_ahead <= 0 || p->n_look_ahead > 64)
        return fail(c, DRP_EINVAL, "bad mpc shape");
    if (p->noise_type < DRP_NOISE_NORMAL || p->noise_type > DRP_NOISE_TOTAL_RAND)
        return fail(c, DRP_EINVAL, "bad noise_type %d", p->noise_type);
    const int nb = p->n_batch, N = p->n_particles, H = p->n_look_ahead, B = p->n_sample * nb;
    CHK(check_bn(c, B, N));
    HIPCHK(c, hipSetDevice(c->device));
    {
        // sampled pushes stay inside the clip box: its two longest diagonals bound every impulse
        const float box[8] = {p->act_lo[0], p->act_lo[1], p->act_hi[2], p->act_hi[3],
                              p->act_hi[0], p->act_hi[1], p->act_lo[2], p->act_lo[3]};
        c->sess_attr_max = max_abs(attr, (size_t)nb * N);
        c->sess_dens_max = max_abs(dens, (size_t)nb);
        CHK(range_check(c, c->sess_attr_max, c->sess_dens_max, push_len_bound(c, box, 2)));
    }
    c->mpc = *p;
    CHK(h2d(c, c->s_in, s0, (size_t)nb * N * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, attr, (size_t)nb * N * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)nb * sizeof(float)));
    CHK(h2d(c, c->nominal, nominal, (size_t)H * 4 * sizeof(double)));
    CHK(ensure(c, c->actions, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure(c, c->partials, (size_t)(6 + 4 * H) * sizeof(double)));
    CHK(ensure(c, c->gathered, (size_t)(6 + 4 * H) * sizeof(double) * (size_t)(c->n_ranks > 0 ? c->n_ranks : 1)));
    CHK(ensure(c, c->stats, 8 * sizeof(double)));
    CHK(ensure_step_ws(c, B, N));
    CHK(ensure(c, c->states, (size_t)B * H * N * 3 * sizeof(float)));
    CHK(ensure(c, c->rewards, (size_t)B * H * sizeof(float)));
    CHK(ensure(c, c->scratch, (size_t)B * sizeof(float)));
    CHK(guarded_wait(c, nullptr));
    c->mpc_pending[0] = c->mpc_pending[1] = false;         // a new problem drops what the last one left in flight
    c->mpc_on = true;
    c->gd_on = false;
    c->mpc_cself_tag = 0;           // new attributes / densities / batch size
    return DRP_OK;
}

int drp_mpc_sample(drp_ctx* c, const float* noise, uint64_t iteration) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    const float* dnoise = nullptr;
    if (noise) {
        CHK(h2d(c, c->noise, noise, (size_t)p.n_sample * p.n_look_ahead * 4 * sizeof(float)));
        dnoise = ptr<float>(c->noise);
    }
    ProbeScope ps(c, KC_MPPI);
    hipLaunchKernelGGL(k_mppi_sample, dim3((4 * p.n_sample + 255) / 256), dim3(256), 0, c->stream,
                       ptr<double>(c->nominal), dnoise, p.n_sample, p.n_batch, p.n_look_ahead, p.sigma,
                       p.beta_filter, make_float4(p.act_lo[0], p.act_lo[1], p.act_lo[2], p.act_lo[3]),
                       make_float4(p.act_hi[0], p.act_hi[1], p.act_hi[2], p.act_hi[3]), p.seed,
                       p.sample_offset, iteration, p.noise_type, ptr<float>(c->actions));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_set_actions(drp_ctx* c, const float* actions) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    if (!actions) return fail(c, DRP_EINVAL, "null actions");
    const drp_mpc_params& p = c->mpc;
    CHK(range_check(c, c->sess_attr_max, c->sess_dens_max,
                    push_len_bound(c, actions, (size_t)p.n_sample * p.n_batch * p.n_look_ahead)));
    CHK(h2d(c, c->actions, actions, (size_t)p.n_sample * p.n_batch * p.n_look_ahead * 4 * sizeof(float)));
    return DRP_OK;
}

int drp_mpc_rollout(drp_ctx* c, int reward_all_steps) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    return run_rollout(c, p.n_batch, p.n_particles, p.n_sample * p.n_batch, p.n_look_ahead,
                       reward_all_steps != 0, reward_all_steps == 0, true);
}

static int launch_partials(drp_ctx* c, double* out) {
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead;
    ProbeScope ps(c, KC_MPPI);
    hipLaunchKernelGGL(k_mppi_partials, dim3(4 * H + 1), dim3(256), 0, c->stream,
                       ptr<float>(c->rewards) + (H - 1), H, ptr<float>(c->actions), p.n_sample, p.n_batch,
                       H, p.reward_weight, p.sample_offset, out);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_partials(drp_ctx* c, double* out) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(launch_partials(c, ptr<double>(c->partials)));
    if (out) {
        CHK(d2h(c, out, c->partials.p, (size_t)(6 + 4 * c->mpc.n_look_ahead) * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

static int launch_update(drp_ctx* c, const double* dev_partials, int n_ranks, int rank_stride = 0) {
    const drp_mpc_params& p = c->mpc;
    ProbeScope ps(c, KC_MPPI);
    c->dv(DV_MPPI_SOFTMAX);
    hipLaunchKernelGGL(k_mppi_update, dim3(1), dim3(128), 0, c->stream, dev_partials, n_ranks,
                       rank_stride > 0 ? rank_stride : 6 + 4 * p.n_look_ahead, p.n_look_ahead, (double)p.n_sample * (double)n_ranks, ptr<double>(c->nominal),
                       ptr<double>(c->stats));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_update(drp_ctx* c, const double* partials, int n_ranks, double* nominal_out) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    if (!partials || n_ranks <= 0) return fail(c, DRP_EINVAL, "bad partials");
    const size_t rec = (size_t)(6 + 4 * c->mpc.n_look_ahead) * sizeof(double);
    CHK(h2d(c, c->gathered, partials, rec * n_ranks));
    CHK(launch_update(c, ptr<double>(c->gathered), n_ranks));
    if (nominal_out) {
        CHK(d2h(c, nominal_out, c->nominal.p, (size_t)c->mpc.n_look_ahead * 4 * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

int drp_mpc_update_device(drp_ctx* c) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm_failed) return comm_failed_error(c);
    CHK(launch_partials(c, ptr<double>(c->partials)));
    const int rec = 6 + 4 * c->mpc.n_look_ahead;
    if (c->comm && (c->n_ranks > 1 || c->comm_always)) {
        CHK(ensure(c, c->gathered, (size_t)rec * sizeof(double) * c->n_ranks));
        RcclApi* R = rccl_api();
        ncclResult_t r = R->AllGather(c->partials.p, c->gathered.p, rec, ncclDouble, c->comm, c->stream);
        if (r != ncclSuccess) return fail(c, DRP_ECOMM, "ncclAllGather: %s", R->GetErrorString(r));
        return launch_update(c, ptr<double>(c->gathered), c->n_ranks);
    }
    return launch_update(c, ptr<double>(c->partials), 1);
}

// ---- elite (CEM-style) update: nominal = mean of the k best sequences over all ranks
static int elite_check(drp_ctx* c, int k) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    if (k < 1 || k > 1024) return fail(c, DRP_EINVAL, "elite size %d outside 1..1024", k);
    if ((size_t)c->mpc.n_sample * 16 + (size_t)k * 4 > 150 * 1024)
        return fail(c, DRP_EINVAL, "elite update supports up to 9 000 samples per rank");
    return DRP_OK;
}

static int pow2_at_least(int n) { int p = 1; while (p < n) p <<= 1; return p; }

static int launch_elite_local(drp_ctx* c, int k, double* out) {
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead;
    // sort path while keys + indices + positions of 2^m >= n_sample entries fit in LDS; k dependent rounds otherwise
    int n2 = pow2_at_least(p.n_sample);
    size_t lds = (size_t)n2 * 20;
    if (lds > 150 * 1024 || k > n2) { n2 = 0; lds = (size_t)p.n_sample * 16 + (size_t)k * 4; }
    ProbeScope ps(c, KC_MPPI);
    c->dv(n2 ? DV_ELITE_SORT : DV_ELITE_ROUNDS);
    hipLaunchKernelGGL(k_elite_local, dim3(1), dim3(256), lds, c->stream, ptr<float>(c->rewards) + (H - 1), H,
                       ptr<float>(c->actions), p.n_sample, p.n_batch, H, k, p.sample_offset, n2, out);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

static int launch_elite_update(drp_ctx* c, const double* dev_records, int n_ranks, int k, int rank_stride = 0) {
    const int HJ = 4 * c->mpc.n_look_ahead, total = n_ranks * k;
    if (rank_stride <= 0) rank_stride = k * (2 + HJ);
    int n2 = pow2_at_least(total);
    size_t lds = (size_t)n2 * 16 + (size_t)((n2 + 1) / 2) * 8 + (size_t)k * HJ * 8;    // keys, indices, positions, k sequences
    if (lds > 150 * 1024) {
        n2 = 0;
        size_t lds_d = (size_t)total * 2;
        if (lds_d < (size_t)k * HJ) lds_d = (size_t)k * HJ;
        lds = lds_d * 8 + (size_t)k * 4;
        if (lds > 150 * 1024) return fail(c, DRP_EINVAL, "too many elite records (%d ranks x %d, horizon %d)", n_ranks, k, c->mpc.n_look_ahead);
    }
    ProbeScope ps(c, KC_MPPI);
    hipLaunchKernelGGL(k_elite_update, dim3(1), dim3(256), lds, c->stream, dev_records, n_ranks, rank_stride, k, c->mpc.n_look_ahead, n2,
                       ptr<double>(c->nominal), ptr<double>(c->stats) + 6);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_elite(drp_ctx* c, int k, double* out) {
    CHK(elite_check(c, k));
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure(c, c->elite, (size_t)k * (2 + 4 * c->mpc.n_look_ahead) * sizeof(double)));
    CHK(launch_elite_local(c, k, ptr<double>(c->elite)));
    if (out) {
        CHK(d2h(c, out, c->elite.p, (size_t)k * (2 + 4 * c->mpc.n_look_ahead) * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

int drp_mpc_update_elite(drp_ctx* c, const double* records, int n_ranks, int k, double* nominal_out) {
    CHK(elite_check(c, k));
    HIPCHK(c, hipSetDevice(c->device));
    if (!records || n_ranks <= 0) return fail(c, DRP_EINVAL, "bad elite records");
    const size_t bytes = (size_t)n_ranks * k * (2 + 4 * c->mpc.n_look_ahead) * sizeof(double);
    CHK(ensure(c, c->elite_all, bytes));
    CHK(h2d(c, c->elite_all, records, bytes));
    CHK(launch_elite_update(c, ptr<double>(c->elite_all), n_ranks, k));
    if (nominal_out) {
        CHK(d2h(c, nominal_out, c->nominal.p, (size_t)c->mpc.n_look_ahead * 4 * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

int drp_mpc_update_elite_device(drp_ctx* c, int k) {
    CHK(elite_check(c, k));
    HIPCHK(c, hipSetDevice(c->device));
    // One message per rank and iteration: [statistics record (6 + 4H) | k elite records (2 + 4H each)],
    // written side by side by the two local kernels, all-gathered with ONE RCCL call, read in place by the
    // two combine kernels (the softmax combine supplies mean / std / max / argmax; its nominal is then replaced
    // by the elite mean).
    if (c->comm_failed) return comm_failed_error(c);
    const int H = c->mpc.n_look_ahead, rec_s = 6 + 4 * H, rec_e = k * (2 + 4 * H), msg = rec_s + rec_e;
    CHK(ensure(c, c->elite, (size_t)msg * sizeof(double)));
    double* mine = ptr<double>(c->elite);
    CHK(launch_partials(c, mine));
    CHK(launch_elite_local(c, k, mine + rec_s));
    const double* all = mine;
    int n_ranks = 1;
    if (c->comm && (c->n_ranks > 1 || c->comm_always)) {
        CHK(ensure(c, c->elite_all, (size_t)msg * sizeof(double) * c->n_ranks));
        RcclApi* R = rccl_api();
        ncclResult_t r = R->AllGather(c->elite.p, c->elite_all.p, msg, ncclDouble, c->comm, c->stream);
        if (r != ncclSuccess) return fail(c, DRP_ECOMM, "ncclAllGather: %s", R->GetErrorString(r));
        all = ptr<double>(c->elite_all);
        n_ranks = c->n_ranks;
    }
    CHK(launch_update(c, all, n_ranks, msg));
    return launch_elite_update(c, all + rec_s, n_ranks, k, msg);
}

int drp_mpc_get(drp_ctx* c, float* actions, float* rewards, float* rewards_all, float* states,
                double* nominal) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead, B = p.n_sample * p.n_batch, N = p.n_particles;
    if (actions) CHK(d2h(c, actions, c->actions.p, (size_t)B * H * 4 * sizeof(float)));
    if (rewards)
        HIPCHK(c, hipMemcpy2DAsync(rewards, sizeof(float), ptr<float>(c->rewards) + (H - 1),
                                   H * sizeof(float), sizeof(float), B, hipMemcpyDeviceToHost, c->stream));
    if (rewards_all) CHK(d2h(c, rewards_all, c->rewards.p, (size_t)B * H * sizeof(float)));
    if (states) CHK(d2h(c, states, c->states.p, (size_t)B * H * N * 3 * sizeof(float)));
    if (nominal) CHK(d2h(c, nominal, c->nominal.p, (size_t)H * 4 * sizeof(double)));
    return drp_sync(c);
}

// The planner's loop reads every iteration's pushes and final rewards (planners.py:721-738) but no iteration waits for
// the host: the copies go to pinned memory behind the iteration's kernels (and before the next sampling overwrites the
// pushes), the caller enqueues the next iteration and then waits for this slot's event.
int drp_mpc_fetch_async(drp_ctx* c, int slot) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    if (slot < 0 || slot > 1) return fail(c, DRP_EINVAL, "slot must be 0 or 1");
    if (c->mpc_pending[slot]) return fail(c, DRP_ESTATE, "slot %d holds an iteration nobody has waited for", slot);
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead, B = p.n_sample * p.n_batch;
    const size_t na = (size_t)B * H * 4, nr = (size_t)B;
    if (c->mpc_pin_floats < na + nr) {
        for (int q = 0; q < 2; ++q) {
            if (c->mpc_pending[q]) return fail(c, DRP_ESTATE, "the batch grew while an iteration was in flight");
            if (c->mpc_pin[q]) HIPCHK(c, hipHostFree(c->mpc_pin[q]));
            c->mpc_pin[q] = nullptr;
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->mpc_pin[q]), (na + nr) * sizeof(float), hipHostMallocDefault));
            if (!c->mpc_ev[q]) HIPCHK(c, hipEventCreateWithFlags(&c->mpc_ev[q], hipEventDisableTiming));
        }
        c->mpc_pin_floats = na + nr;
    }
    HIPCHK(c, hipMemcpyAsync(c->mpc_pin[slot], c->actions.p, na * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->mpc_pin[slot] + na, sizeof(float), ptr<float>(c->rewards) + (H - 1), H * sizeof(float),
                               sizeof(float), B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->mpc_ev[slot], c->stream));
    c->mpc_pending[slot] = true;
    return DRP_OK;
}

int drp_mpc_wait(drp_ctx* c, int slot, float* actions, float* rewards) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    if (slot < 0 || slot > 1 || !c->mpc_pending[slot]) return fail(c, DRP_ESTATE, "no iteration in flight in slot %d", slot);
    HIPCHK(c, hipSetDevice(c->device));
    c->mpc_pending[slot] = false;
    CHK(guarded_wait(c, c->mpc_ev[slot]));
    const drp_mpc_params& p = c->mpc;
    const size_t na = (size_t)p.n_sample * p.n_batch * p.n_look_ahead * 4, nr = (size_t)p.n_sample * p.n_batch;
    if (actions) memcpy(actions, c->mpc_pin[slot], na * sizeof(float));
    if (rewards) memcpy(rewards, c->mpc_pin[slot] + na, nr * sizeof(float));
    return DRP_OK;
}

int drp_fps(drp_ctx* c, const float* pts, int n, int dim, int k, int init_idx, int32_t* idx_out, float* max_dist_out) {
    if (!c || !pts || !idx_out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || k <= 0 || k > n || init_idx < 0 || init_idx >= n || (dim != 2 && dim != 3))
        return fail(c, DRP_EINVAL, "bad fps arguments n=%d dim=%d k=%d init=%d", n, dim, k, init_idx);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->scratch, pts, (size_t)n * dim * sizeof(float)));
    CHK(ensure(c, c->g_agg, (size_t)n * sizeof(float) + (size_t)(k + 1) * sizeof(int)));   // dist | chosen | max
    float* dist = ptr<float>(c->g_agg);
    int* chosen = reinterpret_cast<int*>(dist + n);
    CHK(ensure(c, c->stats, 8 * sizeof(double)));
    float* md = reinterpret_cast<float*>(ptr<double>(c->stats) + 7);
    const bool in_regs = n <= FPS_WIDE_THREADS * FPS_REG_PT(dim);
    if (dim == 2) {
        c->dv(in_regs ? DV_FPS_REG : DV_FPS_MEM);
        if (in_regs) hipLaunchKernelGGL(k_fps_reg<2>, dim3(1), dim3(FPS_WIDE_THREADS), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, chosen, md);
        else hipLaunchKernelGGL(k_fps<2>, dim3(1), dim3(1024), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, dist, chosen, md);
    } else {
        c->dv(in_regs ? DV_FPS_REG : DV_FPS_MEM);
        if (in_regs) hipLaunchKernelGGL(k_fps_reg<3>, dim3(1), dim3(FPS_WIDE_THREADS), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, chosen, md);
        else hipLaunchKernelGGL(k_fps<3>, dim3(1), dim3(1024), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, dist, chosen, md);
    }
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, idx_out, chosen, (size_t)k * sizeof(int)));
    if (max_dist_out) CHK(d2h(c, max_dist_out, md, sizeof(float)));
    return drp_sync(c);
}

// ---- particle extraction (row f2) ---------------------------------------------------------------
namespace {
const long long PX_MAX_CELLS = 1ll << 24;
const float PX_FG_DEPTH = (float)(0.599 / 0.8);    // env/flex_env.py:945, compared in float32

int px_nblk(size_t n) { return (int)((n + PX_TILE - 1) / PX_TILE); }

// depth image on the device -> c->px_pcd [n,3] float64 + per-block bounds; *n_out after a sync
int px_stage_pcd(drp_ctx* c, const float* d_depth, const uint8_t* d_mask, int h, int w, float gs, const double cam[4],
                 int* n_out) {
    const size_t npix = (size_t)h * w;
    const int nblk = px_nblk(npix);
    hipStream_t st = c->stream;
    CHK(ensure(c, c->px_blk, (size_t)(2 * nblk + 2) * sizeof(unsigned long long)));
    unsigned long long* cnt = ptr<unsigned long long>(c->px_blk);
    unsigned long long* off = cnt + nblk;
    hipLaunchKernelGGL(k_px_count, dim3(nblk), dim3(PX_BLOCK), 0, st, d_depth, d_mask, gs, PX_FG_DEPTH, npix, cnt);
    hipLaunchKernelGGL(k_px_scan_u64, dim3(1), dim3(1024), 0, st, cnt, nblk, off);
    HIPCHK(c, hipGetLastError());
    unsigned long long total = 0;
    CHK(d2h(c, &total, off + nblk, sizeof(total)));
    CHK(guarded_wait(c, nullptr));
    if (total > 0x7fffffffull) return fail(c, DRP_EINVAL, "too many foreground pixels");
    const int n = (int)total;
    *n_out = n;
    CHK(ensure(c, c->px_pcd, (size_t)(n > 0 ? n : 1) * 3 * sizeof(double)));
    CHK(ensure(c, c->px_bmin, (size_t)nblk * 3 * sizeof(double)));
    CHK(ensure(c, c->px_bmax, (size_t)nblk * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_px_compact, dim3(nblk), dim3(PX_BLOCK), 0, st, d_depth, d_mask, gs, PX_FG_DEPTH, w, npix,
                       cam[0], cam[1], cam[2], cam[3], off, ptr<double>(c->px_pcd), ptr<double>(c->px_bmin),
                       ptr<double>(c->px_bmax));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

// per-block bounds (c->px_bmin/bmax, nblk blocks) -> voxel grid; cloud d_pcd[n] -> c->px_down[m]
int px_stage_down(drp_ctx* c, const double* d_pcd, int n, int nblk_bounds, double voxel, int* m_out) {
    hipStream_t st = c->stream;
    if (n <= 0) { *m_out = 0; return DRP_OK; }
    CHK(ensure(c, c->px_grid, sizeof(PxGrid)));
    PxGrid* g = ptr<PxGrid>(c->px_grid);
    hipLaunchKernelGGL(k_px_bounds, dim3(1), dim3(64), 0, st, ptr<double>(c->px_bmin), ptr<double>(c->px_bmax),
                       nblk_bounds, n, voxel, g);
    HIPCHK(c, hipGetLastError());
    PxGrid hg;
    CHK(d2h(c, &hg, g, sizeof(hg)));
    CHK(guarded_wait(c, nullptr));
    if (hg.cells <= 0 || hg.cells > PX_MAX_CELLS)
        return fail(c, DRP_EINVAL, "voxel grid %d x %d x %d exceeds %lld cells", hg.dims[0], hg.dims[1], hg.dims[2],
                    PX_MAX_CELLS);
    const long long cells = hg.cells;
    const int cblk = px_nblk((size_t)cells);
    CHK(ensure(c, c->px_keys, (size_t)n * sizeof(int)));
    CHK(ensure(c, c->px_list, (size_t)n * sizeof(int)));
    CHK(ensure(c, c->px_cellcnt, (size_t)cells * sizeof(int)));
    CHK(ensure(c, c->px_cellfill, (size_t)cells * sizeof(int)));
    CHK(ensure(c, c->px_celloff, (size_t)cells * sizeof(unsigned long long)));
    CHK(ensure(c, c->px_blk, (size_t)(2 * cblk + 2) * sizeof(unsigned long long)));
    unsigned long long* bsum = ptr<unsigned long long>(c->px_blk);
    unsigned long long* boff = bsum + cblk;
    HIPCHK(c, hipMemsetAsync(c->px_cellcnt.p, 0, (size_t)cells * sizeof(int), st));
    HIPCHK(c, hipMemsetAsync(c->px_cellfill.p, 0, (size_t)cells * sizeof(int), st));
    const int pblk = (n + 255) / 256;
    hipLaunchKernelGGL(k_px_cell_count, dim3(pblk), dim3(256), 0, st, d_pcd, n, voxel, g, ptr<int>(c->px_keys),
                       ptr<int>(c->px_cellcnt));
    hipLaunchKernelGGL(k_px_cell_blocksum, dim3(cblk), dim3(PX_BLOCK), 0, st, ptr<int>(c->px_cellcnt), cells, bsum);
    hipLaunchKernelGGL(k_px_scan_u64, dim3(1), dim3(1024), 0, st, bsum, cblk, boff);
    hipLaunchKernelGGL(k_px_cell_offsets, dim3(cblk), dim3(PX_BLOCK), 0, st, ptr<int>(c->px_cellcnt), cells, boff,
                       ptr<unsigned long long>(c->px_celloff));
    HIPCHK(c, hipGetLastError());
    unsigned long long total = 0;
    CHK(d2h(c, &total, boff + cblk, sizeof(total)));
    CHK(guarded_wait(c, nullptr));
    const int m = (int)(total >> 32);
    if ((int)(total & 0xffffffffull) != n) return fail(c, DRP_ESTATE, "voxel scan lost points");
    *m_out = m;
    CHK(ensure(c, c->px_down, (size_t)m * 3 * sizeof(double)));
    CHK(ensure(c, c->px_down32, (size_t)m * 3 * sizeof(float)));
    hipLaunchKernelGGL(k_px_cell_fill, dim3(pblk), dim3(256), 0, st, ptr<int>(c->px_keys), n,
                       ptr<unsigned long long>(c->px_celloff), ptr<int>(c->px_cellfill), ptr<int>(c->px_list));
    hipLaunchKernelGGL(k_px_voxel_mean, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, ptr<int>(c->px_cellcnt),
                       ptr<unsigned long long>(c->px_celloff), ptr<int>(c->px_list), d_pcd, cells,
                       ptr<double>(c->px_down), ptr<float>(c->px_down32));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

// sampler + particle_r (+ recentering radius) on a device cloud (float64 + its float32 copy)
int px_stage_fps(drp_ctx* c, const double* d_pcd, const float* d_pcd32, int m, int npoints, int batch,
                 const int32_t* init_idx, uint64_t seed) {
    hipStream_t st = c->stream;
    const int* d_init = nullptr;
    if (init_idx) {
        for (int b = 0; b < batch; ++b)
            if (init_idx[b] < 0 || init_idx[b] >= m)
                return fail(c, DRP_EINVAL, "init_idx[%d]=%d outside the cloud of %d points", b, init_idx[b], m);
        CHK(h2d(c, c->px_init, init_idx, (size_t)batch * sizeof(int)));
        d_init = ptr<int>(c->px_init);
    }
    CHK(ensure(c, c->px_dist, (size_t)batch * m * sizeof(float)));
    CHK(ensure(c, c->px_chosen, (size_t)batch * npoints * sizeof(int)));
    CHK(ensure(c, c->px_pts, (size_t)batch * npoints * 3 * sizeof(float)));
    CHK(ensure(c, c->px_r, (size_t)batch * sizeof(double)));
    CHK(ensure(c, c->px_rr, (size_t)batch * sizeof(double)));
    hipLaunchKernelGGL(k_px_fps, dim3(batch), dim3(1024), 0, st, d_pcd32, m, npoints, d_init,
                       (unsigned long long)seed, ptr<float>(c->px_dist), ptr<int>(c->px_chosen), ptr<float>(c->px_pts));
    hipLaunchKernelGGL(k_px_radius, dim3(batch), dim3(1024), 0, st, d_pcd, m, ptr<float>(c->px_pts), npoints,
                       ptr<double>(c->px_r), ptr<double>(c->px_rr));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int px_check_cloud(drp_ctx* c, int n, int npoints, int batch) {
    if (npoints <= 0 || batch <= 0) return fail(c, DRP_EINVAL, "bad npoints=%d batch=%d", npoints, batch);
    if (n < npoints) return fail(c, DRP_EINVAL, "cloud of %d points, %d particles asked", n, npoints);
    return DRP_OK;
}
}  // namespace

int drp_depth2fgpcd(drp_ctx* c, const float* depth, const uint8_t* mask, int h, int w, const double cam[4],
                    double* pcd_out, int cap, int* n_out) {
    if (!c || !depth || !cam || !n_out) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0) return fail(c, DRP_EINVAL, "bad image size %d x %d", h, w);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)h * w;
    CHK(h2d(c, c->px_depth, depth, npix * sizeof(float)));
    if (mask) CHK(h2d(c, c->px_mask, mask, npix));
    int n = 0;
    CHK(px_stage_pcd(c, ptr<float>(c->px_depth), mask ? ptr<uint8_t>(c->px_mask) : nullptr, h, w, 1.0f, cam, &n));
    *n_out = n;
    if (pcd_out) {
        if (cap < n) return fail(c, DRP_EINVAL, "capacity %d < %d foreground points", cap, n);
        if (n > 0) CHK(d2h(c, pcd_out, c->px_pcd.p, (size_t)n * 3 * sizeof(double)));
    }
    return drp_sync(c);
}

int drp_downsample_pcd(drp_ctx* c, const double* pcd, int n, double voxel, double* out, int cap, int* m_out) {
    if (!c || !pcd || !m_out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || !(voxel > 0.0)) return fail(c, DRP_EINVAL, "bad downsample arguments n=%d voxel=%g", n, voxel);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_pcd, pcd, (size_t)n * 3 * sizeof(double)));
    const int nblk = px_nblk((size_t)n);
    CHK(ensure(c, c->px_bmin, (size_t)nblk * 3 * sizeof(double)));
    CHK(ensure(c, c->px_bmax, (size_t)nblk * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_px_point_bounds, dim3(nblk), dim3(PX_BLOCK), 0, c->stream, ptr<double>(c->px_pcd), n,
                       ptr<double>(c->px_bmin), ptr<double>(c->px_bmax));
    int m = 0;
    CHK(px_stage_down(c, ptr<double>(c->px_pcd), n, nblk, voxel, &m));
    *m_out = m;
    if (out) {
        if (cap < m) return fail(c, DRP_EINVAL, "capacity %d < %d voxels", cap, m);
        CHK(d2h(c, out, c->px_down.p, (size_t)m * 3 * sizeof(double)));
    }
    return drp_sync(c);
}

int drp_fps_pcd(drp_ctx* c, const double* pcd, int n, int npoints, int batch, const int32_t* init_idx,
                uint64_t seed, float* pts_out, double* r_out) {
    if (!c || !pcd || !pts_out) return fail(c, DRP_EINVAL, "null argument");
    CHK(px_check_cloud(c, n, npoints, batch));
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_down, pcd, (size_t)n * 3 * sizeof(double)));
    CHK(ensure(c, c->px_down32, (size_t)n * 3 * sizeof(float)));
    hipLaunchKernelGGL(k_px_to_f32, dim3((unsigned)(((size_t)n * 3 + 255) / 256)), dim3(256), 0, c->stream,
                       ptr<double>(c->px_down), (size_t)n * 3, ptr<float>(c->px_down32));
    CHK(px_stage_fps(c, ptr<double>(c->px_down), ptr<float>(c->px_down32), n, npoints, batch, init_idx, seed));
    CHK(d2h(c, pts_out, c->px_pts.p, (size_t)batch * npoints * 3 * sizeof(float)));
    if (r_out) CHK(d2h(c, r_out, c->px_r.p, (size_t)batch * sizeof(double)));
    return drp_sync(c);
}

int drp_fps_rad(drp_ctx* c, const double* pcd, int n, double radius, int init_idx, int cap, int32_t* idx_out,
                int* count_out) {
    if (!c || !pcd || !idx_out || !count_out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || cap <= 0 || init_idx < 0 || init_idx >= n || !(radius >= 0.0))
        return fail(c, DRP_EINVAL, "bad fps_rad arguments n=%d cap=%d init=%d radius=%g", n, cap, init_idx, radius);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_down, pcd, (size_t)n * 3 * sizeof(double)));
    CHK(ensure(c, c->px_dist, (size_t)n * sizeof(double)));
    CHK(ensure(c, c->px_chosen, (size_t)(cap + 1) * sizeof(int)));
    int* chosen = ptr<int>(c->px_chosen);
    hipLaunchKernelGGL(k_px_fps_rad, dim3(1), dim3(1024), 0, c->stream, ptr<double>(c->px_down), n, radius, init_idx, cap,
                       ptr<double>(c->px_dist), chosen, chosen + cap);
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, count_out, chosen + cap, sizeof(int)));
    CHK(guarded_wait(c, nullptr));
    CHK(d2h(c, idx_out, chosen, (size_t)*count_out * sizeof(int)));
    return drp_sync(c);
}

int drp_recenter(drp_ctx* c, const double* pcd, int n, const float* sampled, int npoints, int batch, const double* r,
                 float* out) {
    if (!c || !pcd || !sampled || !r || !out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || npoints <= 0 || batch <= 0) return fail(c, DRP_EINVAL, "bad recenter arguments");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_down, pcd, (size_t)n * 3 * sizeof(double)));
    CHK(h2d(c, c->px_pts, sampled, (size_t)batch * npoints * 3 * sizeof(float)));
    CHK(h2d(c, c->px_rr, r, (size_t)batch * sizeof(double)));
    CHK(ensure(c, c->px_out, (size_t)batch * npoints * 3 * sizeof(double)));
    float* o32 = ptr<float>(c->px_out);
    hipLaunchKernelGGL(k_px_recenter, dim3((batch * npoints + 3) / 4), dim3(256), 0, c->stream, ptr<double>(c->px_down), n,
                       ptr<float>(c->px_pts), npoints, batch, ptr<double>(c->px_rr), o32, (double*)nullptr);
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, out, o32, (size_t)batch * npoints * 3 * sizeof(float)));
    return drp_sync(c);
}

int drp_obs2ptcl(drp_ctx* c, const float* depth_raw, int h, int w, float global_scale, const double cam[4],
                 int npoints, int batch, const int32_t* init_idx, uint64_t seed, double* ptcl_out, double* r_out,
                 int* n_fg, int* n_down) {
    if (!c || !depth_raw || !cam || !ptcl_out || !r_out) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0 || !(global_scale > 0.0f)) return fail(c, DRP_EINVAL, "bad image %d x %d scale %g", h, w, global_scale);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)h * w;
    CHK(h2d(c, c->px_depth, depth_raw, npix * sizeof(float)));
    int n = 0, m = 0;
    CHK(px_stage_pcd(c, ptr<float>(c->px_depth), nullptr, h, w, global_scale, cam, &n));
    if (n_fg) *n_fg = n;
    if (n <= 0) return fail(c, DRP_EINVAL, "no foreground pixel (depth < 0.599/0.8 of the scaled image)");
    CHK(px_stage_down(c, ptr<double>(c->px_pcd), n, px_nblk(npix), 0.01, &m));   // env/flex_env.py:947
    if (n_down) *n_down = m;
    CHK(px_check_cloud(c, m, npoints, batch));
    CHK(px_stage_fps(c, ptr<double>(c->px_down), ptr<float>(c->px_down32), m, npoints, batch, init_idx, seed));
    CHK(ensure(c, c->px_out, (size_t)batch * npoints * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_px_recenter, dim3((batch * npoints + 3) / 4), dim3(256), 0, c->stream, ptr<double>(c->px_down), m,
                       ptr<float>(c->px_pts), npoints, batch, ptr<double>(c->px_rr), (float*)nullptr,
                       ptr<double>(c->px_out));
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, ptcl_out, c->px_out.p, (size_t)batch * npoints * 3 * sizeof(double)));
    CHK(d2h(c, r_out, c->px_r.p, (size_t)batch * sizeof(double)));
    return drp_sync(c);
}

// ---- goal pre-processing (row f3) ---------------------------------------------------------------
namespace {
// seg (device, [h,w] u8) -> c->gl_dist [h,w] float32
int goal_stage_dt(drp_ctx* c, const uint8_t* d_seg, int h, int w, int mode) {
    const size_t npix = (size_t)h * w;
    hipStream_t st = c->stream;
    CHK(ensure(c, c->gl_tmp, npix * sizeof(int)));
    CHK(ensure(c, c->gl_dist, npix * sizeof(float)));
    if (mode == DRP_DT_CV5) {
        const size_t lds = (size_t)3 * (w + 4) * sizeof(int);
        if (lds > 60000) return fail(c, DRP_EINVAL, "image width %d too large for the chamfer kernel", w);
        c->dv(DV_DT_CV5);
        hipLaunchKernelGGL(k_dt_cv5, dim3(1), dim3(DT_THREADS), lds, st, d_seg, h, w, ptr<int>(c->gl_tmp),
                           ptr<float>(c->gl_dist));
    } else if (mode == DRP_DT_EXACT) {
        if ((size_t)w * sizeof(int) > 60000) return fail(c, DRP_EINVAL, "image width %d too large", w);
        c->dv(DV_DT_EXACT);
        hipLaunchKernelGGL(k_edt_cols, dim3((w + 255) / 256), dim3(256), 0, st, d_seg, h, w, ptr<int>(c->gl_tmp));
        hipLaunchKernelGGL(k_edt_rows, dim3(h), dim3(256), (size_t)w * sizeof(int), st, ptr<int>(c->gl_tmp), h, w,
                           ptr<float>(c->gl_dist));
    } else {
        return fail(c, DRP_EINVAL, "unknown distance transform mode %d", mode);
    }
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}
}  // namespace

int drp_distance_transform(drp_ctx* c, const uint8_t* src, int h, int w, int mode, float* dist_out) {
    if (!c || !src || !dist_out) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0) return fail(c, DRP_EINVAL, "bad image size %d x %d", h, w);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)h * w;
    CHK(h2d(c, c->gl_seg, src, npix));
    CHK(goal_stage_dt(c, ptr<uint8_t>(c->gl_seg), h, w, mode));
    CHK(d2h(c, dist_out, c->gl_dist.p, npix * sizeof(float)));
    return drp_sync(c);
}

int drp_set_goal_image(drp_ctx* c, const float* obs_goal, int h, int w, int mode, int max_goal_pts, int fps_init,
                       float* field_out, float* goal_coor_out, int* m_out) {
    if (!c || !obs_goal) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0 || max_goal_pts <= 0) return fail(c, DRP_EINVAL, "bad goal image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t npix = (size_t)h * w;
    const unsigned eb = (unsigned)((npix + 255) / 256);
    CHK(h2d(c, c->gl_goal, obs_goal, npix * sizeof(float)));
    CHK(ensure(c, c->gl_seg, npix));
    hipLaunchKernelGGL(k_goal_seg, dim3(eb), dim3(256), 0, st, ptr<float>(c->gl_goal), npix, ptr<uint8_t>(c->gl_seg));
    // goal pixels first: an image without any is an error before anything is installed
    const int nblk = px_nblk(npix);
    CHK(ensure(c, c->gl_blk, (size_t)(2 * nblk + 2) * sizeof(unsigned long long) + (size_t)(eb + 1) * sizeof(float)));
    unsigned long long* cnt = ptr<unsigned long long>(c->gl_blk);
    unsigned long long* off = cnt + nblk;
    float* bmin = reinterpret_cast<float*>(off + nblk + 2);
    hipLaunchKernelGGL(k_goal_count, dim3(nblk), dim3(PX_BLOCK), 0, st, ptr<uint8_t>(c->gl_seg), npix, cnt);
    hipLaunchKernelGGL(k_px_scan_u64, dim3(1), dim3(1024), 0, st, cnt, nblk, off);
    HIPCHK(c, hipGetLastError());
    unsigned long long total = 0;
    CHK(d2h(c, &total, off + nblk, sizeof(total)));
    CHK(guarded_wait(c, nullptr));
    const int count = (int)total;
    if (count <= 0) return fail(c, DRP_EINVAL, "the goal image has no pixel below 0.5");
    if (count == (int)npix) return fail(c, DRP_EINVAL, "the goal image has no pixel at or above 0.5");
    if (fps_init < 0 || fps_init >= count) return fail(c, DRP_EINVAL, "fps_init=%d outside the %d goal pixels", fps_init, count);
    const int m = max_goal_pts < count ? max_goal_pts : count;
    CHK(ensure(c, c->gl_pix, (size_t)count * 2 * sizeof(float)));
    hipLaunchKernelGGL(k_goal_compact, dim3(nblk), dim3(PX_BLOCK), 0, st, ptr<uint8_t>(c->gl_seg), w, npix, off,
                       ptr<float>(c->gl_pix));
    CHK(ensure(c, c->gl_fps, (size_t)count * sizeof(float) + (size_t)(m + 2) * sizeof(int)));
    float* fdist = ptr<float>(c->gl_fps);
    int* chosen = reinterpret_cast<int*>(fdist + count);
    float* md = reinterpret_cast<float*>(chosen + m);
    c->dv(count <= FPS_WIDE_THREADS * FPS_REG_PT(2) ? DV_FPS_REG : DV_FPS_MEM);
    if (count <= FPS_WIDE_THREADS * FPS_REG_PT(2))
        hipLaunchKernelGGL(k_fps_reg<2>, dim3(1), dim3(FPS_WIDE_THREADS), 0, st, ptr<float>(c->gl_pix), count, m, fps_init, chosen, md);
    else
        hipLaunchKernelGGL(k_fps<2>, dim3(1), dim3(1024), 0, st, ptr<float>(c->gl_pix), count, m, fps_init, fdist, chosen, md);
    CHK(ensure(c, c->goal_coor, (size_t)m * 2 * sizeof(float)));
    hipLaunchKernelGGL(k_goal_gather, dim3((m + 255) / 256), dim3(256), 0, st, ptr<float>(c->gl_pix), chosen, m,
                       ptr<float>(c->goal_coor));
    // the field
    CHK(goal_stage_dt(c, ptr<uint8_t>(c->gl_seg), h, w, mode));
    CHK(ensure(c, c->goal_field, npix * sizeof(float)));
    hipLaunchKernelGGL(k_goal_sub, dim3(eb), dim3(256), 0, st, ptr<float>(c->gl_goal), ptr<float>(c->gl_dist), npix,
                       ptr<float>(c->goal_field), bmin);
    hipLaunchKernelGGL(k_goal_min, dim3(1), dim3(1024), 0, st, bmin, (int)eb, bmin + eb);
    hipLaunchKernelGGL(k_goal_shift, dim3(eb), dim3(256), 0, st, ptr<float>(c->goal_field), npix, bmin + eb);
    HIPCHK(c, hipGetLastError());
    if (field_out) CHK(d2h(c, field_out, c->goal_field.p, npix * sizeof(float)));
    if (goal_coor_out) CHK(d2h(c, goal_coor_out, c->goal_coor.p, (size_t)m * 2 * sizeof(float)));
    CHK(guarded_wait(c, nullptr));
    if (m_out) *m_out = m;
    c->goal_h = h; c->goal_w = w; c->goal_m = m;
    c->have_goal = true;
    return DRP_OK;
}

// ---- gradient-descent planner (row f1) ----------------------------------------------------------
namespace {
// relation encoder backward on the matrix cores (kmb_edge_encode): one tile of 32 edge slots per wave, the tiles of a
// small batch spread one per CU
void launch_edge_encode_mfma(drp_ctx* c, const float* s_prev, int prev_mod, size_t prev_stride, int nb, const int16_t* idx,
                             const uint8_t* cnt, const float* gah, const unsigned* mht, size_t bn, int N, int B, float* gpos_edge,
                             const KbEdgeDump& dump) {
    const long ntiles = (long)B * (((long)N * DRP_K + 31) / 32);
    const unsigned grid = (unsigned)(ntiles < (long)c->n_cu ? ntiles : (long)c->n_cu);
    hipLaunchKernelGGL(kmb_edge_encode, dim3(grid), dim3(64 * MFMA_WAVES), KMB_EDGE_ENCODE_LDS, c->stream, ptr<float>(c->w_mfma),
                       ptr<float>(c->w_mfma_bwd), s_prev, prev_mod, prev_stride, ptr<float>(c->attr), nb, ptr<float>(c->dens), nb, idx,
                       cnt, gah, mht, bn, N, B, gpos_edge, dump);
}
int gd_forward_backward(drp_ctx* c) {
    const int nb = c->gd_nb, N = c->gd_N, B = c->gd_B, H = c->gd_H;
    const size_t bn = (size_t)B * N;
    const size_t hstride = (size_t)H * N * 3;
    hipStream_t st = c->stream;
    const bool rev_lds = N <= KB_REV_LDS_MAX_N && !c->rev_global_only;
    float* states = ptr<float>(c->states);
    float* eh = ptr<float>(c->eff_hist);
    unsigned* mh = ptr<unsigned>(c->tape_mask);
    auto d2d = [&](void* dst, const void* src, size_t bytes) -> int {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return DRP_OK;
    };
    // ---- forward on the fused engine; km_prop<., TAPE> leaves what the backward pass needs: the
    //      effect after the encoder and after every propagation step, and the ReLU masks of the edges
    const int saved_engine = c->engine;
    c->engine = c->gd_engine;
    // the self-edge constants depend on attributes and densities only: computed once per GD problem,
    // again only if a rollout in between has reused the buffer
    int rc = DRP_OK;
    if (c->gd_cself_tag != c->cself_tag || c->gd_cself_tag == 0) {
        rc = prepare_cself(c, nb, N, B, &c->gd_cself, &c->gd_cself_ok);
        c->gd_cself_tag = c->cself_tag;
    }
    const float* cself = c->gd_cself;
    const uint8_t* cself_ok = c->gd_cself_ok;
    bool rev_built = false;
    for (int t = 0; t < H && rc == DRP_OK; ++t) {
        StepArgs a{};
        if (t == 0) { a.s_prev = ptr<float>(c->s_in); a.prev_mod = nb; a.prev_stride = (size_t)N * 3; }
        else { a.s_prev = states + (size_t)(t - 1) * N * 3; a.prev_mod = B; a.prev_stride = hstride; }
        a.attr = ptr<float>(c->attr); a.attr_mod = nb;
        a.dens = ptr<float>(c->dens); a.dens_mod = nb;
        a.actions = ptr<float>(c->actions) + (size_t)t * 4; a.act_stride = (size_t)H * 4;
        a.build_graph = true;
        a.s_out = states + (size_t)t * N * 3; a.out_stride = hstride;
        a.B = B; a.N = N;
        a.eff_hist = eh + (size_t)t * 4 * bn * 64;
        a.mask_hist = mh + (size_t)t * DRP_PSTEP * bn * DRP_K * 2;
        a.cself = cself; a.cself_ok = cself_ok;
        if (H == 1) { a.rev_off = ptr<int>(c->rev_off); a.rev = ptr<int>(c->rev); a.rev_built = &rev_built; }   // one set of reversed lists: the only step's
        // the step's impulses and neighbour lists are part of the tape: the step writes them there (its
        // workspace pointers are lent the tape's slices for the call) instead of being copied afterwards
        void* const save_sd = c->s_delta.p; void* const save_idx = c->nbr_idx.p; void* const save_cnt = c->nbr_cnt.p;
        c->s_delta.p = ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3;
        c->nbr_idx.p = ptr<int16_t>(c->tape_idx) + (size_t)t * bn * DRP_K;
        c->nbr_cnt.p = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        rc = run_step(c, a);
        c->s_delta.p = save_sd; c->nbr_idx.p = save_idx; c->nbr_cnt.p = save_cnt;
        if (rc != DRP_OK) break;
    }
    c->engine = saved_engine;
    CHK(rc);
    // reward of the final step only (planners.py:436-438) and its gradient, in one launch
    const float* vw = ptr<float>(c->w_valu);
    const float* wraw = ptr<float>(c->w_raw);
    float* g_state = ptr<float>(c->g_state);                 // [H][B,N,3]
    {
        ProbeScope ps(c, KC_BWD_REWARD);
        c->dv(DV_BWD_REWARD);
        hipLaunchKernelGGL(kb_reward, dim3(B), dim3(256), KB_REWARD_LDS(N), st, states + (size_t)(H - 1) * N * 3, hstride,
                           N, ptr<float>(c->goal_field), c->goal_h, c->goal_w, ptr<float>(c->goal_coor), c->goal_m, c->cam,
                           1, g_state + (size_t)(H - 1) * bn * 3, (size_t)N * 3, ptr<float>(c->rewards), c->gd_host_rewards);
    }
    for (int t = H - 1; t >= 0; --t) {
        const float* s_prev = (t == 0) ? ptr<float>(c->s_in) : states + (size_t)(t - 1) * N * 3;
        const int prev_mod = (t == 0) ? nb : B;
        const size_t prev_stride = (t == 0) ? (size_t)N * 3 : hstride;
        float* eht = eh + (size_t)t * 4 * bn * 64;
        const unsigned* mht = mh + (size_t)t * DRP_PSTEP * bn * DRP_K * 2;
        const int16_t* idx = ptr<int16_t>(c->tape_idx) + (size_t)t * bn * DRP_K;
        const uint8_t* cnt = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        float* g_out = g_state + (size_t)t * bn * 3;
        float* gah = ptr<float>(c->g_agg_hist);
        if (!rev_built) {
            ProbeScope ps(c, KC_BWD_LISTS);
            c->dv(N <= 512 ? DV_REV_256 : DV_REV_1024);
            if (N <= 512)
                hipLaunchKernelGGL(kb_reverse_lists<256>, dim3(B), dim3(256), KB_REV_LDS(N, rev_lds), st, idx,
                                   cnt, N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, (const int*)nullptr);
            else
                hipLaunchKernelGGL(kb_reverse_lists<1024>, dim3(B), dim3(1024), KB_REV_LDS(N, rev_lds), st, idx,
                                   cnt, N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, (const int*)nullptr);
        }
        const int spw_b = (B + c->n_cu - 1) / c->n_cu;
        if (c->bwd_fused && c->bwd_rows && N <= KMB_ROWS_MAX) {
            // piles of up to 256 particles: a workgroup takes groups of whole samples with at most 256 rows, a wave keeps
            // its tile's rows in registers through all phases (kmb_rows_bwd).  Samples per group: the fewest that do not
            // add a round of groups over the CUs (fewer waves at work per CU, more CUs at work)
            const int g_max = KMB_ROWS_MAX / N;
            auto rounds = [&](int g) { return (((long)B + g - 1) / g + c->n_cu - 1) / c->n_cu; };
            int gps = g_max;
            while (gps > 1 && rounds(gps - 1) == rounds(g_max)) --gps;
            const long n_groups = ((long)B + gps - 1) / gps;
            ProbeScope ps(c, KC_BWD_NODE);
            c->dv(DV_BWD_ROWS);
            hipLaunchKernelGGL(kmb_rows_bwd, dim3((unsigned)(n_groups < (long)c->n_cu ? n_groups : (long)c->n_cu)), dim3(64 * KMB_FUSED_WAVES),
                               KMB_ROWS_LDS, st, ptr<float>(c->w_mfma), ptr<float>(c->w_mfma_bwd), ptr<uint16_t>(c->w_split6),
                               ptr<uint16_t>(c->w_split6_bwd), eht, mht, cnt, ptr<int>(c->rev_off),
                               ptr<int>(c->rev), g_out, (size_t)N * 3, ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr),
                               nb, ptr<float>(c->dens), nb, N, B, gps, t > 0 ? gah : (float*)nullptr, ptr<float>(c->g_sdelta));
        } else if (c->bwd_fused && whole_samples(c, B, N) && ((long)spw_b * N + 31) / 32 >= c->bwd_fused_min_tiles) {
            // chip-filling batches: everything between the reward's gradient and the impulses' in one launch,
            // a workgroup owning whole samples (kmb_step_bwd)
            ProbeScope ps(c, KC_BWD_NODE);
            c->dv(DV_BWD_STEP);
            hipLaunchKernelGGL(kmb_step_bwd, dim3((unsigned)((B + spw_b - 1) / spw_b)), dim3(64 * KMB_FUSED_WAVES), KMB_FUSED_LDS, st,
                               ptr<float>(c->w_mfma), ptr<float>(c->w_mfma_bwd), eht, mht, cnt, ptr<int>(c->rev_off), ptr<int>(c->rev),
                               g_out, (size_t)N * 3, ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), nb,
                               ptr<float>(c->dens), nb, N, B, spw_b, ptr<float>(c->g_eff), ptr<float>(c->g_cnode), gah,
                               ptr<float>(c->g_sdelta));
        } else if ((long)B * ((N + 31) / 32) >= KMB_MIN_TILES && !c->bwd_valu_stages) {      // node stages on the matrix cores
            const float* mw = ptr<float>(c->w_mfma);
            const float* mb = ptr<float>(c->w_mfma_bwd);
            const long node_tiles = (long)B * ((N + 31) / 32);
            const dim3 ngrid(mfma_grid_spread(c, node_tiles)), nblk(64 * MFMA_WAVES);
            c->dv(DV_BWD_STAGES_MFMA);
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kmb_predict, ngrid, nblk, KMB_PREDICT_LDS, st, mw, mb, eht + 3 * bn * 64, g_out, (size_t)N * 3, N, B,
                               ptr<float>(c->g_eff), (float*)nullptr, (float*)nullptr);
            }
            // update of the last propagation step, then per step: edge terms, and in one launch the
            // projection of this step with the update of the one before
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL((kmb_node_step<false, true>), ngrid, nblk, KMB_STEP_LDS(false, true), st, mb, ptr<float>(c->g_eff),
                               ptr<float>(c->g_eff), (const float*)nullptr, eht + (size_t)DRP_PSTEP * bn * 64, ptr<float>(c->g_cnode), 1,
                               gah + (size_t)(DRP_PSTEP - 1) * bn * 64, N, B);
            }
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn * 64;
                const unsigned* mask_p = mht + (size_t)p * bn * DRP_K * 2;
                { ProbeScope ps(c, KC_BWD_EDGE);
                hipLaunchKernelGGL(kb_edge_terms, dim3(B), dim3(256), 0, st, g_agg_p, mask_p, cnt, ptr<int>(c->rev_off),
                                   ptr<int>(c->rev), N, ptr<float>(c->g_proj), 1);
                }
                if (p > 0)
                    { ProbeScope ps(c, KC_BWD_NODE);
                    hipLaunchKernelGGL((kmb_node_step<true, true>), ngrid, nblk, KMB_STEP_LDS(true, true), st, mb,
                                       ptr<float>(c->g_eff), ptr<float>(c->g_eff), ptr<float>(c->g_proj), eht + (size_t)p * bn * 64,
                                       ptr<float>(c->g_cnode), 0, gah + (size_t)(p - 1) * bn * 64, N, B);
                    }
                else
                    { ProbeScope ps(c, KC_BWD_NODE);
                    hipLaunchKernelGGL((kmb_node_step<true, false>), ngrid, nblk, KMB_STEP_LDS(true, false), st, mb,
                                       ptr<float>(c->g_eff), ptr<float>(c->g_eff), ptr<float>(c->g_proj), (const float*)nullptr, (float*)nullptr, 0,
                                       (float*)nullptr, N, B);
                    }
            }
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kmb_node_encode, ngrid, nblk, KMB_NODE_ENCODE_LDS, st, mw, mb,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), nb, ptr<float>(c->dens), nb,
                               eht, ptr<float>(c->g_eff), ptr<float>(c->g_cnode), N, B, ptr<float>(c->g_sdelta), (float*)nullptr,
                               (float*)nullptr, (float*)nullptr, (float*)nullptr);
            }
        } else {
            c->dv(DV_BWD_STAGES_VALU);
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kb_predict, dim3(B), dim3(256), 0, st, vw, wraw, eht + 3 * bn * 64, g_out, (size_t)N * 3, N,
                               ptr<float>(c->g_eff), (float*)nullptr, (float*)nullptr, 1);
            }
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn * 64;
                const unsigned* mask_p = mht + (size_t)p * bn * DRP_K * 2;
                { ProbeScope ps(c, KC_BWD_NODE);
                hipLaunchKernelGGL(kb_update, dim3(B), dim3(256), 0, st, wraw, eht + (size_t)(p + 1) * bn * 64,
                                   ptr<float>(c->g_eff), ptr<float>(c->g_cnode), p == DRP_PSTEP - 1 ? 1 : 0, N, g_agg_p, 1);
                }
                { ProbeScope ps(c, KC_BWD_EDGE);
                hipLaunchKernelGGL(kb_edge_terms, dim3(B), dim3(256), 0, st, g_agg_p, mask_p, cnt, ptr<int>(c->rev_off),
                                   ptr<int>(c->rev), N, ptr<float>(c->g_proj), 1);
                }
                { ProbeScope ps(c, KC_BWD_NODE);
                hipLaunchKernelGGL(kb_project, dim3(B), dim3(256), 0, st, wraw, ptr<float>(c->g_proj), N, ptr<float>(c->g_eff), 1);
                }
            }
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kb_node_encode, dim3(B), dim3(256), 0, st, vw, wraw,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), nb, ptr<float>(c->dens),
                               nb, eht, ptr<float>(c->g_eff), ptr<float>(c->g_cnode), N, ptr<float>(c->g_sdelta),
                               (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 1);
            }
        }
        float* g_prev = nullptr;
        if (t > 0) {
            // d loss / d state[t-1] = residual share + relation encoder + gen_s_delta's position dependence
            g_prev = g_state + (size_t)(t - 1) * bn * 3;
            CHK(d2d(g_prev, g_out, bn * 3 * sizeof(float)));
            { ProbeScope ps(c, KC_BWD_EDGE);
            c->dv(c->bwd_edge_mfma ? DV_BWD_EDGE_MFMA : DV_BWD_EDGE_VALU);
            if (c->bwd_edge_mfma)
                launch_edge_encode_mfma(c, s_prev, prev_mod, prev_stride, nb, idx, cnt, gah, mht, bn, N, B, ptr<float>(c->gpos_edge), KbEdgeDump{});
            else
                hipLaunchKernelGGL(kb_edge_encode, dim3(B), dim3(256), KB_EDGE_ENCODE_LDS, st, vw, wraw,
                                   s_prev, prev_mod, prev_stride, ptr<float>(c->attr), nb, ptr<float>(c->dens), nb, idx, cnt,
                                   gah, mht, bn, N, g_prev, (size_t)N * 3, ptr<float>(c->gpos_edge), KbEdgeDump{}, 1);
            }
            { ProbeScope ps(c, KC_BWD_EDGE);
            hipLaunchKernelGGL(kb_gather_pos, dim3((N + 255) / 256, B), dim3(256), 0, st, ptr<float>(c->gpos_edge),
                               ptr<int>(c->rev_off), ptr<int>(c->rev), N, g_prev, (size_t)N * 3, c->bwd_edge_mfma ? 1 : 0, cnt);
            }
        }
        { ProbeScope ps(c, KC_BWD_PUSH);
        hipLaunchKernelGGL(kb_sdelta, dim3(B), dim3(256), 0, st, s_prev, prev_mod, prev_stride,
                           ptr<float>(c->actions) + (size_t)t * 4, (size_t)H * 4, ptr<float>(c->g_sdelta), N, c->cam,
                           ptr<float>(c->g_act) + (size_t)t * 4, (size_t)H * 4, g_prev, (size_t)N * 3, t == 0 ? c->gd_adam : KbAdam{});
        }
    }
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}
}  // namespace

int drp_gd_begin(drp_ctx* c, const float* s0, const float* attr, const float* dens, int nb, int N,
                 const float* actions, int B, int H, double lr, const float act_lo[4], const float act_hi[4]) {
    CHK(need(c, true, true, true));
    CHK(check_bn(c, B, N));
    if (!s0 || !attr || !dens || !actions || !act_lo || !act_hi) return fail(c, DRP_EINVAL, "null argument");
    if (H < 1 || H > 64) return fail(c, DRP_EINVAL, "bad horizon H=%d", H);
    if (nb <= 0 || B % nb != 0) return fail(c, DRP_EINVAL, "B must be a multiple of n_batch");
    HIPCHK(c, hipSetDevice(c->device));
    {
        // Adam moves the pushes, the clip keeps them in the box: bound by the box's diagonals and by the initial pushes
        const float box[8] = {act_lo[0], act_lo[1], act_hi[2], act_hi[3], act_hi[0], act_hi[1], act_lo[2], act_lo[3]};
        CHK(pick_tape_engine(c, max_abs(attr, (size_t)nb * N), max_abs(dens, (size_t)nb),
                             fmaxf(push_len_bound(c, box, 2), push_len_bound(c, actions, (size_t)B * H)), &c->gd_engine));
    }
    const size_t bn = (size_t)B * N;
    CHK(h2d(c, c->s_in, s0, (size_t)nb * N * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, attr, (size_t)nb * N * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)nb * sizeof(float)));
    CHK(h2d(c, c->actions, actions, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure_step_ws(c, B, N, c->gd_engine));
    CHK(ensure(c, c->states, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->rewards, (size_t)B * sizeof(float)));
    CHK(ensure(c, c->eff_hist, (size_t)H * 4 * bn * 64 * sizeof(float)));
    CHK(ensure(c, c->tape_sdelta, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->tape_idx, (size_t)H * bn * DRP_K * sizeof(int16_t)));
    CHK(ensure(c, c->tape_cnt, (size_t)H * bn));
    CHK(ensure(c, c->tape_mask, (size_t)H * DRP_PSTEP * bn * DRP_K * 2 * sizeof(unsigned)));
    CHK(ensure(c, c->g_agg_hist, (size_t)DRP_PSTEP * bn * 64 * sizeof(float)));
    CHK(ensure(c, c->rev_off, (size_t)B * (N + 1) * sizeof(int)));
    CHK(ensure(c, c->rev, bn * DRP_K * sizeof(int)));
    CHK(ensure(c, c->gpos_edge, bn * DRP_K * 4 * sizeof(float)));
    CHK(ensure(c, c->g_eff, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->g_cnode, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->g_agg, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->g_proj, bn * 128 * sizeof(float)));
    CHK(ensure(c, c->g_state, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->g_sdelta, bn * 3 * sizeof(float)));
    CHK(ensure(c, c->g_act, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure(c, c->adam_m, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure(c, c->adam_v, (size_t)B * H * 4 * sizeof(float)));
    HIPCHK(c, hipMemsetAsync(c->adam_m.p, 0, (size_t)B * H * 4 * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->adam_v.p, 0, (size_t)B * H * 4 * sizeof(float), c->stream));
    CHK(guarded_wait(c, nullptr));
    c->gd_nb = nb; c->gd_N = N; c->gd_B = B; c->gd_H = H; c->gd_iter = 0; c->gd_lr = lr;
    for (int q = 0; q < DRP_GD_SLOTS; ++q) c->gd_pending[q] = false;           // a new problem drops what the last one left in flight
    c->gd_cself_tag = 0;
    memcpy(c->gd_lo, act_lo, 4 * sizeof(float));
    memcpy(c->gd_hi, act_hi, 4 * sizeof(float));
    c->lastH = H;
    c->gd_on = true;
    c->mpc_on = false;
    return DRP_OK;
}

int drp_gd_grad(drp_ctx* c, float* rewards_out, float* grad_act_out, float* grad_state_out) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(gd_forward_backward(c));
    const size_t bn = (size_t)c->gd_B * c->gd_N;
    if (rewards_out) CHK(d2h(c, rewards_out, c->rewards.p, (size_t)c->gd_B * sizeof(float)));
    if (grad_act_out) CHK(d2h(c, grad_act_out, c->g_act.p, (size_t)c->gd_B * c->gd_H * 4 * sizeof(float)));
    if (grad_state_out) {
        // device layout [H][B,N,3] -> caller layout [B,H,N,3]
        const size_t row = (size_t)c->gd_N * 3 * sizeof(float);
        for (int t = 0; t < c->gd_H; ++t)
            HIPCHK(c, hipMemcpy2DAsync(grad_state_out + (size_t)t * c->gd_N * 3, (size_t)c->gd_H * row,
                                       ptr<float>(c->g_state) + (size_t)t * bn * 3, row, row, c->gd_B,
                                       hipMemcpyDeviceToHost, c->stream));
    }
    return drp_sync(c);
}

namespace {
// one iteration on the stream: forward, backward, Adam, clip -- the optimiser step of a row in the kb_sdelta launch that
// completes the row's gradient (rollout step 0's, the last of the backward pass): one launch fewer per iteration
int gd_iteration(drp_ctx* c) {
    // torch.optim.Adam: step_size = lr / (1 - beta1^t), denom = sqrt(v) / sqrt(1 - beta2^t) + eps
    const double it = (double)(c->gd_iter + 1);
    const double bc1 = 1.0 - pow(0.9, it), bc2 = 1.0 - pow(0.999, it);
    KbAdam a{};
    a.act = ptr<float>(c->actions); a.m = ptr<float>(c->adam_m); a.v = ptr<float>(c->adam_v); a.act_copy = c->gd_host_actions;
    a.n_row = c->gd_H * 4;
    a.step_size = (float)(c->gd_lr / bc1); a.bc2_sqrt = (float)sqrt(bc2); a.b1 = 0.9f;
    a.lo = make_float4(c->gd_lo[0], c->gd_lo[1], c->gd_lo[2], c->gd_lo[3]);
    a.hi = make_float4(c->gd_hi[0], c->gd_hi[1], c->gd_hi[2], c->gd_hi[3]);
    c->gd_adam = a;
    const int rc = gd_forward_backward(c);
    c->gd_adam = KbAdam{};
    CHK(rc);
    c->gd_iter += 1;
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}
}  // namespace

int drp_gd_step(drp_ctx* c, float* rewards_out) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(gd_iteration(c));
    if (rewards_out) {
        CHK(d2h(c, rewards_out, c->rewards.p, (size_t)c->gd_B * sizeof(float)));
        return drp_sync(c);
    }
    return DRP_OK;
}

// The planner's loop needs every iteration's rewards and updated pushes on the host (per-column bookkeeping,
// planners.py:721-727), but no iteration waits for the host: slot s of two takes the iteration's results into pinned
// memory behind the kernels, the caller enqueues the NEXT iteration before it waits for this one.
int drp_gd_step_async(drp_ctx* c, int slot) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    if (slot < 0 || slot >= DRP_GD_SLOTS) return fail(c, DRP_EINVAL, "slot must be 0 .. %d", DRP_GD_SLOTS - 1);
    if (c->gd_pending[slot]) return fail(c, DRP_ESTATE, "slot %d holds an iteration nobody has waited for", slot);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nr = (size_t)c->gd_B, na = (size_t)c->gd_B * c->gd_H * 4;
    if (c->gd_pin_floats < nr + na) {
        for (int q = 0; q < DRP_GD_SLOTS; ++q) {
            if (c->gd_pending[q]) return fail(c, DRP_ESTATE, "the batch grew while an iteration was in flight");
            if (c->gd_pin[q]) HIPCHK(c, hipHostFree(c->gd_pin[q]));
            c->gd_pin[q] = nullptr;
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->gd_pin[q]), (nr + na) * sizeof(float), hipHostMallocDefault));
            if (!c->gd_ev[q]) HIPCHK(c, hipEventCreateWithFlags(&c->gd_ev[q], hipEventDisableTiming));
        }
        c->gd_pin_floats = nr + na;
    }
    // the iteration's own kernels write the slot (pinned host memory is device-visible): kb_reward the rewards, k_adam
    // the updated pushes -- two copies fewer on the stream per iteration (they were 27 of 197 us at 20 particles)
    c->gd_host_rewards = c->gd_pin[slot];
    c->gd_host_actions = c->gd_pin[slot] + nr;
    const int rc_it = gd_iteration(c);
    c->gd_host_rewards = c->gd_host_actions = nullptr;
    CHK(rc_it);
    HIPCHK(c, hipEventRecord(c->gd_ev[slot], c->stream));
    c->gd_pending[slot] = true;
    return DRP_OK;
}

int drp_gd_wait(drp_ctx* c, int slot, float* rewards_out, float* actions_out) {
    if (!c) return DRP_EINVAL;
    if (slot < 0 || slot >= DRP_GD_SLOTS || !c->gd_pending[slot]) return fail(c, DRP_ESTATE, "no iteration in flight in slot %d", slot);
    HIPCHK(c, hipSetDevice(c->device));
    c->gd_pending[slot] = false;
    CHK(guarded_wait(c, c->gd_ev[slot]));
    const size_t nr = (size_t)c->gd_B, na = (size_t)c->gd_B * c->gd_H * 4;
    if (rewards_out) memcpy(rewards_out, c->gd_pin[slot], nr * sizeof(float));
    if (actions_out) memcpy(actions_out, c->gd_pin[slot] + nr, na * sizeof(float));
    return DRP_OK;
}

int drp_gd_get(drp_ctx* c, float* actions_out) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    if (!actions_out) return fail(c, DRP_EINVAL, "null buffer");
    CHK(d2h(c, actions_out, c->actions.p, (size_t)c->gd_B * c->gd_H * 4 * sizeof(float)));
    return drp_sync(c);
}

// ---- training on the same kernels (row f4) ------------------------------------------------------
namespace {
// forward over n_rollout steps (+ loss), optionally the backward pass with weight gradients
int train_forward_backward(drp_ctx* c, int B, int N, bool backward) {
    const int H = c->tr_nroll;
    const size_t bn = (size_t)B * N, bn64 = bn * 64, bnk = bn * DRP_K;
    const size_t hstride = (size_t)H * N * 3;                 // predicted states [B][H][N][3]
    const size_t in_stride = (size_t)(H + 1) * N * 3;         // given states     [B][H+1][N][3]
    hipStream_t st = c->stream;
    const bool rev_lds = N <= KB_REV_LDS_MAX_N && !c->rev_global_only;
    float* states = ptr<float>(c->states);
    const float* given = ptr<float>(c->tr_states);
    float* eh = ptr<float>(c->eff_hist);
    unsigned* mh = ptr<unsigned>(c->tape_mask);
    float* ah = ptr<float>(c->agg_hist);
    float* g_state = ptr<float>(c->g_state);
    double* loss = ptr<double>(c->tr_loss);
    const float scale = 1.0f / (float)(H * B);
    const int saved_engine = c->engine;
    c->engine = c->tr_engine;
    const float* cself = nullptr;
    const uint8_t* cself_ok = nullptr;
    int rc = prepare_cself(c, B, N, B, &cself, &cself_ok);
    for (int t = 0; t < H && rc == DRP_OK; ++t) {
        // with a backward pass to follow, the step's impulses and neighbour lists are part of the tape: its workspace
        // pointers are lent the tape's slices for the call (as the GD planner does) instead of three copies afterwards
        void* const save_sd = c->s_delta.p; void* const save_idx = c->nbr_idx.p; void* const save_cnt = c->nbr_cnt.p;
        struct Lend {
            drp_ctx* c; void* sd; void* idx; void* cnt;
            ~Lend() { c->s_delta.p = sd; c->nbr_idx.p = idx; c->nbr_cnt.p = cnt; }
        } lend{c, save_sd, save_idx, save_cnt};
        if (backward) {
            c->s_delta.p = ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3;
            c->nbr_idx.p = ptr<int16_t>(c->tape_idx) + (size_t)t * bnk;
            c->nbr_cnt.p = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        }
        // this step's impulses are data (train/train_gnn_dyn.py:181)
        hipError_t e = hipMemcpy2DAsync(c->s_delta.p, (size_t)N * 3 * sizeof(float),
                                        ptr<float>(c->tr_sdelta) + (size_t)t * N * 3, hstride * sizeof(float),
                                        (size_t)N * 3 * sizeof(float), B, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { rc = fail(c, DRP_EHIP, "hipMemcpy2DAsync: %s", hipGetErrorString(e)); break; }
        StepArgs a{};
        if (t == 0) { a.s_prev = given; a.prev_mod = B; a.prev_stride = in_stride; }
        else { a.s_prev = states + (size_t)(t - 1) * N * 3; a.prev_mod = B; a.prev_stride = hstride; }
        a.attr = ptr<float>(c->attr); a.attr_mod = B;
        a.dens = ptr<float>(c->dens); a.dens_mod = B;
        a.actions = nullptr; a.act_stride = 0;
        a.build_graph = true;
        a.s_out = states + (size_t)t * N * 3; a.out_stride = hstride;
        a.B = B; a.N = N;
        a.cself = cself; a.cself_ok = cself_ok;
        a.padded = true;                // collate_fn pads with zero rows: coincident particles
        if (backward) {
            a.eff_hist = eh + (size_t)t * 4 * bn64;
            a.mask_hist = mh + (size_t)t * DRP_PSTEP * bnk * 2;
            a.agg_hist = ah + (size_t)t * 3 * bn64;
        }
        rc = run_step(c, a);
        if (rc != DRP_OK) break;
        // loss of this step and d loss / d s_pred_t (train/train_gnn_dyn.py:184-186, :203)
        hipLaunchKernelGGL(kt_mse_grad, dim3(B), dim3(256), 0, st, states + (size_t)t * N * 3, hstride,
                           given + (size_t)(t + 1) * N * 3, in_stride, ptr<int>(c->tr_nums), N, scale,
                           g_state + (size_t)t * bn * 3, loss + (size_t)t * B);
    }
    c->engine = saved_engine;
    CHK(rc);
    HIPCHK(c, hipGetLastError());
    if (!backward) return DRP_OK;

    const float* vw = ptr<float>(c->w_valu);
    const float* wraw = ptr<float>(c->w_raw);
    float* G = ptr<float>(c->tr_grad);
    // a training batch is a handful of samples: split each sample's rows over workgroups
    // (row kernels: one receiver per wave and pass; edge kernels: one receiver per 16 lanes)
    auto pick = [&](int rows_per_block) {
        int ch = (N + rows_per_block - 1) / rows_per_block;
        if (ch > 4096 / B) ch = 4096 / B;
        return ch < 1 ? 1 : ch;
    };
    const int chunks = pick(4), chunks16 = pick(16);
    const dim3 rgrid((unsigned)(B * chunks)), egrid((unsigned)(B * chunks16));
    const float* dens = ptr<float>(c->dens);
    HIPCHK(c, hipMemsetAsync(G, 0, (size_t)W_TOTAL * sizeof(float), st));
    // the reversed lists of ALL rollout steps in one launch (the tape holds every step's lists; a training batch is a handful
    // of workgroups per step)
    c->dv(N <= 512 ? DV_REV_256 : DV_REV_1024);
    if (N <= 512)
        hipLaunchKernelGGL(kb_reverse_lists<256>, dim3(B * H), dim3(256), KB_REV_LDS(N, rev_lds), st, ptr<int16_t>(c->tape_idx),
                           ptr<uint8_t>(c->tape_cnt), N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, ptr<int>(c->tr_nums), B);
    else
        hipLaunchKernelGGL(kb_reverse_lists<1024>, dim3(B * H), dim3(1024), KB_REV_LDS(N, rev_lds), st, ptr<int16_t>(c->tape_idx),
                           ptr<uint8_t>(c->tape_cnt), N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, ptr<int>(c->tr_nums), B);
    // deferred weight gradients: what a job reads keeps a buffer per rollout step t (g_eff and g_proj: per propagation step
    // too; slot 0 of g_eff is the transient copy the predictor writes and the particle encoder reads)
    const bool defer = c->wg_defer_now;
    const size_t per_t = defer ? 1 : 0;
    for (int t = H - 1; t >= 0; --t) {
        const size_t tt = per_t * (size_t)t;
        float* const ge_tmp = ptr<float>(c->g_eff);
        auto ge_v = [&](int v) { return ptr<float>(c->g_eff) + per_t * ((size_t)(t * 3 + v) + 1) * bn64; };   // v = 0, 1, 2: steps 2, 1, 0
        auto gp_v = [&](int p) { return ptr<float>(c->g_proj) + per_t * (size_t)(t * 3 + p) * bn64 * 2; };
        float* const g_cnode_t = ptr<float>(c->g_cnode) + tt * bn64;
        float* const tr_hact_t = ptr<float>(c->tr_hact) + tt * bn64;
        float* const tr_gh_t = ptr<float>(c->tr_gh) + tt * bn64;
        float* const tr_gpe_t = ptr<float>(c->tr_gpe) + tt * bn64;
        float* const tr_a1n_t = ptr<float>(c->tr_a1n) + tt * bn64;
        float* const tr_gh1_t = ptr<float>(c->tr_gh1) + tt * bn64;
        float* const tr_xn_t = ptr<float>(c->tr_xn) + tt * bn * 8;
        KbEdgeDump ed{ptr<float>(c->ed_re) + tt * bnk * 64, ptr<float>(c->ed_a2) + tt * bnk * 64, ptr<float>(c->ed_a1) + tt * bnk * 64,
                      ptr<float>(c->ed_x0) + tt * bnk * 8, ptr<float>(c->ed_gce) + tt * bnk * 64, ptr<float>(c->ed_g3) + tt * bnk * 64,
                      ptr<float>(c->ed_g2) + tt * bnk * 64, ptr<float>(c->ed_g1) + tt * bnk * 64};
        const float* s_prev = (t == 0) ? given : states + (size_t)(t - 1) * N * 3;
        const size_t prev_stride = (t == 0) ? in_stride : hstride;
        float* eht = eh + (size_t)t * 4 * bn64;
        const unsigned* mht = mh + (size_t)t * DRP_PSTEP * bnk * 2;
        float* aht = ah + (size_t)t * 3 * bn64;
        const int16_t* idx = ptr<int16_t>(c->tape_idx) + (size_t)t * bnk;
        const uint8_t* cnt = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        float* g_out = g_state + (size_t)t * bn * 3;
        float* gah = ptr<float>(c->g_agg_hist);
        int* const rev_off_t = ptr<int>(c->rev_off) + (size_t)t * B * (N + 1);
        int* const rev_t = ptr<int>(c->rev) + (size_t)t * bnk;
        // node-level stages: on the matrix cores when the batch has enough 32-row tiles to fill the chip,
        // otherwise the row kernels chunked over (sample, rows)
        if ((long)B * ((N + 31) / 32) >= KMB_MIN_TILES && !c->bwd_valu_stages) {
            const float* mw = ptr<float>(c->w_mfma);
            const float* mb = ptr<float>(c->w_mfma_bwd);
            const long node_tiles = (long)B * ((N + 31) / 32);
            const dim3 ngrid(mfma_grid_spread(c, node_tiles)), nblk(64 * MFMA_WAVES);
            c->dv(DV_TRAIN_NODE_MFMA);
            // predictor
            hipLaunchKernelGGL(kmb_predict, ngrid, nblk, KMB_PREDICT_LDS, st, mw, mb, eht + 3 * bn64, g_out, (size_t)N * 3, N, B,
                               ge_tmp, tr_hact_t, tr_gh_t);
            launch_wgrad<64>(c, tr_gh_t, 64, eht + 3 * bn64, 64, (long)bn, G + W_PR0_W, 64, 1, G + W_PR0_B, nullptr,
                             nullptr, 1, 1);
            launch_wgrad<3>(c, tr_hact_t, 64, g_out, 3, (long)bn, G + W_PR1_W, 1, 64, nullptr, nullptr, nullptr, 1, 1);
            hipLaunchKernelGGL(kt_colsum3, dim3(1), dim3(1024), 0, st, g_out, (long)bn, G + W_PR1_B);
            // update of the last propagation step; then per step the edge terms and, in one launch, the
            // projection of this step with the update of the one before (k_backward_mfma.h)
            hipLaunchKernelGGL((kmb_node_step<false, true>), ngrid, nblk, KMB_STEP_LDS(false, true), st, mb, ge_tmp, ge_v(0),
                               (const float*)nullptr, eht + (size_t)DRP_PSTEP * bn64, g_cnode_t, 1,
                               gah + (size_t)(DRP_PSTEP - 1) * bn64, N, B);
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn64;
                const unsigned* mask_p = mht + (size_t)p * bnk * 2;
                float* const ge_p = ge_v(DRP_PSTEP - 1 - p);     // the pre-activation gradient of step p
                float* const gp_p = gp_v(p);
                // particle propagator, aggregate columns
                launch_wgrad<64>(c, ge_p, 64, aht + (size_t)p * bn64, 64, (long)bn, G + W_PP_W + 64, 129, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                hipLaunchKernelGGL(kb_edge_terms, egrid, dim3(256), 0, st, g_agg_p, mask_p, cnt, rev_off_t,
                                   rev_t, N, gp_p, chunks16);
                // relation propagator, receiver and sender columns
                launch_wgrad<64>(c, gp_p, 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 64, 193, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                launch_wgrad<64>(c, gp_p + 64, 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 128, 193,
                                 1, nullptr, nullptr, nullptr, 1, 1);
                flush_wgrad(c);                          // (not deferred:) before the next kernel overwrites g_eff (and, next step, g_proj)
                if (p > 0)
                    hipLaunchKernelGGL((kmb_node_step<true, true>), ngrid, nblk, KMB_STEP_LDS(true, true), st, mb,
                                       ge_p, ge_v(DRP_PSTEP - p), gp_p, eht + (size_t)p * bn64,
                                       g_cnode_t, 0, gah + (size_t)(p - 1) * bn64, N, B);
                else
                    hipLaunchKernelGGL((kmb_node_step<true, false>), ngrid, nblk, KMB_STEP_LDS(true, false), st, mb,
                                       ge_p, ge_tmp, gp_p, (const float*)nullptr, (float*)nullptr, 0,
                                       (float*)nullptr, N, B);
            }
            // particle propagator, encoder columns + density column + bias; particle encoder
            launch_wgrad<64>(c, g_cnode_t, 64, eht, 64, (long)bn, G + W_PP_W, 129, 1, G + W_PP_B, G + W_PP_W + 128,
                             dens, B, (long)N);
            hipLaunchKernelGGL(kmb_node_encode, ngrid, nblk, KMB_NODE_ENCODE_LDS, st, mw, mb,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), B, dens, B, eht,
                               ge_tmp, g_cnode_t, N, B, ptr<float>(c->g_sdelta), tr_gpe_t,
                               tr_a1n_t, tr_gh1_t, tr_xn_t);
            launch_wgrad<64>(c, tr_gpe_t, 64, tr_a1n_t, 64, (long)bn, G + W_PE2_W, 64, 1, G + W_PE2_B,
                             nullptr, nullptr, 1, 1);
            launch_wgrad<5>(c, tr_gh1_t, 64, tr_xn_t, 8, (long)bn, G + W_PE0_W, 5, 1, G + W_PE0_B,
                            nullptr, nullptr, 1, 1);
            flush_wgrad(c);
        } else {
            c->dv(DV_TRAIN_NODE_VALU);
            // predictor
            hipLaunchKernelGGL(kb_predict, rgrid, dim3(256), 0, st, vw, wraw, eht + 3 * bn64, g_out, (size_t)N * 3, N,
                               ptr<float>(c->g_eff), ptr<float>(c->tr_hact), ptr<float>(c->tr_gh), chunks);
            launch_wgrad<64>(c, ptr<float>(c->tr_gh), 64, eht + 3 * bn64, 64, (long)bn, G + W_PR0_W, 64, 1, G + W_PR0_B, nullptr,
                             nullptr, 1, 1);
            launch_wgrad<3>(c, ptr<float>(c->tr_hact), 64, g_out, 3, (long)bn, G + W_PR1_W, 1, 64, nullptr, nullptr, nullptr, 1, 1);
            hipLaunchKernelGGL(kt_colsum3, dim3(1), dim3(1024), 0, st, g_out, (long)bn, G + W_PR1_B);
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn64;
                const unsigned* mask_p = mht + (size_t)p * bnk * 2;
                hipLaunchKernelGGL(kb_update, rgrid, dim3(256), 0, st, wraw, eht + (size_t)(p + 1) * bn64,
                                   ptr<float>(c->g_eff), ptr<float>(c->g_cnode), p == DRP_PSTEP - 1 ? 1 : 0, N, g_agg_p, chunks);
                // particle propagator, aggregate columns: g_eff now holds the pre-activation gradient
                launch_wgrad<64>(c, ptr<float>(c->g_eff), 64, aht + (size_t)p * bn64, 64, (long)bn, G + W_PP_W + 64, 129, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                hipLaunchKernelGGL(kb_edge_terms, egrid, dim3(256), 0, st, g_agg_p, mask_p, cnt, rev_off_t,
                                   rev_t, N, ptr<float>(c->g_proj), chunks16);
                // relation propagator, receiver and sender columns
                launch_wgrad<64>(c, ptr<float>(c->g_proj), 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 64, 193, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                launch_wgrad<64>(c, ptr<float>(c->g_proj) + 64, 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 128, 193,
                                 1, nullptr, nullptr, nullptr, 1, 1);
                flush_wgrad(c);                          // before kb_project overwrites g_eff
                hipLaunchKernelGGL(kb_project, rgrid, dim3(256), 0, st, wraw, ptr<float>(c->g_proj), N, ptr<float>(c->g_eff), chunks);
            }
            // particle propagator, encoder columns + density column + bias; particle encoder
            launch_wgrad<64>(c, ptr<float>(c->g_cnode), 64, eht, 64, (long)bn, G + W_PP_W, 129, 1, G + W_PP_B, G + W_PP_W + 128,
                             dens, B, (long)N);
            hipLaunchKernelGGL(kb_node_encode, rgrid, dim3(256), 0, st, vw, wraw,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), B, dens, B, eht,
                               ptr<float>(c->g_eff), ptr<float>(c->g_cnode), N, ptr<float>(c->g_sdelta), ptr<float>(c->tr_gpe),
                               ptr<float>(c->tr_a1n), ptr<float>(c->tr_gh1), ptr<float>(c->tr_xn), chunks);
            launch_wgrad<64>(c, ptr<float>(c->tr_gpe), 64, ptr<float>(c->tr_a1n), 64, (long)bn, G + W_PE2_W, 64, 1, G + W_PE2_B,
                             nullptr, nullptr, 1, 1);
            launch_wgrad<5>(c, ptr<float>(c->tr_gh1), 64, ptr<float>(c->tr_xn), 8, (long)bn, G + W_PE0_W, 5, 1, G + W_PE0_B,
                            nullptr, nullptr, 1, 1);
            flush_wgrad(c);
        }
        // the previous step's output feeds this step as s_cur: residual + relation encoder
        float* g_prev = nullptr;
        if (t > 0) {
            g_prev = g_state + (size_t)(t - 1) * bn * 3;
            hipLaunchKernelGGL(kt_add, dim3((unsigned)((bn * 3 + 255) / 256)), dim3(256), 0, st, g_prev, g_out, bn * 3);
        }
        c->dv(c->bwd_edge_mfma ? DV_BWD_EDGE_MFMA : DV_BWD_EDGE_VALU);
        if (c->bwd_edge_mfma)
            launch_edge_encode_mfma(c, s_prev, B, prev_stride, B, idx, cnt, gah, mht, bn, N, B,
                                    g_prev != nullptr ? ptr<float>(c->gpos_edge) : (float*)nullptr, ed);
        else
            hipLaunchKernelGGL(kb_edge_encode, rgrid, dim3(256), KB_EDGE_ENCODE_LDS, st, vw, wraw, s_prev, B,
                               prev_stride, ptr<float>(c->attr), B, dens, B, idx, cnt, gah, mht, bn, N, g_prev, (size_t)N * 3, ptr<float>(c->gpos_edge), ed, chunks);
        if (g_prev != nullptr)
            hipLaunchKernelGGL(kb_gather_pos, dim3((N + 255) / 256, B), dim3(256), 0, st, ptr<float>(c->gpos_edge),
                               rev_off_t, rev_t, N, g_prev, (size_t)N * 3, c->bwd_edge_mfma ? 1 : 0, cnt);
        launch_wgrad<64>(c, ed.gce, 64, ed.re, 64, (long)bnk, G + W_RP_W, 193, 1, G + W_RP_B, G + W_RP_W + 192, dens, B,
                         (long)N * DRP_K);
        launch_wgrad<64>(c, ed.g3, 64, ed.a2, 64, (long)bnk, G + W_RE4_W, 64, 1, G + W_RE4_B, nullptr, nullptr, 1, 1);
        launch_wgrad<64>(c, ed.g2, 64, ed.a1, 64, (long)bnk, G + W_RE2_W, 64, 1, G + W_RE2_B, nullptr, nullptr, 1, 1);
        launch_wgrad<6>(c, ed.g1, 64, ed.x0, 8, (long)bnk, G + W_RE0_W, 6, 1, G + W_RE0_B, nullptr, nullptr, 1, 1);
        flush_wgrad(c);                                  // the next rollout step rewrites the dumps these jobs read
    }
    flush_wgrad(c);
    if (defer) CHK(flush_wgrad_all(c));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

// The packed copies of the weights follow an optimiser step without a round trip of the packers through the host
// (k_train.h): only the blob itself comes back -- the split relation encoder's range shift is a function of the
// weights (set_split_range), and drp_get_weights serves the host copy.
int ensure_repack_maps(drp_ctx* c) {
    if (c->repack_maps_ready) return DRP_OK;
    std::vector<float> probe((size_t)W_TOTAL);
    for (int i = 0; i < (int)W_TOTAL; ++i) probe[i] = (float)(i + 1);          // exact in fp32 (38 403 < 2^24)
    auto to_map = [](const std::vector<float>& packed) {
        std::vector<int> m(packed.size());
        for (size_t i = 0; i < packed.size(); ++i)
            m[i] = packed[i] == 0.0f ? 0 : (packed[i] < 0.0f ? -1 : (int)packed[i]);
        return m;
    };
    std::vector<float> v, m, mb;
    pack_valu(probe.data(), v);
    pack_mfma(probe.data(), m);
    pack_mfma_bwd(probe.data(), mb);
    const std::vector<int> mv = to_map(v), mm = to_map(m), mmb = to_map(mb);
    CHK(h2d(c, c->map_valu, mv.data(), mv.size() * sizeof(int)));
    CHK(h2d(c, c->map_mfma, mm.data(), mm.size() * sizeof(int)));
    CHK(h2d(c, c->map_mfma_bwd, mmb.data(), mmb.size() * sizeof(int)));
    CHK(guarded_wait(c, nullptr));                   // the vectors go out of scope
    if (!c->w_pin) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->w_pin), (size_t)W_TOTAL * sizeof(float), hipHostMallocDefault));
    c->repack_maps_ready = true;
    return DRP_OK;
}

int repack_on_device(drp_ctx* c) {
    CHK(ensure_repack_maps(c));
    hipStream_t st = c->stream;
    const float* w = ptr<float>(c->w_raw);
    hipLaunchKernelGGL(kt_repack_gather, dim3((V_TOTAL + 255) / 256), dim3(256), 0, st, w, ptr<int>(c->map_valu), ptr<float>(c->w_valu), (int)V_TOTAL);
    hipLaunchKernelGGL(kt_repack_gather, dim3((M_TOTAL + 255) / 256), dim3(256), 0, st, w, ptr<int>(c->map_mfma), ptr<float>(c->w_mfma), (int)M_TOTAL);
    hipLaunchKernelGGL(kt_repack_gather, dim3((MB_TOTAL + 255) / 256), dim3(256), 0, st, w, ptr<int>(c->map_mfma_bwd), ptr<float>(c->w_mfma_bwd), (int)MB_TOTAL);
    hipLaunchKernelGGL(kt_repack_split6, dim3(7 * 16), dim3(256), 0, st, w, ptr<uint16_t>(c->w_split6));
    hipLaunchKernelGGL(kt_repack_split6_bwd, dim3(6 * 16), dim3(256), 0, st, w, ptr<uint16_t>(c->w_split6_bwd));
    // the relation encoder's range shift depends on the new weights: fetch the blob (it is the host copy
    // drp_get_weights serves anyway), derive the shift, then pack the split-fp16 fragments with it
    HIPCHK(c, hipMemcpyAsync(c->w_pin, c->w_raw.p, (size_t)W_TOTAL * sizeof(float), hipMemcpyDeviceToHost, st));
    CHK(guarded_wait(c, nullptr));
    c->w_host.assign(c->w_pin, c->w_pin + W_TOTAL);
    set_split_range(c, c->w_host.data());
    hipLaunchKernelGGL(kt_repack_split, dim3(4 * 16), dim3(256), 0, st, w, c->re_range.shift, ptr<uint16_t>(c->w_split));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int install_weights(drp_ctx* c, const std::vector<float>& blob) {
    std::vector<float> tmp(blob);
    return drp_load_weights(c, tmp.data(), tmp.size(), c->adj_thresh);
}
}  // namespace

int drp_train_begin(drp_ctx* c, int n_rollout, double lr, double beta1) {
    CHK(need(c, true, false, false));
    if (n_rollout < 1 || n_rollout > 64 || !(lr > 0.0) || !(beta1 >= 0.0 && beta1 < 1.0))
        return fail(c, DRP_EINVAL, "bad training arguments n_rollout=%d lr=%g beta1=%g", n_rollout, lr, beta1);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure(c, c->tr_grad, (size_t)W_TOTAL * sizeof(float)));
    CHK(ensure(c, c->tr_m, (size_t)W_TOTAL * sizeof(float)));
    CHK(ensure(c, c->tr_v, (size_t)W_TOTAL * sizeof(float)));
    CHK(ensure(c, c->tr_part, (size_t)KT_WGRAD_MAX_JOBS * KT_WGRAD_MAX_BLOCKS * 66 * 64 * sizeof(float)));
    HIPCHK(c, hipMemsetAsync(c->tr_m.p, 0, (size_t)W_TOTAL * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->tr_v.p, 0, (size_t)W_TOTAL * sizeof(float), c->stream));
    CHK(guarded_wait(c, nullptr));
    c->tr_nroll = n_rollout; c->tr_lr = lr; c->tr_beta1 = beta1; c->tr_iter = 0;
    c->tr_on = true;
    c->gd_on = false;
    c->mpc_on = false;
    return DRP_OK;
}

int drp_train_step(drp_ctx* c, const float* states, const float* states_delta, const float* attrs,
                   const int32_t* particle_nums, const float* particle_dens, int B, int N, int mode, double* loss_out,
                   float* grad_out) {
    if (!c || !c->tr_on) return fail(c, DRP_ESTATE, "drp_train_begin not called");
    CHK(check_bn(c, B, N));
    if (!states || !states_delta || !attrs || !particle_nums || !particle_dens) return fail(c, DRP_EINVAL, "null argument");
    if (mode < DRP_TRAIN_EVAL || mode > DRP_TRAIN_UPDATE) return fail(c, DRP_EINVAL, "bad mode %d", mode);
    for (int b = 0; b < B; ++b)
        if (particle_nums[b] <= 0 || particle_nums[b] > N)
            return fail(c, DRP_EINVAL, "particle_nums[%d]=%d outside 1..%d", b, particle_nums[b], N);
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    {
        float amax = 0.0f;                                    // a_cur = attrs[:, 0]
        for (int b = 0; b < B; ++b) amax = fmaxf(amax, max_abs(attrs + (size_t)b * (c->tr_nroll + 1) * N, (size_t)N));
        CHK(pick_tape_engine(c, amax, max_abs(particle_dens, (size_t)B), max_abs(states_delta, (size_t)B * c->tr_nroll * N * 3), &c->tr_engine));
    }
    const int H = c->tr_nroll;
    const size_t bn = (size_t)B * N, bn64 = bn * 64, bnk = bn * DRP_K;
    const bool backward = mode != DRP_TRAIN_EVAL;
    CHK(h2d(c, c->tr_states, states, (size_t)B * (H + 1) * N * 3 * sizeof(float)));
    CHK(h2d(c, c->tr_sdelta, states_delta, (size_t)B * H * N * 3 * sizeof(float)));
    CHK(h2d(c, c->tr_nums, particle_nums, (size_t)B * sizeof(int)));
    CHK(h2d(c, c->dens, particle_dens, (size_t)B * sizeof(float)));
    // a_cur = attrs[:, 0] for every step (train/train_gnn_dyn.py:173)
    CHK(ensure(c, c->attr, bn * sizeof(float)));
    CHK(h2d(c, c->scratch, attrs, (size_t)B * (H + 1) * N * sizeof(float)));
    HIPCHK(c, hipMemcpy2DAsync(c->attr.p, (size_t)N * sizeof(float), c->scratch.p, (size_t)(H + 1) * N * sizeof(float),
                               (size_t)N * sizeof(float), B, hipMemcpyDeviceToDevice, c->stream));
    CHK(ensure_step_ws(c, B, N, c->tr_engine));
    CHK(ensure(c, c->states, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->g_state, (size_t)H * bn * 3 * sizeof(float)));
    c->wg_defer_now = false;
    c->wg_jobs.clear();
    if (backward) {
        // deferred weight gradients keep every job's operands until the end of the backward pass: H copies of the node-level
        // dumps (3 H + 1 of g_eff, 3 H of g_proj) and of the relation encoder's dumps -- 0.24 GB per rollout step at 32 x 300
        const size_t keep_bytes = (size_t)H * (16 * bn64 + 7 * bnk * 64 + bnk * 8 + bn * 8) * sizeof(float);
        const bool defer = c->wgrad_defer && (long)B * ((N + 31) / 32) >= KMB_MIN_TILES && !c->bwd_valu_stages && keep_bytes <= ((size_t)8 << 30);
        c->wg_defer_now = defer;
        const size_t kt = defer ? (size_t)H : 1;
        CHK(ensure(c, c->eff_hist, (size_t)H * 4 * bn64 * sizeof(float)));
        CHK(ensure(c, c->agg_hist, (size_t)H * 3 * bn64 * sizeof(float)));
        CHK(ensure(c, c->tape_sdelta, (size_t)H * bn * 3 * sizeof(float)));
        CHK(ensure(c, c->tape_idx, (size_t)H * bnk * sizeof(int16_t)));
        CHK(ensure(c, c->tape_cnt, (size_t)H * bn));
        CHK(ensure(c, c->tape_mask, (size_t)H * DRP_PSTEP * bnk * 2 * sizeof(unsigned)));
        CHK(ensure(c, c->g_agg_hist, (size_t)DRP_PSTEP * bn64 * sizeof(float)));
        CHK(ensure(c, c->rev_off, (size_t)H * B * (N + 1) * sizeof(int)));
        CHK(ensure(c, c->rev, (size_t)H * bnk * sizeof(int)));
        CHK(ensure(c, c->gpos_edge, bnk * 4 * sizeof(float)));
        CHK(ensure(c, c->g_eff, (defer ? 3 * kt + 1 : 1) * bn64 * sizeof(float)));
        CHK(ensure(c, c->g_cnode, kt * bn64 * sizeof(float)));
        CHK(ensure(c, c->g_agg, bn64 * sizeof(float)));
        CHK(ensure(c, c->g_proj, (defer ? 3 * kt : 1) * bn64 * 2 * sizeof(float)));
        CHK(ensure(c, c->g_sdelta, bn * 3 * sizeof(float)));
        DevBuf* node64[] = {&c->tr_hact, &c->tr_gh, &c->tr_gpe, &c->tr_a1n, &c->tr_gh1};
        for (DevBuf* b : node64) CHK(ensure(c, *b, kt * bn64 * sizeof(float)));
        CHK(ensure(c, c->tr_xn, kt * bn * 8 * sizeof(float)));
        DevBuf* edge64[] = {&c->ed_re, &c->ed_a2, &c->ed_a1, &c->ed_gce, &c->ed_g3, &c->ed_g2, &c->ed_g1};
        for (DevBuf* b : edge64) CHK(ensure(c, *b, kt * bnk * 64 * sizeof(float)));
        CHK(ensure(c, c->ed_x0, kt * bnk * 8 * sizeof(float)));
    }
    CHK(ensure(c, c->tr_loss, (size_t)H * B * sizeof(double)));
    c->lastH = H;
    CHK(train_forward_backward(c, B, N, backward));
    std::vector<double> parts((size_t)H * B);
    if (loss_out) CHK(d2h(c, parts.data(), c->tr_loss.p, parts.size() * sizeof(double)));
    if (grad_out && backward) CHK(d2h(c, grad_out, c->tr_grad.p, (size_t)W_TOTAL * sizeof(float)));
    if (mode == DRP_TRAIN_UPDATE) {
        c->tr_iter += 1;
        const double bc1 = 1.0 - pow(c->tr_beta1, (double)c->tr_iter), bc2 = 1.0 - pow(0.999, (double)c->tr_iter);
        const float inf = __builtin_inff();
        hipLaunchKernelGGL(k_adam, dim3((W_TOTAL + 255) / 256), dim3(256), 0, c->stream, ptr<float>(c->w_raw),
                           ptr<float>(c->tr_grad), ptr<float>(c->tr_m), ptr<float>(c->tr_v), (int)W_TOTAL,
                           (float)(c->tr_lr / bc1), (float)sqrt(bc2), make_float4(-inf, -inf, -inf, -inf),
                           make_float4(inf, inf, inf, inf), (float)c->tr_beta1);
        HIPCHK(c, hipGetLastError());
        // the engines read packed copies of the weights: rebuild them from the updated blob
        if (c->repack_device) {
            CHK(repack_on_device(c));
        } else {
            std::vector<float> blob((size_t)W_TOTAL);
            CHK(d2h(c, blob.data(), c->w_raw.p, (size_t)W_TOTAL * sizeof(float)));
            CHK(guarded_wait(c, nullptr));
            CHK(install_weights(c, blob));
        }
    }
    CHK(drp_sync(c));
    if (loss_out) {
        double total = 0.0;                     // fixed order: step-major, then sample
        for (double v : parts) total += v;
        *loss_out = total;
    }
    return DRP_OK;
}

int drp_train_set_lr(drp_ctx* c, double lr) {
    if (!c || !c->tr_on) return fail(c, DRP_ESTATE, "drp_train_begin not called");
    if (!(lr > 0.0)) return fail(c, DRP_EINVAL, "bad lr %g", lr);
    c->tr_lr = lr;
    return DRP_OK;
}

int drp_get_weights(drp_ctx* c, float* blob_out, size_t n_floats) {
    CHK(need(c, true, false, false));
    if (!blob_out || n_floats != (size_t)W_TOTAL) return fail(c, DRP_EINVAL, "blob_out must hold %d floats", (int)W_TOTAL);
    memcpy(blob_out, c->w_host.data(), n_floats * sizeof(float));
    return DRP_OK;
}

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

// ---- measurement / debugging -----------------------------------------------------------------
#ifdef PROP_STAMPS
int drp_debug_prop_stamps(drp_ctx* c, unsigned long long* out8, int reset) {
    (void)c;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<unsigned long long> h(4096 * 8);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_prop_stamps), h.size() * 8) != hipSuccess) return -1;
    if (out8) {
        for (int q = 0; q < 8; ++q) out8[q] = 0;
        for (size_t i = 0; i < h.size(); ++i) out8[i & 7] += h[i];
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prop_stamps), h.data(), h.size() * 8) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef PROP_STAMPS
int drp_debug_prop_span(drp_ctx* c, unsigned long long* out, int n) {
    (void)c;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (n > 4096 * 2) n = 4096 * 2;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prop_span), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

int drp_probe_begin(drp_ctx* c, const char* kernel_class) {
    if (!c) return DRP_EINVAL;
    c->probe_cls = -1;
    c->probe_used = 0;
    c->probe_count = false;
    if (!kernel_class || !*kernel_class) return DRP_OK;
    if (strcmp(kernel_class, "prop+work") == 0) { kernel_class = "prop"; c->probe_count = true; }
    for (int i = 0; i < KC_COUNT; ++i)
        if (strcmp(kernel_class, kclass_names[i]) == 0) {
            if (i == KC_PROP && c->probe_count) {
                HIPCHK(c, hipSetDevice(c->device));
                CHK(ensure(c, c->probe_work, PROP_WORK_SHARDS * PROP_WORK_STRIDE * sizeof(unsigned long long)));
                HIPCHK(c, hipMemsetAsync(c->probe_work.p, 0, PROP_WORK_SHARDS * PROP_WORK_STRIDE * sizeof(unsigned long long), c->stream));
            }
            c->probe_cls = i;
            return DRP_OK;
        }
    return fail(c, DRP_EINVAL, "unknown kernel class '%s'", kernel_class);
}

int drp_probe_work(drp_ctx* c, unsigned long long out[8]) {
    if (!c || !out) return fail(c, DRP_EINVAL, "null argument");
    if (c->probe_cls != KC_PROP || !c->probe_count || !c->probe_work.p) return fail(c, DRP_ESTATE, "drp_probe_begin(\"prop+work\") not running");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<unsigned long long> sh((size_t)PROP_WORK_SHARDS * PROP_WORK_STRIDE);
    CHK(d2h(c, sh.data(), c->probe_work.p, sh.size() * sizeof(unsigned long long)));
    CHK(guarded_wait(c, nullptr));
    unsigned long long w[PROP_WORK_COUNT] = {};
    for (int q = 0; q < PROP_WORK_SHARDS; ++q)
        for (int i = 0; i < PROP_WORK_COUNT; ++i) w[i] += sh[(size_t)q * PROP_WORK_STRIDE + i];
    for (int i = 0; i < PROP_WORK_COUNT; ++i) out[i] = w[i];
    // the matrix instructions those units are made of (k_mlp_split.h: the chain of an edge slot, the node layers of a tile)
    out[5] = (unsigned long long)PROP_MFMA_CHAIN * w[PROP_WORK_CHAIN_SLOTS] + (unsigned long long)PROP_MFMA_NODE * w[PROP_WORK_TILES] +
             (unsigned long long)PROP_MFMA_NODE_LAST * w[PROP_WORK_TILES_LAST] + (unsigned long long)PROP_MFMA_ENC * w[PROP_WORK_ENC_TILES];
    out[6] = 0; out[7] = 0;
    return DRP_OK;
}

int drp_probe_read(drp_ctx* c, double* total_ms, long* launches) {
    if (!c) return DRP_EINVAL;
    CHK(guarded_wait(c, nullptr));
    double tot = 0.0;
    long n = 0;
    for (size_t i = 0; i + 1 < c->probe_used; i += 2) {
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->probe_ev[i], c->probe_ev[i + 1]));
        tot += ms;
        ++n;
    }
    c->probe_used = 0;
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return DRP_OK;
}

// holds the context's stream for `ms` milliseconds (a kernel spinning on the 100 MHz real-time counter): what a
// collective waiting for a dead peer looks like to the host.  tests/test_gpu_errors.py drives the deadline of
// guarded_wait with it.  ms <= 10 000.
__global__ void k_debug_stall(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int drp_dispatch_reset(drp_ctx* c) {
    if (!c) return DRP_EINVAL;
    memset(c->dv_hit, 0, sizeof(c->dv_hit));
    return DRP_OK;
}

static long dv_join(const unsigned char* hit, bool default_only, char* out, size_t out_len) {
    std::string all;
    char name[96];
    for (int id = 0; id < DV_COUNT; ++id) {
        bool dflt = true;
        dv_name(id, name, sizeof(name), &dflt);
        if (hit ? !hit[id] : (default_only && !dflt)) continue;
        if (!all.empty()) all += ';';
        all += name;
    }
    if (out && out_len) snprintf(out, out_len, "%s", all.c_str());
    return (long)all.size();
}

long drp_last_dispatch(drp_ctx* c, char* out, size_t out_len) {
    if (!c) return DRP_EINVAL;
    return dv_join(c->dv_hit, false, out, out_len);
}

long drp_dispatch_variants(int default_only, char* out, size_t out_len) { return dv_join(nullptr, default_only != 0, out, out_len); }

int drp_range_info(drp_ctx* c, int* shift, double* bound, double* wmax, int* ok) {
    CHK(need(c, true, false, false));
    const SplitRange& r = c->re_range;
    if (shift) *shift = r.shift;
    if (bound) *bound = split_range_bound(r, r.env_attr, r.env_delta, r.env_dens);
    if (wmax) *wmax = (double)r.wmax;
    if (ok) *ok = c->re_ok ? 1 : 0;
    return DRP_OK;
}

int drp_debug_stall(drp_ctx* c, int ms) {
    if (!c || ms < 0 || ms > 10000) return fail(c, DRP_EINVAL, "stall of %d ms outside 0..10000", ms);
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_debug_stall, dim3(1), dim3(1), 0, c->stream, (unsigned long long)ms * 100000ull);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

long drp_debug_fetch(drp_ctx* c, const char* name, void* out, size_t out_bytes) {
    if (!c || !name || !out) return DRP_EINVAL;
    const size_t bn = (size_t)c->lastB * c->lastN;
    const DevBuf* b = nullptr;
    size_t bytes = 0;
    if (!strcmp(name, "s_delta")) { b = &c->s_delta; bytes = bn * 3 * 4; }
    else if (!strcmp(name, "nbr_idx")) { b = &c->nbr_idx; bytes = bn * DRP_K * 2; }
    else if (!strcmp(name, "nbr_cnt")) { b = &c->nbr_cnt; bytes = bn; }
    else if (!strcmp(name, "effect")) { b = &c->eff; bytes = bn * 64 * 4; }
    else if (!strcmp(name, "c_node")) { b = &c->c_node; bytes = bn * 64 * 4; }
    else if (!strcmp(name, "c_edge")) { b = &c->c_edge; bytes = bn * DRP_K * 64 * 4; }
    else if (!strcmp(name, "proj")) { b = &c->proj; bytes = bn * 128 * 4; }
    else if (!strcmp(name, "agg")) { b = &c->agg; bytes = bn * 64 * 4; }
    else if (!strcmp(name, "stats")) { b = &c->stats; bytes = 8 * sizeof(double); }
    // the blob and its packed copies (tests: the device re-pack after an optimiser step against the host packers)
    else if (!strcmp(name, "w_raw")) { b = &c->w_raw; bytes = (size_t)W_TOTAL * 4; }
    else if (!strcmp(name, "w_valu")) { b = &c->w_valu; bytes = (size_t)V_TOTAL * 4; }
    else if (!strcmp(name, "w_mfma")) { b = &c->w_mfma; bytes = (size_t)M_TOTAL * 4; }
    else if (!strcmp(name, "w_mfma_bwd")) { b = &c->w_mfma_bwd; bytes = (size_t)MB_TOTAL * 4; }
    else if (!strcmp(name, "w_split")) { b = &c->w_split; bytes = (size_t)S_ALLOC * 16; }
    else if (!strcmp(name, "w_split6")) { b = &c->w_split6; bytes = (size_t)S6_TOTAL * 16; }
    else if (!strcmp(name, "w_split6_bwd")) { b = &c->w_split6_bwd; bytes = (size_t)SB6_TOTAL * 16; }
    else if (!strcmp(name, "rev_off")) { b = &c->rev_off; bytes = (size_t)c->lastB * (c->lastN + 1) * 4; }
    else if (!strcmp(name, "rev")) { b = &c->rev; bytes = bn * DRP_K * 4; }
    else return fail(c, DRP_EINVAL, "unknown buffer '%s'", name);
    // a GD session keeps every step's impulses and lists in its tape, not in the step workspace: the last step's
    DevBuf tape{};
    if (c->gd_on && c->gd_H > 0 && bn == (size_t)c->gd_B * c->gd_N) {
        const size_t t = (size_t)c->gd_H - 1;
        if (b == &c->s_delta) { tape.p = ptr<float>(c->tape_sdelta) + t * bn * 3; tape.cap = bytes; b = &tape; }
        else if (b == &c->nbr_idx) { tape.p = ptr<int16_t>(c->tape_idx) + t * bn * DRP_K; tape.cap = bytes; b = &tape; }
        else if (b == &c->nbr_cnt) { tape.p = ptr<uint8_t>(c->tape_cnt) + t * bn; tape.cap = bytes; b = &tape; }
    }
    if (!b->p || bytes == 0 || bytes > b->cap) return fail(c, DRP_ESTATE, "buffer '%s' not populated", name);
    if (out_bytes < bytes) return fail(c, DRP_EINVAL, "buffer '%s' needs %zu bytes", name, bytes);
    if (hipMemcpyAsync(out, b->p, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        return fail(c, DRP_EHIP, "debug fetch failed");
    { const int rc = guarded_wait(c, nullptr); if (rc != DRP_OK) return rc; }
    return (long)bytes;
}

}  // extern "C"

#ifdef ROLLOUT_STAMPS
extern "C" int drp_debug_roll_stamps(unsigned long long* out16, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_roll_stamps), 16 * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_roll_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef ROLLOUT_STAMPS
extern "C" int drp_debug_bwd_stamps(unsigned long long* out16, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bwd_stamps), 16 * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef GC_STATS
extern "C" int drp_gc_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gc_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gc_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
