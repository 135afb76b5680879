// C ABI of the engine (include/drp.h): context, device workspaces, kernel pipelines -- ONE translation unit whose text lives in
// the capi_*.h sections included below (context; pipelines; then the entry points by surface: core, planner, pre-processing,
// gradient descent, training, RCCL, measurement).  The propagation kernels' instantiations are translation units of their own.
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

#include "capi_ctx.h"
#include "capi_pipeline.h"

extern "C" {

#include "capi_core.h"
#include "capi_mpc.h"
#include "capi_prep.h"
#include "capi_gd.h"
#include "capi_train.h"
#include "capi_comm.h"
#include "capi_debug.h"

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
