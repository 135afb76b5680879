// Shared definitions for the gfx950 kernels of the particle-GNN rollout engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRP_K 10
#define DRP_F 64
#define DRP_PSTEP 3              // model/gnn_dyn.py:160
#define DRP_DENS_SCALE 5000.0f   // model/gnn_dyn.py:158
#define DRP_PUSHER_W (0.8f / 24.0f)   // planners.py:228
#define DRP_SOFT_SCALE 0.01f     // planners.py:251

// Non-template kernels are DEFINED in these headers.  The translation units that only hold explicit instantiations of the
// propagation kernels (inst_*.hip, k_prop_inst.h) include the same headers: there they get internal linkage (and are dropped).
#ifdef DRP_PROP_INSTANTIATE
#define DRP_GLOBAL static __global__
#else
#define DRP_GLOBAL __global__
#endif

// The correctly rounded square root torch and numpy compute (IEEE sqrtf).  HIP's `__fsqrt_rn` is NOT that: without
// OCML_BASIC_ROUNDED_OPERATIONS it is __ocml_native_sqrt_f32, the bare v_sqrt_f32 (1 ulp) -- enough to move a particle across
// gen_s_delta's hard mask `u < len` (planners.py:248) or to break a near-tie of np.argmax in fps_np.  `__builtin_sqrtf` lowers
// to the llvm.sqrt intrinsic, which the backend expands with its fix-up steps (no fast-math flags in this build).
__device__ __forceinline__ float drp_sqrt_rn(float x) { return __builtin_sqrtf(x); }

// camera constants, passed by value to kernels (planners.py:192-209,
// env/flex_rewards.py:189-193)
struct DrpCam {
    float m[12];     // 3x4 affine world->camera (before the division by gs)
    float gs;        // global_scale
    float fx, fy, cx, cy;
};

// Offsets (in floats) into the state_dict blob, torch Linear layout [out,in]
// (module order of model/gnn_dyn.py:125-145).
enum {
    W_PE0_W = 0,                    // [64,5]
    W_PE0_B = W_PE0_W + 64 * 5,     // [64]
    W_PE2_W = W_PE0_B + 64,         // [64,64]
    W_PE2_B = W_PE2_W + 64 * 64,
    W_RE0_W = W_PE2_B + 64,         // [64,6]
    W_RE0_B = W_RE0_W + 64 * 6,
    W_RE2_W = W_RE0_B + 64,         // [64,64]
    W_RE2_B = W_RE2_W + 64 * 64,
    W_RE4_W = W_RE2_B + 64,         // [64,64]
    W_RE4_B = W_RE4_W + 64 * 64,
    W_PP_W = W_RE4_B + 64,          // [64,129] = [W_pe | W_agg | w_d]
    W_PP_B = W_PP_W + 64 * 129,
    W_RP_W = W_PP_B + 64,           // [64,193] = [W_e | W_r | W_s | w_d]
    W_RP_B = W_RP_W + 64 * 193,
    W_PR0_W = W_RP_B + 64,          // [64,64]
    W_PR0_B = W_PR0_W + 64 * 64,
    W_PR1_W = W_PR0_B + 64,         // [3,64]
    W_PR1_B = W_PR1_W + 3 * 64,
    W_TOTAL = W_PR1_B + 3           // 38403
};
static_assert(W_TOTAL == 38403, "state_dict size");

// Device-side weight pack for the VALU engine: every matrix transposed to
// [in][64] so lane = output feature reads consecutive floats.
enum {
    V_PE0_T = 0,                    // [5][64]
    V_PE0_B = V_PE0_T + 5 * 64,
    V_PE2_T = V_PE0_B + 64,         // [64][64]
    V_PE2_B = V_PE2_T + 4096,
    V_PPE_T = V_PE2_B + 64,         // W_pe^T  [64][64]   (particle propagator, encode part)
    V_PP_WD = V_PPE_T + 4096,       // w_d     [64]
    V_PP_B = V_PP_WD + 64,          // bias    [64]
    V_AGG_T = V_PP_B + 64,          // W_agg^T [64][64]
    V_RE0_T = V_AGG_T + 4096,       // [6][64]
    V_RE0_B = V_RE0_T + 6 * 64,
    V_RE2_T = V_RE0_B + 64,
    V_RE2_B = V_RE2_T + 4096,
    V_RE4_T = V_RE2_B + 64,
    V_RE4_B = V_RE4_T + 4096,
    V_RPE_T = V_RE4_B + 64,         // W_e^T [64][64]   (relation propagator, encode part)
    V_RP_WD = V_RPE_T + 4096,
    V_RP_B = V_RP_WD + 64,
    V_RPR_T = V_RP_B + 64,          // W_r^T [64][64]
    V_RPS_T = V_RPR_T + 4096,       // W_s^T [64][64]
    V_PR0_T = V_RPS_T + 4096,
    V_PR0_B = V_PR0_T + 4096,
    V_PR1_W = V_PR0_B + 64,         // [3][64] (row-major as in torch: dot per output)
    V_PR1_B = V_PR1_W + 192,
    V_TOTAL = V_PR1_B + 4
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float bcast_lane(float v, int lane_const) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_const));
}

// ReLU as ONE instruction: hipcc canonicalises the operand of fmaxf (an extra v_max_f32 x, x in
// front of every v_max_f32 whose input comes from an MFMA or a load: 96 of 613 vector
// instructions per slot in km_prop), and folds v_med3(x, 0, inf) back into that pair.  On the
// bit pattern ReLU is an integer max with 0: non-negative floats are non-negative ints and
// keep their bits, anything with the sign bit set (incl. -0) becomes +0, a +NaN stays a NaN.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// min of two NON-NEGATIVE floats (squared distances): their bit patterns order like ints
__device__ __forceinline__ float min_nonneg(float a, float b) { return __int_as_float(min(__float_as_int(a), __float_as_int(b))); }
