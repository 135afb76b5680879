// The propagation kernels' template instantiations -- 16 of km_prop, 24 of km_prop3, 12 of km_rollout, each a 256-VGPR kernel
// the compiler works on for seconds -- are compiled in translation units of their own (inst_*.hip), in parallel; every other
// translation unit sees them as `extern template` and only launches them.  One list per kernel, used by both sides.
#pragma once
#include "k_mlp_split.h"
#include "k_rollout.h"

#define KM_PROP_SIG(L, T, P, W) __global__ void km_prop<L, T, P, W>( \
    const uint16_t*, const uint16_t*, const float*, const float*, int, size_t, const float*, int, const float*, int, const int16_t*, \
    const uint8_t*, const float*, const float*, const float*, float*, int, int, float*, float*, size_t, const float*, const uint8_t*, \
    unsigned*, float*, float, float, int, unsigned long long*)
#define KM_PROP3_SIG(T, P, E, W, O) __global__ void km_prop3<T, P, E, W, O>( \
    const uint16_t*, const uint16_t*, const float*, const float*, int, size_t, const float*, int, const float*, int, const int16_t*, \
    const uint8_t*, float*, float*, float*, float*, int, int, int, const float*, float*, size_t, const float*, const uint8_t*, \
    unsigned*, float*, float, float, int, float4*, size_t, unsigned long long*, size_t)
#define KM_ROLLOUT_SIG(P, E, W, O) __global__ void km_rollout<P, E, W, O>(const RolloutArgs*)

// X(last, tape, pair, work)
#define KM_PROP_LIST_TAPE(X, T) \
    X(false, T, false, false) X(false, T, false, true) X(false, T, true, false) X(false, T, true, true) \
    X(true, T, false, false) X(true, T, false, true) X(true, T, true, false) X(true, T, true, true)
// X(tape, pair, cache, work, rows kept): the cache's three states are (off), (on), (on + rows kept)
#define KM_PROP3_LIST_TAPE(X, T) \
    X(T, false, false, false, false) X(T, false, false, true, false) X(T, false, true, false, false) X(T, false, true, true, false) \
    X(T, false, true, false, true) X(T, false, true, true, true) \
    X(T, true, false, false, false) X(T, true, false, true, false) X(T, true, true, false, false) X(T, true, true, true, false) \
    X(T, true, true, false, true) X(T, true, true, true, true)
// X(pair, cache, work, rows kept)
#define KM_ROLLOUT_LIST(X) \
    X(false, false, false, false) X(false, false, true, false) X(false, true, false, false) X(false, true, true, false) \
    X(false, true, false, true) X(false, true, true, true) \
    X(true, false, false, false) X(true, false, true, false) X(true, true, false, false) X(true, true, true, false) \
    X(true, true, false, true) X(true, true, true, true)

#define KM_DECL_PROP(L, T, P, W) extern template KM_PROP_SIG(L, T, P, W);
#define KM_DECL_PROP3(T, P, E, W, O) extern template KM_PROP3_SIG(T, P, E, W, O);
#define KM_DECL_ROLLOUT(P, E, W, O) extern template KM_ROLLOUT_SIG(P, E, W, O);
#define KM_INST_PROP(L, T, P, W) template KM_PROP_SIG(L, T, P, W);
#define KM_INST_PROP3(T, P, E, W, O) template KM_PROP3_SIG(T, P, E, W, O);
#define KM_INST_ROLLOUT(P, E, W, O) template KM_ROLLOUT_SIG(P, E, W, O);

// The diagnostic builds (-DROLLOUT_STAMPS, -DPROP_STAMPS, -DGC_STATS) keep their counters in __device__ variables, and a
// __device__ variable is one per translation unit: there everything is instantiated where it is launched (csrc/drp_capi.hip,
// implicitly) and the inst_*.hip units are empty.
#if defined(ROLLOUT_STAMPS) || defined(PROP_STAMPS) || defined(GC_STATS)
#define DRP_UNITY 1
#endif
#if !defined(DRP_PROP_INSTANTIATE) && !defined(DRP_UNITY)
KM_PROP_LIST_TAPE(KM_DECL_PROP, false)
KM_PROP_LIST_TAPE(KM_DECL_PROP, true)
KM_PROP3_LIST_TAPE(KM_DECL_PROP3, false)
KM_PROP3_LIST_TAPE(KM_DECL_PROP3, true)
KM_ROLLOUT_LIST(KM_DECL_ROLLOUT)
#endif
