// Farthest-point subsample (utils.py:451-466 fps_np), used once per planner call to pick
// min(5N, #goal pixels) goal pixels (planners.py:620-624) -- a Python loop of numpy passes in
// the reference (0.15 s at 17.7 k goal pixels -> 1500 points).
//
// One workgroup of 1024 threads; the point set and the running nearest-chosen distance stay
// in L2.  Every iteration: dist = min(dist, |p - last|) with |.| = sqrt of the fp32 sum of
// squares exactly as np.linalg.norm evaluates it, then a block-wide arg-max (first maximum,
// as np.argmax).  The selection is bit-identical to fps_np's.
#pragma once
#include "drp_common.h"

template <int DIM>
__global__ void __launch_bounds__(1024)
k_fps(const float* __restrict__ pts, int n, int k, int init_idx, float* __restrict__ dist, int* __restrict__ chosen,
      float* __restrict__ max_dist_out) {
    __shared__ float sval[16];
    __shared__ int sidx[16];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int last = init_idx;
    if (tid == 0) chosen[0] = init_idx;
    for (int it = 0; it < k; ++it) {
        float lp[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) lp[c] = pts[(size_t)last * DIM + c];
        float best = -1.0f;
        int arg = 0x7fffffff;
        for (int i = tid; i < n; i += 1024) {
            float sq = 0.0f;
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const float d = pts[(size_t)i * DIM + c] - lp[c];
                sq = __fadd_rn(sq, __fmul_rn(d, d));
            }
            float nd = __fsqrt_rn(sq);
            if (it > 0) nd = fminf(dist[i], nd);
            dist[i] = nd;
            if (nd > best) { best = nd; arg = i; }          // ascending i: first maximum wins
        }
        // block arg-max: larger value, then smaller index
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(arg, off, 64);
            if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
        }
        __syncthreads();
        if (lane == 0) { sval[wave] = best; sidx[wave] = arg; }
        __syncthreads();
        if (tid == 0) {
            float bv = sval[0];
            int bi = sidx[0];
            for (int w = 1; w < 16; ++w)
                if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
            s_last = bi;
            if (it + 1 < k) chosen[it + 1] = bi;
            else *max_dist_out = bv;                        // fps_np's second return value: dist.max()
        }
        __syncthreads();
        last = s_last;
    }
}

// Same selection with the point set and the running SQUARED distances held in registers (n <= 512 * PT): an iteration is
// arithmetic plus two block-wide reductions over eight waves, no memory traffic and NO square root per point.
//
// fps_np keeps dist = min(dist, sqrt(sq)) and takes the FIRST maximum of dist.  sqrt (correctly rounded) is monotone, so
// sqrt(min(a, b)) = min(sqrt(a), sqrt(b)) bit for bit: the kernel keeps d2 = min(d2, sq).  What sqrt can do is map two
// different d2 onto one dist -- numpy then takes the smaller index -- so the arg-max is taken in two steps:
//   1.  M2 = max d2 over the block, S = sqrt(M2) (= dist.max()), T = the smallest float whose square root still rounds to S
//       (found by stepping down from M2: at most a few floats);
//   2.  the chosen point = the smallest index with d2 >= T, i.e. with sqrt(d2) == S: numpy's first maximum.
// Round 4's kernel (1 024 threads x 24 points, a correctly rounded square root per point and iteration, four barriers) took
// 4.6 us per iteration -- 27.6 ms for the 6 000 goal pixels of a 1 200-particle plan; this one has two barriers, eight waves
// and about 500 instructions per wave and iteration.
#define FPS_WIDE_THREADS 512
#define FPS_REG_PT(DIM) ((DIM) == 2 ? 48 : 40)      // 512 threads, two waves per SIMD: 256 VGPRs each
template <int DIM>
__global__ void __launch_bounds__(FPS_WIDE_THREADS)
k_fps_reg(const float* __restrict__ pts, int n, int k, int init_idx, int* __restrict__ chosen,
          float* __restrict__ max_dist_out) {
    constexpr int PT = FPS_REG_PT(DIM), NW = FPS_WIDE_THREADS / 64;
    __shared__ float s_m[NW];
    __shared__ int s_i[NW];
    __shared__ float s_c[NW][DIM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float p[PT][DIM], d2[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = tid + q * FPS_WIDE_THREADS;     // ascending in q: a thread's first candidate is its smallest index
#pragma unroll
        for (int c = 0; c < DIM; ++c) p[q][c] = i < n ? pts[(size_t)i * DIM + c] : 0.0f;
        d2[q] = i < n ? __builtin_inff() : -1.0f;     // beyond the set: below every real distance, for ever
    }
    float lp[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) lp[c] = pts[(size_t)init_idx * DIM + c];
    if (tid == 0) chosen[0] = init_idx;
    for (int it = 0; it < k; ++it) {
        // ---- distances to the last chosen point, the thread's largest
        float m = -2.0f;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            float sq = 0.0f;
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const float d = p[q][c] - lp[c];
                sq = __fadd_rn(sq, __fmul_rn(d, d));
            }
            const float nd = fminf(d2[q], sq);
            d2[q] = nd;
            m = fmaxf(m, nd);
        }
        float wm = m;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
        if (lane == 0) s_m[wave] = wm;
        __syncthreads();
        float M2 = s_m[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M2 = fmaxf(M2, s_m[w]);
        const float S = __fsqrt_rn(M2);
        // the smallest float whose square root rounds to S
        float T = M2;
        for (int step = 0; step < 8 && T > 0.0f; ++step) {
            const float below = __uint_as_float(__float_as_uint(T) - 1u);
            if (__fsqrt_rn(below) != S) break;
            T = below;
        }
        // ---- the smallest index at that distance, with its coordinates
        int arg = 0x7fffffff;
        float ac[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) ac[c] = 0.0f;
        if (m >= T) {
#pragma unroll
            for (int q = PT - 1; q >= 0; --q)
                if (d2[q] >= T) {
                    arg = tid + q * FPS_WIDE_THREADS;
#pragma unroll
                    for (int c = 0; c < DIM; ++c) ac[c] = p[q][c];
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int oi = __shfl_xor(arg, off, 64);
            float oc[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) oc[c] = __shfl_xor(ac[c], off, 64);
            if (oi < arg) {
                arg = oi;
#pragma unroll
                for (int c = 0; c < DIM; ++c) ac[c] = oc[c];
            }
        }
        if (lane == 0) {
            s_i[wave] = arg;
#pragma unroll
            for (int c = 0; c < DIM; ++c) s_c[wave][c] = ac[c];
        }
        __syncthreads();
        int best = s_i[0], bw = 0;
#pragma unroll
        for (int w = 1; w < NW; ++w)
            if (s_i[w] < best) { best = s_i[w]; bw = w; }
#pragma unroll
        for (int c = 0; c < DIM; ++c) lp[c] = s_c[bw][c];
        if (tid == 0) {
            if (it + 1 < k) chosen[it + 1] = best;
            else *max_dist_out = S;                         // fps_np's second return value: dist.max()
        }
        // (s_m is rewritten after this iteration's second barrier, s_i / s_c after the next one's first: no reader is behind)
    }
}
