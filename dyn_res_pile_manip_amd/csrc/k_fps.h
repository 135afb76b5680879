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

// Same selection with the point set and the running distances held in registers (n <= 1024 * PT):
// an iteration is then arithmetic plus one block-wide arg-max, no memory traffic.  The goal pixel
// lists of the planner (<= ~22 k pixels of a 720 x 720 goal image) take this path.
// points per thread: 1024 threads leave 128 VGPRs each
#define FPS_REG_PT(DIM) ((DIM) == 2 ? 24 : 16)
template <int DIM>
__global__ void __launch_bounds__(1024)
k_fps_reg(const float* __restrict__ pts, int n, int k, int init_idx, int* __restrict__ chosen,
          float* __restrict__ max_dist_out) {
    __shared__ float sval[16];
    __shared__ int sidx[16];
    __shared__ int s_last;
    __shared__ float s_lp[DIM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PT = FPS_REG_PT(DIM);
    float p[PT][DIM], dist[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = tid + q * 1024;                 // ascending in q: first maximum = smallest q, then smallest tid
#pragma unroll
        for (int c = 0; c < DIM; ++c) p[q][c] = i < n ? pts[(size_t)i * DIM + c] : 0.0f;
        dist[q] = 0.0f;
    }
    if (tid == 0) chosen[0] = init_idx;
    if (tid < DIM) s_lp[tid] = pts[(size_t)init_idx * DIM + tid];
    __syncthreads();
    for (int it = 0; it < k; ++it) {
        float lp[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) lp[c] = s_lp[c];
        float best = -1.0f;
        int arg = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const int i = tid + q * 1024;
            float sq = 0.0f;
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const float d = p[q][c] - lp[c];
                sq = __fadd_rn(sq, __fmul_rn(d, d));
            }
            float nd = __fsqrt_rn(sq);
            if (it > 0) nd = fminf(dist[q], nd);
            dist[q] = nd;
            if (i < n && nd > best) { best = nd; arg = i; }
        }
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
            else *max_dist_out = bv;
        }
        __syncthreads();
        // the owner of the chosen point publishes its coordinates
        const int last = s_last;
        if ((last & 1023) == tid) {
            const int q = last >> 10;
#pragma unroll
            for (int qq = 0; qq < PT; ++qq)
                if (qq == q)
#pragma unroll
                    for (int c = 0; c < DIM; ++c) s_lp[c] = p[qq][c];
        }
        __syncthreads();
    }
}
