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
            float nd = drp_sqrt_rn(sq);
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
#define FPS_REG_PT(DIM) ((DIM) == 2 ? 48 : 32)      // 512 threads, two waves per SIMD: 256 VGPRs each (3-D: 40 points spill)
// wave-wide max / min as a SCALAR: four DPP row rotations (an all-reduce over each row of 16 lanes), then one v_readlane per
// row -- a dozen instructions, against six dependent ds_bpermute round trips of a __shfl_xor ladder (two ladders per
// iteration were most of its 3 us)
template <int R>
__device__ __forceinline__ int fps_ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + R, 0xf, 0xf, true); }
__device__ __forceinline__ float fps_wave_max(float x) {
    int v = __float_as_int(x);
#define FPS_STEP(R) v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(fps_ror<R>(v))))
    FPS_STEP(8); FPS_STEP(4); FPS_STEP(2); FPS_STEP(1);
#undef FPS_STEP
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(v, 0)), __int_as_float(__builtin_amdgcn_readlane(v, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(v, 32)), __int_as_float(__builtin_amdgcn_readlane(v, 48))));
}
__device__ __forceinline__ int fps_wave_min(int v) {
    v = min(v, fps_ror<8>(v)); v = min(v, fps_ror<4>(v)); v = min(v, fps_ror<2>(v)); v = min(v, fps_ror<1>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
typedef float fps_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fps_wave_min_f(float x) { return -fps_wave_max(-x); }
// One CU's vector units and the latency of two block-wide reductions bound an iteration.  Two things keep it short:
//  * the points are held as PAIRS -- (x_q, x_q+1), (y_q, y_q+1) -- so that differences, squares and the sum are packed fp32
//    instructions (v_pk_add_f32 / v_pk_mul_f32: two IEEE results each, nothing fused);
//  * a thread holds CONSECUTIVE points (goal pixels come in raster order: a wave's 64 x PT points are a band of the image) and a
//    wave knows its bounding box: when the newly chosen point is at least as far from the box as the wave's largest d2, no d2
//    of the wave can change (fp32 subtraction, squaring and addition are monotone, so the bound holds for the ROUNDED values:
//    sq_point >= L >= max d2 gives min(d2, sq) = d2 bit for bit) and the wave skips its update -- after the first few hundred
//    picks most waves skip most iterations.
template <int DIM>
__global__ void __launch_bounds__(FPS_WIDE_THREADS)
k_fps_reg(const float* __restrict__ pts, int n, int k, int init_idx, int* __restrict__ chosen,
          float* __restrict__ max_dist_out) {
    constexpr int PT = FPS_REG_PT(DIM), NP = PT / 2, NW = FPS_WIDE_THREADS / 64;
    static_assert(PT % 2 == 0, "points are held in pairs");
    __shared__ float s_m[NW];
    __shared__ int s_i[NW];
    __shared__ float s_c[NW][DIM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + FPS_WIDE_THREADS - 1) / FPS_WIDE_THREADS;         // consecutive points per thread (<= PT: the host's promise)
    const int base = tid * per;
    fps_v2 p[NP][DIM], d2[NP];            // pair h = slots 2h, 2h + 1; slot q is point base + q
    float lo[DIM], hi[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) { lo[c] = __builtin_inff(); hi[c] = -__builtin_inff(); }
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        const int i0 = base + 2 * h, i1 = i0 + 1;
        const bool v0 = 2 * h < per && i0 < n, v1 = 2 * h + 1 < per && i1 < n;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            p[h][c].x = v0 ? pts[(size_t)i0 * DIM + c] : 0.0f;
            p[h][c].y = v1 ? pts[(size_t)i1 * DIM + c] : 0.0f;
            if (v0) { lo[c] = fminf(lo[c], p[h][c].x); hi[c] = fmaxf(hi[c], p[h][c].x); }
            if (v1) { lo[c] = fminf(lo[c], p[h][c].y); hi[c] = fmaxf(hi[c], p[h][c].y); }
        }
        d2[h].x = v0 ? __builtin_inff() : -1.0f;      // beyond the set: below every real distance, for ever
        d2[h].y = v1 ? __builtin_inff() : -1.0f;
    }
    // the wave's bounding box, as scalars (an empty wave: lo = +inf, hi = -inf -- infinitely far from everything)
#pragma unroll
    for (int c = 0; c < DIM; ++c) { lo[c] = fps_wave_min_f(lo[c]); hi[c] = fps_wave_max(hi[c]); }
    float lp[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) lp[c] = pts[(size_t)init_idx * DIM + c];
    if (tid == 0) chosen[0] = init_idx;
    float m = -2.0f, wm = __builtin_inff();           // the thread's and the wave's largest d2
    for (int it = 0; it < k; ++it) {
        // ---- can any d2 of this wave change?  L = the rounded squared distance of the chosen point to the wave's box
        float L = 0.0f;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const float d = lp[c] < lo[c] ? lo[c] - lp[c] : (lp[c] > hi[c] ? hi[c] - lp[c] : 0.0f);
            L = c == 0 ? d * d : L + d * d;
        }
        if (it == 0 || !(L >= wm)) {                  // (the first pass sets every d2, an empty wave's largest to -1)
            // ---- squared distances to the chosen point (np.linalg.norm's sum of squares, left to right), the thread's largest
            m = -2.0f;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                fps_v2 sq;
                {
                    const fps_v2 d = p[h][0] - fps_v2{lp[0], lp[0]};
                    sq = d * d;
                }
#pragma unroll
                for (int c = 1; c < DIM; ++c) {
                    const fps_v2 d = p[h][c] - fps_v2{lp[c], lp[c]};
                    sq = sq + d * d;
                }
                fps_v2 nd;
                nd.x = fminf(d2[h].x, sq.x);
                nd.y = fminf(d2[h].y, sq.y);
                d2[h] = nd;
                m = fmaxf(fmaxf(m, nd.x), nd.y);
            }
            wm = fps_wave_max(m);
        }
        if (lane == 0) s_m[wave] = wm;
        __syncthreads();
        float M2 = s_m[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M2 = fmaxf(M2, s_m[w]);
        const float S = drp_sqrt_rn(M2);
        // the smallest float whose square root rounds to S
        float T = M2;
        for (int step = 0; step < 8 && T > 0.0f; ++step) {
            const float below = __uint_as_float(__float_as_uint(T) - 1u);
            if (drp_sqrt_rn(below) != S) break;
            T = below;
        }
        // ---- the smallest index at that distance, with its coordinates
        int arg = 0x7fffffff;
        float ac[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) ac[c] = 0.0f;
        if (wm >= T) {
            if (m >= T) {
#pragma unroll
                for (int h = NP - 1; h >= 0; --h) {
                    if (d2[h].y >= T) {
                        arg = base + 2 * h + 1;
#pragma unroll
                        for (int c = 0; c < DIM; ++c) ac[c] = p[h][c].y;
                    }
                    if (d2[h].x >= T) {
                        arg = base + 2 * h;
#pragma unroll
                        for (int c = 0; c < DIM; ++c) ac[c] = p[h][c].x;
                    }
                }
            }
            // the wave's smallest candidate as a scalar; its owner (one lane: indices are unique) publishes it with its coordinates
            const int wmin = fps_wave_min(arg);
            if (arg == wmin) {
                s_i[wave] = wmin;
#pragma unroll
                for (int c = 0; c < DIM; ++c) s_c[wave][c] = ac[c];
            }
        } else if (lane == 0) {
            s_i[wave] = 0x7fffffff;
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
