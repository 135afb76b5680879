// Goal pre-processing on the device, once per goal image:
//   * the distance transform inside config_reward_ptcl (env/flex_rewards.py:172-177):
//       G = goal - distanceTransform(goal < 0.5);  G -= min(G)
//   * the goal pixel list and its farthest-point subsample (planners.py:620-624).
// Two transforms:
//   DRP_DT_CV5   cv2.distanceTransform(src, DIST_L2, 5) as OpenCV's distanceTransform_5x5
//                computes it (modules/imgproc/src/distransform.cpp): 16.16 fixed-point chamfer
//                with weights 1, 1.4, 2.1969, one forward and one backward raster pass.  The
//                raster recurrence tmp[j] = min(c[j], tmp[j-1] + a) is a min-plus prefix scan,
//                tmp[j] = a*j + min_{k<=j}(c[k] - a*k), so a row is done by one workgroup in
//                parallel with exactly the integers of the sequential loop.
//   DRP_DT_EXACT the exact Euclidean transform (scipy.ndimage.distance_transform_edt): integer
//                squared distances, separable (columns, then rows), sqrt in float64.
#pragma once
#include "drp_common.h"
#include "k_particles.h"

#define DT_INIT0 (0x7fffffff >> 2)     // OpenCV INIT_DIST0
#define DT_HV 65536                    // CV_FLT_TO_FIX(1.0f, 16)
#define DT_DIAG 91750                  // cvRound(1.4f * 65536)
#define DT_LONG 143976                 // cvRound(2.1969f * 65536)
#define DT_THREADS 1024

// block-wide inclusive prefix-min of one value per thread (DT_THREADS threads), plus a carry
__device__ __forceinline__ int dt_block_prefix_min(int v, int* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v = min(v, o);
    }
    __syncthreads();
    if (lane == 63) s_w[wave] = v;
    __syncthreads();
    int pre = 0x7fffffff;
    for (int w = 0; w < wave; ++w) pre = min(pre, s_w[w]);
    return min(v, pre);
}

// One workgroup walks the rows.  tmp: [h][w] int32 work image (forward result, then final).
// LDS: three rows with a 2-pixel border each.
__global__ void __launch_bounds__(DT_THREADS)
k_dt_cv5(const uint8_t* __restrict__ src, int h, int w, int* __restrict__ tmp, float* __restrict__ dist) {
    extern __shared__ int s_rows[];
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x;
    const int ld = w + 4;
    int* r2 = s_rows;              // row i-2 (forward) / i+2 (backward)
    int* r1 = s_rows + ld;         // row i-1 / i+1
    int* rc = s_rows + 2 * ld;     // current row
    for (int j = tid; j < 2 * ld; j += DT_THREADS) s_rows[j] = DT_INIT0;
    __syncthreads();
    // ---- forward pass ----
    for (int i = 0; i < h; ++i) {
        if (tid == 0) s_carry = 0x7fffffff;
        for (int j = tid; j < 2; j += DT_THREADS) { rc[j] = DT_INIT0; rc[w + 2 + j] = DT_INIT0; }
        __syncthreads();
        for (int base = 0; base < w; base += DT_THREADS) {
            const int j = base + tid;
            int c = 0x7fffffff, v = 0x7fffffff;
            if (j < w) {
                if (!src[(size_t)i * w + j]) c = 0;
                else {
                    const int* p2 = r2 + 2 + j;
                    const int* p1 = r1 + 2 + j;
                    c = p2[-1] + DT_LONG;
                    c = min(c, p2[1] + DT_LONG);
                    c = min(c, p1[-2] + DT_LONG);
                    c = min(c, p1[-1] + DT_DIAG);
                    c = min(c, p1[0] + DT_HV);
                    c = min(c, p1[1] + DT_DIAG);
                    c = min(c, p1[2] + DT_LONG);
                }
                v = c - DT_HV * j;
            }
            const int carry = s_carry;
            int pm = min(dt_block_prefix_min(v, s_w), carry);
            if (j < w) {
                int t = pm + DT_HV * j;
                t = min(t, DT_INIT0 + DT_HV * (j + 1));        // the left border pixel, tmp[-1] + a*(j+1)
                rc[2 + j] = t;
                tmp[(size_t)i * w + j] = t;
            }
            __syncthreads();
            if (tid == DT_THREADS - 1) s_carry = pm;
            __syncthreads();
        }
        int* t = r2; r2 = r1; r1 = rc; rc = t;
    }
    // ---- backward pass ----
    __syncthreads();
    for (int j = tid; j < ld; j += DT_THREADS) { r2[j] = DT_INIT0; r1[j] = DT_INIT0; }
    __syncthreads();
    for (int i = h - 1; i >= 0; --i) {
        if (tid == 0) s_carry = 0x7fffffff;
        for (int j = tid; j < 2; j += DT_THREADS) { rc[j] = DT_INIT0; rc[w + 2 + j] = DT_INIT0; }
        __syncthreads();
        for (int base = 0; base < w; base += DT_THREADS) {
            const int jr = base + tid;                 // distance from the right end
            const int j = w - 1 - jr;
            int v = 0x7fffffff;
            if (jr < w) {
                const int* n2 = r2 + 2 + j;
                const int* n1 = r1 + 2 + j;
                int c = tmp[(size_t)i * w + j];
                c = min(c, n2[1] + DT_LONG);
                c = min(c, n2[-1] + DT_LONG);
                c = min(c, n1[2] + DT_LONG);
                c = min(c, n1[1] + DT_DIAG);
                c = min(c, n1[0] + DT_HV);
                c = min(c, n1[-1] + DT_DIAG);
                c = min(c, n1[-2] + DT_LONG);
                v = c - DT_HV * jr;
            }
            const int carry = s_carry;
            int pm = min(dt_block_prefix_min(v, s_w), carry);
            if (jr < w) {
                int t = pm + DT_HV * jr;
                t = min(t, DT_INIT0 + DT_HV * (jr + 1));
                rc[2 + j] = t;
                tmp[(size_t)i * w + j] = t;
                dist[(size_t)i * w + j] = (float)t * (1.0f / 65536.0f);
            }
            __syncthreads();
            if (tid == DT_THREADS - 1) s_carry = pm;
            __syncthreads();
        }
        int* t = r2; r2 = r1; r1 = rc; rc = t;
    }
}

// exact transform, phase 1: per column, distance to the nearest zero pixel of the column
#define EDT_INF 0x3fffffff
__global__ void __launch_bounds__(256)
k_edt_cols(const uint8_t* __restrict__ src, int h, int w, int* __restrict__ g) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    int d = EDT_INF;
    for (int y = 0; y < h; ++y) {
        d = src[(size_t)y * w + x] ? (d == EDT_INF ? EDT_INF : d + 1) : 0;
        g[(size_t)y * w + x] = d;
    }
    d = EDT_INF;
    for (int y = h - 1; y >= 0; --y) {
        d = src[(size_t)y * w + x] ? (d == EDT_INF ? EDT_INF : d + 1) : 0;
        if (d < g[(size_t)y * w + x]) g[(size_t)y * w + x] = d;
    }
}

// phase 2: per row, d2[x] = min_x' (x - x')^2 + g[x']^2 ; out = float32(sqrt(float64(d2)))
__global__ void __launch_bounds__(256)
k_edt_rows(const int* __restrict__ g, int h, int w, float* __restrict__ dist) {
    extern __shared__ int s_g[];
    const int y = blockIdx.x;
    for (int x = threadIdx.x; x < w; x += 256) s_g[x] = g[(size_t)y * w + x];
    __syncthreads();
    for (int x = threadIdx.x; x < w; x += 256) {
        long long best = 0x7fffffffffffffffll;
        for (int xp = 0; xp < w; ++xp) {
            const int gv = s_g[xp];
            if (gv == EDT_INF) continue;
            const long long dx = x - xp;
            const long long v = dx * dx + (long long)gv * gv;
            best = v < best ? v : best;
        }
        dist[(size_t)y * w + x] = (float)sqrt((double)best);
    }
}

__global__ void k_goal_seg(const float* __restrict__ goal, size_t n, uint8_t* __restrict__ seg) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) seg[i] = goal[i] < 0.5f ? 1 : 0;         // env/flex_rewards.py:173
}

// G = goal - dist (env/flex_rewards.py:175) + per-block minima
__global__ void __launch_bounds__(256)
k_goal_sub(const float* __restrict__ goal, const float* __restrict__ dist, size_t n, float* __restrict__ field,
           float* __restrict__ blk_min) {
    __shared__ float s_w[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float v = INFINITY;
    if (i < n) { v = __fsub_rn(goal[i], dist[i]); field[i] = v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) blk_min[blockIdx.x] = fminf(fminf(s_w[0], s_w[1]), fminf(s_w[2], s_w[3]));
}

__global__ void __launch_bounds__(1024) k_goal_min(const float* __restrict__ blk_min, int nblk, float* __restrict__ out) {
    __shared__ float s_w[16];
    float v = INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 1024) v = fminf(v, blk_min[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) v = fminf(v, s_w[w]);
        *out = v;
    }
}

__global__ void k_goal_shift(float* __restrict__ field, size_t n, const float* __restrict__ mn) {   // :176
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) field[i] = __fsub_rn(field[i], *mn);
}

// goal pixels (goal < 0.5) in row-major order as (col, row) float32 (planners.py:620-621)
__global__ void __launch_bounds__(PX_BLOCK)
k_goal_count(const uint8_t* __restrict__ seg, size_t npix, unsigned long long* __restrict__ blk_cnt) {
    const size_t base = (size_t)blockIdx.x * PX_TILE + (size_t)threadIdx.x * PX_PER_THREAD;
    int c = 0;
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q)
        if (base + q < npix && seg[base + q]) ++c;
    int total;
    (void)px_block_scan(c, total);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = (unsigned long long)total;
}

__global__ void __launch_bounds__(PX_BLOCK)
k_goal_compact(const uint8_t* __restrict__ seg, int w, size_t npix, const unsigned long long* __restrict__ blk_off,
               float* __restrict__ pix) {
    const size_t base = (size_t)blockIdx.x * PX_TILE + (size_t)threadIdx.x * PX_PER_THREAD;
    int c = 0;
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q)
        if (base + q < npix && seg[base + q]) ++c;
    int total;
    size_t pos = (size_t)blk_off[blockIdx.x] + (size_t)px_block_scan(c, total);
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q) {
        const size_t i = base + q;
        if (i < npix && seg[i]) {
            const int row = (int)(i / (size_t)w), col = (int)(i - (size_t)row * w);
            pix[pos * 2] = (float)col;
            pix[pos * 2 + 1] = (float)row;
            ++pos;
        }
    }
}

__global__ void k_goal_gather(const float* __restrict__ pix, const int* __restrict__ chosen, int k, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < k) { out[i * 2] = pix[(size_t)chosen[i] * 2]; out[i * 2 + 1] = pix[(size_t)chosen[i] * 2 + 1]; }
}
