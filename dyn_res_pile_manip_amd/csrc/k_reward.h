// Particle reward, one workgroup per state row.
//
//   config_reward_ptcl   env/flex_rewards.py:156-214, downstream of the distance transform:
//     pix    = (x fx / z + cx, y fy / z + cy)                              :189-193
//     r1     = sum_n bilinear(G, pix_n)   grid_sample(border, align_corners=False)  :197-199
//     r2     = sum_m min_n |goal_coor_m - pix_n|                            :207-209
//     reward = -(r1 + r2) / N                                               :211-214
//
// The reference materialises dist[B',M,N] (18 GB at 1024 samples x 10 steps x 300
// particles); here a row's N pixel positions sit in LDS and each thread walks them for
// its goal points.  Sums use a fixed reduction tree, so results are reproducible.
#pragma once
#include "drp_common.h"

__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats */) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

// state row r lives at state + r * row_stride (floats); reward_out[r].
__global__ void __launch_bounds__(256)
k_reward(const float* __restrict__ state, size_t row_stride, int N, const float* __restrict__ G,
         int Hh, int Ww, const float* __restrict__ goal_coor, int M, DrpCam cam, int normalize,
         float* __restrict__ reward_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Na = (N + 3) & ~3;                     // py and the reduction scratch stay 16-B aligned
    float* px = lds;
    float* py = lds + Na;
    float* red = lds + 2 * Na;
    const float* s = state + (size_t)blockIdx.x * row_stride;
    float r1 = 0.0f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float x = s[n * 3 + 0], y = s[n * 3 + 1], z = s[n * 3 + 2];
        const float u = __fadd_rn(__fdiv_rn(__fmul_rn(x, cam.fx), z), cam.cx);
        const float v = __fadd_rn(__fdiv_rn(__fmul_rn(y, cam.fy), z), cam.cy);
        px[n] = u;
        py[n] = v;
        // normalise with H for both axes (:197), then grid_sample's un-normalisation
        const float nx = __fsub_rn(__fmul_rn(__fdiv_rn(u, (float)Hh), 2.0f), 1.0f);
        const float ny = __fsub_rn(__fmul_rn(__fdiv_rn(v, (float)Hh), 2.0f), 1.0f);
        float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(nx, 1.0f), (float)Ww), 1.0f), 2.0f);
        float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(ny, 1.0f), (float)Hh), 1.0f), 2.0f);
        ix = fminf(fmaxf(ix, 0.0f), (float)(Ww - 1));        // padding_mode='border'
        iy = fminf(fmaxf(iy, 0.0f), (float)(Hh - 1));
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float tx = ix - x0f, ty = iy - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = min(x0 + 1, Ww - 1), y1 = min(y0 + 1, Hh - 1);
        const float g00 = G[(size_t)y0 * Ww + x0], g01 = G[(size_t)y0 * Ww + x1];
        const float g10 = G[(size_t)y1 * Ww + x0], g11 = G[(size_t)y1 * Ww + x1];
        r1 += g00 * ((1.0f - tx) * (1.0f - ty)) + g01 * (tx * (1.0f - ty)) +
              g10 * ((1.0f - tx) * ty) + g11 * (tx * ty);
    }
    __syncthreads();
    float r2 = 0.0f;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const float gx = goal_coor[m * 2 + 0], gy = goal_coor[m * 2 + 1];
        float best = __builtin_inff();
        int n = 0;
        for (; n + 4 <= N; n += 4) {                 // four particles per pair of broadcast LDS reads
            const float4 x4 = *reinterpret_cast<const float4*>(px + n), y4 = *reinterpret_cast<const float4*>(py + n);
            const float dx0 = gx - x4.x, dy0 = gy - y4.x, dx1 = gx - x4.y, dy1 = gy - y4.y;
            const float dx2 = gx - x4.z, dy2 = gy - y4.z, dx3 = gx - x4.w, dy3 = gy - y4.w;
            const float d0 = __fadd_rn(__fmul_rn(dx0, dx0), __fmul_rn(dy0, dy0)), d1 = __fadd_rn(__fmul_rn(dx1, dx1), __fmul_rn(dy1, dy1));
            const float d2 = __fadd_rn(__fmul_rn(dx2, dx2), __fmul_rn(dy2, dy2)), d3 = __fadd_rn(__fmul_rn(dx3, dx3), __fmul_rn(dy3, dy3));
            best = fminf(fminf(best, fminf(d0, d1)), fminf(d2, d3));
        }
        for (; n < N; ++n) {
            const float dx = gx - px[n], dy = gy - py[n];
            best = fminf(best, __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
        }
        r2 += drp_sqrt_rn(best);       // min of sqrt == sqrt of min (sqrt is monotone)
    }
    const float t1 = block_sum_256(r1, red);
    const float t2 = block_sum_256(r2, red);
    if (threadIdx.x == 0) {
        float r = t1 + t2;
        if (normalize) r = __fdiv_rn(r, (float)N);
        reward_out[blockIdx.x] = -r;
    }
}
