// Push impulse + neighbour graph, one workgroup per sample.
//
//   gen_s_delta        planners.py:211-257   (push -> per-particle impulse)
//   graph build        model/gnn_dyn.py:223-251 (radius AND top-10, receiver-major lists)
//
// The reference materialises [B,N,N,3] pairwise tensors and dense one-hot Rr/Rs; here a
// sample's N displaced positions sit in LDS (12 B each), every thread owns one
// receiver and sweeps the senders with broadcast LDS reads, and the result is
// nbr_idx[B,N,10] int16 + nbr_cnt[B,N] uint8: ascending sender index, the order
// `nonzero()` enumerates a receiver's edges in (model/gnn_dyn.py:247).
//
// Distances are evaluated exactly as torch does: ((dx*dx + dy*dy) + dz*dz) in fp32 with
// no FMA contraction, and the radius test is (dis - thr) < 0 (model/gnn_dyn.py:229-236).
#pragma once
#include "drp_common.h"

struct PushFrame {           // per-sample push geometry in the camera frame
    float sx, sy, sz;        // start
    float ex, ey, ez;        // end
    float dx, dy, dz;        // unit direction
    float len;
};

__device__ __forceinline__ void cam_point(const DrpCam& c, float x, float y, float z, float& ox,
                                          float& oy, float& oz) {
    // (M [p;1])[:3] / gs, planners.py:206
    ox = __fdiv_rn(fmaf(c.m[2], z, fmaf(c.m[1], y, fmaf(c.m[0], x, c.m[3]))), c.gs);
    oy = __fdiv_rn(fmaf(c.m[6], z, fmaf(c.m[5], y, fmaf(c.m[4], x, c.m[7]))), c.gs);
    oz = __fdiv_rn(fmaf(c.m[10], z, fmaf(c.m[9], y, fmaf(c.m[8], x, c.m[11]))), c.gs);
}

__device__ __forceinline__ PushFrame push_frame(const DrpCam& c, const float* act) {
    PushFrame f;
    // s_3d = (sx, 0, -sy), e_3d = (ex, 0, -ey): planners.py:231-234
    cam_point(c, act[0], 0.0f, -act[1], f.sx, f.sy, f.sz);
    cam_point(c, act[2], 0.0f, -act[3], f.ex, f.ey, f.ez);
    float vx = f.ex - f.sx, vy = f.ey - f.sy, vz = f.ez - f.sz;
    f.len = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz)));
    // zero-length push: 0/0 = NaN, exactly as the reference (planners.py:240)
    f.dx = __fdiv_rn(vx, f.len);
    f.dy = __fdiv_rn(vy, f.len);
    f.dz = __fdiv_rn(vz, f.len);
    return f;
}

__device__ __forceinline__ void push_delta(const PushFrame& f, float px, float py, float pz,
                                           float& ox, float& oy, float& oz) {
    // ortho = (-dir_y, dir_x, 0): planners.py:242
    float rx = px - f.sx, ry = py - f.sy, rz = pz - f.sz;
    float v = __fadd_rn(__fadd_rn(__fmul_rn(rx, -f.dy), __fmul_rn(ry, f.dx)), __fmul_rn(rz, 0.0f));
    float u = __fadd_rn(__fadd_rn(__fmul_rn(rx, f.dx), __fmul_rn(ry, f.dy)), __fmul_rn(rz, f.dz));
    float hard = (u < f.len && u > 0.0f) ? 1.0f : 0.0f;                       // :248
    float soft = fmaxf(fmaxf(-DRP_PUSHER_W - v, 0.0f), fmaxf(v - DRP_PUSHER_W, 0.0f));  // :249-250
    soft = expf(__fdiv_rn(-soft, DRP_SOFT_SCALE));                            // :251
    float tx = f.ex - px, ty = f.ey - py, tz = f.ez - pz;
    float to_end = __fadd_rn(__fadd_rn(__fmul_rn(tx, f.dx), __fmul_rn(ty, f.dy)), __fmul_rn(tz, f.dz));
    ox = __fmul_rn(__fmul_rn(__fmul_rn(to_end, f.dx), hard), soft);          // :254
    oy = __fmul_rn(__fmul_rn(__fmul_rn(to_end, f.dy), hard), soft);
    oz = __fmul_rn(__fmul_rn(__fmul_rn(to_end, f.dz), hard), soft);
}

// s_delta only (drp_gen_s_delta): grid B, any block size
__global__ void k_sdelta(const float* __restrict__ s_cur, const float* __restrict__ action, int N,
                         float* __restrict__ s_delta, DrpCam cam) {
    const int b = blockIdx.x;
    const PushFrame f = push_frame(cam, action + (size_t)b * 4);
    const float* s = s_cur + (size_t)b * N * 3;
    float* o = s_delta + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float x, y, z;
        push_delta(f, s[i * 3 + 0], s[i * 3 + 1], s[i * 3 + 2], x, y, z);
        o[i * 3 + 0] = x;
        o[i * 3 + 1] = y;
        o[i * 3 + 2] = z;
    }
}

__device__ __forceinline__ float pair_dis(float xi, float yi, float zi, float xj, float yj, float zj) {
    float dx = xj - xi, dy = yj - yi, dz = zj - zi;     // sender - receiver, gnn_dyn.py:230
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// grid = B, block = min(1024, N rounded up to a wave) threads (one receiver per thread in a
// single pass whenever N <= 1024), dynamic LDS = 3*N floats.
//   s_prev  : sample b reads row (b % prev_mod) at stride prev_stride floats
//   actions : if not null, s_delta is generated from actions[b*act_stride ..+4] and
//             written to s_delta; otherwise s_delta is read.
__global__ void __launch_bounds__(1024)
k_graph(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
        const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
        int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, DrpCam cam, float thr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* px = lds;
    float* py = lds + N;
    float* pz = lds + 2 * N;
    const int b = blockIdx.x;
    const int BLOCK = blockDim.x;
    const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
    float* sd = s_delta + (size_t)b * N * 3;

    if (actions != nullptr) {
        const PushFrame f = push_frame(cam, actions + (size_t)b * act_stride);
        for (int i = threadIdx.x; i < N; i += BLOCK) {
            float x = s[i * 3 + 0], y = s[i * 3 + 1], z = s[i * 3 + 2];
            float ox, oy, oz;
            push_delta(f, x, y, z, ox, oy, oz);
            sd[i * 3 + 0] = ox;
            sd[i * 3 + 1] = oy;
            sd[i * 3 + 2] = oz;
            px[i] = __fadd_rn(x, ox);      // p = s_cur + s_delta, gnn_dyn.py:224
            py[i] = __fadd_rn(y, oy);
            pz[i] = __fadd_rn(z, oz);
        }
    } else {
        for (int i = threadIdx.x; i < N; i += BLOCK) {
            px[i] = __fadd_rn(s[i * 3 + 0], sd[i * 3 + 0]);
            py[i] = __fadd_rn(s[i * 3 + 1], sd[i * 3 + 1]);
            pz[i] = __fadd_rn(s[i * 3 + 2], sd[i * 3 + 2]);
        }
    }
    __syncthreads();

    for (int i = threadIdx.x; i < N; i += BLOCK) {
        const float xi = px[i], yi = py[i], zi = pz[i];
        // pass 1: the 10 smallest distances, ascending, in registers (static indices only)
        float best[DRP_K];
#pragma unroll
        for (int q = 0; q < DRP_K; ++q) best[q] = __builtin_inff();
        for (int j = 0; j < N; ++j) {
            const float d = pair_dis(xi, yi, zi, px[j], py[j], pz[j]);
            if (d < best[DRP_K - 1]) {
#pragma unroll
                for (int q = DRP_K - 1; q > 0; --q)
                    best[q] = (d < best[q - 1]) ? best[q - 1] : fminf(best[q], d);
                best[0] = fminf(best[0], d);
            }
        }
        // pass 2: ascending sender index; keep j if it is among the 10 nearest and inside
        // the radius.  (N < 10: best[9] stays +inf and every particle is "top-k".)
        const float kth = best[DRP_K - 1];
        int16_t* out = nbr_idx + ((size_t)b * N + i) * DRP_K;
        int cnt = 0;
        for (int j = 0; j < N; ++j) {
            const float d = pair_dis(xi, yi, zi, px[j], py[j], pz[j]);
            if (d <= kth && __fsub_rn(d, thr) < 0.0f && cnt < DRP_K) out[cnt++] = (int16_t)j;
        }
        nbr_cnt[(size_t)b * N + i] = (uint8_t)cnt;
        for (int q = cnt; q < DRP_K; ++q) out[q] = -1;
    }
}
