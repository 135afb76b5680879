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

// Neighbour lists, one thread per receiver, two sweeps over the sample's senders (positions
// broadcast from LDS, one ds_read_b128 each):
//   1. the ten smallest in-radius distances, kept ascending in ten registers.  Inserting d
//      into an ascending list is new[q] = clamp(d, old[q-1], old[q]) = v_med3_f32: ten
//      instructions per sender, no branch, no divergence.  The list starts at thr, so
//      senders outside the radius (which can never be edges: adj = (dis - thr < 0) * topk)
//      leave it unchanged and list[9] ends as min(thr, 10th smallest in-radius distance).
//   2. senders in ascending index with d <= list[9] and d - thr < 0 are the edges (at most
//      ten; ties at the cut go to the lower index).
// Both sweeps evaluate the distance with the same expression, so the result is exactly
// the reference's radius AND top-10 mask.
// grid = B * ceil(N / GRAPH_THREADS): a workgroup owns GRAPH_THREADS consecutive receivers of
// one sample and stages all N displaced positions (16 B each) in LDS; the first workgroup
// of a sample writes s_delta.  dynamic LDS = 4*N floats.
#define GRAPH_THREADS 128

__global__ void __launch_bounds__(GRAPH_THREADS)
k_graph(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
        const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
        int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, DrpCam cam, float thr, int chunks,
        int self_first) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* p4 = reinterpret_cast<float4*>(lds);                       // [N] displaced positions
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
    float* sd = s_delta + (size_t)b * N * 3;

    if (actions != nullptr) {
        const PushFrame f = push_frame(cam, actions + (size_t)b * act_stride);
        for (int i = threadIdx.x; i < N; i += GRAPH_THREADS) {
            float x = s[i * 3 + 0], y = s[i * 3 + 1], z = s[i * 3 + 2];
            float ox, oy, oz;
            push_delta(f, x, y, z, ox, oy, oz);
            if (chunk == 0) {
                sd[i * 3 + 0] = ox;
                sd[i * 3 + 1] = oy;
                sd[i * 3 + 2] = oz;
            }
            p4[i] = make_float4(__fadd_rn(x, ox), __fadd_rn(y, oy), __fadd_rn(z, oz), 0.0f);  // gnn_dyn.py:224
        }
    } else {
        for (int i = threadIdx.x; i < N; i += GRAPH_THREADS)
            p4[i] = make_float4(__fadd_rn(s[i * 3 + 0], sd[i * 3 + 0]), __fadd_rn(s[i * 3 + 1], sd[i * 3 + 1]),
                                __fadd_rn(s[i * 3 + 2], sd[i * 3 + 2]), 0.0f);
    }
    __syncthreads();

    const int i = chunk * GRAPH_THREADS + threadIdx.x;
    if (i >= N) return;
    const float4 pi = p4[i];
    float best[DRP_K];
#pragma unroll
    for (int q = 0; q < DRP_K; ++q) best[q] = thr;
    int j = 0;
    for (; j + 4 <= N; j += 4) {
        const float4 q0 = p4[j], q1 = p4[j + 1], q2 = p4[j + 2], q3 = p4[j + 3];
        const float d4[4] = {pair_dis(pi.x, pi.y, pi.z, q0.x, q0.y, q0.z), pair_dis(pi.x, pi.y, pi.z, q1.x, q1.y, q1.z),
                             pair_dis(pi.x, pi.y, pi.z, q2.x, q2.y, q2.z), pair_dis(pi.x, pi.y, pi.z, q3.x, q3.y, q3.z)};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = DRP_K - 1; q > 0; --q) best[q] = __builtin_amdgcn_fmed3f(d4[u], best[q - 1], best[q]);
            best[0] = min_nonneg(d4[u], best[0]);
        }
    }
    for (; j < N; ++j) {
        const float4 pj = p4[j];
        const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
#pragma unroll
        for (int q = DRP_K - 1; q > 0; --q) best[q] = __builtin_amdgcn_fmed3f(d, best[q - 1], best[q]);
        best[0] = min_nonneg(d, best[0]);
    }
    const float kth = best[DRP_K - 1];
    int16_t* out = nbr_idx + ((size_t)b * N + i) * DRP_K;
    int cnt = 0;
    // self_first (fused engine with a per-sample self-edge constant, km_prop): the self loop --
    // distance 0, inside any positive radius and never beyond the 10th smallest -- takes slot 0
    // and the other senders follow in ascending index.  The sum over a receiver's edges does not
    // depend on their order; the C ABI's drp_build_graph keeps the reference's ascending order.
    const int skip = (self_first && thr > 0.0f) ? i : -1;
    if (skip >= 0) out[cnt++] = (int16_t)i;
    for (j = 0; j + 4 <= N; j += 4) {
        const float4 q0 = p4[j], q1 = p4[j + 1], q2 = p4[j + 2], q3 = p4[j + 3];
        const float d4[4] = {pair_dis(pi.x, pi.y, pi.z, q0.x, q0.y, q0.z), pair_dis(pi.x, pi.y, pi.z, q1.x, q1.y, q1.z),
                             pair_dis(pi.x, pi.y, pi.z, q2.x, q2.y, q2.z), pair_dis(pi.x, pi.y, pi.z, q3.x, q3.y, q3.z)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (d4[u] <= kth && __fsub_rn(d4[u], thr) < 0.0f && cnt < DRP_K && j + u != skip) out[cnt++] = (int16_t)(j + u);
    }
    for (; j < N; ++j) {
        const float4 pj = p4[j];
        const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
        if (d <= kth && __fsub_rn(d, thr) < 0.0f && cnt < DRP_K && j != skip) out[cnt++] = (int16_t)j;
    }
    nbr_cnt[(size_t)b * N + i] = (uint8_t)cnt;
    for (int q = cnt; q < DRP_K; ++q) out[q] = -1;
}
