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

// Neighbour lists, two phases per receiver (one thread each):
//   1. sweep all senders (one ds_read_b128 broadcast per sender), append the index of every
//      sender inside the radius to the thread's own candidate list in LDS.  Only in-radius
//      senders can become edges: adj = (dis - thr < 0) * topk, so the edge set is "the up to
//      10 nearest among the in-radius senders" -- the candidates (38 of 300 on average on the
//      0.4 x 0.4 workspace) are all that the top-k selection has to look at.
//   2. sorted insertion of the candidates' (distance, index) pairs into ten registers, then
//      a small sorting network puts the survivors in ascending sender index.
// A thread whose candidate list overflows (very dense piles) falls back to the exact
// two-sweep selection over all senders.  Distances are recomputed with the same
// expression in both phases, so the selection is bit-identical to the one-phase form.
// grid = B * ceil(N / GRAPH_THREADS): a workgroup owns GRAPH_THREADS consecutive receivers of one
// sample (small workgroups, 21 KB of LDS at N = 300: seven per CU, no occupancy tail);
// every workgroup stages all N displaced positions, the first one of a sample writes s_delta.
// dynamic LDS = 4*N floats + GRAPH_THREADS * GRAPH_CAP int16.
#define GRAPH_CAP 64
#define GRAPH_THREADS 128

__device__ __forceinline__ void cswap_idx(int& a, int& b) {
    const int lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}

__global__ void __launch_bounds__(GRAPH_THREADS)
k_graph(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
        const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
        int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, DrpCam cam, float thr, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* p4 = reinterpret_cast<float4*>(lds);                       // [N] displaced positions
    int16_t* cand = reinterpret_cast<int16_t*>(lds + 4 * N);           // [GRAPH_CAP][blockDim]
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int BLOCK = GRAPH_THREADS;
    const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
    float* sd = s_delta + (size_t)b * N * 3;

    if (actions != nullptr) {
        const PushFrame f = push_frame(cam, actions + (size_t)b * act_stride);
        for (int i = threadIdx.x; i < N; i += BLOCK) {
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
        for (int i = threadIdx.x; i < N; i += BLOCK)
            p4[i] = make_float4(__fadd_rn(s[i * 3 + 0], sd[i * 3 + 0]), __fadd_rn(s[i * 3 + 1], sd[i * 3 + 1]),
                                __fadd_rn(s[i * 3 + 2], sd[i * 3 + 2]), 0.0f);
    }
    __syncthreads();

    {
        const int i = chunk * GRAPH_THREADS + threadIdx.x;
        if (i >= N) return;
        const float4 pi = p4[i];
        // phase 1: candidates inside the radius (four senders per trip: the LDS reads are issued
        // together, the appends stay in ascending order)
        int nc = 0;
        int j = 0;
        for (; j + 4 <= N; j += 4) {
            const float4 q0 = p4[j], q1 = p4[j + 1], q2 = p4[j + 2], q3 = p4[j + 3];
            const float d0 = pair_dis(pi.x, pi.y, pi.z, q0.x, q0.y, q0.z);
            const float d1 = pair_dis(pi.x, pi.y, pi.z, q1.x, q1.y, q1.z);
            const float d2 = pair_dis(pi.x, pi.y, pi.z, q2.x, q2.y, q2.z);
            const float d3 = pair_dis(pi.x, pi.y, pi.z, q3.x, q3.y, q3.z);
            if (__fsub_rn(d0, thr) < 0.0f) { if (nc < GRAPH_CAP) cand[nc * BLOCK + threadIdx.x] = (int16_t)j; ++nc; }
            if (__fsub_rn(d1, thr) < 0.0f) { if (nc < GRAPH_CAP) cand[nc * BLOCK + threadIdx.x] = (int16_t)(j + 1); ++nc; }
            if (__fsub_rn(d2, thr) < 0.0f) { if (nc < GRAPH_CAP) cand[nc * BLOCK + threadIdx.x] = (int16_t)(j + 2); ++nc; }
            if (__fsub_rn(d3, thr) < 0.0f) { if (nc < GRAPH_CAP) cand[nc * BLOCK + threadIdx.x] = (int16_t)(j + 3); ++nc; }
        }
        for (; j < N; ++j) {
            const float4 pj = p4[j];
            const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
            if (__fsub_rn(d, thr) < 0.0f) {
                if (nc < GRAPH_CAP) cand[nc * BLOCK + threadIdx.x] = (int16_t)j;
                ++nc;
            }
        }
        float bd[DRP_K];
        int bj[DRP_K];
#pragma unroll
        for (int q = 0; q < DRP_K; ++q) { bd[q] = __builtin_inff(); bj[q] = 0x7fff; }
        if (nc <= GRAPH_CAP) {
            // phase 2: ten nearest candidates, ties to the lower index (candidates come in ascending j)
            for (int c = 0; c < nc; ++c) {
                const int j = cand[c * BLOCK + threadIdx.x];
                const float4 pj = p4[j];
                const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
                if (d < bd[DRP_K - 1]) {
#pragma unroll
                    for (int q = DRP_K - 1; q > 0; --q) {
                        const bool up = d < bd[q - 1];            // everything from q-1 on shifts up
                        const bool here = !up && d < bd[q];
                        bj[q] = up ? bj[q - 1] : (here ? j : bj[q]);
                        bd[q] = up ? bd[q - 1] : (here ? d : bd[q]);
                    }
                    if (d < bd[0]) { bd[0] = d; bj[0] = j; }
                }
            }
        } else {
            // exact fallback: 10 smallest in-radius distances, then ascending index sweep
            for (int j = 0; j < N; ++j) {
                const float4 pj = p4[j];
                const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
                if (__fsub_rn(d, thr) < 0.0f && d < bd[DRP_K - 1]) {
#pragma unroll
                    for (int q = DRP_K - 1; q > 0; --q) bd[q] = (d < bd[q - 1]) ? bd[q - 1] : fminf(bd[q], d);
                    bd[0] = fminf(bd[0], d);
                }
            }
            const float kth = bd[DRP_K - 1];
            int cnt2 = 0;
            for (int j = 0; j < N; ++j) {
                const float4 pj = p4[j];
                const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
                if (d <= kth && __fsub_rn(d, thr) < 0.0f && cnt2 < DRP_K) {
#pragma unroll
                    for (int q = 0; q < DRP_K; ++q)
                        if (q == cnt2) bj[q] = j;
                    ++cnt2;
                }
            }
        }
        const int cnt = min(nc, DRP_K);
        // ascending sender index (empty slots hold 0x7fff and sink to the end): 10-input network
        cswap_idx(bj[0], bj[1]); cswap_idx(bj[2], bj[3]); cswap_idx(bj[4], bj[5]); cswap_idx(bj[6], bj[7]); cswap_idx(bj[8], bj[9]);
        cswap_idx(bj[0], bj[2]); cswap_idx(bj[1], bj[3]); cswap_idx(bj[4], bj[6]); cswap_idx(bj[5], bj[7]);
        cswap_idx(bj[1], bj[2]); cswap_idx(bj[5], bj[6]); cswap_idx(bj[0], bj[4]); cswap_idx(bj[3], bj[7]);
        cswap_idx(bj[1], bj[5]); cswap_idx(bj[2], bj[6]);
        cswap_idx(bj[1], bj[4]); cswap_idx(bj[3], bj[6]);
        cswap_idx(bj[2], bj[4]); cswap_idx(bj[3], bj[5]);
        cswap_idx(bj[3], bj[4]);
        // bj[0..7] sorted; merge the sorted pair (8,9) in by insertion
#pragma unroll
        for (int e = 8; e < DRP_K; ++e)
#pragma unroll
            for (int q = e; q > 0; --q) cswap_idx(bj[q - 1], bj[q]);
        int16_t* out = nbr_idx + ((size_t)b * N + i) * DRP_K;
#pragma unroll
        for (int q = 0; q < DRP_K; ++q) out[q] = (q < cnt) ? (int16_t)bj[q] : (int16_t)-1;
        nbr_cnt[(size_t)b * N + i] = (uint8_t)cnt;
    }
}
