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
#include <type_traits>
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
    f.len = drp_sqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz)));
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
DRP_GLOBAL void k_sdelta(const float* __restrict__ s_cur, const float* __restrict__ action, int N,
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

// Work items of unequal weight (the chunks of a sample: the last one is partly empty).  The hardware hands workgroups
// out statically -- id mod 8 picks the XCD, (id / 8) mod 4 one of its four shader-engine queues -- so with 2, 4 or 8
// chunks per sample all the light ones land on the same queues and the heavy ones share the rest: measured 37 - 45 %
// on k_graph_cells at 4 chunks per sample against 3 or 5, and on k_graph_strips at 400 particles (4 chunks) against
// 350 (3).  Queue q takes the q-th thirty-second of the item list instead: grids are rounded up to a multiple of 32.
#define SPREAD_GRID(n_items) (((n_items) + 31) & ~31)
__device__ __forceinline__ int spread_item() {
    return ((int)blockIdx.x & 31) * ((int)gridDim.x >> 5) + ((int)blockIdx.x >> 5);
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
//   2. senders in ascending index with d - thr < 0 and d < list[9], plus the senders AT list[9]
//      while slots are left (ties at the cut go to the lower index), are the edges.
// Both sweeps evaluate the distance with the same expression, so the result is exactly
// the reference's radius AND top-10 mask.
// grid = B * ceil(N / GRAPH_THREADS): a workgroup owns GRAPH_THREADS consecutive receivers of
// one sample and stages all N displaced positions (16 B each) in LDS; the first workgroup
// of a sample writes s_delta.  dynamic LDS = 4*N floats.
#ifndef GRAPH_THREADS
#define GRAPH_THREADS 128
#endif

// one receiver: both sweeps over the N displaced positions of its sample in `p4` (LDS), list to out[10] / cnt_out
__device__ __forceinline__ void graph_receiver(const float4* __restrict__ p4, int N, int i, float thr, int self_first,
                                               int16_t* __restrict__ out, uint8_t* __restrict__ cnt_out) {
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
    int cnt = 0;
    // self_first (fused engine with a per-sample self-edge constant, km_prop): the self loop --
    // distance 0, inside any positive radius and never beyond the 10th smallest -- takes slot 0
    // and the other senders follow in ascending index.  The sum over a receiver's edges does not
    // depend on their order; the C ABI's drp_build_graph keeps the reference's ascending order.
    const int skip = (self_first && thr > 0.0f) ? i : -1;
    // every sender strictly nearer than kth is in; the senders AT kth share the slots that leaves, lowest index
    // first (exact ties beyond the one that defines kth need coincident particles)
    int ties_left = DRP_K;
#pragma unroll
    for (int q = 0; q < DRP_K - 1; ++q) ties_left -= (best[q] < kth) ? 1 : 0;
    if (skip >= 0) {
        out[cnt++] = (int16_t)i;
        if (!(0.0f < kth)) --ties_left;              // the self loop (distance 0) is itself one of the ties
    }
    // One flat predicate per sender and ONE predicated store: written as nested ifs (in radius and not beyond kth / slots left /
    // not the self loop / strictly nearer, else a tie with budget left) the sweep compiled to five levels of exec-mask
    // branches that a wave walks for nearly every sender as soon as a few lanes are active -- 11.5 us of a 106-us rollout
    // step at 50 particles.  The same decisions, the same lists.
    auto visit = [&](float d, int js) {
        const bool in = d <= kth && __fsub_rn(d, thr) < 0.0f && cnt < DRP_K && js != skip;
        const bool strict = d < kth;
        const bool take = in && (strict || ties_left > 0);
        if (take) out[cnt] = (int16_t)js;
        ties_left -= (take && !strict) ? 1 : 0;
        cnt += take ? 1 : 0;
    };
    for (j = 0; j + 4 <= N; j += 4) {
        const float4 q0 = p4[j], q1 = p4[j + 1], q2 = p4[j + 2], q3 = p4[j + 3];
        const float d4[4] = {pair_dis(pi.x, pi.y, pi.z, q0.x, q0.y, q0.z), pair_dis(pi.x, pi.y, pi.z, q1.x, q1.y, q1.z),
                             pair_dis(pi.x, pi.y, pi.z, q2.x, q2.y, q2.z), pair_dis(pi.x, pi.y, pi.z, q3.x, q3.y, q3.z)};
#pragma unroll
        for (int u = 0; u < 4; ++u) visit(d4[u], j + u);
    }
    for (; j < N; ++j) {
        const float4 pj = p4[j];
        visit(pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z), j);
    }
    *cnt_out = (uint8_t)cnt;
    for (int q = cnt; q < DRP_K; ++q) out[q] = -1;
}

// ---- reversed neighbour lists of ONE sample by its workgroup of T threads: for every sender j the edge slots (i*10 + k)
// it feeds, ascending (kb_reverse_lists, k_backward.h, is the launch of its own; k_graph_rev below runs it behind the
// lists' construction).  Out-degrees counted and scanned in LDS; the lists are filled and put in order in LDS too when they
// fit (s_rev: 2*N ints, + 10*N ints with `in_lds`).
template <int T>
__device__ __forceinline__ void reverse_lists(const int16_t* nb, const uint8_t* nc /* may have been written by this workgroup just before (k_graph_rev) */, int N,
                                              int* __restrict__ ro /* [N+1] */, int* __restrict__ rv /* [N*10] */, int in_lds,
                                              int n_recv, int* s_rev) {
    __shared__ int s_w[T / 64];
    int* deg = s_rev;
    int* off = s_rev + N;
    const int tid = threadIdx.x;
    int* fill = in_lds ? s_rev + 2 * N : rv;
    for (int i = tid; i < N; i += T) deg[i] = 0;
    __syncthreads();
    for (int e = tid; e < N * DRP_K; e += T) {
        const int i = e / DRP_K, k = e - i * DRP_K;
        if (k < nc[i] && i < n_recv) atomicAdd(&deg[nb[e]], 1);
    }
    __syncthreads();
    // exclusive scan of deg: every thread owns a contiguous segment
    const int seg = (N + T - 1) / T;
    const int lo = min(tid * seg, N), hi = min(lo + seg, N);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += deg[i];
    int inc = sum;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = inc - sum, total = 0;
    for (int w = 0; w < T / 64; ++w) {
        if (w < wave) base += s_w[w];
        total += s_w[w];
    }
    for (int i = lo; i < hi; ++i) { off[i] = base; base += deg[i]; }
    __syncthreads();
    for (int i = tid; i < N; i += T) { ro[i] = off[i]; deg[i] = 0; }
    if (tid == 0) ro[N] = total;
    __syncthreads();
    for (int e = tid; e < N * DRP_K; e += T) {
        const int i = e / DRP_K, k = e - i * DRP_K;
        if (k < nc[i] && i < n_recv) {
            const int j = nb[e];
            fill[off[j] + atomicAdd(&deg[j], 1)] = e;
        }
    }
    __syncthreads();
    for (int j = tid; j < N; j += T) {  // fixed order inside every sender's list
        int* seg_j = fill + off[j];
        const int n = deg[j];
        for (int a = 1; a < n; ++a) {
            const int v = seg_j[a];
            int c = a - 1;
            while (c >= 0 && seg_j[c] > v) { seg_j[c + 1] = seg_j[c]; --c; }
            seg_j[c + 1] = v;
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int p = tid; p < total; p += T) rv[p] = fill[p];
    }
}

template <bool REV>
__device__ __forceinline__ void graph_sample(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
        const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
        int16_t* nbr_idx, uint8_t* nbr_cnt, DrpCam cam, float thr, int chunks,
        int n_items /* B * chunks; grid = SPREAD_GRID(n_items) */, int self_first, int* rev_off, int* rev) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* p4 = reinterpret_cast<float4*>(lds);                       // [N] displaced positions
    const int item = spread_item();
    if (item >= n_items) return;
    const int b = item / chunks, chunk = item - b * chunks;
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
    if (i < N) graph_receiver(p4, N, i, thr, self_first, nbr_idx + ((size_t)b * N + i) * DRP_K, nbr_cnt + (size_t)b * N + i);
    if (REV) {
        // the sample's reversed lists behind its lists, by the same workgroup (one chunk: N <= GRAPH_THREADS): the GD
        // planner's backward pass wants them, and a launch of their own costs more than the work (9.5 us at 20 particles)
        __threadfence_block();
        __syncthreads();                                               // the lists are written; the positions in LDS are dead
        reverse_lists<GRAPH_THREADS>(nbr_idx + (size_t)b * N * DRP_K, nbr_cnt + (size_t)b * N, N, rev_off + (size_t)b * (N + 1),
                                     rev + (size_t)b * N * DRP_K, 1, N, reinterpret_cast<int*>(lds));
    }
}
DRP_GLOBAL void __launch_bounds__(GRAPH_THREADS)
k_graph(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
        const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
        int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, DrpCam cam, float thr, int chunks,
        int n_items /* B * chunks; grid = SPREAD_GRID(n_items) */, int self_first) {
    graph_sample<false>(s_prev, prev_mod, prev_stride, actions, act_stride, s_delta, N, nbr_idx, nbr_cnt, cam, thr, chunks, n_items,
                        self_first, nullptr, nullptr);
}
// dynamic LDS: 12 * N ints (>= the 4 * N floats of the positions)
DRP_GLOBAL void __launch_bounds__(GRAPH_THREADS)
k_graph_rev(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
            const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
            int16_t* nbr_idx, uint8_t* nbr_cnt, DrpCam cam, float thr, int n_items /* B: one chunk per sample */, int self_first,
            int* __restrict__ rev_off /* [B][N+1] */, int* __restrict__ rev /* [B][N*10] */) {
    graph_sample<true>(s_prev, prev_mod, prev_stride, actions, act_stride, s_delta, N, nbr_idx, nbr_cnt, cam, thr, 1, n_items,
                       self_first, rev_off, rev);
}

// ---- the plain sweep for a HANDFUL of samples: four threads per receiver -------------------------------------------
// k_graph gives a receiver one thread that walks all N senders twice.  With a batch of thousands of samples that is what
// fills the chip; with a training batch (4 samples of 300 - 1000 particles: a dozen workgroups) it is one long serial walk
// per thread on an otherwise empty chip -- 57 us per rollout step at 300 particles, 144 us at 1000, a sixth of a training
// iteration.  Here a workgroup of 512 threads owns the same 128 receivers and thread (r, q) walks only quarter q of the
// senders (a quarter is wave-uniform: the reads stay broadcasts).  The four partial lists of ten meet in LDS:
//   1. ten smallest in-radius distances of the own quarter (graph_receiver's first sweep)               -> LDS
//   2. every thread of a receiver merges the four lists (30 insertions): kth and the tie budget
//   3. own quarter again: senders strictly nearer than kth, and (separately, up to ten) senders AT kth    -> LDS
//   4. thread q = 0 walks the quarters in order, merging each quarter's two ascending lists, under graph_receiver's own
//      emission rule (strictly nearer: in; at kth: while the tie budget lasts; never more than ten) -- the quarters are
//      ascending index ranges, so the walk meets the senders in ascending index, as the single sweep does.
// The same lists as k_graph for every input, coincident particles and lattices included (tests/test_gpu_graph_strips.py).
#define GRAPH_Q4_THREADS 512
#define GRAPH_Q4_LDS(N) ((size_t)4 * ((N) + 3) * sizeof(float) + (size_t)128 * 4 * 10 * sizeof(float) + (size_t)128 * 4 * 22 * sizeof(int16_t))
// (the body of workgroup `blk`: k_graph_q4 below, and km_graph_q4_encode of k_rollout.h, where the particle encoder's tiles
// share the launch)
__device__ __forceinline__ void
graph_q4_block(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
               const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
               int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, const DrpCam& cam, float thr, int chunks, int self_first,
               int blk, float* lds) {
    float4* p4 = reinterpret_cast<float4*>(lds);                       // [N] displaced positions
    float* bestq = lds + 4 * ((N + 3) & ~3);                           // [128][4][10]
    int16_t* listq = reinterpret_cast<int16_t*>(bestq + 128 * 4 * 10); // [128][4][22]: 10 strict, 10 ties, 2 counts
    const int b = blk / chunks, chunk = blk - b * chunks;
    const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
    float* sd = s_delta + (size_t)b * N * 3;
    if (actions != nullptr) {
        const PushFrame f = push_frame(cam, actions + (size_t)b * act_stride);
        for (int i = threadIdx.x; i < N; i += GRAPH_Q4_THREADS) {
            float x = s[i * 3 + 0], y = s[i * 3 + 1], z = s[i * 3 + 2];
            float ox, oy, oz;
            push_delta(f, x, y, z, ox, oy, oz);
            if (chunk == 0) { sd[i * 3 + 0] = ox; sd[i * 3 + 1] = oy; sd[i * 3 + 2] = oz; }
            p4[i] = make_float4(__fadd_rn(x, ox), __fadd_rn(y, oy), __fadd_rn(z, oz), 0.0f);
        }
    } else {
        for (int i = threadIdx.x; i < N; i += GRAPH_Q4_THREADS)
            p4[i] = make_float4(__fadd_rn(s[i * 3 + 0], sd[i * 3 + 0]), __fadd_rn(s[i * 3 + 1], sd[i * 3 + 1]),
                                __fadd_rn(s[i * 3 + 2], sd[i * 3 + 2]), 0.0f);
    }
    __syncthreads();
    const int r = threadIdx.x & 127, q = threadIdx.x >> 7;
    const int i = chunk * 128 + r;
    const bool have = i < N;
    const int nq = (((N + 3) >> 2) + 3) & ~3;                          // senders per quarter, a multiple of 4
    const int lo = min(q * nq, N), hi = min(lo + nq, N);
    const float4 pi = p4[have ? i : 0];
    float best[DRP_K];
#pragma unroll
    for (int t = 0; t < DRP_K; ++t) best[t] = thr;
    {
        int j = lo;
        for (; j + 4 <= hi; j += 4) {
            const float4 q0 = p4[j], q1 = p4[j + 1], q2 = p4[j + 2], q3 = p4[j + 3];
            const float d4[4] = {pair_dis(pi.x, pi.y, pi.z, q0.x, q0.y, q0.z), pair_dis(pi.x, pi.y, pi.z, q1.x, q1.y, q1.z),
                                 pair_dis(pi.x, pi.y, pi.z, q2.x, q2.y, q2.z), pair_dis(pi.x, pi.y, pi.z, q3.x, q3.y, q3.z)};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = DRP_K - 1; t > 0; --t) best[t] = __builtin_amdgcn_fmed3f(d4[u], best[t - 1], best[t]);
                best[0] = min_nonneg(d4[u], best[0]);
            }
        }
        for (; j < hi; ++j) {
            const float4 pj = p4[j];
            const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
#pragma unroll
            for (int t = DRP_K - 1; t > 0; --t) best[t] = __builtin_amdgcn_fmed3f(d, best[t - 1], best[t]);
            best[0] = min_nonneg(d, best[0]);
        }
    }
    float* mine = bestq + (r * 4 + q) * 10;
#pragma unroll
    for (int t = 0; t < DRP_K; ++t) mine[t] = best[t];
    __syncthreads();
    // 2. the other three quarters' lists into this one (every thread of the receiver: no broadcast needed afterwards)
#pragma unroll
    for (int o = 1; o < 4; ++o) {
        const float* other = bestq + (r * 4 + ((q + o) & 3)) * 10;
#pragma unroll
        for (int t2 = 0; t2 < DRP_K; ++t2) {
            const float d = other[t2];
#pragma unroll
            for (int t = DRP_K - 1; t > 0; --t) best[t] = __builtin_amdgcn_fmed3f(d, best[t - 1], best[t]);
            best[0] = min_nonneg(d, best[0]);
        }
    }
    const float kth = best[DRP_K - 1];
    const int skip = (self_first && thr > 0.0f) ? i : -1;
    // 3. the own quarter's senders: strictly nearer than kth / at kth (the first ten of either are all that can matter)
    int16_t* ls = listq + (r * 4 + q) * 22;
    int ns = 0, nt = 0;
    for (int j = lo; j < hi; ++j) {
        const float4 pj = p4[j];
        const float d = pair_dis(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z);
        if (d <= kth && __fsub_rn(d, thr) < 0.0f && j != skip) {
            if (d < kth) { if (ns < DRP_K) ls[ns++] = (int16_t)j; }
            else if (nt < DRP_K) ls[10 + nt++] = (int16_t)j;
        }
    }
    ls[20] = (int16_t)ns;
    ls[21] = (int16_t)nt;
    __syncthreads();
    if (q != 0 || !have) return;
    // 4. graph_receiver's emission rule over the four quarters in order
    int16_t* out = nbr_idx + ((size_t)b * N + i) * DRP_K;
    int cnt = 0;
    int ties_left = DRP_K;
#pragma unroll
    for (int t = 0; t < DRP_K - 1; ++t) ties_left -= (best[t] < kth) ? 1 : 0;
    if (skip >= 0) {
        out[cnt++] = (int16_t)i;
        if (!(0.0f < kth)) --ties_left;
    }
    for (int qq = 0; qq < 4; ++qq) {
        const int16_t* l = listq + (r * 4 + qq) * 22;
        const int n_s = l[20], n_t = l[21];
        int a = 0, t = 0;
        while ((a < n_s || t < n_t) && cnt < DRP_K) {
            const bool take_s = a < n_s && (t >= n_t || l[a] < l[10 + t]);
            if (take_s) out[cnt++] = l[a++];
            else {
                if (ties_left > 0) { out[cnt++] = l[10 + t]; --ties_left; }
                ++t;
            }
        }
    }
    nbr_cnt[(size_t)b * N + i] = (uint8_t)cnt;
    for (int t = cnt; t < DRP_K; ++t) out[t] = -1;
}
DRP_GLOBAL void __launch_bounds__(GRAPH_Q4_THREADS)
k_graph_q4(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
           const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
           int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, DrpCam cam, float thr, int chunks, int self_first) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    graph_q4_block(s_prev, prev_mod, prev_stride, actions, act_stride, s_delta, N, nbr_idx, nbr_cnt, cam, thr, chunks, self_first,
                   (int)blockIdx.x, lds);
}

// ---- the same lists with the senders bucketed into x strips --------------------------------------------
// An edge needs |dx| < radius, so a receiver only has to look at senders whose x is within the radius of
// its own.  k_graph_sort (one workgroup per sample) writes the sample's displaced positions into 64 fixed
// strips of x (counting sort: histogram, scan, scatter; 1 cm wide in the camera frame, the ends clamped)
// and k_graph_strips hands a workgroup 128 CONSECUTIVE receivers of that order, so the receivers of a wave sit
// in a few neighbouring strips and the wave sweeps one contiguous range of the sorted senders: the strips of
// its first and last receiver widened by the radius (+ one strip against rounding).  The range is
// wave-uniform, so the positions still arrive by broadcast LDS reads (per-lane windows do not pay, DESIGN.md
// 9b), and a workgroup stages only the range its two waves need.  Senders outside the range are farther than
// the radius in x alone: (dis - thr < 0) is false for them and, being farther than every in-radius sender,
// they cannot displace one from the ten nearest -- the result is exactly k_graph's.
//   uniform piles over the +-0.2 workspace: 66 % of the pairs at N = 300, 50 % at N = 1200; a pile narrower
//   than two radii gains nothing and pays the sort.
// The order inside a strip is whatever the scatter's atomics produced (one order per sample: that is why the
// sort is its own launch and not redone by every workgroup); nothing depends on it.  The sweeps meet the
// senders in strip order, not index order: the (at most ten) chosen indices are staged in LDS and leave
// through a sorting network in ascending index, as k_graph emits them.  The 10th-nearest distance itself
// always ties with kth; a SECOND sender at exactly that distance (coincident particles, e.g. zero-padded
// training rows) takes the slow path that picks the remaining lowest indices one sweep at a time -- the host
// keeps k_graph for padded batches.
#define GRAPH_STRIPS 64
#define GRAPH_STRIP_X0 (-0.32f)
#define GRAPH_STRIP_INV_W 100.0f
#define GRAPH_SORT_THREADS 256

__device__ __forceinline__ int graph_strip(float x) {
    // any monotone map of x is valid; NaN lands in strip 0
    const float t = __fmul_rn(__fsub_rn(x, GRAPH_STRIP_X0), GRAPH_STRIP_INV_W);
    return (int)fminf(fmaxf(t, 0.0f), (float)(GRAPH_STRIPS - 1));
}

__device__ __forceinline__ void sort2(int& a, int& b) {
    const int lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
}

// sorted[b][Np] = (x, y, z, index) in strip order (Np = N rounded up to 4, the padding 1e18 away),
// starts[b][GRAPH_STRIPS + 1] = first slot of every strip; also writes s_delta when the push is given.
DRP_GLOBAL void __launch_bounds__(GRAPH_SORT_THREADS)
k_graph_sort(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
             const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
             DrpCam cam, float4* __restrict__ sorted, int* __restrict__ starts) {
    __shared__ int cursor[GRAPH_STRIPS];
    const int b = blockIdx.x, Np = (N + 3) & ~3;
    const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
    float* sd = s_delta + (size_t)b * N * 3;
    float4* q4 = sorted + (size_t)b * Np;
    int* sstart = starts + (size_t)b * (GRAPH_STRIPS + 1);
    PushFrame f = {};
    if (actions != nullptr) f = push_frame(cam, actions + (size_t)b * act_stride);
    auto displaced = [&](int i, float& x, float& y, float& z, bool write_delta) {
        const float sx = s[i * 3 + 0], sy = s[i * 3 + 1], sz = s[i * 3 + 2];
        float ox, oy, oz;
        if (actions != nullptr) {
            push_delta(f, sx, sy, sz, ox, oy, oz);
            if (write_delta) {
                sd[i * 3 + 0] = ox;
                sd[i * 3 + 1] = oy;
                sd[i * 3 + 2] = oz;
            }
        } else {
            ox = sd[i * 3 + 0]; oy = sd[i * 3 + 1]; oz = sd[i * 3 + 2];
        }
        x = __fadd_rn(sx, ox);                                    // gnn_dyn.py:224
        y = __fadd_rn(sy, oy);
        z = __fadd_rn(sz, oz);
    };
    if (threadIdx.x < GRAPH_STRIPS) cursor[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += GRAPH_SORT_THREADS) {
        float x, y, z;
        displaced(i, x, y, z, false);
        atomicAdd(&cursor[graph_strip(x)], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                       // exclusive scan of the 64 counts by the first wave
        const int cnt = cursor[threadIdx.x];
        int v = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(v, o, 64);
            if ((int)threadIdx.x >= o) v += u;
        }
        sstart[threadIdx.x + 1] = v;
        if (threadIdx.x == 0) sstart[0] = 0;
        cursor[threadIdx.x] = v - cnt;
    }
    if ((int)threadIdx.x < Np - N) q4[N + threadIdx.x] = make_float4(1e18f, 0.0f, 0.0f, __int_as_float(-1));
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += GRAPH_SORT_THREADS) {
        float x, y, z;
        displaced(i, x, y, z, true);
        const int slot = atomicAdd(&cursor[graph_strip(x)], 1);
        q4[slot] = make_float4(x, y, z, __int_as_float(i));
    }
}

// T receivers per workgroup: 128, or 256 for large samples (fewer workgroups stage overlapping ranges: graph build
// 6.0 -> 5.1 ms per iteration at 1200 particles, but 2 % slower at 300)
#define GS_LIST 11                                                 // ten entries + the slot a full list keeps overwriting
#define GRAPH_STRIPS_LDS(N, T) ((size_t)((((N) + 3) & ~3) + 4) * 16 + (GRAPH_STRIPS + 1) * 4 + (T) * GS_LIST * 2 + 16)

// ---- k_graph_strips_q: the sweep over x strips, ranges per QUARTER wave (the wave-wide variant of round 2, 0.715 vs 0.674 ms per
// iteration at 300 particles, is gone with its switch) ----
template <int T>
__global__ void __launch_bounds__(T)
k_graph_strips_q(const float4* __restrict__ sorted, const int* __restrict__ starts, int N,
               int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, float thr, int chunks,
               int n_items /* B * chunks; grid = SPREAD_GRID(n_items) */, int self_first) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Np = (N + 3) & ~3;
    float4* q4 = reinterpret_cast<float4*>(lds);                 // the staged range of the sorted senders
    int* sstart = reinterpret_cast<int*>(q4 + Np + 4);           // [GRAPH_STRIPS + 1]; q4[Np .. Np + 3]: the sentinel block
    int16_t* lst = reinterpret_cast<int16_t*>(sstart + GRAPH_STRIPS + 1);   // [T][GS_LIST] chosen indices, unsorted
    const int item = spread_item();
    if (item >= n_items) return;
    const int b = item / chunks, chunk = item - b * chunks;
    const float4* g4 = sorted + (size_t)b * Np;
    if (threadIdx.x <= GRAPH_STRIPS) sstart[threadIdx.x] = starts[(size_t)b * (GRAPH_STRIPS + 1) + threadIdx.x];
    const int s_first = chunk * T;
    const int s_last = min(s_first + T, N) - 1;
    const float radius = __fsqrt_rn(fmaxf(thr, 0.0f)) * 1.000001f;
    const int reach = (int)fminf(ceilf(radius * GRAPH_STRIP_INV_W), (float)GRAPH_STRIPS) + 1;
    // the workgroup's range: strips of its first and last receiver (the order is by strip), widened by the radius
    const int wg_smin = graph_strip(g4[s_first].x), wg_smax = graph_strip(g4[s_last].x);
    __syncthreads();
    const int wlo = sstart[max(wg_smin - reach, 0)] & ~3;
    const int whi = (sstart[min(wg_smax + reach, GRAPH_STRIPS - 1) + 1] + 3) & ~3;
    for (int j = wlo + (int)threadIdx.x; j < whi; j += T) q4[j] = g4[j];
    if (threadIdx.x < 4) q4[Np + threadIdx.x] = make_float4(1e18f, 0.0f, 0.0f, __int_as_float(-1));   // what a lane past its range reads
    __syncthreads();

    const int si = s_first + threadIdx.x;
    const bool valid = si < N;
    const unsigned long long act = __ballot(valid);
    if (act == 0) return;
    const float4 pi = q4[valid ? si : s_last];                    // a receiver is inside its own workgroup's range
    const int i = __float_as_int(pi.w);
    const int my_strip = graph_strip(pi.x);
    // QUARTER-wave ranges: the sixteen lanes of a DPP row are sixteen consecutive receivers of the order, and their strips
    // (+ halo) are a quarter of the span the wave's 64 cover; a read of four different addresses, each shared by the
    // sixteen lanes of a row, costs the LDS what one broadcast costs.  Invalid lanes carry the last receiver's strip.
    const int lane = threadIdx.x & 63;
    const int smin = __shfl(my_strip, lane & ~15, 64);
    const int smax = __shfl(my_strip, lane | 15, 64);
    auto wave_max4 = [&](int v) {                          // v is uniform over each row
        return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                   max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
    };
    auto row_max = [&](float v) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        return v;
    };
    const int jlo = max(sstart[max(smin - reach, 0)] & ~3, wlo);
    const int jhi = min((sstart[min(smax + reach, GRAPH_STRIPS - 1) + 1] + 3) & ~3, whi);

    float best[DRP_K];
#pragma unroll
    for (int q = 0; q < DRP_K; ++q) best[q] = thr;
    auto sweep1 = [&](int ja, int jb) {                   // [ja, jb): this row's range, a multiple of four long
        const int trip = wave_max4(jb - ja);
        for (int t = 0; t < trip; t += 4) {
            const int j = (ja + t < jb) ? ja + t : Np;
            const float4 q0 = q4[j], q1 = q4[j + 1], q2 = q4[j + 2], q3 = q4[j + 3];
            const float d4[4] = {pair_dis(pi.x, pi.y, pi.z, q0.x, q0.y, q0.z), pair_dis(pi.x, pi.y, pi.z, q1.x, q1.y, q1.z),
                                 pair_dis(pi.x, pi.y, pi.z, q2.x, q2.y, q2.z), pair_dis(pi.x, pi.y, pi.z, q3.x, q3.y, q3.z)};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int q = DRP_K - 1; q > 0; --q) best[q] = __builtin_amdgcn_fmed3f(d4[u], best[q - 1], best[q]);
                best[0] = min_nonneg(d4[u], best[0]);
            }
        }
    };
    // first sweep in two stages: the strips within half the radius give a provisional 10th-nearest distance (an upper
    // bound of the final one: more candidates can only lower it); a sender that could still enter the list is nearer
    // than that in x, so the stage that follows only covers the strips within the wave's largest provisional distance
    // -- none at all in a dense pile.  The list of the ten smallest does not depend on the order of insertion.
    const int reach_a = (reach + 1) >> 1;
    const int ja = max(sstart[max(smin - reach_a, 0)] & ~3, jlo);
    const int jb = min((sstart[min(smax + reach_a, GRAPH_STRIPS - 1) + 1] + 3) & ~3, jhi);
    sweep1(ja, jb);
    {
        const float kprov = row_max(valid ? best[DRP_K - 1] : 0.0f);
        const int reach_b = min(reach, (int)ceilf(__fsqrt_rn(fmaxf(kprov, 0.0f)) * 1.000001f * GRAPH_STRIP_INV_W) + 1);
        const bool ring = reach_b > reach_a;               // per row
        const int jl = ring ? max(sstart[max(smin - reach_b, 0)] & ~3, jlo) : ja;
        const int jh = ring ? min((sstart[min(smax + reach_b, GRAPH_STRIPS - 1) + 1] + 3) & ~3, jhi) : jb;
        if (wave_max4(ring ? 1 : 0)) {
            sweep1(jl, ja);
            sweep1(jb, jh);
        }
    }
    const float kth = best[DRP_K - 1];
    // second sweep: senders strictly nearer than kth are in; the ones AT kth fill what is left, lowest index first.
    // Nothing farther than sqrt(kth) is emitted, so the sweep narrows to the strips within the wave's largest
    // 10th-nearest distance (dense piles: a quarter of the radius)
    const float kmax = row_max(valid ? kth : 0.0f);
    const int reach2 = min(reach, (int)ceilf(__fsqrt_rn(fmaxf(kmax, 0.0f)) * 1.000001f * GRAPH_STRIP_INV_W) + 1);
    const int jlo2 = max(sstart[max(smin - reach2, 0)] & ~3, jlo);
    const int jhi2 = min((sstart[min(smax + reach2, GRAPH_STRIPS - 1) + 1] + 3) & ~3, jhi);
    const int skip = (self_first && thr > 0.0f) ? i : -1;
    // kth is the tenth smallest in-radius distance (or thr when there are fewer): at most nine senders are nearer, and
    // when no more than ten are at or below it every one of them is an edge -- the common case, one compare per
    // candidate: d <= kle with kle = kth below the radius, else the largest float under thr (d <= kle <=> d < thr).
    // The entry is written unconditionally and kept only if the count moves on.  More than ten (several senders at
    // exactly kth): the list is rebuilt by the exact rule.
    const float kle = (kth < thr) ? kth : (thr > 0.0f ? __int_as_float(__float_as_int(thr) - 1) : -1.0f);
    int16_t* mine = lst + threadIdx.x * GS_LIST;
    int cnt = 0;
    const int trip2 = wave_max4(jhi2 - jlo2);
    for (int t = 0; t < trip2; t += 4) {
        const int j = (jlo2 + t < jhi2) ? jlo2 + t : Np;
        const float4 q0 = q4[j], q1 = q4[j + 1], q2 = q4[j + 2], q3 = q4[j + 3];
        const float d4[4] = {pair_dis(pi.x, pi.y, pi.z, q0.x, q0.y, q0.z), pair_dis(pi.x, pi.y, pi.z, q1.x, q1.y, q1.z),
                             pair_dis(pi.x, pi.y, pi.z, q2.x, q2.y, q2.z), pair_dis(pi.x, pi.y, pi.z, q3.x, q3.y, q3.z)};
        const int o4[4] = {__float_as_int(q0.w), __float_as_int(q1.w), __float_as_int(q2.w), __float_as_int(q3.w)};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            mine[min(cnt, DRP_K)] = (int16_t)o4[u];
            cnt += (d4[u] <= kle) ? 1 : 0;
        }
    }
    if (cnt > DRP_K) {
        cnt = 0;
        for (int j = jlo2; j < jhi2; ++j) {
            const float4 q = q4[j];
            const float d = pair_dis(pi.x, pi.y, pi.z, q.x, q.y, q.z);
            const int o = __float_as_int(q.w);
            if (__fsub_rn(d, thr) < 0.0f && d < kth && o != skip && cnt < DRP_K) mine[cnt++] = (int16_t)o;
        }
        const int room = DRP_K - cnt - (skip >= 0 ? 1 : 0);
        int last = -1;
        for (int r = 0; r < room; ++r) {                         // the lowest indices among the senders at kth, one sweep each
            int nxt = 0x7fff;
            for (int j = jlo2; j < jhi2; ++j) {
                const float4 q = q4[j];
                const float d = pair_dis(pi.x, pi.y, pi.z, q.x, q.y, q.z);
                const int o = __float_as_int(q.w);
                if (__fsub_rn(d, thr) < 0.0f && d == kth && o != skip && o > last) nxt = min(nxt, o);
            }
            if (nxt == 0x7fff) break;
            mine[cnt++] = (int16_t)nxt;
            last = nxt;
        }
    }
    if (skip >= 0) {                                               // the receiver itself is slot 0 of the output, not an entry
        int at = -1;
#pragma unroll
        for (int q = 0; q < DRP_K; ++q)
            if (q < cnt && (int)mine[q] == skip) at = q;
        if (at >= 0) {
            mine[at] = mine[cnt - 1];
            --cnt;
        }
    }
    if (!valid) return;
    // ascending index through a 10-input sorting network (29 compare-exchanges); empty slots sort last
    int v[DRP_K];
#pragma unroll
    for (int q = 0; q < DRP_K; ++q) v[q] = (q < cnt) ? (int)mine[q] : 0x7fff;
#define CE(a, b) sort2(v[a], v[b])
    CE(0, 5); CE(1, 6); CE(2, 7); CE(3, 8); CE(4, 9);
    CE(0, 3); CE(1, 4); CE(5, 8); CE(6, 9);
    CE(0, 2); CE(3, 6); CE(7, 9);
    CE(0, 1); CE(2, 4); CE(5, 7); CE(8, 9);
    CE(1, 2); CE(3, 5); CE(4, 6); CE(7, 8);
    CE(1, 3); CE(2, 5); CE(4, 7); CE(6, 8);
    CE(2, 3); CE(4, 5); CE(6, 7);
    CE(3, 4); CE(5, 6);
#undef CE
    int16_t* out = nbr_idx + ((size_t)b * N + i) * DRP_K;
    int w = 0;
    if (skip >= 0) out[w++] = (int16_t)i;
#pragma unroll
    for (int q = 0; q < DRP_K; ++q)
        if (q < cnt && w < DRP_K) out[w++] = (int16_t)v[q];
    nbr_cnt[(size_t)b * N + i] = (uint8_t)w;
    for (int q = w; q < DRP_K; ++q) out[q] = -1;
}

// ---- the same lists with the senders bucketed into two-dimensional cells ------------------------------------
// What the sweep costs is receivers x candidates x (8 distance + 10 insertion) vector instructions, and the
// candidates of a lane are the senders of the region its GROUP sweeps.  With x strips that region is (strips of the
// wave's 64 receivers + halo) x the WHOLE y extent.  Here the positions are sorted into cells -- y bands of height hb
// (the host picks hb ~ sqrt(16 / density), so that 16 consecutive particles of a band span about hb in x too), each
// band ordered by the 1-cm x strip (k_graph_sort2) -- and a QUARTER wave (16 lanes, 16 consecutive receivers of the
// order: a compact block) sweeps its own region: for every band within the halo of its receivers' bands, the run of
// strips within the halo of their strips -- a contiguous run of the order per band.  A 16-receiver block + halo is a
// quarter of the area a 64-receiver strip range + halo x full height covers at 1 200 particles.
// Same two-stage first sweep (a halo of the expected 10th-nearest distance -> provisional 10th distance -> the ring
// that is still missing, swept as band runs minus the runs already done) and the same narrowed second sweep as
// k_graph_strips; runs are swept exactly (a sender met twice would take two places of the ten).  A quarter never
// straddles two bands (the end of one band and the start of the next are the two ENDS of the workspace in x: such a
// block would sweep the whole width): every band's receivers are dealt to quarters of their own, the last one of a
// band partly idle.  Lists identical to k_graph's.
#define GC_XS 64
#define GC_MAX_BANDS 32

__device__ __forceinline__ int graph_band(float y, float inv_hb, int gy) {
    const float t = __fmul_rn(__fsub_rn(y, GRAPH_STRIP_X0), inv_hb);
    return (int)fminf(fmaxf(t, 0.0f), (float)(gy - 1));
}

// sorted[b][Np] = (x, y, z, index) in cell order (band-major, strip-minor), starts[b][gy * 64 + 1]
DRP_GLOBAL void __launch_bounds__(GRAPH_SORT_THREADS)
k_graph_sort2(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride,
              const float* __restrict__ actions, size_t act_stride, float* __restrict__ s_delta, int N,
              DrpCam cam, int gy, float inv_hb, float4* __restrict__ sorted, int* __restrict__ starts) {
    __shared__ int cursor[GC_MAX_BANDS * GC_XS];
    __shared__ int s_w[GRAPH_SORT_THREADS / 64];
    const int b = blockIdx.x, Np = (N + 3) & ~3, ncell = gy * GC_XS;
    const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
    float* sd = s_delta + (size_t)b * N * 3;
    float4* q4 = sorted + (size_t)b * Np;
    int* sstart = starts + (size_t)b * (ncell + 1);
    PushFrame f = {};
    if (actions != nullptr) f = push_frame(cam, actions + (size_t)b * act_stride);
    auto displaced = [&](int i, float& x, float& y, float& z, bool write_delta) {
        const float sx = s[i * 3 + 0], sy = s[i * 3 + 1], sz = s[i * 3 + 2];
        float ox, oy, oz;
        if (actions != nullptr) {
            push_delta(f, sx, sy, sz, ox, oy, oz);
            if (write_delta) {
                sd[i * 3 + 0] = ox;
                sd[i * 3 + 1] = oy;
                sd[i * 3 + 2] = oz;
            }
        } else {
            ox = sd[i * 3 + 0]; oy = sd[i * 3 + 1]; oz = sd[i * 3 + 2];
        }
        x = __fadd_rn(sx, ox);                                    // gnn_dyn.py:224
        y = __fadd_rn(sy, oy);
        z = __fadd_rn(sz, oz);
    };
    for (int c = threadIdx.x; c < ncell; c += GRAPH_SORT_THREADS) cursor[c] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += GRAPH_SORT_THREADS) {
        float x, y, z;
        displaced(i, x, y, z, false);
        atomicAdd(&cursor[graph_band(y, inv_hb, gy) * GC_XS + graph_strip(x)], 1);
    }
    __syncthreads();
    {   // exclusive scan of the cell counts: every thread owns a contiguous segment of cells
        const int per = (ncell + GRAPH_SORT_THREADS - 1) / GRAPH_SORT_THREADS;
        const int lo = min((int)threadIdx.x * per, ncell), hi = min(lo + per, ncell);
        int sum = 0;
        for (int c = lo; c < hi; ++c) sum += cursor[c];
        int inc = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int base = inc - sum;
        for (int w = 0; w < wave; ++w) base += s_w[w];
        if (threadIdx.x == 0) sstart[0] = 0;
        for (int c = lo; c < hi; ++c) {
            const int n = cursor[c];
            cursor[c] = base;
            base += n;
            sstart[c + 1] = base;
        }
    }
    if ((int)threadIdx.x < Np - N) q4[N + threadIdx.x] = make_float4(1e18f, 0.0f, 0.0f, __int_as_float(-1));
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += GRAPH_SORT_THREADS) {
        float x, y, z;
        displaced(i, x, y, z, true);
        const int slot = atomicAdd(&cursor[graph_band(y, inv_hb, gy) * GC_XS + graph_strip(x)], 1);
        q4[slot] = make_float4(x, y, z, __int_as_float(i));
    }
}

// ---- k_graph_cells: a quarter is a DPP row --------------------------------------------------------------------
// Lane l of the quarter loads candidate 16 c + l of the quarter's region (one coalesced global load per lane and
// sixteen candidates, the next chunk requested before this one is used), and the sixteen lanes see all sixteen
// through `row_ror:r`, r = 0 ... 15, folded into the subtraction that starts the distance (v_sub_f32_dpp: no
// instruction more).  Every lane meets the same sixteen candidates in a rotated order; the ten smallest distances
// and the emitted set do not depend on the order.  A quarter's region is ONE sequence -- its runs listed with their
// prefix sums in a small LDS table, position -> slot by a scan of that table -- so the wave's trip count is the
// longest of its four quarters' TOTALS, and a run's end costs no masked tail: positions past the total read as
// x = 1e18.  No positions in LDS: 18 KB per workgroup whatever N, eight waves per SIMD.
// (The first version broadcast every candidate from LDS to the 16 lanes -- one ds_read_b128 per candidate and lane,
// the sample's positions staged by every workgroup, 19 KB at 1 200 particles, the four quarters in lockstep band by
// band -- 3.9 ms per iteration at 1 200 x 512 x 20 against 3.0 now)
#define GC_THREADS 256
#define GC_RUNS (2 * GC_MAX_BANDS)
#define GC_LIST 11                                                // ten entries + the slot a full list keeps overwriting
#define GRAPH_CELLS_LDS(ncell) \
    ((((size_t)(ncell) + 4) & ~(size_t)3) * 4 + (size_t)(GC_THREADS / 16) * GC_RUNS * 8 + (size_t)GC_THREADS * GC_LIST * 2 + 16)

template <int R>
__device__ __forceinline__ int row_ror_i(int v) {
    if constexpr (R == 0) return v;
    else return __builtin_amdgcn_update_dpp(0, v, 0x120 + R, 0xf, 0xf, true);
}
template <int R>
__device__ __forceinline__ float row_ror(float v) { return __int_as_float(row_ror_i<R>(__float_as_int(v))); }
template <int R, class F>
__device__ __forceinline__ void for_rot16(F&& f) {
    f(std::integral_constant<int, R>{});
    if constexpr (R + 1 < 16) for_rot16<R + 1>(f);
}
// all-reduce over the 16 lanes of a row: four rotations, each folded into its min / max
__device__ __forceinline__ int row_min_i(int v) {
    v = min(v, row_ror_i<8>(v)); v = min(v, row_ror_i<4>(v)); v = min(v, row_ror_i<2>(v)); return min(v, row_ror_i<1>(v));
}
__device__ __forceinline__ int row_max_i(int v) {
    v = max(v, row_ror_i<8>(v)); v = max(v, row_ror_i<4>(v)); v = max(v, row_ror_i<2>(v)); return max(v, row_ror_i<1>(v));
}
__device__ __forceinline__ float row_max_f(float v) {
    v = fmaxf(v, row_ror<8>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<2>(v)); return fmaxf(v, row_ror<1>(v));
}
// largest of a value that is uniform within every row: a scalar
__device__ __forceinline__ int rows_max_i(int v) {
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

#ifdef GC_STATS
__device__ unsigned long long gc_stats[16];
#define GC_STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&gc_stats[k], (unsigned long long)(v)); } while (0)
#define GC_STATQ(k, v) do { if ((threadIdx.x & 15) == 0) atomicAdd(&gc_stats[k], (unsigned long long)(v)); } while (0)
#else
#define GC_STAT(k, v)
#define GC_STATQ(k, v)
#endif
DRP_GLOBAL void __launch_bounds__(GC_THREADS)
k_graph_cells(const float4* __restrict__ sorted, const int* __restrict__ starts, int N, int gy, float inv_hb,
               int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, float thr, int chunks, int n_items /* B * chunks */,
               int self_first, float halo_first) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = GC_THREADS;
    const int Np = (N + 3) & ~3, ncell = gy * GC_XS;
    int* cs = reinterpret_cast<int*>(lds);                                   // [ncell + 1] first slot of every cell
    int2* tabs = reinterpret_cast<int2*>(cs + ((ncell + 4) & ~3));           // [T / 16][GC_RUNS] (first slot - prefix, prefix)
    int16_t* lst = reinterpret_cast<int16_t*>(tabs + (T / 16) * GC_RUNS);   // [T][GC_LIST] chosen indices, unsorted
    const int item = spread_item();
    if (item >= n_items) return;
    const int b = item / chunks, chunk = item - b * chunks;
    const float4* g4 = sorted + (size_t)b * Np;
    const int* gs = starts + (size_t)b * (ncell + 1);
    for (int c = threadIdx.x; c <= ncell; c += T) cs[c] = gs[c];
    for (int c = threadIdx.x; c < (T / 16) * GC_RUNS; c += T) tabs[c] = make_int2(0, 0x7fffffff);   // no run starts here
    __shared__ int qstart[GC_MAX_BANDS + 1];
    __syncthreads();
    if (threadIdx.x == 0) {
        int q = 0;
        for (int bb = 0; bb < gy; ++bb) {
            qstart[bb] = q;
            q += (cs[(bb + 1) * GC_XS] - cs[bb * GC_XS] + 15) >> 4;
        }
        qstart[gy] = q;
    }
    __syncthreads();
    const int nq = qstart[gy];
    const int qid = chunk * (T / 16) + ((int)threadIdx.x >> 4);
    const bool q_on = qid < nq;
    if (__ballot(q_on) == 0) return;                             // no barrier below: a wave is on its own from here
    int qb = 0;
    {
        const int qq = min(qid, nq - 1);
        while (qb + 1 < gy && qstart[qb + 1] <= qq) ++qb;
    }
    const float radius = __fsqrt_rn(fmaxf(thr, 0.0f)) * 1.000001f;
    auto reach_x = [&](float hh) { return (int)fminf(ceilf(hh * GRAPH_STRIP_INV_W + 1e-3f), (float)GC_XS); };
    auto reach_y = [&](float hh) { return (int)fminf(ceilf(hh * inv_hb + 1e-3f), (float)gy); };
    const int band_lo = cs[qb * GC_XS], band_hi = cs[(qb + 1) * GC_XS];
    const int l16 = (int)threadIdx.x & 15;
    const int si = band_lo + ((min(qid, nq - 1) - qstart[qb]) << 4) + l16;
    const bool valid = q_on && si < band_hi;
    const int si_safe = valid ? si : max(band_hi - 1, band_lo);  // an idle lane mirrors its band's last receiver
    const float4 pi = g4[si_safe];
    const int i = __float_as_int(pi.w);
    const int xs = graph_strip(pi.x), yb = graph_band(pi.y, inv_hb, gy);
    const int xs_min = row_min_i(xs), xs_max = row_max_i(xs), yb_min = row_min_i(yb), yb_max = row_max_i(yb);
    // the quarter's region -- bands [yb_min - rb, yb_max + rb], in each the strips [xs_min - rx, xs_max + rx], minus
    // (excl) the region of the halo (rbE, rxE) swept before -- as a table of runs: position p of the region, with
    // prefix <= p < next prefix, is slot p + (first slot - prefix); entries past the last run keep prefix = INT_MAX
    int2* tab = tabs + ((int)threadIdx.x >> 4) * GC_RUNS;
    int nr = 0, nr_hi = 0, total = 0;
    auto build = [&](int rb, int rx, bool excl, int rbE, int rxE) {
        const int b_lo = max(yb_min - rb, 0), b_hi = min(yb_max + rb, gy - 1);
        const int x0 = max(xs_min - rx, 0), x1 = min(xs_max + rx, GC_XS - 1);
        const int eb_lo = max(yb_min - rbE, 0), eb_hi = min(yb_max + rbE, gy - 1);
        const int e0 = max(xs_min - rxE, 0), e1 = min(xs_max + rxE, GC_XS - 1);
        nr = 0;
        total = 0;
        for (int bb = b_lo; bb <= b_hi; ++bb) {
            const int* cb = cs + bb * GC_XS;
            int ra, re, ta = 0, tb = 0;
            if (excl && bb >= eb_lo && bb <= eb_hi) {
                ra = cb[x0]; re = cb[e0];                          // strips x0 .. e0 - 1
                ta = cb[e1 + 1]; tb = cb[x1 + 1];                  // strips e1 + 1 .. x1
            } else {
                ra = cb[x0]; re = cb[x1 + 1];
            }
            if (re > ra) { tab[nr] = make_int2(ra - total, total); ++nr; total += re - ra; }
            if (tb > ta) { tab[nr] = make_int2(ta - total, total); ++nr; total += tb - ta; }
        }
        for (int r = nr; r < nr_hi; ++r) tab[r] = make_int2(0, 0x7fffffff);
        nr_hi = max(nr_hi, nr);
    };
    int stat_slot = 0;
    auto sweep = [&](auto body16) {
        const int nch = rows_max_i((total + 15) >> 4);
        const int nrm = rows_max_i(nr);
        GC_STAT(stat_slot, nch);                                  // chunks the wave runs
        GC_STATQ(stat_slot + 1, q_on ? total : 0);                // candidates of the quarter's region
        GC_STATQ(stat_slot + 2, q_on ? nr : 0);
        auto fetch = [&](int pos) {
            int dlt = 0;
            for (int r = 0; r < nrm; ++r) {
                const int2 t = tab[r];
                dlt = (pos >= t.y) ? t.x : dlt;
            }
            const bool in = pos < total;
            float4 q = g4[in ? pos + dlt : si_safe];
            q.x = in ? q.x : 1e18f;
            return q;
        };
        if (nch == 0) return;
        float4 cur = fetch(l16);
        for (int c = 0; c < nch; ++c) {
            float4 nxt = cur;
            if (c + 1 < nch) nxt = fetch(((c + 1) << 4) + l16);
            body16(cur);
            cur = nxt;
        }
    };

    float best[DRP_K];
#pragma unroll
    for (int q = 0; q < DRP_K; ++q) best[q] = thr;
    auto insert16 = [&](const float4& q) {
        for_rot16<0>([&](auto rr) {
            constexpr int R = decltype(rr)::value;
            const float d = pair_dis(pi.x, pi.y, pi.z, row_ror<R>(q.x), row_ror<R>(q.y), row_ror<R>(q.z));
#pragma unroll
            for (int k = DRP_K - 1; k > 0; --k) best[k] = __builtin_amdgcn_fmed3f(d, best[k - 1], best[k]);
            best[0] = min_nonneg(d, best[0]);
        });
    };
    // first sweep, stage A: a halo of the expected 10th-nearest distance; stage B: what the provisional 10th distance
    // found there still allows
    const float halo_req = fminf(fmaxf(halo_first, 0.0f), radius);
    const int rba = reach_y(halo_req), rxa = reach_x(halo_req);
    // the cells of that reach hold every sender within halo_a of a receiver of the block: whole cells are swept
    // anyway, so the first stage vouches for all they cover (and the second one runs that much less often)
    const float halo_a = fminf(radius, fminf(rxa >= GC_XS ? radius : ((float)rxa - 2e-3f) * (1.0f / GRAPH_STRIP_INV_W),
                                             rba >= gy ? radius : ((float)rba - 2e-3f) / inv_hb));
    build(rba, rxa, false, 0, 0);
    GC_STAT(12, 1);
    GC_STATQ(13, q_on ? 1 : 0);
    sweep(insert16);
    {
        const float kprov = row_max_f(valid ? best[DRP_K - 1] : 0.0f);
        const float halo_b = fminf(radius, __fsqrt_rn(fmaxf(kprov, 0.0f)) * 1.000001f);
        const bool more = halo_b > halo_a;
        if (__any(more)) {
            const int rbb = more ? max(reach_y(halo_b), rba) : rba, rxb = more ? max(reach_x(halo_b), rxa) : rxa;
            build(rbb, rxb, true, rba, rxa);
            stat_slot = 3;
            GC_STAT(14, 1);
            GC_STATQ(15, (q_on && more) ? 1 : 0);
            sweep(insert16);
        }
    }
    const float kth = best[DRP_K - 1];
    // second sweep over what the quarter's largest 10th-nearest distance still reaches.  Edges are the senders with
    // d - thr < 0 that are nearer than kth, plus the ones AT kth while slots are left, lowest index first.  kth is the
    // tenth smallest in-radius distance (or thr when there are fewer): at most nine are nearer, and when no more than
    // ten are at or below it every one of them is an edge -- the common case, one compare per candidate: d <= kle with
    // kle = kth below the radius, else the largest float under thr (d <= kle <=> d < thr).  More than ten (several
    // senders at exactly kth): the list is rebuilt by the exact rule.
    const float halo_2 = fminf(radius, __fsqrt_rn(fmaxf(row_max_f(valid ? kth : 0.0f), 0.0f)) * 1.000001f);
    build(reach_y(halo_2), reach_x(halo_2), false, 0, 0);
    const int skip = (self_first && thr > 0.0f) ? i : -1;
    const float kle = (kth < thr) ? kth : (thr > 0.0f ? __int_as_float(__float_as_int(thr) - 1) : -1.0f);
    int16_t* mine = lst + threadIdx.x * GC_LIST;
    int cnt = 0;
    stat_slot = 6;
    sweep([&](const float4& q) {
        for_rot16<0>([&](auto rr) {
            constexpr int R = decltype(rr)::value;
            const float d = pair_dis(pi.x, pi.y, pi.z, row_ror<R>(q.x), row_ror<R>(q.y), row_ror<R>(q.z));
            mine[min(cnt, DRP_K)] = (int16_t)row_ror_i<R>(__float_as_int(q.w));     // kept only if the count moves on
            cnt += (d <= kle) ? 1 : 0;
        });
    });
    if (cnt > DRP_K) {
        // the exact rule, one walk over the lane's own region per pass
        auto walk = [&](auto f) {
            for (int u = 0; u < nr; ++u) {
                const int2 t = tab[u];
                const int len = ((u + 1 < nr) ? tab[u + 1].y : total) - t.y;
                for (int j = t.x + t.y; j < t.x + t.y + len; ++j) {
                    const float4 q = g4[j];
                    f(pair_dis(pi.x, pi.y, pi.z, q.x, q.y, q.z), __float_as_int(q.w));
                }
            }
        };
        cnt = 0;
        walk([&](float d, int o) {
            if (__fsub_rn(d, thr) < 0.0f && d < kth && o != skip && cnt < DRP_K) mine[cnt++] = (int16_t)o;
        });
        const int room = DRP_K - cnt - (skip >= 0 ? 1 : 0);
        int last = -1;
        for (int r = 0; r < room; ++r) {
            int nxt = 0x7fff;
            walk([&](float d, int o) {
                if (__fsub_rn(d, thr) < 0.0f && d == kth && o != skip && o > last) nxt = min(nxt, o);
            });
            if (nxt == 0x7fff) break;
            mine[cnt++] = (int16_t)nxt;
            last = nxt;
        }
    }
    if (skip >= 0) {                                               // the receiver itself is slot 0 of the output, not an entry
        int at = -1;
#pragma unroll
        for (int q = 0; q < DRP_K; ++q)
            if (q < cnt && (int)mine[q] == skip) at = q;
        if (at >= 0) {
            mine[at] = mine[cnt - 1];
            --cnt;
        }
    }
    if (!valid) return;
    int v[DRP_K];
#pragma unroll
    for (int q = 0; q < DRP_K; ++q) v[q] = (q < cnt) ? (int)mine[q] : 0x7fff;
#define CE(a, b) sort2(v[a], v[b])
    CE(0, 5); CE(1, 6); CE(2, 7); CE(3, 8); CE(4, 9);
    CE(0, 3); CE(1, 4); CE(5, 8); CE(6, 9);
    CE(0, 2); CE(3, 6); CE(7, 9);
    CE(0, 1); CE(2, 4); CE(5, 7); CE(8, 9);
    CE(1, 2); CE(3, 5); CE(4, 6); CE(7, 8);
    CE(1, 3); CE(2, 5); CE(4, 7); CE(6, 8);
    CE(2, 3); CE(4, 5); CE(6, 7);
    CE(3, 4); CE(5, 6);
#undef CE
    int16_t* out = nbr_idx + ((size_t)b * N + i) * DRP_K;
    int w = 0;
    if (skip >= 0) out[w++] = (int16_t)i;
#pragma unroll
    for (int q = 0; q < DRP_K; ++q)
        if (q < cnt && w < DRP_K) out[w++] = (int16_t)v[q];
    nbr_cnt[(size_t)b * N + i] = (uint8_t)w;
    for (int q = w; q < DRP_K; ++q) out[q] = -1;
}

// ---- mean in-degree of a batch's lists, for the host's choice between paired and unpaired tiles (prop_pair, capi_ctx.h)
// one workgroup; out (host memory the device can write) = sum | rows << 24 | N << 48 in ONE 64-bit store
#define DEG_STAT_MAX_ROWS 65536
DRP_GLOBAL void __launch_bounds__(1024)
k_deg_stat(const uint8_t* __restrict__ nbr_cnt, int rows, int N, unsigned long long* __restrict__ out) {
    __shared__ int part[16];
    int s = 0;
    for (int r = threadIdx.x; r < rows; r += 1024) s += nbr_cnt[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        *out = (unsigned long long)(unsigned)t | ((unsigned long long)(unsigned)rows << 24) | ((unsigned long long)(unsigned)N << 48);
        __threadfence_system();
    }
}

