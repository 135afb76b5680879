// fp32 VALU engine for the PropNet MLPs (model/gnn_dyn.py:147-198) in factored form.
//
// Layout: one wavefront = 64 output features of R rows at a time.  A weight matrix
// sits in LDS transposed to [in][64] (lane = output feature reads consecutive floats,
// conflict-free); an input row lives across the wave (lane k holds x[k]) and is
// broadcast with v_readlane.  These kernels are the simple, obviously-correct engine
// the MFMA engine is cross-checked against on the device.
#pragma once
#include "drp_common.h"

// acc[r] = init[r] + sum_k x[r][k] * Wt[k][lane];  x[r] distributed one feature per lane.
template <int IN, int R>
__device__ __forceinline__ void dense_bcast(const float* __restrict__ Wt, const float (&x)[R],
                                            float (&acc)[R], int lane) {
#pragma unroll
    for (int k = 0; k < IN; ++k) {
        const float w = Wt[k * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fmaf(bcast_lane(x[r], k), w, acc[r]);
    }
}

__device__ __forceinline__ void lds_copy(float* dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// ---- particle encoder + loop-invariant node constant -----------------------------------
//   pe     = relu(W2 relu(W1 [s_delta, a, d] + b1) + b2)         gnn_dyn.py:174-175
//   c_node = W_pe pe + w_d d + b                                 (particle propagator,
//            the part of gnn_dyn.py:191-193 that does not change over the 3 steps)
//   eff    = pe                                                  gnn_dyn.py:176
// grid = B (one workgroup per sample), block = 256.
template <int R>
__global__ void __launch_bounds__(256)
k_node_encode(const float* __restrict__ vw, const float* __restrict__ s_delta,
              const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens,
              int dens_mod, int N, float* __restrict__ eff, float* __restrict__ c_node) {
    __shared__ float w0[5 * 64], w2[4096], wpe[4096];
    lds_copy(w0, vw + V_PE0_T, 5 * 64);
    lds_copy(w2, vw + V_PE2_T, 4096);
    lds_copy(wpe, vw + V_PPE_T, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int b = blockIdx.x;
    const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
    const float b0 = vw[V_PE0_B + lane], b2 = vw[V_PE2_B + lane];
    const float cb = fmaf(d, vw[V_PP_WD + lane], vw[V_PP_B + lane]);
    const float* sd = s_delta + (size_t)b * N * 3;
    const float* at = attr + (size_t)(b % attr_mod) * N;
    for (int base = wave * R; base < N; base += nwave * R) {
        float x[R], h[R], o[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = min(base + r, N - 1);
            float v = 0.0f;
            if (lane < 3) v = sd[i * 3 + lane];
            else if (lane == 3) v = at[i];
            else if (lane == 4) v = d;
            x[r] = v;
            h[r] = b0;
        }
        dense_bcast<5, R>(w0, x, h, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) { h[r] = fmaxf(h[r], 0.0f); o[r] = b2; }
        dense_bcast<64, R>(w2, h, o, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) { o[r] = fmaxf(o[r], 0.0f); h[r] = cb; }
        dense_bcast<64, R>(wpe, o, h, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = base + r;
            if (i < N) {
                eff[((size_t)b * N + i) * 64 + lane] = o[r];
                c_node[((size_t)b * N + i) * 64 + lane] = h[r];
            }
        }
    }
}

// ---- relation encoder + loop-invariant edge constant ------------------------------------
//   re     = 3 x (Linear+ReLU) on [a_r, a_s, s_r - s_s, d]        gnn_dyn.py:166-171,179-180
//   c_edge = W_e re + w_d d + b   (relation propagator, constant part of :186-187)
// One wave = the 10 slots of one receiver.  grid = B, block = 256.
__global__ void __launch_bounds__(256)
k_edge_encode(const float* __restrict__ vw, const float* __restrict__ s_cur, int s_mod,
              size_t s_stride, const float* __restrict__ attr, int attr_mod,
              const float* __restrict__ dens, int dens_mod, const int16_t* __restrict__ nbr_idx,
              const uint8_t* __restrict__ nbr_cnt, int N, float* __restrict__ c_edge) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w0 = lds;               // [6][64]
    float* w2 = w0 + 6 * 64;
    float* w4 = w2 + 4096;
    float* we = w4 + 4096;
    lds_copy(w0, vw + V_RE0_T, 6 * 64);
    lds_copy(w2, vw + V_RE2_T, 4096);
    lds_copy(w4, vw + V_RE4_T, 4096);
    lds_copy(we, vw + V_RPE_T, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int b = blockIdx.x;
    const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
    const float b0 = vw[V_RE0_B + lane], b2 = vw[V_RE2_B + lane], b4 = vw[V_RE4_B + lane];
    const float cb = fmaf(d, vw[V_RP_WD + lane], vw[V_RP_B + lane]);
    const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
    const float* at = attr + (size_t)(b % attr_mod) * N;
    constexpr int R = DRP_K;
    for (int i = wave; i < N; i += nwave) {
        const int cnt = nbr_cnt[(size_t)b * N + i];
        const int16_t* nb = nbr_idx + ((size_t)b * N + i) * DRP_K;
        float x[R], h[R], o[R];
        const float ar = at[i];
        const float sr = (lane >= 2 && lane < 5) ? s[i * 3 + lane - 2] : 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = (r < cnt) ? (int)nb[r] : i;     // padded slots: harmless self edge
            float v = 0.0f;
            if (lane == 0) v = ar;
            else if (lane == 1) v = at[j];
            else if (lane < 5) v = sr - s[j * 3 + lane - 2];   // receiver - sender, :180
            else if (lane == 5) v = d;
            x[r] = v;
            h[r] = b0;
        }
        dense_bcast<6, R>(w0, x, h, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) { h[r] = fmaxf(h[r], 0.0f); o[r] = b2; }
        dense_bcast<64, R>(w2, h, o, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) { o[r] = fmaxf(o[r], 0.0f); h[r] = b4; }
        dense_bcast<64, R>(w4, o, h, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) { h[r] = fmaxf(h[r], 0.0f); o[r] = cb; }
        dense_bcast<64, R>(we, h, o, lane);
        float* out = c_edge + ((size_t)b * N + i) * DRP_K * 64;
#pragma unroll
        for (int r = 0; r < R; ++r) out[r * 64 + lane] = o[r];
    }
}

// ---- per-step node projections:  proj[b,i] = [W_r eff | W_s eff]  (128 floats) -----------
template <int R>
__global__ void __launch_bounds__(256)
k_project(const float* __restrict__ vw, const float* __restrict__ eff, int N,
          float* __restrict__ proj) {
    __shared__ float wr[4096], ws[4096];
    lds_copy(wr, vw + V_RPR_T, 4096);
    lds_copy(ws, vw + V_RPS_T, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int b = blockIdx.x;
    for (int base = wave * R; base < N; base += nwave * R) {
        float x[R], pr[R], ps[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = min(base + r, N - 1);
            x[r] = eff[((size_t)b * N + i) * 64 + lane];
            pr[r] = 0.0f;
            ps[r] = 0.0f;
        }
        dense_bcast<64, R>(wr, x, pr, lane);
        dense_bcast<64, R>(ws, x, ps, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = base + r;
            if (i < N) {
                proj[((size_t)b * N + i) * 128 + lane] = pr[r];
                proj[((size_t)b * N + i) * 128 + 64 + lane] = ps[r];
            }
        }
    }
}

// ---- node update:  eff = relu(c_node + W_agg agg + eff)   gnn_dyn.py:191-193,:82-85 -------
template <int R>
__global__ void __launch_bounds__(256)
k_update(const float* __restrict__ vw, const float* __restrict__ agg,
         const float* __restrict__ c_node, int N, float* __restrict__ eff) {
    __shared__ float wa[4096];
    lds_copy(wa, vw + V_AGG_T, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int b = blockIdx.x;
    for (int base = wave * R; base < N; base += nwave * R) {
        float x[R], o[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t row = (size_t)b * N + min(base + r, N - 1);
            x[r] = agg[row * 64 + lane];
            o[r] = c_node[row * 64 + lane] + eff[row * 64 + lane];
        }
        dense_bcast<64, R>(wa, x, o, lane);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (base + r < N) eff[((size_t)b * N + base + r) * 64 + lane] = fmaxf(o[r], 0.0f);
    }
}

// ---- predictor:  s_pred = W1 relu(W0 eff + b0) + b1 + s_cur   gnn_dyn.py:196-198 -----------
template <int R>
__global__ void __launch_bounds__(256)
k_predict(const float* __restrict__ vw, const float* __restrict__ eff,
          const float* __restrict__ s_cur, int s_mod, size_t s_stride, int N,
          float* __restrict__ s_out, size_t out_stride) {
    __shared__ float w0[4096];
    lds_copy(w0, vw + V_PR0_T, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int b = blockIdx.x;
    const float b0 = vw[V_PR0_B + lane];
    const float w1x = vw[V_PR1_W + lane], w1y = vw[V_PR1_W + 64 + lane], w1z = vw[V_PR1_W + 128 + lane];
    const float b1 = (lane < 3) ? vw[V_PR1_B + lane] : 0.0f;
    const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
    float* so = s_out + (size_t)b * out_stride;
    for (int base = wave * R; base < N; base += nwave * R) {
        float x[R], h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x[r] = eff[((size_t)b * N + min(base + r, N - 1)) * 64 + lane];
            h[r] = b0;
        }
        dense_bcast<64, R>(w0, x, h, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float hv = fmaxf(h[r], 0.0f);
            const float ox = wave_sum(hv * w1x), oy = wave_sum(hv * w1y), oz = wave_sum(hv * w1z);
            const int i = base + r;
            if (i < N && lane < 3) {
                const float v = (lane == 0) ? ox : (lane == 1) ? oy : oz;
                so[i * 3 + lane] = (v + b1) + s[i * 3 + lane];
            }
        }
    }
}
