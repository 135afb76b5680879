// Split-bf16 ("bf16x3") MFMA version of the relation encoder chain.
//
// fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950 and there is no TF32 path, so the
// fp32 chain of k_mlp_mfma.h is bound by the matrix pipe (69 % of the fp32 peak measured).
// Here every fp32 operand is split into two bf16 pieces, x = x_hi + x_lo (round-to-nearest
// twice, |x - x_hi - x_lo| <= 2^-18 |x|), weights are split the same way on the host, and
// each product is formed as  W_lo x_hi + W_hi x_lo + W_hi x_hi  with fp32 accumulation inside
// v_mfma_f32_32x32x16_bf16 (products of bf16 pairs are exact in fp32).  The dropped
// W_lo x_lo term is <= 2^-18 relative: measured against the reference, the predicted
// displacement differs by 2.2e-6 relative (fp32 chain: 5.9e-7; parity bound 1e-4), i.e.
// 9e-9 absolute on positions whose fp32 ulp is 6e-8.  3 bf16 MFMAs of 32 cycles replace
// 8 fp32 MFMAs of 64 cycles: 5.3x less matrix-pipe time.
//
// Same transposed register chain as k_mlp_mfma.h.  For v_mfma_f32_32x32x16_bf16 the B
// operand of lane (col j, half h) is 8 consecutive k: k = 8h + jj; the C/D registers
// 8(s&1)..8(s&1)+7 of output block ob = s>>1 are fed as k-step s, so the weights are packed
// with   feature(s,h,jj) = 32(s>>1) + (r&3) + 8(r>>2) + 4h,  r = 8(s&1) + jj.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "k_mlp_mfma.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- packed split weights, in units of bf16x8 (16 bytes).  64x64: [part 2][ob 2][s 4][lane 64]
enum {
    S_RE0 = 0,                 // first layer, one k-step: [part 2][ob 2][lane 64]
    S_RE2 = S_RE0 + 256,
    S_RE4 = S_RE2 + 1024,
    S_RPE = S_RE4 + 1024,
    S_TOTAL = S_RPE + 1024     // x 16 bytes
};

inline uint16_t host_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

inline float host_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline int split_feature(int s, int h, int jj) {
    const int r = 8 * (s & 1) + jj;
    return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// host: state_dict blob -> split-bf16 fragments (uint16 storage, 8 per bf16x8)
inline void pack_split(const float* w, std::vector<uint16_t>& out) {
    out.assign((size_t)S_TOTAL * 8, 0);
    auto put = [&](int unit, int jj, int part, float v) {
        // unit = index of the hi bf16x8; the lo copy sits `part_stride` units later (given by caller)
        (void)part;
        out[(size_t)unit * 8 + jj] = host_bf16_rne(v);
    };
    auto P64 = [&](int dst, int src, int ld, int col0) {
        for (int ob = 0; ob < 2; ++ob)
            for (int s = 0; s < 4; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int i = lane & 31, h = lane >> 5;
                        const float v = w[src + (32 * ob + i) * ld + col0 + split_feature(s, h, jj)];
                        const float hi = host_bf16_to_f32(host_bf16_rne(v));
                        put(dst + ((0 * 2 + ob) * 4 + s) * 64 + lane, jj, 0, hi);
                        put(dst + ((1 * 2 + ob) * 4 + s) * 64 + lane, jj, 1, v - hi);
                    }
    };
    // relation encoder layer 0: inputs [a_r, a_s, dx, dy, dz, d, 1(bias), 0], k = 8h + jj, h = 1 unused
    for (int ob = 0; ob < 2; ++ob)
        for (int lane = 0; lane < 64; ++lane)
            for (int jj = 0; jj < 8; ++jj) {
                const int i = lane & 31, h = lane >> 5, o = 32 * ob + i;
                float v = 0.0f;
                if (h == 0 && jj < 6) v = w[W_RE0_W + o * 6 + jj];
                else if (h == 0 && jj == 6) v = w[W_RE0_B + o];
                const float hi = host_bf16_to_f32(host_bf16_rne(v));
                put(S_RE0 + (0 * 2 + ob) * 64 + lane, jj, 0, hi);
                put(S_RE0 + (1 * 2 + ob) * 64 + lane, jj, 1, v - hi);
            }
    P64(S_RE2, W_RE2_W, 64, 0);
    P64(S_RE4, W_RE4_W, 64, 0);
    P64(S_RPE, W_RP_W, 193, 0);
}

struct FragB {
    bf16x8 hi[4], lo[4];      // per 16-deep k-step
};

template <bool RELU>
__device__ __forceinline__ void split_frag(const Frag& in, FragB& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            float x = in.v[s >> 1][8 * (s & 1) + jj];
            if (RELU) x = fmaxf(x, 0.0f);
            const __bf16 hi = (__bf16)x;
            o.hi[s][jj] = hi;
            o.lo[s][jj] = (__bf16)(x - (float)hi);
        }
}

// acc += W x, W packed as bf16x8[(part*2 + ob)*4 + s][lane]
__device__ __forceinline__ void mfma_layer64_split(const bf16x8* __restrict__ wp, const FragB& b, Frag& acc, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const bf16x8 a_hi = wp[((0 * 2 + ob) * 4 + s) * 64 + lane];
            const bf16x8 a_lo = wp[((1 * 2 + ob) * 4 + s) * 64 + lane];
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b.hi[s], acc.v[ob], 0, 0, 0);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.lo[s], acc.v[ob], 0, 0, 0);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.hi[s], acc.v[ob], 0, 0, 0);
        }
    }
}

// first layer: one k-step over [a_r, a_s, dx, dy, dz, d, 1, 0] (lanes of half 1 supply zeros)
__device__ __forceinline__ void mfma_layer8_split(const bf16x8* __restrict__ wp, const float (&x)[8], int h, Frag& acc, int lane) {
    bf16x8 bhi, blo;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const float v = (h == 0) ? x[jj] : 0.0f;
        const __bf16 hi = (__bf16)v;
        bhi[jj] = hi;
        blo[jj] = (__bf16)(v - (float)hi);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const bf16x8 a_hi = wp[(0 * 2 + ob) * 64 + lane];
        const bf16x8 a_lo = wp[(1 * 2 + ob) * 64 + lane];
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bhi, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, blo, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bhi, acc.v[ob], 0, 0, 0);
    }
}

// the relation-encoder chain of one tile: inputs -> c_edge fragment
__device__ __forceinline__ void edge_chain_split(const bf16x8* __restrict__ wsp /*LDS, S_* offsets*/,
                                                 const float* __restrict__ rows /*b2,b4,b_rp,wd_rp*/,
                                                 const float (&x)[8], float d, int h, int lane, Frag& out) {
    Frag a, c;
    FragB fb;
    frag_zero(a);
    mfma_layer8_split(wsp + S_RE0, x, h, a, lane);
    split_frag<true>(a, fb);
    frag_from_row(rows + 0, h, c);
    mfma_layer64_split(wsp + S_RE2, fb, c, lane);
    split_frag<true>(c, fb);
    frag_from_row(rows + 64, h, a);
    mfma_layer64_split(wsp + S_RE4, fb, a, lane);
    split_frag<true>(a, fb);
    frag_bias_dens(rows + 128, rows + 192, d, h, out);
    mfma_layer64_split(wsp + S_RPE, fb, out, lane);
}

// same contract as km_edge_encode (k_mlp_mfma.h)
__global__ void __launch_bounds__(64 * MFMA_WAVES)
km_edge_encode_split(const uint16_t* __restrict__ sw, const float* __restrict__ mw,
                     const float* __restrict__ s_cur, int s_mod, size_t s_stride,
                     const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
                     const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt, int N, int B,
                     float* __restrict__ c_edge) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wsp_f = lds;                          // S_TOTAL * 4 floats
    float* rows = wsp_f + S_TOTAL * 4;           // b2, b4, b_rp, wd_rp
    float* tiles = rows + 256;
    lds_fill(wsp_f, reinterpret_cast<const float*>(sw), S_TOTAL * 4);
    lds_fill(rows, mw + R_RE2_B, 256);
    __syncthreads();
    const bf16x8* wsp = reinterpret_cast<const bf16x8*>(wsp_f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    float* tile = tiles + wave * TILE_FLOATS;
    const int nslots = N * DRP_K;
    const int tps = (nslots + 31) >> 5;
    const long ntiles = (long)B * tps;
    for (long gt = (long)blockIdx.x * MFMA_WAVES + wave; gt < ntiles; gt += (long)gridDim.x * MFMA_WAVES) {
        const int b = (int)(gt / tps), t = (int)(gt - (long)b * tps);
        const int slot = min(t * 32 + j, nslots - 1);
        const int i = slot / DRP_K, k = slot - i * DRP_K;
        const int cnt = nbr_cnt[(size_t)b * N + i];
        const int jn = (k < cnt) ? (int)nbr_idx[((size_t)b * N + i) * DRP_K + k] : i;
        const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
        const float* at = attr + (size_t)(b % attr_mod) * N;
        const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
        float x[8];
        x[0] = at[i];
        x[1] = at[jn];
        x[2] = s[i * 3 + 0] - s[jn * 3 + 0];
        x[3] = s[i * 3 + 1] - s[jn * 3 + 1];
        x[4] = s[i * 3 + 2] - s[jn * 3 + 2];
        x[5] = d;
        x[6] = 1.0f;
        x[7] = 0.0f;
        Frag c;
        edge_chain_split(wsp, rows, x, d, h, lane, c);
        const int rows_valid = min(32, nslots - t * 32);
        frag_store_tile(c, c_edge + ((size_t)b * nslots + (size_t)t * 32) * 64, 64, rows_valid, tile, lane);
    }
}

#define KM_EDGE_SPLIT_LDS ((S_TOTAL * 4 + 256 + MFMA_WAVES * TILE_FLOATS) * sizeof(float))

// ---- fused: relation encoder recomputed per propagation step + segmented aggregate -----------
// With the split chain the encoder is cheap enough to recompute in each of the three
// propagation steps, so the [B,N,10,64] edge-constant buffer (786 MB at 1024 x 300, written
// once and read three times per rollout step: 55 % of all HBM traffic of the unfused
// pipeline) never exists.  One workgroup per sample: the sample's W_s eff rows and particle
// positions are staged in LDS once, then a wave owns a tile of 32 receivers, loops over
// the 10 slots, runs the chain for the 32 edges (slot k of each receiver) and accumulates
//     agg[i] += relu(c_edge + (W_r eff)[i] + (W_s eff)[send])      gnn_dyn.py:183-189
// in the accumulator layout (receiver on the lane, 32 features in registers).  The sender
// rows are gathered from LDS with ds_read_b128 (row stride 68 floats spreads the rows over
// the banks).  HBM traffic per launch: proj once (157 MB) + agg (79 MB).
// Needs N*272 + 60 KB of LDS: N <= KM_FUSED_MAX_N.
#define KM_FUSED_MAX_N 368
#define PS_LD 68

__global__ void __launch_bounds__(64 * MFMA_WAVES)
km_edge_agg_split(const uint16_t* __restrict__ sw, const float* __restrict__ mw,
                  const float* __restrict__ s_cur, int s_mod, size_t s_stride,
                  const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
                  const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt,
                  const float* __restrict__ proj, int N, int B, float* __restrict__ agg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wsp_f = lds;                          // S_TOTAL * 4 floats
    float* rows = wsp_f + S_TOTAL * 4;           // b2, b4, b_rp, wd_rp
    float* pos = rows + 256;                     // [N][4] = x, y, z, attr
    float* ps = pos + ((N * 4 + 3) & ~3);        // [N][PS_LD]
    lds_fill(wsp_f, reinterpret_cast<const float*>(sw), S_TOTAL * 4);
    lds_fill(rows, mw + R_RE2_B, 256);
    const bf16x8* wsp = reinterpret_cast<const bf16x8*>(wsp_f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tps = (N + 31) >> 5;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                         // previous sample's readers are done
        const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
        const float* at = attr + (size_t)(b % attr_mod) * N;
        const float4* pj = reinterpret_cast<const float4*>(proj) + (size_t)b * N * 32;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            *reinterpret_cast<float4*>(pos + i * 4) = make_float4(s[i * 3], s[i * 3 + 1], s[i * 3 + 2], at[i]);
        for (int idx = threadIdx.x; idx < N * 16; idx += blockDim.x)
            *reinterpret_cast<float4*>(ps + (idx >> 4) * PS_LD + (idx & 15) * 4) = pj[(size_t)(idx >> 4) * 32 + 16 + (idx & 15)];
        __syncthreads();
        const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
        for (int t = wave; t < tps; t += MFMA_WAVES) {
            const int i = min(t * 32 + j, N - 1);
            const int cnt = nbr_cnt[(size_t)b * N + i];
            const int16_t* nb = nbr_idx + ((size_t)b * N + i) * DRP_K;
            // bpr = b + d w_d + (W_r eff)[i]: the receiver's share of every in-edge, used as the
            // initial accumulator of the chain's last layer
            Frag bpr, acc;
            {
                Frag pr;
                frag_bias_dens(rows + 128, rows + 192, d, h, bpr);
                frag_from_row(proj + ((size_t)b * N + i) * 128, h, pr);
#pragma unroll
                for (int r = 0; r < 16; ++r) { bpr.v[0][r] += pr.v[0][r]; bpr.v[1][r] += pr.v[1][r]; }
            }
            frag_zero(acc);
            const float4 pi = *reinterpret_cast<const float4*>(pos + i * 4);
            int jn = (0 < cnt) ? (int)nb[0] : i;
#pragma unroll 1
            for (int k = 0; k < DRP_K; ++k) {
                // keep the (loop-invariant) packed-weight reads inside the loop: hoisted, they
                // would occupy several hundred VGPRs
                asm volatile("" ::: "memory");
                const int jcur = jn;
                if (k + 1 < DRP_K) jn = (k + 1 < cnt) ? (int)nb[k + 1] : i;    // prefetch next index
                const float4 pn = *reinterpret_cast<const float4*>(pos + jcur * 4);
                float x[8];
                x[0] = pi.w; x[1] = pn.w;
                x[2] = pi.x - pn.x; x[3] = pi.y - pn.y; x[4] = pi.z - pn.z;
                x[5] = d; x[6] = 1.0f; x[7] = 0.0f;
                Frag a, c;
                FragB fb;
                frag_zero(a);
                mfma_layer8_split(wsp + S_RE0, x, h, a, lane);
                split_frag<true>(a, fb);
                frag_from_row(rows + 0, h, c);
                mfma_layer64_split(wsp + S_RE2, fb, c, lane);
                split_frag<true>(c, fb);
                frag_from_row(rows + 64, h, a);
                mfma_layer64_split(wsp + S_RE4, fb, a, lane);
                split_frag<true>(a, fb);
                c = bpr;
                mfma_layer64_split(wsp + S_RPE, fb, c, lane);
                const float keep = (k < cnt) ? 1.0f : 0.0f;
                const float* srow = ps + jcur * PS_LD + 4 * h;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 sv = *reinterpret_cast<const float4*>(srow + 32 * ob + 8 * g);
                        acc.v[ob][4 * g + 0] += keep * fmaxf(c.v[ob][4 * g + 0] + sv.x, 0.0f);
                        acc.v[ob][4 * g + 1] += keep * fmaxf(c.v[ob][4 * g + 1] + sv.y, 0.0f);
                        acc.v[ob][4 * g + 2] += keep * fmaxf(c.v[ob][4 * g + 2] + sv.z, 0.0f);
                        acc.v[ob][4 * g + 3] += keep * fmaxf(c.v[ob][4 * g + 3] + sv.w, 0.0f);
                    }
            }
            if (t * 32 + j < N) frag_to_row(agg + ((size_t)b * N + i) * 64, h, acc);
        }
    }
}

#define KM_FUSED_LDS(N) ((size_t)(S_TOTAL * 4 + 256 + (((N) * 4 + 3) & ~3) + (N) * PS_LD) * sizeof(float))
