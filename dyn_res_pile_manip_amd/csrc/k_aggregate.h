// Gather + segmented sum over a receiver's in-edges: the "scatter-add" of the reference.
//
//   effect_rel[e]  = relu(c_edge[e] + (W_r eff)[recv e] + (W_s eff)[send e])   gnn_dyn.py:183-187
//   agg[i]         = sum over e with recv e == i of effect_rel[e]              gnn_dyn.py:189
//
// The reference does this with dense one-hot bmm's (Rr, Rs, Rr^T).  Here edges are
// receiver-major with at most 10 per receiver, so the scatter-add is a segmented sum of
// <= 10 rows with no atomics: 16 lanes own one receiver (a float4 = 4 features per
// lane), a wave owns 4 receivers, every load is a full 16 B/lane = 1 KiB per wave
// instruction.  HBM/L2-bound: per receiver it reads its own projected row (256 B), K
// sender rows (256 B each, gathered), K edge-constant rows (256 B each, streamed) and
// writes one row: (2K+2) * 256 B, no arithmetic to speak of.
//
// grid = B (one workgroup per sample keeps the gathered rows of a sample in one
// XCD's L2), block = 256 threads = 16 receivers per pass.
#pragma once
#include "drp_common.h"

__global__ void __launch_bounds__(256)
k_aggregate(const float* __restrict__ c_edge, const float* __restrict__ proj,
            const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt, int N,
            float* __restrict__ agg, int chunks /* workgroups per sample: > 1 when the batch alone cannot fill the chip */) {
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int len = (((N + chunks - 1) / chunks) + 15) & ~15;
    const int lo = ch * len, hi = min(N, lo + len);
    const int q = threadIdx.x & 15;          // float4 column within the 64-feature row
    const int g = threadIdx.x >> 4;          // receiver slot within the pass (0..15)
    const float4* ce = reinterpret_cast<const float4*>(c_edge) + (size_t)b * N * DRP_K * 16;
    const float4* pj = reinterpret_cast<const float4*>(proj) + (size_t)b * N * 32;
    float4* out = reinterpret_cast<float4*>(agg) + (size_t)b * N * 16;
    const int16_t* nb = nbr_idx + (size_t)b * N * DRP_K;
    const uint8_t* nc = nbr_cnt + (size_t)b * N;
    for (int i = lo + g; i < hi; i += 16) {
        const int cnt = nc[i];
        const float4 pr = pj[(size_t)i * 32 + q];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int js[DRP_K];
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) js[k] = (k < cnt) ? (int)nb[i * DRP_K + k] : i;
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) {
            const float4 c = ce[((size_t)i * DRP_K + k) * 16 + q];
            const float4 ps = pj[(size_t)js[k] * 32 + 16 + q];
            if (k < cnt) {
                acc.x += fmaxf((c.x + pr.x) + ps.x, 0.0f);
                acc.y += fmaxf((c.y + pr.y) + ps.y, 0.0f);
                acc.z += fmaxf((c.z + pr.z) + ps.z, 0.0f);
                acc.w += fmaxf((c.w + pr.w) + ps.w, 0.0f);
            }
        }
        out[(size_t)i * 16 + q] = acc;
    }
}

// The same segmented sum, also leaving what the reverse-mode kernels need of the edge stage: the 64 ReLU bits of every
// edge slot in the tape's layout (k_backward.h kb_mask_bit: feature f in word (f >> 2) & 1 at bit
// 31 - (16 (f >> 5) + (f & 3) + 4 ((f & 31) >> 3))).  The gradient-descent planner and the trainer write their tape with the
// fused engine's km_prop<., TAPE>; this kernel serves them on the fp32 engine when the split-fp16 relation encoder refuses
// the weights or the inputs (DRP_ERANGE): slower (the edge constants are materialised), no range limit.
__global__ void __launch_bounds__(256)
k_aggregate_tape(const float* __restrict__ c_edge, const float* __restrict__ proj,
                 const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt, int N,
                 float* __restrict__ agg, int chunks, unsigned* __restrict__ mask /* [B*N*10][2] */) {
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int len = (((N + chunks - 1) / chunks) + 15) & ~15;
    const int lo = ch * len, hi = min(N, lo + len);
    const int q = threadIdx.x & 15;
    const int g = threadIdx.x >> 4;
    const float4* ce = reinterpret_cast<const float4*>(c_edge) + (size_t)b * N * DRP_K * 16;
    const float4* pj = reinterpret_cast<const float4*>(proj) + (size_t)b * N * 32;
    float4* out = reinterpret_cast<float4*>(agg) + (size_t)b * N * 16;
    const int16_t* nb = nbr_idx + (size_t)b * N * DRP_K;
    const uint8_t* nc = nbr_cnt + (size_t)b * N;
    unsigned* mk = mask + (size_t)b * N * DRP_K * 2;
    const int shift = 28 - 16 * (q >> 3) - 4 * ((q >> 1) & 3);      // this lane's nibble: bit 3 = x ... bit 0 = w
    // every lane of a wave runs the same number of passes (the shuffles below need all 16 lanes of a receiver, and a
    // receiver's 16 lanes are either all inside the range or all outside)
    for (int i = lo + g; i < hi; i += 16) {
        const int cnt = nc[i];
        const float4 pr = pj[(size_t)i * 32 + q];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < DRP_K; ++k) {
            unsigned w = 0u;
            if (k < cnt) {
                const int j = (int)nb[i * DRP_K + k];
                const float4 c = ce[((size_t)i * DRP_K + k) * 16 + q];
                const float4 ps = pj[(size_t)j * 32 + 16 + q];
                const float x = (c.x + pr.x) + ps.x, y = (c.y + pr.y) + ps.y, z = (c.z + pr.z) + ps.z, v = (c.w + pr.w) + ps.w;
                acc.x += fmaxf(x, 0.0f); acc.y += fmaxf(y, 0.0f); acc.z += fmaxf(z, 0.0f); acc.w += fmaxf(v, 0.0f);
                w = ((x > 0.0f ? 8u : 0u) | (y > 0.0f ? 4u : 0u) | (z > 0.0f ? 2u : 0u) | (v > 0.0f ? 1u : 0u)) << shift;
            }
            // OR over the receiver's lanes of equal parity (lanes q and q ^ 1 own the two words)
            w |= __shfl_xor(w, 2, 64);
            w |= __shfl_xor(w, 4, 64);
            w |= __shfl_xor(w, 8, 64);
            if (q < 2) mk[((size_t)i * DRP_K + k) * 2 + q] = w;
        }
        out[(size_t)i * 16 + q] = acc;
    }
}

// Variant for samples whose sender rows fit in LDS (N <= 600): the workgroup first copies
// the sample's [N][64] W_s eff rows into LDS (one compulsory read), then every gather is a
// conflict-free ds_read_b128 (a row is 256 B = all 64 banks, 16 lanes read it whole), so
// HBM sees only the compulsory bytes: c_edge stream + proj once + agg write.
// grid = B, block = 512 threads (32 receivers per pass), dynamic LDS = N * 256 bytes.
__global__ void __launch_bounds__(512)
k_aggregate_lds(const float* __restrict__ c_edge, const float* __restrict__ proj,
                const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt, int N,
                float* __restrict__ agg) {
    extern __shared__ __attribute__((aligned(16))) float4 ps[];      // [N][16]
    const int b = blockIdx.x;
    const int q = threadIdx.x & 15;
    const int g = threadIdx.x >> 4;          // 0..31
    const float4* ce = reinterpret_cast<const float4*>(c_edge) + (size_t)b * N * DRP_K * 16;
    const float4* pj = reinterpret_cast<const float4*>(proj) + (size_t)b * N * 32;
    float4* out = reinterpret_cast<float4*>(agg) + (size_t)b * N * 16;
    const int16_t* nb = nbr_idx + (size_t)b * N * DRP_K;
    const uint8_t* nc = nbr_cnt + (size_t)b * N;
    for (int idx = threadIdx.x; idx < N * 16; idx += 512)
        ps[idx] = pj[(size_t)(idx >> 4) * 32 + 16 + (idx & 15)];
    __syncthreads();
    for (int i = g; i < N; i += 32) {
        const int cnt = nc[i];
        const float4 pr = pj[(size_t)i * 32 + q];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int js[DRP_K];
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) js[k] = (k < cnt) ? (int)nb[i * DRP_K + k] : i;
        float4 c[DRP_K];
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) c[k] = ce[((size_t)i * DRP_K + k) * 16 + q];
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) {
            const float4 s = ps[js[k] * 16 + q];
            if (k < cnt) {
                acc.x += fmaxf((c[k].x + pr.x) + s.x, 0.0f);
                acc.y += fmaxf((c[k].y + pr.y) + s.y, 0.0f);
                acc.z += fmaxf((c[k].z + pr.z) + s.z, 0.0f);
                acc.w += fmaxf((c[k].w + pr.w) + s.w, 0.0f);
            }
        }
        out[(size_t)i * 16 + q] = acc;
    }
}
#define K_AGG_LDS_MAX_N 600
