// Proxy for km_prop's per-slot chain without memory traffic: 3 layers of (3-term fp16 MFMAs -> ReLU ->
// split to fp16 hi/lo) on a 32-item tile, with 32x32x16 MFMAs (2 accumulators of 16 registers) or
// 16x16x32 MFMAs (8 accumulators of 4 registers).  Same FLOPs, same vector work: does the finer
// granularity let the split of finished accumulators overlap the remaining MFMAs better?
//   hipcc --offload-arch=gfx950 -O3 tools/chain_proxy.hip -o tools/chain_proxy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ void split_pair(float x0, float x1, f16x8& hi, f16x8& lo, int q) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(r0), "=&v"(r1) : "v"(h), "v"(x0), "v"(x1));
    const fp16x2_t l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi[2 * q] = (_Float16)h[0]; hi[2 * q + 1] = (_Float16)h[1];
    lo[2 * q] = (_Float16)l[0]; lo[2 * q + 1] = (_Float16)l[1];
}

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, const f16x8* __restrict__ wsrc, int iters) {
    __shared__ f16x8 w[2 * 2 * 4 * 64];            // one layer's packed weights [part][ob][s][lane]
    for (int i = threadIdx.x; i < 2 * 2 * 4 * 64; i += blockDim.x) w[i] = wsrc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8 bh[4], bl[4];
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 8; ++j) { bh[s][j] = (_Float16)(0.01f * (lane + j + s)); bl[s][j] = (_Float16)1e-5f; }
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int layer = 0; layer < 3; ++layer) {
            if (MODE == 0) {
                f32x16 acc[2];
                for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) acc[ob][r] = 0.1f;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const f16x8 whi = w[((0 * 2 + ob) * 4 + s) * 64 + lane], wlo = w[((1 * 2 + ob) * 4 + s) * 64 + lane];
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh[s], acc[ob], 0, 0, 0);
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bl[s], acc[ob], 0, 0, 0);
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh[s], acc[ob], 0, 0, 0);
                    }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        split_pair(relu1(acc[s >> 1][8 * (s & 1) + 2 * q]), relu1(acc[s >> 1][8 * (s & 1) + 2 * q + 1]), bh[s], bl[s], q);
            } else {
                // 16x16x32: accumulators [ob 4][ib 2], k-steps 2; B operand per (ib, ks): 8 values
                f32x4 acc[4][2];
                for (int ob = 0; ob < 4; ++ob) for (int ib = 0; ib < 2; ++ib) for (int r = 0; r < 4; ++r) acc[ob][ib][r] = 0.1f;
#pragma unroll
                for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const f16x8 whi = w[((0 * 2 + (ob & 1)) * 4 + 2 * (ob >> 1) + ks) * 64 + lane];
                        const f16x8 wlo = w[((1 * 2 + (ob & 1)) * 4 + 2 * (ob >> 1) + ks) * 64 + lane];
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) {
                            acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, bh[2 * ib + ks], acc[ob][ib], 0, 0, 0);
                            acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, bl[2 * ib + ks], acc[ob][ib], 0, 0, 0);
                            acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, bh[2 * ib + ks], acc[ob][ib], 0, 0, 0);
                        }
                    }
                // next layer's B operand (ib, ks): registers of out-blocks 2ks and 2ks+1
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ob = 2 * ks + (q >> 1), r = 2 * (q & 1);
                            split_pair(relu1(acc[ob][ib][r]), relu1(acc[ob][ib][r + 1]), bh[2 * ib + ks], bl[2 * ib + ks], q);
                        }
            }
        }
        keep += (float)bh[0][0];
    }
    if (keep == 123.456f) out[0] = keep;
}

// MODE 2: the full slot shape -- first layer (6 MFMAs), three 24-MFMA layers, bias rows from LDS,
// epilogue acc += relu(c + sv) -- with sv from registers; MODE 3: sv gathered from global memory through an
// index loaded from global memory (the dependent loads of the real kernel), issued at the top of the slot
template <int MODE>
__global__ void __launch_bounds__(512) k2(float* out, const f16x8* __restrict__ wsrc, const float* __restrict__ rows_g,
                                          const int* __restrict__ idx, const float* __restrict__ proj, int iters) {
    __shared__ f16x8 w[4][2 * 2 * 4 * 64];
    __shared__ float rows[256];
    for (int i = threadIdx.x; i < 4 * 2 * 2 * 4 * 64; i += blockDim.x) w[i / 1024][i % 1024] = wsrc[i % 1024];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) rows[i] = rows_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, wave = threadIdx.x >> 6;
    const int item = (blockIdx.x * 8 + wave) * 32 + (lane & 31);
    f32x16 acc[2], bpr[2];
    for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) { acc[ob][r] = 0.f; bpr[ob][r] = 0.01f * r; }
    for (int it = 0; it < iters; ++it) {
        f32x16 sv[2];
        if (MODE & 1) {
            const int j = idx[(item * 10 + (it % 10)) & 0xfffff];
            const float* row = proj + (size_t)j * 128 + 64;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = *reinterpret_cast<const float4*>(row + 32 * ob + 8 * g + 4 * h);
                    sv[ob][4 * g] = t.x; sv[ob][4 * g + 1] = t.y; sv[ob][4 * g + 2] = t.z; sv[ob][4 * g + 3] = t.w;
                }
        } else {
            for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) sv[ob][r] = 0.001f * (r + it);
        }
        f16x8 bh[4], bl[4];
        f32x16 c[2];
        // first layer: one k-step, 3 terms, 2 blocks
        for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) c[ob][r] = 0.01f * (r + it);
        if (!(MODE & 4)) {
            f16x8 xh, xl;
#pragma unroll
            for (int q = 0; q < 4; ++q) split_pair(0.01f * (lane + q + it), 0.02f * (q + it), xh, xl, q);
            for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) c[ob][r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const f16x8 whi = w[0][(0 * 2 + ob) * 4 * 64 + lane], wlo = w[0][(1 * 2 + ob) * 4 * 64 + lane];
                c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh, c[ob], 0, 0, 0);
                c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xl, c[ob], 0, 0, 0);
                c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xh, c[ob], 0, 0, 0);
            }
        }
#pragma unroll
        for (int layer = 1; layer < 4; ++layer) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    split_pair(relu1(c[s >> 1][8 * (s & 1) + 2 * q]), relu1(c[s >> 1][8 * (s & 1) + 2 * q + 1]), bh[s], bl[s], q);
            if (layer < 3 && (MODE & 8)) {
                for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) c[ob][r] = 0.02f * r;
            } else if (layer < 3) {
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 t = *reinterpret_cast<const float4*>(rows + 64 * (layer - 1) + 32 * ob + 8 * g + 4 * h);
                        c[ob][4 * g] = t.x; c[ob][4 * g + 1] = t.y; c[ob][4 * g + 2] = t.z; c[ob][4 * g + 3] = t.w;
                    }
            } else {
                c[0] = bpr[0]; c[1] = bpr[1];
            }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f16x8 whi = w[layer][((0 * 2 + ob) * 4 + s) * 64 + lane], wlo = w[layer][((1 * 2 + ob) * 4 + s) * 64 + lane];
                    c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh[s], c[ob], 0, 0, 0);
                    c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bl[s], c[ob], 0, 0, 0);
                    c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh[s], c[ob], 0, 0, 0);
                }
        }
        if (MODE & 2) {
            acc[0][it & 15] += c[0][0] + c[1][1] + sv[0][2] + sv[1][3];
        } else {
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ob][r] += relu1(c[ob][r] + sv[ob][r]);
        }
    }
    float keep = 0.f;
    for (int r = 0; r < 16; ++r) keep += acc[0][r] + acc[1][r];
    if (keep == 123.456f) out[0] = keep;
}

// The slot loop restructured: accumulators start from the inline constant 0 (no LDS bias rows and no
// register copy ahead of a layer's first MFMA; the bias joins in the split, bias + P_r + P_s in the
// epilogue), and the NEXT slot's short first layer + its split are issued between this slot's last
// layer and its epilogue, so neither the 6-MFMA layer nor the epilogue waits on a drained pipe.
template <int MODE>
__global__ void __launch_bounds__(512) k3(float* out, const f16x8* __restrict__ wsrc, const float* __restrict__ rows_g,
                                          const int* __restrict__ idx, const float* __restrict__ proj, int iters) {
    __shared__ f16x8 w[4][2 * 2 * 4 * 64];
    __shared__ float rows[256];
    for (int i = threadIdx.x; i < 4 * 2 * 2 * 4 * 64; i += blockDim.x) w[i / 1024][i % 1024] = wsrc[i % 1024];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) rows[i] = rows_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, wave = threadIdx.x >> 6;
    const int item = (blockIdx.x * 8 + wave) * 32 + (lane & 31);
    f32x16 acc[2], bpr[2];
    for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) { acc[ob][r] = 0.f; bpr[ob][r] = 0.01f * r; }
    f16x8 bh[4], bl[4];
    auto first_layer = [&](int it) {
        f16x8 xh, xl;
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair(0.01f * (lane + q + it), 0.02f * (q + it), xh, xl, q);
        f32x16 a[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const f16x8 whi = w[0][(0 * 2 + ob) * 4 * 64 + lane], wlo = w[0][(1 * 2 + ob) * 4 * 64 + lane];
            f32x16 z;
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            a[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh, z, 0, 0, 0);
            a[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xl, a[ob], 0, 0, 0);
            a[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xh, a[ob], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                split_pair(relu1(a[s >> 1][8 * (s & 1) + 2 * q]), relu1(a[s >> 1][8 * (s & 1) + 2 * q + 1]), bh[s], bl[s], q);
    };
    first_layer(0);
    for (int it = 0; it < iters; ++it) {
        f32x16 sv[2];
        if (MODE & 1) {
            const int j = idx[(item * 10 + (it % 10)) & 0xfffff];
            const float* row = proj + (size_t)j * 128 + 64;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = *reinterpret_cast<const float4*>(row + 32 * ob + 8 * g + 4 * h);
                    sv[ob][4 * g] = t.x; sv[ob][4 * g + 1] = t.y; sv[ob][4 * g + 2] = t.z; sv[ob][4 * g + 3] = t.w;
                }
        } else {
            for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) sv[ob][r] = 0.001f * (r + it);
        }
        f32x16 c[2];
#pragma unroll
        for (int layer = 1; layer < 4; ++layer) {
            if (layer > 1) {
                // bias of the previous layer joins here (its LDS reads were issued long ago)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 b0 = *reinterpret_cast<const float4*>(rows + 64 * (layer - 2) + 32 * (s >> 1) + 8 * (2 * (s & 1)) + 4 * h);
                    const float4 b1 = *reinterpret_cast<const float4*>(rows + 64 * (layer - 2) + 32 * (s >> 1) + 8 * (2 * (s & 1) + 1) + 4 * h);
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        split_pair(relu1(c[s >> 1][8 * (s & 1) + 2 * q] + bb[2 * q]), relu1(c[s >> 1][8 * (s & 1) + 2 * q + 1] + bb[2 * q + 1]),
                                   bh[s], bl[s], q);
                }
            }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                f32x16 z;
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f16x8 whi = w[layer][((0 * 2 + ob) * 4 + s) * 64 + lane], wlo = w[layer][((1 * 2 + ob) * 4 + s) * 64 + lane];
                    c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh[s], s == 0 ? z : c[ob], 0, 0, 0);
                    c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bl[s], c[ob], 0, 0, 0);
                    c[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh[s], c[ob], 0, 0, 0);
                }
            }
        }
        // bias + P_r + P_s, ready long before the MFMAs finish
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[ob][r] += bpr[ob][r];
        if (MODE & 2) first_layer(it + 1);            // next slot's first layer before this slot's epilogue
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ob][r] += relu1(c[ob][r] + sv[ob][r]);
        if (!(MODE & 2)) first_layer(it + 1);
    }
    float keep = 0.f;
    for (int r = 0; r < 16; ++r) keep += acc[0][r] + acc[1][r];
    if (keep == 123.456f) out[0] = keep;
}


// ---- hand-ordered slot (k4): every step = [LDS reads two steps ahead] + [one MFMA group with the VALU pieces of
// one split chunk between its MFMAs], fenced with sched_barrier so the order below is the order issued.
// Split with v_fma_mixlo/mixhi_f16 (residual and its fp16 conversion in one instruction, ReLU of the residual by
// the clamp modifier, ReLU of the hi pair by one v_pk_max_i16): 2 vector instructions per value instead of 3.
// Biases by one MFMA per output block (A = [b_hi, b_lo, 0..], B = [1, 1, 0..]) into a zero accumulator.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
struct W2 { f16x8 hi, lo; };
__device__ __forceinline__ f16x8 asf16(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ unsigned pk_rtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ unsigned mix_resid_relu(unsigned h, float x0, float x1) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
        : "=&v"(l) : "v"(h), "v"(x0), "v"(x1));
    return l;
}
__device__ __forceinline__ unsigned mix_resid(unsigned h, float x0, float x1) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l) : "v"(h), "v"(x0), "v"(x1));
    return l;
}
__device__ __forceinline__ unsigned pk_relu(unsigned h) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, h), z));
}
// piece p (0..2) of a split chunk: registers 8*half .. 8*half+7 of src -> hi/lo operand of one k-step
__device__ __forceinline__ void chunk_piece(const f32x16& src, int half, u32x4& hi, u32x4& lo, int p) {
    if (p == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) hi[q] = pk_rtz(src[8 * half + 2 * q], src[8 * half + 2 * q + 1]);
    } else if (p == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) lo[q] = mix_resid_relu(hi[q], src[8 * half + 2 * q], src[8 * half + 2 * q + 1]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) hi[q] = pk_relu(hi[q]);
    }
}
struct NoValu { __device__ __forceinline__ void operator()(int) const {} };
// one MFMA group (a k-step of one output block: 3 MFMAs) with three VALU pieces between
template <class F>
__device__ __forceinline__ void step3(const W2& w, const u32x4& bh, const u32x4& bl, f32x16& out, const f32x16& cin, F valu) {
    out = MF(w.lo, asf16(bh), cin); FENCE(); valu(0); FENCE();
    out = MF(w.hi, asf16(bl), out); FENCE(); valu(1); FENCE();
    out = MF(w.hi, asf16(bh), out); FENCE(); valu(2); FENCE();
}
__device__ __forceinline__ W2 ldw(const f16x8* wl, int s, int ob, int lane) {
    W2 r;
    r.hi = wl[((0 * 2 + ob) * 4 + s) * 64 + lane];
    r.lo = wl[((1 * 2 + ob) * 4 + s) * 64 + lane];
    return r;
}
// A 64-wide layer up to its second-to-last group.  Entering: fh/fl[0] split, IN[0], IN[1] final, r0 = weights of
// group (s0, ob0), r2 = bias operands when BIASED.  Leaving: group (s3, ob1) still to run with the weights in r1;
// the caller pairs it with the first chunk of the next split.  tailA / tailB: the caller's LDS reads for what follows.
template <bool BIASED, class TA, class TB>
__device__ __forceinline__ void layer64(const f16x8* wl, int lane, W2& r0, W2& r1, W2& r2, u32x4 (&fh)[4], u32x4 (&fl)[4],
                                        const f32x16 (&IN)[2], f32x16 (&OUT)[2], const f32x16& cin0, const f32x16& cin1,
                                        const u32x4& ones, const f32x16& zero, TA tailA, TB tailB) {
    if (BIASED) {
        r1 = ldw(wl, 0, 1, lane); FENCE();
        OUT[0] = MF(r2.hi, asf16(ones), zero); FENCE(); chunk_piece(IN[0], 1, fh[1], fl[1], 0); FENCE();
        OUT[1] = MF(r2.lo, asf16(ones), zero); FENCE(); chunk_piece(IN[0], 1, fh[1], fl[1], 1); chunk_piece(IN[0], 1, fh[1], fl[1], 2); FENCE();
        r2 = ldw(wl, 1, 0, lane); FENCE();
        step3(r0, fh[0], fl[0], OUT[0], OUT[0], [&](int p) { chunk_piece(IN[1], 0, fh[2], fl[2], p); });
        r0 = ldw(wl, 1, 1, lane); FENCE();
        step3(r1, fh[0], fl[0], OUT[1], OUT[1], [&](int p) { chunk_piece(IN[1], 1, fh[3], fl[3], p); });
        r1 = ldw(wl, 2, 0, lane); FENCE();
        step3(r2, fh[1], fl[1], OUT[0], OUT[0], NoValu());
    } else {
        r1 = ldw(wl, 0, 1, lane); FENCE();
        r2 = ldw(wl, 1, 0, lane); FENCE();
        step3(r0, fh[0], fl[0], OUT[0], cin0, [&](int p) { chunk_piece(IN[0], 1, fh[1], fl[1], p); });
        r0 = ldw(wl, 1, 1, lane); FENCE();
        step3(r1, fh[0], fl[0], OUT[1], cin1, [&](int p) { chunk_piece(IN[1], 0, fh[2], fl[2], p); });
        r1 = ldw(wl, 2, 0, lane); FENCE();
        step3(r2, fh[1], fl[1], OUT[0], OUT[0], [&](int p) { chunk_piece(IN[1], 1, fh[3], fl[3], p); });
    }
    r2 = ldw(wl, 3, 0, lane); FENCE();
    step3(r0, fh[1], fl[1], OUT[1], OUT[1], NoValu());
    r0 = ldw(wl, 2, 1, lane); FENCE();
    step3(r1, fh[2], fl[2], OUT[0], OUT[0], NoValu());
    r1 = ldw(wl, 3, 1, lane); FENCE();
    step3(r2, fh[3], fl[3], OUT[0], OUT[0], NoValu());
    tailA(); FENCE();
    step3(r0, fh[2], fl[2], OUT[1], OUT[1], NoValu());
    tailB(); FENCE();
}

template <int MODE>
__global__ void __launch_bounds__(512) k4(float* out, const f16x8* __restrict__ wsrc, const float* __restrict__ rows_g,
                                          const int* __restrict__ idx, const float* __restrict__ proj, int iters) {
    __shared__ f16x8 w[4][2 * 2 * 4 * 64];
    __shared__ f16x8 wb[2][2][64];                    // bias operands [layer 2|3][ob][lane]
    for (int i = threadIdx.x; i < 4 * 2 * 2 * 4 * 64; i += blockDim.x) w[i / 1024][i % 1024] = wsrc[i % 1024];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) wb[i >> 7][(i >> 6) & 1][i & 63] = wsrc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, wave = threadIdx.x >> 6;
    const int item = (blockIdx.x * 8 + wave) * 32 + (lane & 31);
    f32x16 acc[2], bpr[2];
    for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) { acc[ob][r] = 0.f; bpr[ob][r] = 0.01f * r; }
    f32x16 zero;
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    u32x4 ones;
    ones[0] = h ? 0u : 0x3c003c00u; ones[1] = 0; ones[2] = 0; ones[3] = 0;
    W2 r0, r1, r2;
    r0 = ldw(w[0], 0, 0, lane); r1 = ldw(w[0], 0, 1, lane);
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        f32x16 sv[2];
        if (!(MODE & 1)) {
            const int j = idx[(item * 10 + (it % 10)) & 0xfffff];
            const float* row = proj + (size_t)j * 128 + 64;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = *reinterpret_cast<const float4*>(row + 32 * ob + 8 * g + 4 * h);
                    sv[ob][4 * g] = t.x; sv[ob][4 * g + 1] = t.y; sv[ob][4 * g + 2] = t.z; sv[ob][4 * g + 3] = t.w;
                }
        } else {
            for (int ob = 0; ob < 2; ++ob) for (int r = 0; r < 16; ++r) sv[ob][r] = 0.001f * (r + it);
        }
        // layer-1 operand: 8 inputs of half 0 (no ReLU: differences are signed)
        u32x4 xh, xl;
        {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = h ? 0.f : 0.01f * (lane + q + it);
#pragma unroll
            for (int q = 0; q < 4; ++q) { xh[q] = pk_rtz(x[2 * q], x[2 * q + 1]); xl[q] = mix_resid(xh[q], x[2 * q], x[2 * q + 1]); }
        }
        u32x4 fh[4], fl[4];
        f32x16 a[2], c[2];
        FENCE();
        // first layer: two groups of one k-step; the split of block 0 rides on block 1's MFMAs
        r2.hi = wb[0][0][lane]; r2.lo = wb[0][1][lane]; FENCE();
        if (MODE & 4) {
            for (int r = 0; r < 16; ++r) { a[0][r] = 0.01f * (r + it) + __builtin_bit_cast(float, xh[0]); a[1][r] = 0.02f * (r + it); }
            r0 = ldw(w[1], 0, 0, lane); FENCE();
            chunk_piece(a[0], 0, fh[0], fl[0], 0); chunk_piece(a[0], 0, fh[0], fl[0], 1); chunk_piece(a[0], 0, fh[0], fl[0], 2); FENCE();
        } else {
        a[0] = MF(r0.lo, asf16(xh), zero); a[0] = MF(r0.hi, asf16(xl), a[0]); a[0] = MF(r0.hi, asf16(xh), a[0]); FENCE();
        r0 = ldw(w[1], 0, 0, lane); FENCE();
        step3(r1, xh, xl, a[1], zero, [&](int p) { chunk_piece(a[0], 0, fh[0], fl[0], p); });
        }
        layer64<true>(w[1], lane, r0, r1, r2, fh, fl, a, c, zero, zero, ones, zero,
                      [&]() { r2.hi = wb[1][0][lane]; r2.lo = wb[1][1][lane]; }, [&]() { r0 = ldw(w[2], 0, 0, lane); });
        step3(r1, fh[3], fl[3], c[1], c[1], [&](int p) { chunk_piece(c[0], 0, fh[0], fl[0], p); });
        layer64<true>(w[2], lane, r0, r1, r2, fh, fl, c, a, zero, zero, ones, zero,
                      [&]() {}, [&]() { r0 = ldw(w[3], 0, 0, lane); });
        step3(r1, fh[3], fl[3], a[1], a[1], [&](int p) { chunk_piece(a[0], 0, fh[0], fl[0], p); });
        if (!(MODE & 8)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sv[0][r] += bpr[0][r]; sv[1][r] += bpr[1][r]; }
        }
        FENCE();
        W2 n0, n1;
        layer64<false>(w[3], lane, r0, r1, r2, fh, fl, a, c, sv[0], sv[1], ones, zero,
                       [&]() { n0 = ldw(w[0], 0, 0, lane); }, [&]() { n1 = ldw(w[0], 0, 1, lane); });
        if (MODE & 2) {
            step3(r1, fh[3], fl[3], c[1], c[1], NoValu());
            r0 = n0; r1 = n1;
            acc[0][it & 15] += c[0][0] + c[1][1];
        } else {
        step3(r1, fh[3], fl[3], c[1], c[1], [&](int p) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (r % 3 == p) acc[0][r] += relu1(c[0][r]);
        });
        r0 = n0; r1 = n1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][r] += relu1(c[1][r]);
        }
        FENCE();
    }
    float keep = 0.f;
    for (int r = 0; r < 16; ++r) keep += acc[0][r] + acc[1][r];
    if (keep == 123.456f) out[0] = keep;
}

template <int MODE>
void run4(const char* name, float* d, f16x8* w, float* rows, int* idx, float* proj, int threads = 512) {
    const int iters = 3000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k4<MODE>, dim3(256), dim3(threads), 0, 0, d, w, rows, idx, proj, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k4<MODE>, dim3(256), dim3(threads), 0, 0, d, w, rows, idx, proj, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s threads %d: %8.0f cycles per slot per SIMD (at 2.4 GHz)\n", name, threads, ms * 1e-3 * 2.4e9 / iters / (threads / 256));
}


// 16-item tile: the 3-layer chain on v_mfma_f32_16x16x32_f16 with four 4-register accumulators per layer
// (half the registers of the 32-item tile), launched with 2, 3 or 4 waves per SIMD.  Same FLOPs per item, twice the
// LDS weight reads per item.  B operand of k-step ks, lane group g = lane >> 4: k = 8g + jj <-> registers r = jj & 3 of
// output blocks 2ks + (jj >> 2).
template <int WPS>
__global__ void __launch_bounds__(256 * WPS) k5(float* out, const f16x8* __restrict__ wsrc, int iters) {
    __shared__ f16x8 w[2 * 4 * 2 * 64];            // one layer: [part][ob 4][ks 2][lane]
    for (int i = threadIdx.x; i < 2 * 4 * 2 * 64; i += blockDim.x) w[i] = wsrc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8 bh[2], bl[2];
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 8; ++j) { bh[s][j] = (_Float16)(0.01f * (lane + j + s)); bl[s][j] = (_Float16)1e-5f; }
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int layer = 0; layer < 3; ++layer) {
            f32x4 acc[4];
            for (int ob = 0; ob < 4; ++ob) for (int r = 0; r < 4; ++r) acc[ob][r] = 0.1f;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f16x8 whi = w[((0 * 4 + ob) * 2 + ks) * 64 + lane], wlo = w[((1 * 4 + ob) * 2 + ks) * 64 + lane];
                    acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, bh[ks], acc[ob], 0, 0, 0);
                    acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, bl[ks], acc[ob], 0, 0, 0);
                    acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, bh[ks], acc[ob], 0, 0, 0);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ob = 2 * ks + (q >> 1), r = 2 * (q & 1);
                    split_pair(relu1(acc[ob][r]), relu1(acc[ob][r + 1]), bh[ks], bl[ks], q);
                }
        }
        keep += (float)bh[0][0];
    }
    if (keep == 123.456f) out[0] = keep;
}

template <int WPS>
void run5(float* d, f16x8* w) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k5<WPS>, dim3(256), dim3(256 * WPS), 0, 0, d, w, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k5<WPS>, dim3(256), dim3(256 * WPS), 0, 0, d, w, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("16-item tile, 16x16x32 f16, %d waves/SIMD: %8.1f cycles per item per SIMD (at 2.4 GHz) [32-item tile, 2 waves: see first block / 64]\n", WPS,
           ms * 1e-3 * 2.4e9 / iters / (16.0 * WPS));
}

template <int MODE>
void run3(const char* name, float* d, f16x8* w, float* rows, int* idx, float* proj) {
    const int iters = 3000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3<MODE>, dim3(256), dim3(512), 0, 0, d, w, rows, idx, proj, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k3<MODE>, dim3(256), dim3(512), 0, 0, d, w, rows, idx, proj, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s waves/SIMD 2: %8.0f cycles per slot per SIMD (two waves; at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / iters / 2);
}

template <int MODE>
void run2(const char* name, float* d, f16x8* w, float* rows, int* idx, float* proj) {
    const int iters = 3000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k2<MODE>, dim3(256), dim3(512), 0, 0, d, w, rows, idx, proj, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k2<MODE>, dim3(256), dim3(512), 0, 0, d, w, rows, idx, proj, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s waves/SIMD 2: %8.0f cycles per slot per SIMD (two waves; at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / iters / 2);
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* d, f16x8* w) {
    const int iters = 4000, threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, w, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, w, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s waves/SIMD %d: %8.0f cycles per 3-layer chain per SIMD (at 2.4 GHz)\n", name, waves_per_simd,
           ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    float* d; f16x8* w;
    hipMalloc(&d, 4); hipMalloc(&w, 2 * 2 * 4 * 64 * 16);
    hipMemset(w, 0x11, 2 * 2 * 4 * 64 * 16);
    for (int wv = 1; wv <= 2; ++wv) {
        run<0>("32x32x16 f16, 2 accumulators", wv, d, w);
        run<1>("16x16x32 f16, 8 accumulators", wv, d, w);
    }
    float *rows, *proj; int* idx;
    hipMalloc(&rows, 1024); hipMemset(rows, 0, 1024);
    const size_t nrows = 307200;                      // 1024 samples x 300 particles
    hipMalloc(&proj, nrows * 128 * 4); hipMemset(proj, 0, nrows * 128 * 4);
    hipMalloc(&idx, (1 << 20) * 4);
    {
        std::vector<int> hi(1 << 20);
        for (int i = 0; i < (1 << 20); ++i) hi[i] = (int)(((long)i * 2654435761u) % nrows);
        hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    }
    run2<0>("slot shape: 78 MFMAs, LDS bias rows, epilogue, sv in registers", d, w, rows, idx, proj);
    run2<1>("  + index -> gathered sv row from global memory", d, w, rows, idx, proj);
    run2<2>("  without the epilogue", d, w, rows, idx, proj);
    run2<4>("  without the first layer", d, w, rows, idx, proj);
    run2<8>("  bias rows as constants instead of LDS reads", d, w, rows, idx, proj);
    run2<14>("  without all three", d, w, rows, idx, proj);
    run3<0>("restructured: zero-start accumulators, bias in the split", d, w, rows, idx, proj);
    run3<2>("  + next slot's first layer before the epilogue", d, w, rows, idx, proj);
    run3<3>("  + gathered sv", d, w, rows, idx, proj);
    run5<2>(d, w); run5<3>(d, w); run5<4>(d, w);
    run4<0>("hand-ordered slot, mixlo/mixhi split, bias MFMAs, gathered sv", d, w, rows, idx, proj);
    run4<0>("  one wave per SIMD", d, w, rows, idx, proj, 256);
    run4<1>("  sv from registers", d, w, rows, idx, proj);
    run4<2>("  no epilogue", d, w, rows, idx, proj);
    run4<4>("  no first layer", d, w, rows, idx, proj);
    run4<8>("  no sv + bpr", d, w, rows, idx, proj);
    run4<15>("  none of the four", d, w, rows, idx, proj);
    run4<15>("  none of the four, one wave per SIMD", d, w, rows, idx, proj, 256);
    return 0;
}
