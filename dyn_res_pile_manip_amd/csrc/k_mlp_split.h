// Split-precision MFMA version of the relation encoder chain.
//
// fp32 MFMA runs at 1/16 of the 16-bit MFMA rate on gfx950 and there is no TF32 path, so the
// fp32 chain of k_mlp_mfma.h is bound by the matrix pipe (69 % of the fp32 peak measured).
// Here every fp32 operand is written as a sum of two fp16 pieces, x = x_hi + x_lo, weights are
// split the same way on the host, and each product is formed as
//     W_lo x_hi + W_hi x_lo + W_hi x_hi
// with fp32 accumulation inside v_mfma_f32_32x32x16_f16 (a product of two fp16 values has 22
// significant bits: exact in fp32).  x_hi is x rounded toward zero to fp16 (v_cvt_pkrtz_f16_f32,
// two values per instruction), the residual x - x_hi is exact in fp32 and is itself rounded to
// fp16, so |x - x_hi - x_lo| < 2^-20 |x|; the dropped W_lo x_lo term is < 2^-21 relative.  The
// residuals are fp16 SUBNORMALS for |x| < 0.06: the matrix cores honour them (checked operand by
// operand, tools/f16_denorm_test.hip).  3 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64 cycles: 5.3x less
// matrix-pipe time.
//
// RANGE.  fp16 tops out at 65 504 (x_hi would saturate: round toward zero never produces inf -- a wrong but
// finite result) and loses the residual's bits below 6e-5.  The hidden activations are therefore carried
// through the chain multiplied by an exact power of two, 2^k, chosen on the host from the weights
// (split_range_shift below: the largest k that keeps a proven bound of every hidden activation under 2^15):
//     W1' = 2^k W1, b1' = 2^k b1, b2' = 2^k b2, b3' = 2^k b3   (host, exact)      h_l' = 2^k h_l
//     c   = 2^-k (2^k (b + d w_d + P_r) + W_e h3')                                  (one fma where an add was)
// ReLU is positively homogeneous and scaling by 2^k commutes with every fp32 rounding, so the result is the
// unscaled chain's bit for bit wherever that one neither saturates nor touches subnormals, and correct
// beyond: weights 1e7 times larger than a trained network's still give the fp32 engines' answer.  A call
// whose inputs exceed the envelope the bound was proven for is refused with DRP_ERANGE (range_check, capi_pipeline.h).
// (The first version split into bf16 pairs: same MFMA count, but v_cvt_pk_bf16_f32 issues at about a
// third of the rate of v_cvt_pkrtz_f16_f32 -- tools/mfma_bench.hip -- and carries 3 bits less.)
// The node layers (6-term split further down) stay on three bf16 pieces.
//
// Same transposed register chain as k_mlp_mfma.h.  For v_mfma_f32_32x32x16_{f16,bf16} the B
// operand of lane (col j, half h) is 8 consecutive k: k = 8h + jj; the C/D registers
// 8(s&1)..8(s&1)+7 of output block ob = s>>1 are fed as k-step s, so the weights are packed
// with   feature(s,h,jj) = 32(s>>1) + (r&3) + 8(r>>2) + 4h,  r = 8(s&1) + jj.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "k_mlp_mfma.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

// ---- packed split weights of the edge chain, in units of f16x8 (16 bytes).  64x64: [part 2][ob 2][s 4][lane 64]
enum {
    S_RE0 = 0,                 // first layer, one k-step: [part 2][ob 2][lane 64]
    S_RE2 = S_RE0 + 256,
    S_RE4 = S_RE2 + 1024,
    S_RPE = S_RE4 + 1024,
    S_TOTAL = S_RPE + 1024,    // x 16 bytes
    S_ROWS = S_TOTAL,          // then 256 floats: 2^k b2, 2^k b4, b_rp, wd_rp (the chain's bias rows, scaled)
    S_ALLOC = S_ROWS + 64      // units in the device buffer
};

// ---- range of the split relation encoder (host) ------------------------------------------------
// Bound of the hidden activations for inputs |attr| <= A, |s_r - s_s| <= D per coordinate, d <= dm:
//   h1 <= max_o (|w_o0| + |w_o1|) A + (|w_o2| + |w_o3| + |w_o4|) D + |w_o5| dm + |b_o|,
//   h_{l+1} <= max_o sum_k |W_l[o,k]| h_l + |b_l[o]|.
struct SplitRange {
    float a1[64], d1[64], m1[64], b1[64];     // first layer: row coefficients of A, D, dm, |bias|
    float rs2[64], b2[64], rs4[64], b4[64];   // row abs sums and |bias| of the two hidden layers
    float wmax;                               // largest |weight| of the unscaled matrices packed as fp16
    bool finite;                              // no NaN / Inf among the relation encoder's and propagator's parameters
    int shift;                                // k
    float env_attr, env_delta, env_dens;      // the envelope k was chosen for
};
inline double split_range_bound(const SplitRange& r, double A, double D, double dm) {
    double h1 = 0, h2 = 0, h3 = 0;
    for (int o = 0; o < 64; ++o) h1 = fmax(h1, r.a1[o] * A + r.d1[o] * D + r.m1[o] * dm + r.b1[o]);
    for (int o = 0; o < 64; ++o) h2 = fmax(h2, r.rs2[o] * h1 + r.b2[o]);
    for (int o = 0; o < 64; ++o) h3 = fmax(h3, r.rs4[o] * h2 + r.b4[o]);
    return fmax(fmax(h1, h2), fmax(h3, 1.0));      // 1.0: the constant input column of the first layer
}
// k: 2^k bound <= 2^15, a factor two under fp16's end.  floor(log2(32768 / bound)) read off the quotient's exponent field
// (exact, and the same instruction sequence on host and device: kt_repack_all derives the shift after an optimiser step)
#define SPLIT_ENV_ATTR 2.0
#define SPLIT_ENV_DELTA 1.5
#define SPLIT_ENV_DENS 2.0
__host__ __device__ inline int range_shift_of(double bound) {
    const double x = 32768.0 / bound;
    if (!(x > 0.0)) return -60;                       // bound = inf or NaN
    unsigned long long u;
    memcpy(&u, &x, sizeof(u));
    const int e = (int)((u >> 52) & 0x7ffu);
    if (e == 0x7ff) return 14;                        // bound = 0 cannot happen (>= 1.0), kept total
    int k = e - 1023;                                 // subnormal quotients read -1023: clamped below
    if (k > 14) k = 14;
    if (k < -60) k = -60;
    return k;
}
inline void split_range_init(const float* w, SplitRange& r, double A, double D, double dm) {
    r.wmax = 0.0f;
    // fmaxf / fmax drop a NaN operand, so the sums and maxima below would let NaN weights through as a finite bound:
    // look for them explicitly
    r.finite = true;
    const int spans[][2] = {{W_RE0_W, 64 * 6}, {W_RE0_B, 64}, {W_RE2_W, 4096}, {W_RE2_B, 64}, {W_RE4_W, 4096}, {W_RE4_B, 64},
                            {W_RP_W, 64 * 193}, {W_RP_B, 64}};
    for (const auto& sp : spans)
        for (int i = 0; i < sp[1]; ++i)
            if (!std::isfinite(w[sp[0] + i])) r.finite = false;
    for (int o = 0; o < 64; ++o) {
        const float* w1 = w + W_RE0_W + o * 6;
        r.a1[o] = fabsf(w1[0]) + fabsf(w1[1]);
        r.d1[o] = fabsf(w1[2]) + fabsf(w1[3]) + fabsf(w1[4]);
        r.m1[o] = fabsf(w1[5]);
        r.b1[o] = fabsf(w[W_RE0_B + o]);
        float s2 = 0, s4 = 0;
        for (int k = 0; k < 64; ++k) {
            s2 += fabsf(w[W_RE2_W + o * 64 + k]);
            s4 += fabsf(w[W_RE4_W + o * 64 + k]);
            r.wmax = fmaxf(r.wmax, fmaxf(fabsf(w[W_RE2_W + o * 64 + k]), fmaxf(fabsf(w[W_RE4_W + o * 64 + k]), fabsf(w[W_RP_W + o * 193 + k]))));
        }
        r.rs2[o] = s2; r.b2[o] = fabsf(w[W_RE2_B + o]);
        r.rs4[o] = s4; r.b4[o] = fabsf(w[W_RE4_B + o]);
    }
    r.env_attr = (float)A; r.env_delta = (float)D; r.env_dens = (float)dm;
    const double bound = split_range_bound(r, A, D, dm);
    r.shift = range_shift_of(bound);
}

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

inline uint16_t host_f16_rne(float f) {      // host pass of hipcc is clang: _Float16 converts with round-to-nearest-even
    const _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

inline float host_f16_to_f32(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

inline int split_feature(int s, int h, int jj) {
    const int r = 8 * (s & 1) + jj;
    return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// host: state_dict blob -> split-fp16 fragments (uint16 storage, 8 per f16x8); `shift` = k of the RANGE note:
// first layer and the hidden biases carry 2^k
inline void pack_split(const float* w, std::vector<uint16_t>& out, int shift) {
    out.assign((size_t)S_ALLOC * 8, 0);
    float* srows = reinterpret_cast<float*>(out.data() + (size_t)S_ROWS * 8);
    for (int o = 0; o < 64; ++o) {
        srows[o] = ldexpf(w[W_RE2_B + o], shift);
        srows[64 + o] = ldexpf(w[W_RE4_B + o], shift);
        srows[128 + o] = w[W_RP_B + o];
        srows[192 + o] = w[W_RP_W + o * 193 + 192];
    }
    auto put = [&](int unit, int jj, int part, float v) {
        // unit = index of the hi f16x8; the lo copy sits `part_stride` units later (given by caller)
        (void)part;
        out[(size_t)unit * 8 + jj] = host_f16_rne(v);
    };
    auto P64 = [&](int dst, int src, int ld, int col0) {
        for (int ob = 0; ob < 2; ++ob)
            for (int s = 0; s < 4; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int i = lane & 31, h = lane >> 5;
                        const float v = w[src + (32 * ob + i) * ld + col0 + split_feature(s, h, jj)];
                        const float hi = host_f16_to_f32(host_f16_rne(v));
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
                v = ldexpf(v, shift);
                const float hi = host_f16_to_f32(host_f16_rne(v));
                put(S_RE0 + (0 * 2 + ob) * 64 + lane, jj, 0, hi);
                put(S_RE0 + (1 * 2 + ob) * 64 + lane, jj, 1, v - hi);
            }
    P64(S_RE2, W_RE2_W, 64, 0);
    P64(S_RE4, W_RE4_W, 64, 0);
    P64(S_RPE, W_RP_W, 193, 0);
}

struct FragB {
    f16x8 hi[4], lo[4];      // per 16-deep k-step
};

// two values -> their fp16 hi pair (round toward zero) and the fp16 pair of the exact residuals
__device__ __forceinline__ void split_pair(float x0, float x1, f16x8& hi, f16x8& lo, int q) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    // residuals x - hi in one v_fma_mix_f32 each (the fp16 half is read in place; written as asm because
    // the compiler folds the multiplication by -1 into a convert + subtract)
    float r0, r1;
    asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(r0), "=&v"(r1) : "v"(h), "v"(x0), "v"(x1));
    const fp16x2_t l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi[2 * q] = (_Float16)h[0]; hi[2 * q + 1] = (_Float16)h[1];
    lo[2 * q] = (_Float16)l[0]; lo[2 * q + 1] = (_Float16)l[1];
}

template <bool RELU>
__device__ __forceinline__ void split_frag(const Frag& in, FragB& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x0 = in.v[s >> 1][8 * (s & 1) + 2 * q], x1 = in.v[s >> 1][8 * (s & 1) + 2 * q + 1];
            if (RELU) { x0 = relu1(x0); x1 = relu1(x1); }
            split_pair(x0, x1, o.hi[s], o.lo[s], q);
        }
}

// acc += W x, W packed as f16x8[(part*2 + ob)*4 + s][lane]
// Packed-weight operands of one MFMA group (3 MFMAs: one k-step of one output block).
struct WOp {
    f16x8 hi, lo;
};

__device__ __forceinline__ WOp wop_load(const f16x8* __restrict__ wp, int s, int ob, int lane) {
    WOp w;
    w.hi = wp[((0 * 2 + ob) * 4 + s) * 64 + lane];
    w.lo = wp[((1 * 2 + ob) * 4 + s) * 64 + lane];
    return w;
}

__device__ __forceinline__ void mfma_group(const WOp& w, const FragB& b, Frag& acc, int s, int ob) {
    acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.lo, b.hi[s], acc.v[ob], 0, 0, 0);
    acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, b.lo[s], acc.v[ob], 0, 0, 0);
    acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.hi, b.hi[s], acc.v[ob], 0, 0, 0);
}

// acc += W x.  24 MFMAs in 8 groups; the LDS reads of group g+1 are issued before the MFMAs
// of group g (an un-prefetched ds_read_b128 pair costs more than the 96 cycles its three
// MFMAs take), `first` holds group 0 (loaded by the caller while the previous layer's
// output was being split) and `next` receives group 0 of the following layer.
// Group order: k-steps 0,1 for both output blocks (they need only the first half of the
// previous layer's split), then k-steps 2,3 of block 0, then of block 1.
__device__ __forceinline__ void mfma_layer64_split(const f16x8* __restrict__ wp, const FragB& b, Frag& acc, int lane,
                                                   const WOp& first, const f16x8* __restrict__ wp_next, WOp& next) {
    WOp w1 = wop_load(wp, 0, 1, lane);
    mfma_group(first, b, acc, 0, 0);
    WOp w2 = wop_load(wp, 1, 0, lane);
    mfma_group(w1, b, acc, 0, 1);
    w1 = wop_load(wp, 1, 1, lane);
    mfma_group(w2, b, acc, 1, 0);
    w2 = wop_load(wp, 2, 0, lane);
    mfma_group(w1, b, acc, 1, 1);
    w1 = wop_load(wp, 3, 0, lane);
    mfma_group(w2, b, acc, 2, 0);
    w2 = wop_load(wp, 2, 1, lane);
    mfma_group(w1, b, acc, 3, 0);
    w1 = wop_load(wp, 3, 1, lane);
    mfma_group(w2, b, acc, 2, 1);
    if (wp_next != nullptr) next = wop_load(wp_next, 0, 0, lane);
    mfma_group(w1, b, acc, 3, 1);
}

__device__ __forceinline__ void mfma_layer64_split(const f16x8* __restrict__ wp, const FragB& b, Frag& acc, int lane) {
    WOp first = wop_load(wp, 0, 0, lane), next;
    mfma_layer64_split(wp, b, acc, lane, first, nullptr, next);
}

// first layer: one k-step over [a_r, a_s, dx, dy, dz, d, 1, 0] (lanes of half 1 supply zeros)
__device__ __forceinline__ void mfma_layer8_split(const f16x8* __restrict__ wp, const float (&x)[8], int h, Frag& acc, int lane) {
    f16x8 bhi, blo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v0 = (h == 0) ? x[2 * q] : 0.0f, v1 = (h == 0) ? x[2 * q + 1] : 0.0f;
        split_pair(v0, v1, bhi, blo, q);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const f16x8 a_hi = wp[(0 * 2 + ob) * 64 + lane];
        const f16x8 a_lo = wp[(1 * 2 + ob) * 64 + lane];
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, bhi, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, blo, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, bhi, acc.v[ob], 0, 0, 0);
    }
}

// the relation-encoder chain of one tile: inputs -> c_edge fragment
__device__ __forceinline__ void edge_chain_split(const f16x8* __restrict__ wsp /*LDS, S_* offsets*/,
                                                 const float* __restrict__ rows /*2^k b2, 2^k b4, b_rp, wd_rp*/,
                                                 const float (&x)[8], float d, int h, int lane, float sc, float inv,
                                                 Frag& out) {
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
#pragma unroll
    for (int r = 0; r < 16; ++r) { out.v[0][r] *= sc; out.v[1][r] *= sc; }
    mfma_layer64_split(wsp + S_RPE, fb, out, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) { out.v[0][r] *= inv; out.v[1][r] *= inv; }
}

// same contract as km_edge_encode (k_mlp_mfma.h)
DRP_GLOBAL void __launch_bounds__(64 * MFMA_WAVES)
km_edge_encode_split(const uint16_t* __restrict__ sw, const float* __restrict__ mw,
                     const float* __restrict__ s_cur, int s_mod, size_t s_stride,
                     const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
                     const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt, int N, int B,
                     float* __restrict__ c_edge, float re_scale, float re_inv) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wsp_f = lds;                          // S_TOTAL * 4 floats
    float* rows = wsp_f + S_TOTAL * 4;           // 2^k b2, 2^k b4, b_rp, wd_rp
    float* tiles = rows + 256;
    lds_fill(wsp_f, reinterpret_cast<const float*>(sw), (S_TOTAL + 64) * 4);      // the rows follow the fragments
    __syncthreads();
    const f16x8* wsp = reinterpret_cast<const f16x8*>(wsp_f);
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
        edge_chain_split(wsp, rows, x, d, h, lane, re_scale, re_inv, c);
        const int rows_valid = min(32, nslots - t * 32);
        frag_store_tile(c, c_edge + ((size_t)b * nslots + (size_t)t * 32) * 64, 64, rows_valid, tile, lane);
    }
}

#define KM_EDGE_SPLIT_LDS ((S_TOTAL * 4 + 256 + MFMA_WAVES * TILE_FLOATS) * sizeof(float))

// ---- whole propagation step in one kernel -----------------------------------------------------
// km_prop<LAST>: per tile of 32 receivers
//     agg  = sum_k relu(c_edge_k + (W_r eff)[i] + (W_s eff)[send_k])   (chain recomputed, above)
//     eff  = relu(c_node + W_agg agg + eff)                             gnn_dyn.py:191-193
//     !LAST: proj_next = [W_r eff | W_s eff]    (next step's node terms; ping-pong buffer, other
//                                                tiles still gather this step's rows)
//     LAST : s_pred = W1 relu(W0 eff + b0) + b1 + s_cur                 gnn_dyn.py:196-198
// agg never leaves the registers: the accumulator layout of the segmented sum is the
// B-operand layout of the next MFMA.  The node layers use a three-way bf16 split with six
// products (W_hi x_hi, W_hi x_mid, W_mid x_hi, W_mid x_mid, W_hi x_lo, W_lo x_hi): error
// <= 2^-24 relative per product, indistinguishable from the fp32 chain (measured 6.4e-7 vs
// 5.9e-7 on the displacement), at 6/16 of the fp32 MFMA time.
enum {                        // units of bf16x8; 64x64: [part 3][ob 2][s 4][lane 64]
    S6_AGG = 0,
    S6_RPR = S6_AGG + 1536,
    S6_RPS = S6_RPR + 1536,
    S6_PR0 = S6_RPS + 1536,
    S6_PE2 = S6_PR0 + 1536,
    S6_PPE = S6_PE2 + 1536,
    S6_PE0 = S6_PPE + 1536,      // first layer, one k-step: [part 3][ob 2][lane 64]
    S6_TOTAL = S6_PE0 + 384
};

inline void pack_split6(const float* w, std::vector<uint16_t>& out) {
    out.assign((size_t)S6_TOTAL * 8, 0);
    auto P = [&](int dst, int src, int ld, int col0) {
        for (int ob = 0; ob < 2; ++ob)
            for (int s = 0; s < 4; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int i = lane & 31, h = lane >> 5;
                        float v = w[src + (32 * ob + i) * ld + col0 + split_feature(s, h, jj)];
                        for (int part = 0; part < 3; ++part) {
                            const uint16_t q = host_bf16_rne(v);
                            out[((size_t)dst + ((part * 2 + ob) * 4 + s) * 64 + lane) * 8 + jj] = q;
                            v -= host_bf16_to_f32(q);
                        }
                    }
    };
    P(S6_AGG, W_PP_W, 129, 64);
    P(S6_RPR, W_RP_W, 193, 64);
    P(S6_RPS, W_RP_W, 193, 128);
    P(S6_PR0, W_PR0_W, 64, 0);
    P(S6_PE2, W_PE2_W, 64, 0);
    P(S6_PPE, W_PP_W, 129, 0);
    // particle encoder layer 0: inputs [sdx, sdy, sdz, a, d, 1(bias), 0, 0], k = 8h + jj, h = 1 unused
    for (int ob = 0; ob < 2; ++ob)
        for (int lane = 0; lane < 64; ++lane)
            for (int jj = 0; jj < 8; ++jj) {
                const int i = lane & 31, h = lane >> 5, o = 32 * ob + i;
                float v = 0.0f;
                if (h == 0 && jj < 5) v = w[W_PE0_W + o * 5 + jj];
                else if (h == 0 && jj == 5) v = w[W_PE0_B + o];
                for (int part = 0; part < 3; ++part) {
                    const uint16_t q = host_bf16_rne(v);
                    out[((size_t)S6_PE0 + (part * 2 + ob) * 64 + lane) * 8 + jj] = q;
                    v -= host_bf16_to_f32(q);
                }
            }
}

struct FragB6 {
    bf16x8 p[3][4];          // [part][k-step]
};

__device__ __forceinline__ void split_frag6(const Frag& in, FragB6& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float x = in.v[s >> 1][8 * (s & 1) + jj];
            const __bf16 hi = (__bf16)x;
            const float r1 = x - (float)hi;
            const __bf16 mid = (__bf16)r1;
            o.p[0][s][jj] = hi;
            o.p[1][s][jj] = mid;
            o.p[2][s][jj] = (__bf16)(r1 - (float)mid);
        }
}

__device__ __forceinline__ void mfma_layer64_split6(const bf16x8* __restrict__ wp, const FragB6& b, Frag& acc, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const bf16x8 w0 = wp[((0 * 2 + ob) * 4 + s) * 64 + lane];
            const bf16x8 w1 = wp[((1 * 2 + ob) * 4 + s) * 64 + lane];
            const bf16x8 w2 = wp[((2 * 2 + ob) * 4 + s) * 64 + lane];
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b.p[0][s], acc.v[ob], 0, 0, 0);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b.p[2][s], acc.v[ob], 0, 0, 0);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b.p[1][s], acc.v[ob], 0, 0, 0);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b.p[0][s], acc.v[ob], 0, 0, 0);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b.p[1][s], acc.v[ob], 0, 0, 0);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b.p[0][s], acc.v[ob], 0, 0, 0);
        }
    }
}

// first layer of the particle encoder on the 6-term split (km_node_encode_split, and km_prop3's phase E)
__device__ __forceinline__ void mfma_layer8_split6(const bf16x8* __restrict__ wp, const float (&x)[8], int h, Frag& acc, int lane) {
    bf16x8 b0, b1, b2;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const float v = (h == 0) ? x[jj] : 0.0f;
        const __bf16 hi = (__bf16)v;
        const float r1 = v - (float)hi;
        const __bf16 mid = (__bf16)r1;
        b0[jj] = hi;
        b1[jj] = mid;
        b2[jj] = (__bf16)(r1 - (float)mid);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const bf16x8 w0 = wp[(0 * 2 + ob) * 64 + lane];
        const bf16x8 w1 = wp[(1 * 2 + ob) * 64 + lane];
        const bf16x8 w2 = wp[(2 * 2 + ob) * 64 + lane];
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b0, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b2, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b1, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b0, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b1, acc.v[ob], 0, 0, 0);
        acc.v[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b0, acc.v[ob], 0, 0, 0);
    }
}

#ifndef PROP_WAVES
#define PROP_WAVES 8
#endif
// Self-edge constant.  The self loop i <- i feeds the relation encoder [a_i, a_i, 0, 0, 0, d]
// (gnn_dyn.py:179-180 with s_r - s_s = 0): when a sample's attributes are all equal (they are
// zeros on the whole MPC path, env/flex_env.py:1044) that input, hence W_e . RelationEncoder(.),
// is ONE vector per sample, the same at every rollout step and propagation step.  k_cself
// evaluates it once per rollout in plain fp32 (lane = feature) and flags the samples where it
// holds; km_prop then starts a receiver's aggregate from relu(c_self + bias + P_r + P_s[i])
// and skips the self slot: one relation-encoder chain in ten never runs.
DRP_GLOBAL void __launch_bounds__(64)
k_cself(const float* __restrict__ vw, const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens,
        int dens_mod, int N, float* __restrict__ cself, uint8_t* __restrict__ ok) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* at = attr + (size_t)(b % attr_mod) * N;
    const float a = at[0];
    bool same = true;
    for (int i = lane; i < N; i += 64) same = same && (at[i] == a);
    const bool uniform = __all(same);
    const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
    float h = vw[V_RE0_B + lane];
    h = fmaf(a, vw[V_RE0_T + 0 * 64 + lane], h);
    h = fmaf(a, vw[V_RE0_T + 1 * 64 + lane], h);
    h = fmaf(d, vw[V_RE0_T + 5 * 64 + lane], h);
    float x = fmaxf(h, 0.0f);
    h = vw[V_RE2_B + lane];
    for (int k = 0; k < 64; ++k) h = fmaf(__shfl(x, k, 64), vw[V_RE2_T + k * 64 + lane], h);
    x = fmaxf(h, 0.0f);
    h = vw[V_RE4_B + lane];
    for (int k = 0; k < 64; ++k) h = fmaf(__shfl(x, k, 64), vw[V_RE4_T + k * 64 + lane], h);
    x = fmaxf(h, 0.0f);
    h = 0.0f;
    for (int k = 0; k < 64; ++k) h = fmaf(__shfl(x, k, 64), vw[V_RPE_T + k * 64 + lane], h);
    cself[(size_t)b * 64 + lane] = h;
    if (lane == 0) ok[b] = uniform ? 1 : 0;
}

// ReLU mask of an edge's 64 relation-effect features, as the gradient-descent planner's and the
// trainer's backward pass want it (k_backward.h): two 32-bit words per edge slot, word h = the
// half-wave that holds the features, bit 31 - (16*ob + r) <-> accumulator register r of output
// block ob, i.e. feature 32*ob + (r&3) + 8*(r>>2) + 4*h.
// m <- (m << 1) | [t > 0] for a relu'd t (> 0 <=> its bits != 0 <=> 0 - bits is negative): a subtraction and one
// v_alignbit_b32 ({m, s} >> 31).  Written in C (shift, or, compare) the compiler makes a compare, a select through an SGPR
// pair, a shift and an OR of it -- six issue slots per element with the s_nops between them, a fifth of the tape-writing
// kernel's slot loop.  (The word only ever goes to a store: no matrix instruction reads what these statements write.)
__device__ __forceinline__ unsigned push_positive_bit(unsigned m, float t) {
    unsigned s;
    asm("v_sub_u32 %1, 0, %2\n\tv_alignbit_b32 %0, %0, %1, 31" : "+v"(m), "=&v"(s) : "v"(t));
    return m;
}
__device__ __forceinline__ unsigned frag_positive_bits(const Frag& f) {
    unsigned m = 0;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = push_positive_bit(m, f.v[ob][r]);
    return m;
}

#ifdef PROP_STAMPS
// Diagnostic build only (tools/prop_stamps.py): shader-clock and 100 MHz wall stamps around the slot loop and the
// node part of every tile, summed here; nothing the kernel computes reads them.
__device__ unsigned long long g_prop_span[4096 * 2];     // last launch: wall stamps (100 MHz) at wave entry / exit
__device__ unsigned long long g_prop_stamps[4096 * 8];   // [workgroup * waves + wave][8], summed over launches by the wave itself
#endif

// ---- the tile loop of a propagation step, shared by km_prop (one step per launch, tiles of all samples dealt
// over the chip) and km_prop3 (the three steps of a rollout step in one launch, a workgroup owning whole samples)
struct PropArgs {
    const float* mw;
    const float* s_cur; int s_mod; size_t s_stride;
    const float* attr; int attr_mod;
    const float* dens; int dens_mod;
    const int16_t* nbr_idx; const uint8_t* nbr_cnt;
    const float* proj; const float* c_node; const float* eff_in; float* eff;
    int N, B;
    float* proj_next; float* s_out; size_t out_stride;
    const float* cself; const uint8_t* cself_ok;
    unsigned* mask_out; float* agg_out;
    float re_scale, re_inv;      // 2^k and 2^-k of the relation encoder's range shift
    float4* ecache;              // EC != 0: this workgroup's edge-chain cache, [tile][iteration][8][64 lanes] float4
    unsigned long long* work;    // WORK: counters of what the launch executes (PROP_WORK_*, drp_probe_work)
};
// What the WORK instantiations of the propagation kernels count (one wave-uniform pass over the tile's in-degrees and two
// atomic adds per tile): the roofline's numerator is what the kernel EXECUTED, read from here.  Instantiations of their own,
// launched for the iteration drp_probe_begin("prop+work") brackets: the same tiles, orders and loops as the kernels they
// shadow, whose code (and register allocation: the 300-particle launch is 8 % slower with a dormant branch in it) stays as it is.
enum {
    PROP_WORK_CHAIN_SLOTS = 0,   // slot iterations that ran the relation encoder's chain (78 MFMAs: 6 + 3 x 24)
    PROP_WORK_CACHED_SLOTS,      // slot iterations served by the edge-chain cache (no MFMA)
    PROP_WORK_TILES,             // node parts of a propagation step that is not the last (144 MFMAs: W_agg, W_r, W_s, 6-term split)
    PROP_WORK_TILES_LAST,        // node parts of the last step (96: W_agg, predictor layer 0)
    PROP_WORK_ENC_TILES,         // particle-encoder tiles inside the launch (204: 12 + 4 x 48)
    PROP_WORK_COUNT
};
// the counters are kept in PROP_WORK_SHARDS copies, a workgroup adding to copy blockIdx.x % PROP_WORK_SHARDS (every tile of the
// chip adding to ONE address costs a fifth of the launch: the adds of one address are serial at the memory side)
#define PROP_WORK_SHARDS 256
#define PROP_WORK_STRIDE 8                     // 64-bit words per copy: a 64-B line of its own
__device__ __forceinline__ unsigned long long* prop_work_shard(unsigned long long* work) {
    return work + (size_t)(blockIdx.x % PROP_WORK_SHARDS) * PROP_WORK_STRIDE;
}
// ... and in words 5 / 6 of its copy the shader-clock cycles (s_memtime) and the 100 MHz ticks (s_memrealtime) the workgroup's
// first thread saw between the launch's entry and its exit: their ratio over all workgroups is the clock the kernel ran at
// (bench.py's sclk_mhz_under_load; tools/clock_probe.hip is the same pair on a spin loop)
#define PROP_WORK_CLK_CYCLES 5
#define PROP_WORK_CLK_TICKS 6
struct WorkClock { unsigned long long c0, r0; };
__device__ __forceinline__ WorkClock work_clock_begin() {
    WorkClock w;
    w.c0 = __builtin_amdgcn_s_memtime();
    w.r0 = __builtin_amdgcn_s_memrealtime();
    return w;
}
__device__ __forceinline__ void work_clock_end(const WorkClock& w, unsigned long long* work) {
    if (threadIdx.x == 0 && work) {
        atomicAdd(prop_work_shard(work) + PROP_WORK_CLK_CYCLES, (unsigned long long)(__builtin_amdgcn_s_memtime() - w.c0));
        atomicAdd(prop_work_shard(work) + PROP_WORK_CLK_TICKS, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - w.r0));
    }
}
#define PROP_MFMA_CHAIN 78
#define PROP_MFMA_NODE 144
#define PROP_MFMA_NODE_LAST 96
#define PROP_MFMA_ENC 204
struct PropLds {
    const f16x8* wsp;         // edge chain, S_* offsets
    const bf16x8* w_agg;      // W_agg
    const bf16x8* w_x;        // !LAST: W_r | W_s;  LAST: predictor layer 0
    const float* rows;        // 2^k b2, 2^k b4, b_rp, wd_rp
    const float* rows_pr;     // LAST: b_pr0, w_pr1[3], b_pr1
    int* tile_ctr;            // next tile of this workgroup's share
};
struct TileId {
    bool valid;
    int b, t;                 // km_prop: sample and tile of the sample; km_prop3: t = tile of the workgroup's rows
};
// Which receiver a lane of a tile works on.  km_prop's tiles are 32 consecutive receivers of ONE sample (b uniform);
// km_prop3's are 32 entries of its workgroup's row order, which runs over all of the workgroup's samples, so the
// sample, and with it the position / attribute / projection bases and the density, are per-lane quantities.
struct LaneRow {
    int b, i;                 // sample, receiver within the sample
    bool live;                // false: a clamped duplicate past the end (computed, never stored)
};

// CARRY (the whole-sample kernels of small workgroups): what a wave's FIRST tile needs before its first slot -- in-degree,
// first neighbours, own and first senders' positions: a chain of three dependent round trips, which later tiles have
// requested during the previous tile's node part -- does not change between the propagation steps of a rollout step (the
// lists and the positions are the step's, only the projections move).  The last tile's node part of step p requests it
// again for the first tile of step p + 1 and hands it over the workgroup barrier in registers: the first tile of steps 2
// and 3 starts like any other (1024 samples x 32 particles + 6 %, x 20 / 50 + 1 %; the same bits).
struct HeadCarry {
    bool ready;
    int b, i, live, cnt, ok, ks, j0, j1;
    const float* s; const float* at;
    float d, pix, piy, piz, pia, p0x, p0y, p0z, p0a;
    unsigned nbw0, nbw1, nbw2;
    // ROWS (the cached kernels with one tile per wave): what a tile's node part produces for its OWN rows -- the receiver's
    // projection P_r, its own sender projection P_s (the self loop's term) and its effect -- is what the same wave's same
    // tile starts the next propagation step with: handed over in registers instead of stored, waited for at the workgroup
    // barrier and loaded again (two round trips of 1.5 - 2 us per step where a step is 10 - 25 us; the same values, the
    // same bits).  rows_ok: the workgroup has no more tiles than waves (a wave keeps its tile: first_of); rows_ready: filled.
    bool rows_ok, rows_ready;
    Frag rpr, rps, re;
};
// PAIR: a tile is 16 receivers, and the 32 item columns of the chain are 16 receivers x two CONSECUTIVE slots -- column j
// belongs to receiver (j & 7) + 8 (j >> 4) and runs the slots of parity (j >> 3) & 1 -- so that a tile needs half the
// slot iterations.  For batches of so few rows that most waves of the chip would have no tile at all (a workgroup's 128
// rows are four tiles of 32 for eight waves; a training batch is one tile per CU) this halves the dependent chain that a
// tile's latency is made of.  The receiver's lane adds its own column's term, then its partner's (eight lanes on, the
// same DPP row: `row_ror:8` inside the add): slot k before slot k + 1, the order of the unpaired loop -- the same bits.
__device__ __forceinline__ float dpp_ror8(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
}
// EC (edge-chain cache; the whole-sample kernels of small piles).  The relation encoder's chain W_e . RelationEncoder(.) of an
// edge depends on the rollout step's positions only -- model/gnn_dyn.py:179-180 computes it ONCE, in front of the loop over the
// propagation steps (:182-193) -- so a workgroup that owns its samples can run it in the first propagation step only
// (EC = 1: the chain starts from a zero accumulator and its raw output q = 2^k W_e h_3 goes to a workgroup-private buffer, in
// the accumulator's own layout: eight coalesced 1-KB stores per slot iteration) and read it back in the other two (EC = 2: no
// chain, no matrix work in the slot loop at all -- q and the sender's row are loaded two iterations ahead), with
//     relu(q 2^-k + ((b + d w_d + P_r[i]) + P_s[j]))
// in all three steps.  That differs from the recomputing loop (EC = 0: the bias and P_r ride in the chain's initial
// accumulator) in the last place of a sum's rounding: EC kernels are compared with the oracle, not bit for bit with EC = 0.
// Pays where the chain's LATENCY is the bound -- one tile per wave -- and the buffer (2.5 KB per receiver) stays in the
// last-level cache; at 300 particles x 1024 samples it is 700 MB per rollout step and recomputing costs the same.
#define EC_UNITS 512             // float4 per (tile, slot iteration): 8 per lane
template <bool LAST, bool TAPE, bool PAIR, bool CARRY, int EC, bool WORK, bool ONE /* a wave has at most ONE tile per step: no queue, no loop */, class First, class Decode, class RowOf>
__device__ __forceinline__ void prop_tiles(const PropArgs& A, const PropLds& L, First first_of /* this wave's first tile */,
                                           Decode decode /* the others: draws from the workgroup's queue */, RowOf row_of, int lane,
                                           HeadCarry& hc
#ifdef PROP_STAMPS
                                           , unsigned long long (&st_sum)[8]
#endif
) {
    const float* mw = A.mw;
    const float* s_cur = A.s_cur; const int s_mod = A.s_mod; const size_t s_stride = A.s_stride;
    const float* attr = A.attr; const int attr_mod = A.attr_mod;
    const float* dens = A.dens; const int dens_mod = A.dens_mod;
    const int16_t* nbr_idx = A.nbr_idx; const uint8_t* nbr_cnt = A.nbr_cnt;
    const float* proj = A.proj; const float* c_node = A.c_node; const float* eff_in = A.eff_in; float* eff = A.eff;
    const int N = A.N;
    float* proj_next = A.proj_next; float* s_out = A.s_out; const size_t out_stride = A.out_stride;
    const float* cself = A.cself; const uint8_t* cself_ok = A.cself_ok;
    unsigned* mask_out = A.mask_out; float* agg_out = A.agg_out;
    const f16x8* wsp = L.wsp;
    const float* rows = L.rows;
    const int j = lane & 31, h = lane >> 5;
    const int jr = PAIR ? ((j & 7) | ((j >> 4) << 3)) : j;      // the tile's receiver this column works for
    const int par = PAIR ? ((j >> 3) & 1) : 0;                  // and which of an iteration's slots it runs
    constexpr int KS = PAIR ? 2 : 1;
    // attr_mod == dens_mod (n_batch or B) for every caller; s_mod is one of the two as well
    const float inv_mod = 1.0f / (float)attr_mod;
    const bool s_by_sample = s_mod != attr_mod;            // states of a running rollout: one block per sample
    // per-lane bases of a receiver's sample
    struct Bases {
        const float* s;       // positions of the sample (s_cur + (b % s_mod) * s_stride)
        const float* at;      // attributes of the sample
        float d;              // density feature
    };
    auto bases_of = [&](int b) {
        int q, bm;
        divmod_small(b, attr_mod, inv_mod, q, bm);
        Bases B_;
        B_.s = s_cur + (size_t)(s_by_sample ? b : bm) * s_stride;   // s_mod is B or attr_mod (every caller)
        B_.at = attr + (size_t)bm * N;
        B_.d = dens[bm] / DRP_DENS_SCALE;
        return B_;
    };
    // What a tile needs before its first slot can start hangs on a chain of dependent global loads
    // (in-degree and first neighbours -> self-loop test -> sender positions: three round trips of 2-3 us
    // each under load, measured 8 us per tile with tools/prop_stamps.py).  The chain is software-pipelined
    // across the tiles of a wave: the next tile's head (in-degree, first three neighbours, own position) is
    // requested when this tile's node part starts, the positions of its first two senders when that part
    // ends, so a tile begins with everything but its P_r / P_s rows on hand.
    struct TileHead {
        LaneRow lr;
        Bases bs;
        int cnt, ok;
        unsigned nbw0, nbw1;          // neighbours 0..3 (int16 pairs)
        float pix, piy, piz, pia;
    };
    auto tile_head = [&](const TileId& id) {
        TileHead hd;
        hd.lr = row_of(id, jr);
        const int b = hd.lr.b, i = hd.lr.i;
        const size_t row = (size_t)b * N + i;
        hd.bs = bases_of(b);
        hd.cnt = nbr_cnt[row];
        hd.ok = (cself != nullptr) ? (int)cself_ok[b] : 0;
        const unsigned* nbw = reinterpret_cast<const unsigned*>(nbr_idx + row * DRP_K);   // rows are 20 B: dword aligned
        hd.nbw0 = nbw[0];
        hd.nbw1 = nbw[1];
        hd.pix = hd.bs.s[i * 3 + 0]; hd.piy = hd.bs.s[i * 3 + 1]; hd.piz = hd.bs.s[i * 3 + 2];
        hd.pia = hd.bs.at[i];
        return hd;
    };
    struct TileFirst {
        int ks, j0, j1;
        float p0x, p0y, p0z, p0a;
    };
    // PAIR: neighbours 4 and 5 of the head (a column's second slot can be slot 4), requested with it; kept beside the
    // struct, whose layout the unpaired kernels' register allocation hangs on
    auto head_nbw2 = [&](const TileHead& hd) {
        return reinterpret_cast<const unsigned*>(nbr_idx + ((size_t)hd.lr.b * N + hd.lr.i) * DRP_K)[2];
    };

    auto tile_first = [&](const TileHead& hd, unsigned nbw2) {
        const int i = hd.lr.i;
        const int nb0 = (int)(hd.nbw0 & 0xffffu), nb1 = (int)(hd.nbw0 >> 16), nb2 = (int)(hd.nbw1 & 0xffffu);
        TileFirst f;
        // self slot first (k_graph self_first) and a per-sample self-edge constant: the self loop is skipped
        f.ks = __all(hd.ok && hd.cnt > 0 && nb0 == i) ? 1 : 0;
        if (!PAIR) {
            f.j0 = (f.ks < hd.cnt) ? (f.ks ? nb1 : nb0) : i;
            f.j1 = (f.ks + 1 < hd.cnt) ? (f.ks ? nb2 : nb1) : i;
        } else {                                          // this column's first two slots: ks + par, ks + par + 2
            const int nb3 = (int)(hd.nbw1 >> 16), nb4 = (int)(nbw2 & 0xffffu);
            const int s0 = f.ks + par;                    // 0 .. 2
            f.j0 = (s0 < hd.cnt) ? (s0 == 0 ? nb0 : (s0 == 1 ? nb1 : nb2)) : i;
            f.j1 = (s0 + 2 < hd.cnt) ? (s0 == 0 ? nb2 : (s0 == 1 ? nb3 : nb4)) : i;
        }
        f.p0x = hd.bs.s[f.j0 * 3 + 0]; f.p0y = hd.bs.s[f.j0 * 3 + 1]; f.p0z = hd.bs.s[f.j0 * 3 + 2]; f.p0a = hd.bs.at[f.j0];
        return f;
    };
    // A workgroup's share of the tiles is handed out to its waves on demand through a counter in LDS: the two
    // waves of a SIMD do not advance evenly (tools/prop_stamps.py: with five tiles each the first wave of the
    // chip was done at 0.7 of the last one's time, and a SIMD with one wave left runs at about 0.6 of its
    // two-wave rate); on demand the waves end within 10 us of each other.
    constexpr bool ROWS = EC != 0 && ONE;             // HeadCarry's rows: the cached kernels whose waves keep one tile
    bool first_tile = true;
    TileId cur = first_of(), nxt = {false, 0, 0};
    TileHead hd_next = {};
    TileFirst tf_next = {};
    unsigned nbw2_next = 0u;
    const TileId first_id = cur;
    if (CARRY && hc.ready) {
        if (cur.valid) {
            hd_next.lr.b = hc.b; hd_next.lr.i = hc.i; hd_next.lr.live = hc.live != 0;
            hd_next.bs.s = hc.s; hd_next.bs.at = hc.at; hd_next.bs.d = hc.d;
            hd_next.cnt = hc.cnt; hd_next.ok = hc.ok; hd_next.nbw0 = hc.nbw0; hd_next.nbw1 = hc.nbw1;
            hd_next.pix = hc.pix; hd_next.piy = hc.piy; hd_next.piz = hc.piz; hd_next.pia = hc.pia;
            tf_next.ks = hc.ks; tf_next.j0 = hc.j0; tf_next.j1 = hc.j1;
            tf_next.p0x = hc.p0x; tf_next.p0y = hc.p0y; tf_next.p0z = hc.p0z; tf_next.p0a = hc.p0a;
            nbw2_next = hc.nbw2;
        }
    } else if (cur.valid) {
        hd_next = tile_head(cur);
        if (PAIR) nbw2_next = head_nbw2(hd_next);
        tf_next = tile_first(hd_next, nbw2_next);
    }
    for (; cur.valid; cur = nxt) {
        const TileHead hd = hd_next;
        const TileFirst tf = tf_next;
        const int b = hd.lr.b, i = hd.lr.i;
        const bool live_row = hd.lr.live;
        const bool live = PAIR ? (live_row && par == 0) : live_row;   // the lanes that hold a receiver's aggregate
        const float* s = hd.bs.s;
        const float* at = hd.bs.at;
        const float d = hd.bs.d;
        const size_t row = (size_t)b * N + i;
        const float* pj = proj + ((size_t)b * N) * 128;
        const int cnt = hd.cnt;
        const int16_t* nb = nbr_idx + row * DRP_K;
        const int ks = tf.ks;
        const float inv = A.re_inv;
        const int kfirst = ks + par;                       // this column's slots: kfirst, kfirst + KS, ... while below cnt
        // EC: this tile's part of the workgroup's cache, one unit per slot ITERATION of the tile
        float4* const ec = (EC != 0) ? A.ecache + (size_t)cur.t * (PAIR ? 5 : DRP_K) * EC_UNITS + lane : nullptr;
        // ---- EC = 2: the chain's outputs come from the cache: a slot is 16 loads of 16 B and 130 vector instructions, no
        // matrix work.  The slot loop (further down) works on HALF slots (one output block: 4 + 4 loads, 32 registers), two in
        // flight: a half's registers are requested again, for the next slot, as soon as they have been added.  The first slot
        // is requested HERE, with the tile's own rows: its addresses hang on nothing but the tile's number and the head's
        // first sender.  A slot past the tile's last is still requested (from addresses that exist: the cache lines of the
        // iteration before, its sender row the sink) -- a request under a branch would make the buffers merge points, i.e.
        // copies behind a full wait; the loads go through global-address-space pointers (flat loads return out of order: a
        // wait for one would be a wait for all).
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef const f32x4 __attribute__((address_space(1))) * GPtr4;
        struct Half { f32x4 q[4], s[4]; };
        Half ha, hb;
        const GPtr4 ecg = (GPtr4)reinterpret_cast<const f32x4*>(ec);
        const GPtr4 pjg = (GPtr4)reinterpret_cast<const f32x4*>(pj + 64 + 4 * h);
        const GPtr4 sinkg = (GPtr4)reinterpret_cast<const f32x4*>(mw + R_SINK + 4 * h);
        // the column's slot k with sender jc; it_q: the iteration whose cache lines are read
        auto issue_half = [&](int k, int it_q, int ob, int jc, Half& H) {
            const GPtr4 src = ecg + (size_t)it_q * EC_UNITS + ob * 4 * 64;
            const GPtr4 rowp = ((k < cnt) ? pjg + (size_t)jc * 32 : sinkg) + 8 * ob;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                H.q[g] = src[g * 64];
                H.s[g] = rowp[2 * g];
            }
        };
        if constexpr (EC == 2) {
            issue_half(kfirst, 0, 0, tf.j0, ha);
            issue_half(kfirst, 0, 1, tf.j0, hb);
        }
        if (WORK) {                                        // the counting instantiation: what this tile will execute
            int n = 0;
            for (int k = kfirst; (k - par) < DRP_K && !__all(k >= cnt); k += KS) ++n;
            if (lane == 0) {
                unsigned long long* w = prop_work_shard(A.work);
                atomicAdd(w + (EC == 2 ? PROP_WORK_CACHED_SLOTS : PROP_WORK_CHAIN_SLOTS), (unsigned long long)n);
                atomicAdd(w + (LAST ? PROP_WORK_TILES_LAST : PROP_WORK_TILES), 1ull);
            }
        }
        const bool use_rows = ROWS && EC == 2 && first_tile && hc.rows_ready;      // wave-uniform
        Frag acc, bpr;
        {
            Frag pr;
            frag_bias_dens(rows + 128, rows + 192, d, h, bpr);
            if (use_rows) pr = hc.rpr;
            else frag_from_row(pj + (size_t)i * 128, h, pr);
#pragma unroll
            for (int r = 0; r < 16; ++r) { bpr.v[0][r] += pr.v[0][r]; bpr.v[1][r] += pr.v[1][r]; }
        }
        // the self loop's effect is relu(c_self + bias + P_r[i] + P_s[i]) without running the encoder chain
        if (ks) {
            const float* csr = cself + (size_t)b * 64;
            const float* psr = pj + (size_t)i * 128 + 64;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 cs = *reinterpret_cast<const float4*>(csr + 32 * ob + 8 * g + 4 * h);
                    float4 ps;
                    if (use_rows) ps = make_float4(hc.rps.v[ob][4 * g + 0], hc.rps.v[ob][4 * g + 1], hc.rps.v[ob][4 * g + 2], hc.rps.v[ob][4 * g + 3]);
                    else ps = *reinterpret_cast<const float4*>(psr + 32 * ob + 8 * g + 4 * h);
                    acc.v[ob][4 * g + 0] = relu1((bpr.v[ob][4 * g + 0] + cs.x) + ps.x);
                    acc.v[ob][4 * g + 1] = relu1((bpr.v[ob][4 * g + 1] + cs.y) + ps.y);
                    acc.v[ob][4 * g + 2] = relu1((bpr.v[ob][4 * g + 2] + cs.z) + ps.z);
                    acc.v[ob][4 * g + 3] = relu1((bpr.v[ob][4 * g + 3] + cs.w) + ps.w);
                }
            if (TAPE && live) mask_out[(row * DRP_K + 0) * 2 + h] = frag_positive_bits(acc);
        } else {
            frag_zero(acc);
        }
        if (EC == 0) {
            // the chain below runs on activations scaled by 2^k: so does its initial accumulator
            const float sc = A.re_scale;
#pragma unroll
            for (int r = 0; r < 16; ++r) { bpr.v[0][r] *= sc; bpr.v[1][r] *= sc; }
        }
        const float pix = hd.pix, piy = hd.piy, piz = hd.piz, pia = hd.pia;
        // two-deep software pipeline on the dependent loads (index -> sender position): the
        // position of slot k+1 and the index of slot k+2 are requested while slot k computes
        int j0 = tf.j0, j1 = tf.j1;
        float p0x = tf.p0x, p0y = tf.p0y, p0z = tf.p0z, p0a = tf.p0a;
#ifdef PROP_STAMPS
        const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
        int st_slots = 0;
#endif
        if constexpr (EC == 2) {
            unsigned mbits = 0u;
            auto consume_half = [&](int k, int ob, const Half& H) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float qv[4] = {H.q[g].x, H.q[g].y, H.q[g].z, H.q[g].w};
                    const float sv[4] = {H.s[g].x, H.s[g].y, H.s[g].z, H.s[g].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const float t = relu1(fmaf(qv[e], inv, bpr.v[ob][r] + sv[e]));
                        acc.v[ob][r] += t;
                        if (PAIR) acc.v[ob][r] += dpp_ror8(t);
                        if (TAPE) mbits = push_positive_bit(mbits, t);
                    }
                }
                if (TAPE && ob == 1 && live_row && (!PAIR || k < DRP_K)) mask_out[(row * DRP_K + k) * 2 + h] = mbits;
            };
            int k = kfirst;
#pragma unroll 1
            for (int it = 0; it < (PAIR ? 5 : DRP_K); ++it, k += KS) {
                if (__all(k >= cnt)) break;                      // wave-uniform trip count (the columns move together)
                const int j2 = (k + 2 * KS < cnt) ? (int)nb[min(k + 2 * KS, DRP_K - 1)] : i;
                const int it_q = (it + 1 < (PAIR ? 5 : DRP_K) && !__all(k + KS >= cnt)) ? it + 1 : it;
                consume_half(k, 0, ha);
                issue_half(k + KS, it_q, 0, j1, ha);
                consume_half(k, 1, hb);
                issue_half(k + KS, it_q, 1, j1, hb);
                j1 = j2;
#ifdef PROP_STAMPS
                ++st_slots;
#endif
            }
        } else {
        int it = 0;                                      // EC = 1: the iteration's unit of the cache
#pragma unroll 1
        for (int k = kfirst; k < DRP_K + par; k += KS) {     // k: this column's slot (PAIR: k may reach DRP_K, a padded slot)
            if (__all(k >= cnt)) break;             // no column of this tile has a slot k
#ifdef PROP_STAMPS
            ++st_slots;
#endif
            asm volatile("" ::: "memory");          // keep the packed-weight reads inside the loop
            const int jcur = j0;
            const float p1x = s[j1 * 3 + 0], p1y = s[j1 * 3 + 1], p1z = s[j1 * 3 + 2], p1a = at[j1];
            const int j2 = (k + 2 * KS < cnt) ? (int)nb[min(k + 2 * KS, DRP_K - 1)] : i;
            float x[8];
            x[0] = pia; x[1] = p0a;
            x[2] = pix - p0x; x[3] = piy - p0y; x[4] = piz - p0z;
            x[5] = d; x[6] = 1.0f; x[7] = 0.0f;
            Frag sv;                                 // issued now, consumed after the chain; a padded slot
            // reads the sink row (-1e30), so its relu(c + sv) is exactly 0
#ifdef PROP_NOGATHER   // timing experiment: the receiver's own row instead of the sender's
            frag_from_row((k < cnt) ? pj + (size_t)i * 128 + 64 + 0 * jcur : mw + R_SINK, h, sv);
#else
            frag_from_row((k < cnt) ? pj + (size_t)jcur * 128 + 64 : mw + R_SINK, h, sv);
#endif
            Frag a, c;
            FragB fb;
            frag_zero(a);
            mfma_layer8_split(wsp + S_RE0, x, h, a, lane);
            WOp w0 = wop_load(wsp + S_RE2, 0, 0, lane), wn;
            split_frag<true>(a, fb);
            frag_from_row(rows + 0, h, c);
            mfma_layer64_split(wsp + S_RE2, fb, c, lane, w0, wsp + S_RE4, wn);
            split_frag<true>(c, fb);
            frag_from_row(rows + 64, h, a);
            mfma_layer64_split(wsp + S_RE4, fb, a, lane, wn, wsp + S_RPE, w0);
            split_frag<true>(a, fb);
            if (EC == 0) c = bpr; else frag_zero(c);
            mfma_layer64_split(wsp + S_RPE, fb, c, lane, w0, nullptr, wn);
            if (EC == 1) {
                // the chain's raw output, for the other two propagation steps; and this step's term the way they form it
                float4* dst = ec + (size_t)it * EC_UNITS;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        dst[(ob * 4 + g) * 64] = make_float4(c.v[ob][4 * g + 0], c.v[ob][4 * g + 1], c.v[ob][4 * g + 2], c.v[ob][4 * g + 3]);
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) { sv.v[0][r] += bpr.v[0][r]; sv.v[1][r] += bpr.v[1][r]; }
            }
            if (!TAPE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!PAIR) {
                        acc.v[0][r] += relu1(fmaf(c.v[0][r], inv, sv.v[0][r]));
                        acc.v[1][r] += relu1(fmaf(c.v[1][r], inv, sv.v[1][r]));
                    } else {
                        const float t0 = relu1(fmaf(c.v[0][r], inv, sv.v[0][r]));
                        const float t1 = relu1(fmaf(c.v[1][r], inv, sv.v[1][r]));
                        acc.v[0][r] += t0;
                        acc.v[1][r] += t1;
                        acc.v[0][r] += dpp_ror8(t0);
                        acc.v[1][r] += dpp_ror8(t1);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    c.v[0][r] = relu1(fmaf(c.v[0][r], inv, sv.v[0][r]));
                    c.v[1][r] = relu1(fmaf(c.v[1][r], inv, sv.v[1][r]));
                    acc.v[0][r] += c.v[0][r];
                    acc.v[1][r] += c.v[1][r];
                    if (PAIR) {
                        acc.v[0][r] += dpp_ror8(c.v[0][r]);
                        acc.v[1][r] += dpp_ror8(c.v[1][r]);
                    }
                }
                if (live_row && (!PAIR || k < DRP_K)) mask_out[(row * DRP_K + k) * 2 + h] = frag_positive_bits(c);
            }
            j0 = j1; j1 = j2;
            p0x = p1x; p0y = p1y; p0z = p1z; p0a = p1a;
            if (EC == 1) ++it;
        }
        }
        if (ROWS && PAIR) {
            // a receiver's second column takes the FIRST column's aggregate (its own was summed odd slot first: equal to the last
            // place only): from here on both columns compute the same bits, so the rows the second column keeps for the next
            // step are what it would have loaded back -- the first column's stores -- and paired tiles keep giving the
            // unpaired bits.  The rotation runs with every lane on (under the selection's mask its source lanes would be off
            // and read as zero): taken first, made opaque, selected afterwards.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float o0 = dpp_ror8(acc.v[0][r]), o1 = dpp_ror8(acc.v[1][r]);
                asm volatile("" : "+v"(o0), "+v"(o1));
                acc.v[0][r] = par ? o0 : acc.v[0][r];
                acc.v[1][r] = par ? o1 : acc.v[1][r];
            }
        }
        // ---- node update on the aggregate still in registers
        asm volatile("" ::: "memory");
#ifdef PROP_STAMPS
        const unsigned long long st_c1 = __builtin_amdgcn_s_memtime();
#endif
        if (!ONE) {
            int li = 0;
            if (lane == 0) li = __hip_atomic_fetch_add(L.tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nxt = decode(__builtin_amdgcn_readfirstlane(li));        // past the share: invalid, and so is every later draw
        } else {
            nxt.valid = false;                                       // the tile loop runs once: straight-line code
        }
        const bool more = nxt.valid;
        if (more) hd_next = tile_head(nxt);
        if (PAIR && more) nbw2_next = head_nbw2(hd_next);
        if (CARRY && !LAST && !more) {                     // the wave's last tile of this step: the head of its first tile of the next
            hd_next = tile_head(first_id);
            if (PAIR) nbw2_next = head_nbw2(hd_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        Frag e;
        {
            Frag cn;
            if (use_rows) e = hc.re;
            else frag_from_row((TAPE ? eff_in : eff) + row * 64, h, e);    // in place unless a tape is written
            frag_from_row(c_node + row * 64, h, cn);
#pragma unroll
            for (int r = 0; r < 16; ++r) { e.v[0][r] += cn.v[0][r]; e.v[1][r] += cn.v[1][r]; }
        }
        if (TAPE && agg_out != nullptr && live) frag_to_row(agg_out + row * 64, h, acc);
        FragB6 f6;
        split_frag6(acc, f6);
        mfma_layer64_split6(L.w_agg, f6, e, lane);
        frag_relu(e);
        if (live) frag_to_row(eff + row * 64, h, e);
        const bool keep_rows = ROWS && !LAST && first_tile && hc.rows_ok;        // wave-uniform
        if (keep_rows) hc.re = e;
        split_frag6(e, f6);
        __builtin_amdgcn_sched_barrier(0);
        if (more || (CARRY && !LAST)) tf_next = tile_first(hd_next, nbw2_next);          // the head has landed by now
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) {
            Frag p;
            frag_zero(p);
            mfma_layer64_split6(L.w_x, f6, p, lane);
            if (live) frag_to_row(proj_next + row * 128, h, p);
            if (keep_rows) hc.rpr = p;
            frag_zero(p);
            mfma_layer64_split6(L.w_x + 1536, f6, p, lane);
            if (live) frag_to_row(proj_next + row * 128 + 64, h, p);
            if (keep_rows) hc.rps = p;
        } else {
            Frag hp;
            frag_from_row(L.rows_pr, h, hp);
            mfma_layer64_split6(L.w_x, f6, hp, lane);
            frag_relu(hp);
            float out[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                Frag w;
                frag_from_row(L.rows_pr + 64 + 64 * o, h, w);
                float p = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) p = fmaf(hp.v[0][r], w.v[0][r], p);
#pragma unroll
                for (int r = 0; r < 16; ++r) p = fmaf(hp.v[1][r], w.v[1][r], p);
                out[o] = p + __shfl_xor(p, 32, 64);
            }
            if (h == 0 && live) {
                float* so = s_out + (size_t)b * out_stride + (size_t)i * 3;
#pragma unroll
                for (int o = 0; o < 3; ++o) so[o] = (out[o] + L.rows_pr[64 + 192 + o]) + s[i * 3 + o];
            }
        }
#ifdef PROP_STAMPS
        {
            const unsigned long long st_c2 = __builtin_amdgcn_s_memtime(), st_r2 = __builtin_amdgcn_s_memrealtime();
            if (EC == 2) st_sum[6] += st_c1 - st_c0;   // shader cycles in the slot loops that read the cache
            else st_sum[0] += st_c1 - st_c0;           // shader cycles in slot loops
            st_sum[1] += (unsigned long long)st_slots;
            st_sum[2] += st_c2 - st_c1;        // shader cycles in node parts
            st_sum[3] += st_c2 - st_c0;        // shader cycles, whole tile
            st_sum[4] += st_r2 - st_r0;        // 100 MHz ticks, whole tile
            st_sum[5] += 1ull;
        }
#endif
        first_tile = false;
    }
    if (ROWS) hc.rows_ready = !LAST && hc.rows_ok && first_id.valid;
    if (CARRY && !LAST && first_id.valid) {
        hc.ready = true;
        hc.b = hd_next.lr.b; hc.i = hd_next.lr.i; hc.live = hd_next.lr.live ? 1 : 0;
        hc.s = hd_next.bs.s; hc.at = hd_next.bs.at; hc.d = hd_next.bs.d;
        hc.cnt = hd_next.cnt; hc.ok = hd_next.ok; hc.nbw0 = hd_next.nbw0; hc.nbw1 = hd_next.nbw1; hc.nbw2 = nbw2_next;
        hc.pix = hd_next.pix; hc.piy = hd_next.piy; hc.piz = hd_next.piz; hc.pia = hd_next.pia;
        hc.ks = tf_next.ks; hc.j0 = tf_next.j0; hc.j1 = tf_next.j1;
        hc.p0x = tf_next.p0x; hc.p0y = tf_next.p0y; hc.p0z = tf_next.p0z; hc.p0a = tf_next.p0a;
    }
}

#ifdef PROP_STAMPS
#define PROP_STAMPS_ARG , st_sum
#else
#define PROP_STAMPS_ARG
#endif

template <bool LAST, bool TAPE, bool PAIR /* tiles of 16 receivers x two slots (prop_tiles); with `spread` only */, bool WORK>
__global__ void __launch_bounds__(64 * PROP_WAVES)
km_prop(const uint16_t* __restrict__ sw, const uint16_t* __restrict__ sw6, const float* __restrict__ mw,
        const float* __restrict__ s_cur, int s_mod, size_t s_stride,
        const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
        const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt,
        const float* __restrict__ proj, const float* __restrict__ c_node, const float* __restrict__ eff_in,
        float* __restrict__ eff, int N, int B, float* __restrict__ proj_next, float* __restrict__ s_out,
        size_t out_stride, const float* __restrict__ cself /* nullable [B,64] */,
        const uint8_t* __restrict__ cself_ok,
        unsigned* __restrict__ mask_out /* TAPE: [B*N*10][2] */, float* __restrict__ agg_out /* TAPE, nullable: [B*N,64] */,
        float re_scale, float re_inv, int spread /* few tiles: one per workgroup first (see decode) */,
        unsigned long long* __restrict__ work /* WORK: PROP_WORK_* counters */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    WorkClock wclk;
    if constexpr (WORK) wclk = work_clock_begin();
#ifdef PROP_STAMPS
    const unsigned long long st_k0 = __builtin_amdgcn_s_memtime(), st_w0 = __builtin_amdgcn_s_memrealtime();
#endif
    float* wsp_f = lds;                              // edge chain, S_TOTAL units
    float* w6_f = wsp_f + S_TOTAL * 4;               // node layers: AGG | (RPR RPS) or (PR0)
    float* rows = w6_f + (LAST ? 2 : 3) * 1536 * 4;  // b2,b4,b_rp,wd_rp | b_pr0, w_pr1[3], b_pr1
    lds_fill(wsp_f, reinterpret_cast<const float*>(sw), S_TOTAL * 4);
    lds_fill(w6_f, reinterpret_cast<const float*>(sw6) + S6_AGG * 4, 1536 * 4);
    if (LAST) {
        lds_fill(w6_f + 1536 * 4, reinterpret_cast<const float*>(sw6) + S6_PR0 * 4, 1536 * 4);
        lds_fill(rows + 256, mw + R_PR0_B, 260);
    } else {
        lds_fill(w6_f + 1536 * 4, reinterpret_cast<const float*>(sw6) + S6_RPR * 4, 2 * 1536 * 4);
    }
    lds_fill(rows, reinterpret_cast<const float*>(sw) + S_ROWS * 4, 256);
    int* tile_ctr = reinterpret_cast<int*>(rows + 516);
    if (threadIdx.x == 0) *tile_ctr = PROP_WAVES;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int tile_rows = PAIR ? 16 : 32;
    const int tps = PAIR ? (N + 15) >> 4 : (N + 31) >> 5;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share
    // one and its L2), so the tiles of a sample -- which gather the same ~150 KB of sender rows -- all go
    // to workgroups of one residue class of blockIdx.  A placement guess only: wrong means slower, not wrong.
    const int ngroups = min(8, (int)gridDim.x);
    const int grp = blockIdx.x % ngroups;
    const int blocks_in_grp = ((int)gridDim.x - grp + ngroups - 1) / ngroups;
    const int samples_in_grp = (B - grp + ngroups - 1) / ngroups;
    const long grp_tiles = (long)(samples_in_grp > 0 ? samples_in_grp : 0) * tps;
    const long lt_step = (long)blocks_in_grp * PROP_WAVES;
    const long wg_base = (long)(blockIdx.x / ngroups) * PROP_WAVES;
    // the workgroup's share: what a static deal of the group's tiles over its waves would give it
    // A batch of few tiles (training: 4 samples of 300 particles are 38) is latency, not throughput: `spread` deals the
    // tiles workgroup-cyclically -- tile = block + grid x draw -- so that each runs alone on its CU (a wave that shares
    // its SIMD with another runs a slot iteration in 3.3 us, alone in 2.2) instead of eight to a workgroup.
    const long all_tiles = (long)B * tps;
    auto decode = [&](int li) {
        TileId id;
        if (spread) {
            const long lt = (long)blockIdx.x + (long)gridDim.x * li;
            id.valid = lt < all_tiles;
            id.b = (int)(lt / tps);
            id.t = (int)(lt - (long)id.b * tps);
            return id;
        }
        const long lt = wg_base + (li & (PROP_WAVES - 1)) + (long)(li / PROP_WAVES) * lt_step;
        id.valid = lt < grp_tiles;
        const int m = (int)(lt / tps);
        id.t = (int)(lt - (long)m * tps);
        id.b = grp + ngroups * m;
        return id;
    };
#ifdef PROP_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_k1 = __builtin_amdgcn_s_memtime();
#endif
    const PropArgs A = {mw, s_cur, s_mod, s_stride, attr, attr_mod, dens, dens_mod, nbr_idx, nbr_cnt, proj, c_node, eff_in, eff,
                        N, B, proj_next, s_out, out_stride, cself, cself_ok, mask_out, agg_out, re_scale, re_inv, nullptr, WORK ? work : nullptr};
    const PropLds L = {reinterpret_cast<const f16x8*>(wsp_f), reinterpret_cast<const bf16x8*>(w6_f),
                       reinterpret_cast<const bf16x8*>(w6_f) + 1536, rows, rows + 256, tile_ctr};
    // a tile = 32 (PAIR: 16) consecutive receivers of one sample
    auto row_of = [&](const TileId& id, int j) {
        LaneRow r;
        r.b = id.b;
        r.live = (id.t * tile_rows + j) < N;
        r.i = min(id.t * tile_rows + j, N - 1);
        return r;
    };
    HeadCarry hc_none;
    prop_tiles<LAST, TAPE, PAIR, false, 0, WORK, false>(A, L, [&]() { return decode(wave); }, decode, row_of, lane, hc_none PROP_STAMPS_ARG);
    if constexpr (WORK) work_clock_end(wclk, work);
#ifdef PROP_STAMPS
    st_sum[6] = st_k1 - st_k0;                               // entry -> weights in LDS
    st_sum[7] = __builtin_amdgcn_s_memtime() - st_k1;        // all tiles of this wave
    if (lane == 0 && blockIdx.x * PROP_WAVES + wave < 4096) {
        g_prop_span[(blockIdx.x * PROP_WAVES + wave) * 2 + 0] = st_w0;
        g_prop_span[(blockIdx.x * PROP_WAVES + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
        for (int q = 0; q < 8; ++q) g_prop_stamps[(blockIdx.x * PROP_WAVES + wave) * 8 + q] += st_sum[q];
    }
#endif
}
#define KM_PROP_LDS(LAST) ((size_t)(S_TOTAL * 4 + ((LAST) ? 2 : 3) * 1536 * 4 + 256 + 260 + 4) * sizeof(float))

// km_prop3: the three propagation steps of a rollout step in ONE launch.  A step needs the previous step's
// W_s eff rows of the SAME sample only, so a workgroup that owns whole samples needs no chip-wide barrier
// between steps: __syncthreads() (workgroup-scope release / acquire: the waves of a workgroup share their CU's
// L1) orders its own stores and gathers.  Saves two launches per rollout step, each with its refill of the
// packed weights (5 us), its launch and its end-of-launch imbalance (tools/prop_stamps.py, DESIGN_NOTES.md 5b).
// All four node matrices stay in LDS (153.6 KB).  grid = ceil(B / spw) workgroups, spw = samples per
// workgroup; the host uses it when every CU gets at least one sample and a workgroup at least PROP_WAVES
// tiles per step, and the one-step kernels otherwise (small batches spread by tiles, not by samples).
//
// Tiles.  The workgroup's receivers -- nb x N rows, contiguous in every per-row buffer -- are ONE list cut into
// tiles of 32, so only the list's last tile has idle lanes (4 x 300 rows: 38 tiles instead of 4 x 10; 4 x 50:
// 7 instead of 8).  A tile runs as many slot iterations as its largest in-degree, so the list is ordered by
// in-degree, largest first (a stable counting sort over 0..10 by the whole workgroup, once per launch: the lists
// do not change between the propagation steps; `perm` in LDS, 2 B per row): a tile's slot count is then its
// receivers' own, and the longest tiles are drawn first.  Per-row results do not depend on which tile a row
// is in (lanes are independent columns of every MFMA), so the order changes no bit -- except that the self-edge
// shortcut needs every lane of a tile to qualify, which the order does not disturb for uniform batches.
// Lists too long for the LDS left over (PROP3_PERM_MAX rows) keep the natural order.
#define PROP3_PERM_MAX 4900
#ifdef ROLLOUT_STAMPS
// Diagnostic build only (tools/rollout_stamps.py): 100 MHz wall stamps between the phases of a rollout step, summed by
// wave 0 of every 32nd workgroup; nothing the kernels compute reads them.
__device__ unsigned long long g_roll_stamps[16];
#define ROLL_STAMP(q) do { if (roll_on) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
                                          atomicAdd(&g_roll_stamps[q], now_ - roll_t); roll_t = now_; } } while (0)
#else
#define ROLL_STAMP(q) do { } while (0)
#endif
// LDS map of the whole-sample kernels (km_prop3, km_rollout), in floats from the start of dynamic LDS
struct Prop3Lds {
    float* wsp_f;      // edge chain (S_TOTAL units); phase E borrows it for the particle encoder's two 64x64 layers
    float* w6_f;       // AGG | RPR | RPS | PR0
    float* rows;       // 2^k b2, 2^k b4, b_rp, wd_rp | b_pr0, w_pr1[3], b_pr1 | tile counter at +516
    float* pe0_f;      // phase E: first layer of the particle encoder, then b_pe2, b_pp, wd_pp; afterwards the row order
    float* rows_e;
    int* tile_ctr;
};
__device__ __forceinline__ Prop3Lds prop3_lds(float* lds) {
    Prop3Lds P;
    P.wsp_f = lds;
    P.w6_f = P.wsp_f + S_TOTAL * 4;
    P.rows = P.w6_f + 4 * 1536 * 4;
    P.pe0_f = P.rows + 520;
    P.rows_e = P.pe0_f + 384 * 4;
    P.tile_ctr = reinterpret_cast<int*>(P.rows + 516);
    return P;
}
// the matrices and rows that stay put for a whole launch (S6_AGG, RPR, RPS, PR0 are consecutive in the packed blob)
__device__ __forceinline__ void prop3_fill_resident(const Prop3Lds& P, const uint16_t* __restrict__ sw, const uint16_t* __restrict__ sw6,
                                                    const float* __restrict__ mw) {
    lds_fill(P.w6_f, reinterpret_cast<const float*>(sw6) + S6_AGG * 4, 4 * 1536 * 4);
    lds_fill(P.rows, reinterpret_cast<const float*>(sw) + S_ROWS * 4, 256);
    lds_fill(P.rows + 256, mw + R_PR0_B, 260);
}

// One rollout step's MLP work for the workgroup's samples [b0, b0 + nb): [phase E: particle encoder] -> row order ->
// three propagation steps, the last one writing s_out.  On entry the resident part of LDS is filled (or being filled:
// `entry_sync` = the caller has not synchronised since) and the edge-chain region holds nothing this function relies
// on; on exit every wave has passed its last tile (no barrier after it).
template <bool TAPE, bool PAIR, bool CARRY, bool ECACHE, bool WORK, bool ONE /* the workgroup has no more tiles than waves (the host's promise) */,
          bool ENC_PRE = false /* km_rollout<pair>: the caller has filled the encoder's matrices and reset the tile counter, one barrier ago */,
          class Aux = int /* ENC_PRE: aux(w, nw) -- work for the waves without an encoder tile (the neighbour lists), w-th of nw */>
__device__ __forceinline__ void prop3_step(const Prop3Lds& P, const uint16_t* __restrict__ sw, const uint16_t* __restrict__ sw6,
                                           const float* __restrict__ mw,
                                           const float* __restrict__ s_cur, int s_mod, size_t s_stride,
                                           const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
                                           const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt,
                                           float* __restrict__ proj_a, float* __restrict__ proj_b, float* __restrict__ c_node,
                                           float* __restrict__ eff, int N, int B, int spw, const float* __restrict__ s_delta,
                                           float* __restrict__ s_out, size_t out_stride, const float* __restrict__ cself,
                                           const uint8_t* __restrict__ cself_ok, unsigned* __restrict__ mask_hist,
                                           float* __restrict__ agg_hist, float re_scale, float re_inv, int order_rows, int tid /* tid */,
                                           float4* __restrict__ ecache /* ECACHE: this workgroup's edge-chain cache (prop_tiles) */,
                                           unsigned long long* __restrict__ work /* WORK: PROP_WORK_* counters */
#ifdef PROP_STAMPS
                                           , unsigned long long (&st_sum)[8]
#endif
                                           , Aux aux = Aux()
                                           , size_t hist_rows = 0 /* TAPE: rows (samples x particles) of the WHOLE batch the history buffers
                                                                     are laid out for -- a launch may cover a block of its samples */
) {
    float* wsp_f = P.wsp_f;
    float* w6_f = P.w6_f;
    float* rows = P.rows;
    float* pe0_f = P.pe0_f;
    float* rows_e = P.rows_e;
    int* tile_ctr = P.tile_ctr;
#ifdef ROLLOUT_STAMPS
    const bool roll_on = tid == 0 && (blockIdx.x & 31) == 0;
    unsigned long long roll_t = __builtin_amdgcn_s_memrealtime();
    const unsigned long long roll_c0 = __builtin_amdgcn_s_memtime(), roll_r0 = roll_t;
#endif
    const bool phase_e = s_delta != nullptr;
    if (!ENC_PRE) {
        if (phase_e) {
            // the particle encoder's two 64x64 layers borrow the edge chain's region; W_r and W_s are resident anyway
            lds_fill(wsp_f, reinterpret_cast<const float*>(sw6) + S6_PE2 * 4, 2 * 1536 * 4, tid);
            lds_fill(pe0_f, reinterpret_cast<const float*>(sw6) + S6_PE0 * 4, 384 * 4, tid);
            lds_fill(rows_e, mw + R_PE2_B, 192, tid);
        } else {
            lds_fill(wsp_f, reinterpret_cast<const float*>(sw), S_TOTAL * 4, tid);
        }
        if (tid == 0) *tile_ctr = PROP_WAVES;
        __syncthreads();
    }
    ROLL_STAMP(3);                                   // encoder weights in LDS
    const int lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * spw, nb = min(spw, B - b0);
    const int wg_rows = (nb > 0 ? nb : 0) * N;       // this workgroup's receivers: rows b0*N .. b0*N + wg_rows
    const int enc_tiles = (wg_rows + 31) >> 5;
    // PAIR (the host's choice for a launch whose workgroups hold up to four tiles of 32 rows for their eight waves, see
    // prop_pair() in capi_ctx.h): the propagation steps run tiles of 16 receivers x two slots (prop_tiles) -- twice
    // the waves at work, half the slot iterations each.  A kernel of its own: the register allocation of the other one
    // is not to move, and both tile loops in one kernel with a per-workgroup choice run 15 % slower, either of them.
    constexpr int tile_rows = PAIR ? 16 : 32;
    const int wg_tiles = PAIR ? (wg_rows + 15) >> 4 : enc_tiles;
    const float inv_N = 1.0f / (float)N;
    if (phase_e) {
        // ---- phase E: the particle encoder over this workgroup's rows (km_node_encode_split's arithmetic per row)
        const bf16x8* wpe2 = reinterpret_cast<const bf16x8*>(wsp_f);
        const bf16x8* wrs = reinterpret_cast<const bf16x8*>(w6_f) + 1536;
        const bf16x8* wpe0 = reinterpret_cast<const bf16x8*>(pe0_f);
        const int j = lane & 31, h = lane >> 5;
        const float inv_mod = 1.0f / (float)attr_mod;
        for (int li = wave; li < enc_tiles;) {
            asm volatile("" ::: "memory");
            const bool live = (li * 32 + j) < wg_rows;
            const int r = min(li * 32 + j, wg_rows - 1);
            int m, i, q, bm;
            divmod_small(r, N, inv_N, m, i);
            const int b = b0 + m;
            divmod_small(b, attr_mod, inv_mod, q, bm);
            const size_t row = (size_t)b0 * N + r;
            const float d = dens[bm] / DRP_DENS_SCALE;
            const float* sd = s_delta + row * 3;
            float x[8] = {sd[0], sd[1], sd[2], attr[(size_t)bm * N + i], d, 1.0f, 0.0f, 0.0f};
            Frag a, pe, c;
            FragB6 f6;
            frag_zero(a);
            mfma_layer8_split6(wpe0, x, h, a, lane);
            frag_relu(a);
            split_frag6(a, f6);
            frag_from_row(rows_e + 0, h, pe);
            mfma_layer64_split6(wpe2, f6, pe, lane);
            frag_relu(pe);
            if (live) frag_to_row(eff + row * 64, h, pe);          // TAPE: slot 0 of the effect history
            split_frag6(pe, f6);
            frag_bias_dens(rows_e + 64, rows_e + 128, d, h, c);
            mfma_layer64_split6(wpe2 + 1536, f6, c, lane);
            if (live) frag_to_row(c_node + row * 64, h, c);
            frag_zero(c);
            mfma_layer64_split6(wrs, f6, c, lane);
            if (live) frag_to_row(proj_a + row * 128, h, c);
            frag_zero(c);
            mfma_layer64_split6(wrs + 1536, f6, c, lane);
            if (live) frag_to_row(proj_a + row * 128 + 64, h, c);
            int qn = 0;
            if (lane == 0) qn = __hip_atomic_fetch_add(tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            li = __builtin_amdgcn_readfirstlane(qn);
            if (WORK && lane == 0) atomicAdd(prop_work_shard(work) + PROP_WORK_ENC_TILES, 1ull);
        }
        if constexpr (ENC_PRE) {
            ROLL_STAMP(4);                               // (wave 0's encoder tile; its share of the lists goes to stamp 14)
            // the waves without an encoder tile build the neighbour lists meanwhile (the encoder reads impulses, attributes
            // and densities, the lists positions: nothing of one another); with a tile for every wave, all of them afterwards
            // -- unless the tile-less waves would each have more than one round of rows (seven tiles at 50 particles: one wave,
            // 200 rows: 14 % slower than everybody taking a share behind its tile)
            if (enc_tiles < PROP_WAVES && wg_rows <= (PROP_WAVES - enc_tiles) * 64) { if (wave >= enc_tiles) aux(wave - enc_tiles, PROP_WAVES - enc_tiles); }
            else aux(wave, PROP_WAVES);
            ROLL_STAMP(14);
        }
        ROLL_STAMP(4);                                   // wave 0's encoder tiles
        __syncthreads();                                 // the encoder's rows of this workgroup's samples are written (ENC_PRE: and the lists)
        ROLL_STAMP(5);                                   // waiting for the other waves' encoder tiles
        lds_fill(wsp_f, reinterpret_cast<const float*>(sw), S_TOTAL * 4, tid);
        if (tid == 0) *tile_ctr = PROP_WAVES;
    }
    // ---- row order: in-degree descending, row ascending within a degree (deterministic; the encoder's LDS is free now)
    uint16_t* perm = reinterpret_cast<uint16_t*>(pe0_f);
    bool ordered = order_rows != 0 && wg_rows <= PROP3_PERM_MAX && wg_rows > 0;
    if (ordered) {
        int* hist = reinterpret_cast<int*>(perm + ((PROP3_PERM_MAX + 1) & ~1));      // [PROP_WAVES][11]
        const uint8_t* cnt_rows = nbr_cnt + (size_t)b0 * N;
        const int groups = (wg_rows + 63) >> 6;
        const int g_lo = (groups * wave) / PROP_WAVES, g_hi = (groups * (wave + 1)) / PROP_WAVES;   // this wave's 64-row groups
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        int mine[DRP_K + 1];
#pragma unroll
        for (int v = 0; v <= DRP_K; ++v) mine[v] = 0;
        for (int g = g_lo; g < g_hi; ++g) {
            const int r = g * 64 + lane;
            const int c = (r < wg_rows) ? min((int)cnt_rows[r], DRP_K) : -1;
#pragma unroll
            for (int v = 0; v <= DRP_K; ++v) mine[v] += __popcll(__ballot(c == v));
        }
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v <= DRP_K; ++v) hist[wave * (DRP_K + 1) + v] = mine[v];
        }
        __syncthreads();
        // first position of (degree v, this wave): every higher degree of every wave, then degree v of the waves before
        int base[DRP_K + 1];
        int top = 0;                                   // rows of the largest in-degree present
        {
            int above = 0;
#pragma unroll
            for (int v = DRP_K; v >= 0; --v) {
                int before = 0, total = 0;
                for (int w = 0; w < PROP_WAVES; ++w) {
                    const int t = hist[w * (DRP_K + 1) + v];
                    before += (w < wave) ? t : 0;
                    total += t;
                }
                base[v] = above + before;
                if (above == 0) top = total;
                above += total;
            }
        }
        // a saturated pile (15 of 16 receivers at the largest in-degree: every tile runs that many slots whatever the
        // order) keeps the natural order, whose rows are consecutive in memory
        ordered = top * 16 < wg_rows * 15;
        for (int g = g_lo; g < g_hi && ordered; ++g) {
            const int r = g * 64 + lane;
            const int c = (r < wg_rows) ? min((int)cnt_rows[r], DRP_K) : -1;
            int pos = 0;
#pragma unroll
            for (int v = 0; v <= DRP_K; ++v) {
                const unsigned long long m = __ballot(c == v);
                if (c == v) pos = base[v] + __popcll(m & lt_mask);
                base[v] += __popcll(m);
            }
            if (c >= 0) perm[pos] = (uint16_t)r;
        }
    }
    __syncthreads();
    ROLL_STAMP(6);                                   // edge-chain weights back in LDS, rows ordered
    // Waves w and w + 4 of a workgroup share a SIMD (tools/hwid.hip).  With no more tiles than waves every wave runs one
    // tile per step, and in the in-degree order tile t is heavier than tile t + 1.  A slot iteration takes 2.2 us with
    // the SIMD to itself and 3.3 us next to another wave's, so a SIMD with tiles of a >= b slots needs about
    // 2.2 a + 1.1 b: the 8 - wg_tiles heaviest tiles keep their SIMD to themselves and waves 4 ... 7 take the rest from
    // the light end, the lightest next to the heaviest that gets company at all (4 x 50 particles, 7 tiles: 0, 1 + 6,
    // 2 + 5, 3 + 4; eight tiles: 0 + 7, 1 + 6, 2 + 5, 3 + 4).  With more tiles than waves the queue hands them out
    // heaviest first as before.
    auto first_of = [&]() {
        TileId first;
        first.b = b0;
        first.t = wave;
        first.valid = wave < wg_tiles;
        if (ordered && wg_tiles <= PROP_WAVES && wave >= PROP_WAVES / 2) {
            const int sd = wave - PROP_WAVES / 2, alone = PROP_WAVES - wg_tiles;     // SIMDs 0 .. alone - 1 run one tile
            first.t = wg_tiles - 1 - (sd - alone);
            first.valid = sd >= alone && first.t >= PROP_WAVES / 2;
        }
        return first;
    };
    // (the queue's decoder keeps the branch it has always had -- dead now, a draw is past PROP_WAVES -- because the slot
    // loop's register allocation moves with it: without the branch the 300-particle launch is 0.3 % slower, with the
    // first tile's mapping inside it 7 %)
    const bool snake = ordered && wg_tiles <= PROP_WAVES;
    auto decode = [&](int li) {
        TileId id;
        id.b = b0;
        if (snake && li >= PROP_WAVES / 2) {
            id.t = wg_tiles - 1 - (li - PROP_WAVES / 2);
            id.valid = li < PROP_WAVES && id.t >= PROP_WAVES / 2;
        } else {
            id.t = li;
            id.valid = li < wg_tiles;
        }
        return id;
    };
    auto row_of = [&](const TileId& id, int j) {
        LaneRow lr;
        const int g = id.t * tile_rows + j;
        lr.live = g < wg_rows;
        const int gc = min(g, wg_rows - 1);
        const int r = ordered ? (int)perm[gc] : gc;
        int m;
        divmod_small(r, N, inv_N, m, lr.i);
        lr.b = b0 + m;
        return lr;
    };
    const size_t bn64 = (TAPE && hist_rows != 0 ? hist_rows : (size_t)B * N) * 64;      // stride between the history's slots
    PropArgs A = {mw, s_cur, s_mod, s_stride, attr, attr_mod, dens, dens_mod, nbr_idx, nbr_cnt, proj_a, c_node, eff, eff,
                  N, B, proj_b, s_out, out_stride, cself, cself_ok, nullptr, nullptr, re_scale, re_inv, ecache, WORK ? work : nullptr};
    PropLds L = {reinterpret_cast<const f16x8*>(wsp_f), reinterpret_cast<const bf16x8*>(w6_f),
                 reinterpret_cast<const bf16x8*>(w6_f) + 1536, rows, rows + 256, tile_ctr};
    HeadCarry hc;
    hc.ready = false;
    hc.rows_ready = false;
    hc.rows_ok = ONE;                                // every wave keeps its one tile from step to step (first_of)
#ifdef ROLLOUT_STAMPS
#define PROP3_STEP_STAMP(p_) do { if (roll_on) atomicAdd(&g_roll_stamps[11 + (p_)], __builtin_amdgcn_s_memrealtime() - roll_t); } while (0)   /* ... by propagation step */
#else
#define PROP3_STEP_STAMP(p_) do { } while (0)
#endif
#define PROP3_STEP(p_) { \
        if ((p_) > 0) { \
            __syncthreads();                         /* step p-1's rows of this workgroup's samples are written */ \
            ROLL_STAMP(8);                           /* waiting for the other waves at a propagation step's end */ \
            if (tid == 0) *tile_ctr = PROP_WAVES; \
            __syncthreads(); \
        } \
        A.proj = ((p_) & 1) ? proj_b : proj_a; \
        A.proj_next = ((p_) & 1) ? proj_a : proj_b; \
        if (TAPE) { \
            A.eff_in = eff + (size_t)(p_) * bn64; \
            A.eff = eff + (size_t)((p_) + 1) * bn64; \
            A.mask_out = mask_hist + (size_t)(p_) * (bn64 / 64) * DRP_K * 2; \
            A.agg_out = agg_hist ? agg_hist + (size_t)(p_) * bn64 : nullptr; \
        } \
        if (ECACHE && (p_) == 0) { \
            prop_tiles<false, TAPE, PAIR, CARRY, 1, WORK, ONE>(A, L, first_of, decode, row_of, lane, hc PROP_STAMPS_ARG); \
        } else if ((p_) + 1 < DRP_PSTEP) { \
            prop_tiles<false, TAPE, PAIR, CARRY, ECACHE ? 2 : 0, WORK, ONE>(A, L, first_of, decode, row_of, lane, hc PROP_STAMPS_ARG); \
        } else { \
            L.w_x = reinterpret_cast<const bf16x8*>(w6_f) + 3 * 1536; \
            prop_tiles<true, TAPE, PAIR, CARRY, ECACHE ? 2 : 0, WORK, ONE>(A, L, first_of, decode, row_of, lane, hc PROP_STAMPS_ARG); \
        } \
        PROP3_STEP_STAMP(p_); \
        ROLL_STAMP(7);                               /* wave 0's tiles of a propagation step */ \
    }
    // The cached kernels run the three steps as three straight-line blocks: the rows a step hands to the next in registers
    // (HeadCarry) are then live from one node part to the next step's start and nowhere else -- around the back edge of a
    // rolled loop they would be live through the first step's chain loop too (900 spilled registers).  The others keep the
    // rolled loop: its two bodies (not the last step / the last) are 20 KB each.
    if constexpr (ECACHE) {
        PROP3_STEP(0) PROP3_STEP(1) PROP3_STEP(2)
    } else {
#pragma unroll 1
        for (int p = 0; p < DRP_PSTEP; ++p) PROP3_STEP(p)
    }
#undef PROP3_STEP
#undef PROP3_STEP_STAMP
#ifdef ROLLOUT_STAMPS
    if (roll_on) {                                   // shader clock over wall clock for the whole step
        atomicAdd(&g_roll_stamps[9], __builtin_amdgcn_s_memtime() - roll_c0);
        atomicAdd(&g_roll_stamps[10], __builtin_amdgcn_s_memrealtime() - roll_r0);
    }
#endif
}

template <bool TAPE, bool PAIR, bool ECACHE, bool WORK, bool ONE /* no more tiles than waves per workgroup (the host's promise; cached kernels) */>
__global__ void __launch_bounds__(64 * PROP_WAVES)
km_prop3(const uint16_t* __restrict__ sw, const uint16_t* __restrict__ sw6, const float* __restrict__ mw,
         const float* __restrict__ s_cur, int s_mod, size_t s_stride,
         const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
         const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt,
         float* __restrict__ proj_a, float* __restrict__ proj_b, float* __restrict__ c_node,
         float* __restrict__ eff /* !TAPE: in place; TAPE: effect history [4][B*N,64] */, int N, int B, int spw,
         const float* __restrict__ s_delta /* not null: the particle encoder runs here first (phase E) */,
         float* __restrict__ s_out, size_t out_stride, const float* __restrict__ cself, const uint8_t* __restrict__ cself_ok,
         unsigned* __restrict__ mask_hist /* TAPE: [3][B*N*10][2] */, float* __restrict__ agg_hist /* TAPE, nullable: [3][B*N,64] */,
         float re_scale, float re_inv, int order_rows,
         float4* __restrict__ ecache /* ECACHE: [workgroup][ec_stride] */, size_t ec_stride,
         unsigned long long* __restrict__ work /* WORK: PROP_WORK_* counters */,
         size_t hist_rows /* TAPE: rows of the whole batch behind the history buffers (0: this launch's B * N) */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef PROP_STAMPS
    const unsigned long long st_k0 = __builtin_amdgcn_s_memtime(), st_w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    WorkClock wclk;
    if constexpr (WORK) wclk = work_clock_begin();
    const Prop3Lds P = prop3_lds(lds);
    prop3_fill_resident(P, sw, sw6, mw);
    prop3_step<TAPE, PAIR, PAIR && !TAPE /* the tape's kernel has no register to spare for the carried head; the big kernel's
                                            allocation is not to move (with it: 256 VGPRs) */, ECACHE, WORK, ONE>(P, sw, sw6, mw, s_cur, s_mod, s_stride, attr, attr_mod, dens, dens_mod, nbr_idx, nbr_cnt, proj_a, proj_b, c_node,
                     eff, N, B, spw, s_delta, s_out, out_stride, cself, cself_ok, mask_hist, agg_hist, re_scale, re_inv, order_rows,
                     (int)threadIdx.x, ECACHE ? ecache + (size_t)blockIdx.x * ec_stride : nullptr, work PROP_STAMPS_ARG, 0, TAPE ? hist_rows : 0);
    if constexpr (WORK) work_clock_end(wclk, work);
#ifdef PROP_STAMPS
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        st_sum[7] = __builtin_amdgcn_s_memtime() - st_k0;        // the whole launch of this wave, fills and barrier waits included
        if (lane == 0 && blockIdx.x * PROP_WAVES + wave < 4096) {
            g_prop_span[(blockIdx.x * PROP_WAVES + wave) * 2 + 0] = st_w0;
            g_prop_span[(blockIdx.x * PROP_WAVES + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
            for (int q = 0; q < 8; ++q) g_prop_stamps[(blockIdx.x * PROP_WAVES + wave) * 8 + q] += st_sum[q];
        }
    }
#endif
}
// the encoder's first layer + rows (6 912 B) are followed by slack up to the row order's 9 800 + 352 B
#define KM_PROP3_LDS ((size_t)(S_TOTAL * 4 + 4 * 1536 * 4 + 256 + 260 + 4) * sizeof(float) + (size_t)(((PROP3_PERM_MAX + 1) & ~1) * 2 + PROP_WAVES * (DRP_K + 1) * 4))


// ---- particle encoder, node constant and first projections on the 6-term split --------------
// Same contract as km_node_encode (k_mlp_mfma.h); outputs go straight from the accumulator
// layout to their rows (no LDS transposition tiles: LDS holds the 126 KB of packed weights).
// (the body of workgroup `blk` of `nblk`, 64 * MFMA_WAVES threads: km_node_encode_split below, and km_graph_q4_encode of k_rollout.h)
__device__ __forceinline__ void
node_encode_split_block(const uint16_t* __restrict__ sw6, const float* __restrict__ mw,
                        const float* __restrict__ s_delta, const float* __restrict__ attr, int attr_mod,
                        const float* __restrict__ dens, int dens_mod, int N, int B,
                        float* __restrict__ eff, float* __restrict__ c_node, float* __restrict__ proj, int blk, int nblk, float* lds) {
    float* w6_f = lds;                         // PE2 | PPE | RPR | RPS (4 x 1536 units) | PE0 (384)
    float* rows = w6_f + (4 * 1536 + 384) * 4; // b_pe2, b_pp, wd_pp
    lds_fill(w6_f, reinterpret_cast<const float*>(sw6) + S6_PE2 * 4, 2 * 1536 * 4);
    lds_fill(w6_f + 2 * 1536 * 4, reinterpret_cast<const float*>(sw6) + S6_RPR * 4, 2 * 1536 * 4);
    lds_fill(w6_f + 4 * 1536 * 4, reinterpret_cast<const float*>(sw6) + S6_PE0 * 4, 384 * 4);
    lds_fill(rows, mw + R_PE2_B, 192);
    __syncthreads();
    const bf16x8* w6 = reinterpret_cast<const bf16x8*>(w6_f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tps = (N + 31) >> 5;
    const long ntiles = (long)B * tps;
    for (long gt = (long)blk + (long)nblk * wave; gt < ntiles; gt += (long)nblk * MFMA_WAVES) {   // workgroup-cyclic first: few tiles spread one per CU
        asm volatile("" ::: "memory");      // keep the packed-weight reads inside the loop (see km_prop)
        const int b = (int)(gt / tps), t = (int)(gt - (long)b * tps);
        const int i = min(t * 32 + j, N - 1);
        const bool live = (t * 32 + j) < N;
        const size_t row = (size_t)b * N + i;
        const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
        const float* sd = s_delta + row * 3;
        float x[8] = {sd[0], sd[1], sd[2], attr[(size_t)(b % attr_mod) * N + i], d, 1.0f, 0.0f, 0.0f};
        Frag a, pe, c;
        FragB6 f6;
        frag_zero(a);
        mfma_layer8_split6(w6 + 4 * 1536, x, h, a, lane);
        frag_relu(a);
        split_frag6(a, f6);
        frag_from_row(rows + 0, h, pe);
        mfma_layer64_split6(w6, f6, pe, lane);
        frag_relu(pe);
        if (live) frag_to_row(eff + row * 64, h, pe);
        split_frag6(pe, f6);
        frag_bias_dens(rows + 64, rows + 128, d, h, c);
        mfma_layer64_split6(w6 + 1536, f6, c, lane);
        if (live) frag_to_row(c_node + row * 64, h, c);
        frag_zero(c);
        mfma_layer64_split6(w6 + 2 * 1536, f6, c, lane);
        if (live) frag_to_row(proj + row * 128, h, c);
        frag_zero(c);
        mfma_layer64_split6(w6 + 3 * 1536, f6, c, lane);
        if (live) frag_to_row(proj + row * 128 + 64, h, c);
    }
}
DRP_GLOBAL void __launch_bounds__(64 * MFMA_WAVES)
km_node_encode_split(const uint16_t* __restrict__ sw6, const float* __restrict__ mw,
                     const float* __restrict__ s_delta, const float* __restrict__ attr, int attr_mod,
                     const float* __restrict__ dens, int dens_mod, int N, int B,
                     float* __restrict__ eff, float* __restrict__ c_node, float* __restrict__ proj) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    node_encode_split_block(sw6, mw, s_delta, attr, attr_mod, dens, dens_mod, N, B, eff, c_node, proj, (int)blockIdx.x, (int)gridDim.x, lds);
}
#define KM_NODE_SPLIT_LDS ((size_t)((4 * 1536 + 384) * 4 + 192) * sizeof(float))
