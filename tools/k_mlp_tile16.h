// ARCHIVED EXPERIMENT -- not part of the product library ("16-item tiles").  It was built as
// csrc/k_mlp_tile16.h behind DRP_TILE16=1 in round 1: correct to 6e-8, 9.35 vs 7.5 ms for the 32-item tile.
// Kept for the record of the measurement; to run it again include it from drp_capi.hip after k_mlp_split.h.
// The propagation kernel on 16-item tiles: v_mfma_f32_16x16x32_{f16,bf16}, 16 accumulator registers per
// 64-feature fragment instead of 32, so the slot loop fits 168 VGPRs and THREE waves share a SIMD (12 per
// workgroup) where the 32-item tile of k_mlp_split.h has two.  Same arithmetic per item (3-term fp16 split on
// the relation encoder, 6-term bf16 split on the node layers), same sample-local three-steps-per-launch
// structure as km_prop3; twice the LDS weight reads per item and a first layer that costs twice as much per
// item (its K = 8 inputs occupy a 32-deep k-step).  Forward without a tape only: the tape's mask layout is the
// 32-item fragment's, so planner gradients and training stay on km_prop / km_prop3.
//
// Fragment: v[ob][r] = feature 16*ob + 4*g + r of item (lane & 15), g = lane >> 4  (C/D layout of the
// instruction: row = 4*(lane>>4) + r, col = lane & 15).  The B operand of k-step ks holds k = 8g + jj, jj = 0..7:
// it is registers r = jj & 3 of output blocks 2ks + (jj >> 2) of the previous layer, so the next layer's weights
// are packed with   kidx(ks, g, jj) = 16*(2ks + (jj>>2)) + 4g + (jj&3).
#pragma once
#include "k_mlp_split.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef PROP16_WAVES
#define PROP16_WAVES 12
#endif

struct Frag16 {
    f32x4v v[4];
};

enum {                         // units of f16x8
    T_RE0 = 0,                 // first layer, one k-step: [part 2][ob 4][lane 64]
    T_RE2 = T_RE0 + 512,       // 64x64: [part 2][ob 4][ks 2][lane 64]
    T_RE4 = T_RE2 + 1024,
    T_RPE = T_RE4 + 1024,
    T_TOTAL = T_RPE + 1024
};
enum {                         // units of bf16x8; 64x64: [part 3][ob 4][ks 2][lane 64]
    T6_AGG = 0,
    T6_RPR = T6_AGG + 1536,
    T6_RPS = T6_RPR + 1536,
    T6_PR0 = T6_RPS + 1536,
    T6_TOTAL = T6_PR0 + 1536
};

inline int t16_kidx(int ks, int g, int jj) { return 16 * (2 * ks + (jj >> 2)) + 4 * g + (jj & 3); }

inline void pack_tile16(const float* w, std::vector<uint16_t>& out) {
    out.assign((size_t)T_TOTAL * 8, 0);
    auto put2 = [&](int hi_unit, int lo_unit, int jj, float v) {
        const float hi = host_f16_to_f32(host_f16_rne(v));
        out[(size_t)hi_unit * 8 + jj] = host_f16_rne(hi);
        out[(size_t)lo_unit * 8 + jj] = host_f16_rne(v - hi);
    };
    auto P64 = [&](int dst, int src, int ld, int col0) {
        for (int ob = 0; ob < 4; ++ob)
            for (int ks = 0; ks < 2; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int i = lane & 15, g = lane >> 4;
                        const float v = w[src + (16 * ob + i) * ld + col0 + t16_kidx(ks, g, jj)];
                        put2(dst + ((0 * 4 + ob) * 2 + ks) * 64 + lane, dst + ((1 * 4 + ob) * 2 + ks) * 64 + lane, jj, v);
                    }
    };
    // relation encoder layer 0: inputs [a_r, a_s, dx, dy, dz, d, 1(bias), 0] on lane group 0, k = 8g + jj
    for (int ob = 0; ob < 4; ++ob)
        for (int lane = 0; lane < 64; ++lane)
            for (int jj = 0; jj < 8; ++jj) {
                const int i = lane & 15, g = lane >> 4, o = 16 * ob + i;
                float v = 0.0f;
                if (g == 0 && jj < 6) v = w[W_RE0_W + o * 6 + jj];
                else if (g == 0 && jj == 6) v = w[W_RE0_B + o];
                put2(T_RE0 + (0 * 4 + ob) * 64 + lane, T_RE0 + (1 * 4 + ob) * 64 + lane, jj, v);
            }
    P64(T_RE2, W_RE2_W, 64, 0);
    P64(T_RE4, W_RE4_W, 64, 0);
    P64(T_RPE, W_RP_W, 193, 0);
}

inline void pack_tile16_6(const float* w, std::vector<uint16_t>& out) {
    out.assign((size_t)T6_TOTAL * 8, 0);
    auto P = [&](int dst, int src, int ld, int col0) {
        for (int ob = 0; ob < 4; ++ob)
            for (int ks = 0; ks < 2; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int i = lane & 15, g = lane >> 4;
                        float v = w[src + (16 * ob + i) * ld + col0 + t16_kidx(ks, g, jj)];
                        for (int part = 0; part < 3; ++part) {
                            const uint16_t q = host_bf16_rne(v);
                            out[((size_t)dst + ((part * 4 + ob) * 2 + ks) * 64 + lane) * 8 + jj] = q;
                            v -= host_bf16_to_f32(q);
                        }
                    }
    };
    P(T6_AGG, W_PP_W, 129, 64);
    P(T6_RPR, W_RP_W, 193, 64);
    P(T6_RPS, W_RP_W, 193, 128);
    P(T6_PR0, W_PR0_W, 64, 0);
}

__device__ __forceinline__ void frag16_zero(Frag16& f) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) f.v[ob][r] = 0.0f;
}

// 64 floats in natural feature order (LDS or global, 16-B aligned) -> this lane's 16 features
__device__ __forceinline__ void frag16_from_row(const float* row, int g, Frag16& f) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const float4 t = *reinterpret_cast<const float4*>(row + 16 * ob + 4 * g);
        f.v[ob][0] = t.x; f.v[ob][1] = t.y; f.v[ob][2] = t.z; f.v[ob][3] = t.w;
    }
}

__device__ __forceinline__ void frag16_to_row(float* row, int g, const Frag16& f) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
        *reinterpret_cast<float4*>(row + 16 * ob + 4 * g) = make_float4(f.v[ob][0], f.v[ob][1], f.v[ob][2], f.v[ob][3]);
}

__device__ __forceinline__ void frag16_bias_dens(const float* b_row, const float* wd_row, float d, int g, Frag16& f) {
    Frag16 w;
    frag16_from_row(b_row, g, f);
    frag16_from_row(wd_row, g, w);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) f.v[ob][r] = fmaf(d, w.v[ob][r], f.v[ob][r]);
}

struct FragB16 {
    f16x8 hi[2], lo[2];      // per 32-deep k-step
};

template <bool RELU>
__device__ __forceinline__ void split16(const Frag16& in, FragB16& o) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ob = 2 * ks + (q >> 1), r = 2 * (q & 1);
            float x0 = in.v[ob][r], x1 = in.v[ob][r + 1];
            if (RELU) { x0 = relu1(x0); x1 = relu1(x1); }
            split_pair(x0, x1, o.hi[ks], o.lo[ks], q);
        }
}

#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MB16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// acc += W x, W packed as f16x8[((part*4 + ob)*2 + ks)*64 + lane]; 24 MFMAs.  Output blocks 0,1 (the next
// layer's first k-step) finish first.
__device__ __forceinline__ void mfma16_layer64(const f16x8* __restrict__ wp, const FragB16& b, Frag16& acc, int lane) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 whi = wp[((0 * 4 + ob) * 2 + ks) * 64 + lane];
            const f16x8 wlo = wp[((1 * 4 + ob) * 2 + ks) * 64 + lane];
            acc.v[ob] = MF16(wlo, b.hi[ks], acc.v[ob]);
            acc.v[ob] = MF16(whi, b.lo[ks], acc.v[ob]);
            acc.v[ob] = MF16(whi, b.hi[ks], acc.v[ob]);
        }
}

// first layer: one k-step over [a_r, a_s, dx, dy, dz, d, 1, 0] on lane group 0 (the other groups supply zeros)
__device__ __forceinline__ void mfma16_layer8(const f16x8* __restrict__ wp, const float (&x)[8], int g, Frag16& acc, int lane) {
    f16x8 bhi, blo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v0 = (g == 0) ? x[2 * q] : 0.0f, v1 = (g == 0) ? x[2 * q + 1] : 0.0f;
        split_pair(v0, v1, bhi, blo, q);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const f16x8 whi = wp[(0 * 4 + ob) * 64 + lane];
        const f16x8 wlo = wp[(1 * 4 + ob) * 64 + lane];
        acc.v[ob] = MF16(wlo, bhi, acc.v[ob]);
        acc.v[ob] = MF16(whi, blo, acc.v[ob]);
        acc.v[ob] = MF16(whi, bhi, acc.v[ob]);
    }
}

struct FragB16x6 {
    bf16x8 p[3][2];          // [part][k-step]
};

__device__ __forceinline__ void split16_6(const Frag16& in, FragB16x6& o) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float x = in.v[2 * ks + (jj >> 2)][jj & 3];
            const __bf16 hi = (__bf16)x;
            const float r1 = x - (float)hi;
            const __bf16 mid = (__bf16)r1;
            o.p[0][ks][jj] = hi;
            o.p[1][ks][jj] = mid;
            o.p[2][ks][jj] = (__bf16)(r1 - (float)mid);
        }
}

__device__ __forceinline__ void mfma16_layer64_6(const bf16x8* __restrict__ wp, const FragB16x6& b, Frag16& acc, int lane) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 w0 = wp[((0 * 4 + ob) * 2 + ks) * 64 + lane];
            const bf16x8 w1 = wp[((1 * 4 + ob) * 2 + ks) * 64 + lane];
            const bf16x8 w2 = wp[((2 * 4 + ob) * 2 + ks) * 64 + lane];
            acc.v[ob] = MB16(w2, b.p[0][ks], acc.v[ob]);
            acc.v[ob] = MB16(w0, b.p[2][ks], acc.v[ob]);
            acc.v[ob] = MB16(w1, b.p[1][ks], acc.v[ob]);
            acc.v[ob] = MB16(w1, b.p[0][ks], acc.v[ob]);
            acc.v[ob] = MB16(w0, b.p[1][ks], acc.v[ob]);
            acc.v[ob] = MB16(w0, b.p[0][ks], acc.v[ob]);
        }
}

struct Prop16Lds {
    const f16x8* wsp;         // edge chain, T_* offsets
    const bf16x8* w_agg;
    const bf16x8* w_x;        // !LAST: W_r | W_s;  LAST: predictor layer 0
    const float* rows;        // b2, b4, b_rp, wd_rp
    const float* rows_pr;     // b_pr0, w_pr1[3], b_pr1
    int* tile_ctr;
};

// the tile loop of one propagation step over this workgroup's samples [b0, b0 + nb): prop_tiles of
// k_mlp_split.h on 16 receivers per wave
template <bool LAST>
__device__ __forceinline__ void prop_tiles16(const PropArgs& A, const Prop16Lds& L, int b0, int nb, int lane, int wave) {
    const float* mw = A.mw;
    const float* s_cur = A.s_cur; const int s_mod = A.s_mod; const size_t s_stride = A.s_stride;
    const float* attr = A.attr; const int attr_mod = A.attr_mod;
    const float* dens = A.dens; const int dens_mod = A.dens_mod;
    const int16_t* nbr_idx = A.nbr_idx; const uint8_t* nbr_cnt = A.nbr_cnt;
    const float* proj = A.proj; const float* c_node = A.c_node; float* eff = A.eff;
    const int N = A.N;
    float* proj_next = A.proj_next; float* s_out = A.s_out; const size_t out_stride = A.out_stride;
    const float* cself = A.cself; const uint8_t* cself_ok = A.cself_ok;
    const f16x8* wsp = L.wsp;
    const float* rows = L.rows;
    const int j = lane & 15, g = lane >> 4;
    const int tps = (N + 15) >> 4;
    const int wg_tiles = (nb > 0 ? nb : 0) * tps;
    struct TileHead {
        int cnt, ok;
        unsigned nbw0, nbw1;
        float pix, piy, piz, pia;
    };
    auto tile_head = [&](int li) {
        const int m = li / tps, t = li - m * tps;
        const int b = b0 + m;
        const int i = min(t * 16 + j, N - 1);
        const size_t row = (size_t)b * N + i;
        const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
        TileHead hd;
        hd.cnt = nbr_cnt[row];
        hd.ok = (cself != nullptr) ? (int)cself_ok[b] : 0;
        const unsigned* nbw = reinterpret_cast<const unsigned*>(nbr_idx + row * DRP_K);
        hd.nbw0 = nbw[0];
        hd.nbw1 = nbw[1];
        hd.pix = s[i * 3 + 0]; hd.piy = s[i * 3 + 1]; hd.piz = s[i * 3 + 2];
        hd.pia = attr[(size_t)(b % attr_mod) * N + i];
        return hd;
    };
    struct TileFirst {
        int ks, j0, j1;
        float p0x, p0y, p0z, p0a;
    };
    auto tile_first = [&](int li, const TileHead& hd) {
        const int m = li / tps, t = li - m * tps;
        const int b = b0 + m;
        const int i = min(t * 16 + j, N - 1);
        const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
        const float* at = attr + (size_t)(b % attr_mod) * N;
        const int nb0 = (int)(hd.nbw0 & 0xffffu), nb1 = (int)(hd.nbw0 >> 16), nb2 = (int)(hd.nbw1 & 0xffffu);
        TileFirst f;
        f.ks = (hd.ok && __all(hd.cnt > 0 && nb0 == i)) ? 1 : 0;
        f.j0 = (f.ks < hd.cnt) ? (f.ks ? nb1 : nb0) : i;
        f.j1 = (f.ks + 1 < hd.cnt) ? (f.ks ? nb2 : nb1) : i;
        f.p0x = s[f.j0 * 3 + 0]; f.p0y = s[f.j0 * 3 + 1]; f.p0z = s[f.j0 * 3 + 2]; f.p0a = at[f.j0];
        return f;
    };
    int li = wave, li_next = 0;
    TileHead hd_next = {};
    TileFirst tf_next = {};
    if (li < wg_tiles) {
        hd_next = tile_head(li);
        tf_next = tile_first(li, hd_next);
    }
    for (; li < wg_tiles; li = li_next) {
        const int m = li / tps, t = li - m * tps;
        const int b = b0 + m;
        const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
        const float* at = attr + (size_t)(b % attr_mod) * N;
        const float* pj = proj + (size_t)b * N * 128;
        const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
        const int i = min(t * 16 + j, N - 1);
        const bool live = (t * 16 + j) < N;
        const size_t row = (size_t)b * N + i;
        const TileHead hd = hd_next;
        const TileFirst tf = tf_next;
        const int cnt = hd.cnt;
        const int16_t* nb = nbr_idx + row * DRP_K;
        Frag16 acc, bpr;
        {
            Frag16 pr;
            frag16_bias_dens(rows + 128, rows + 192, d, g, bpr);
            frag16_from_row(pj + (size_t)i * 128, g, pr);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) bpr.v[ob][r] += pr.v[ob][r];
        }
        const int ks = tf.ks;
        if (ks) {
            Frag16 cs, ps;
            frag16_from_row(cself + (size_t)b * 64, g, cs);
            frag16_from_row(pj + (size_t)i * 128 + 64, g, ps);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc.v[ob][r] = relu1((bpr.v[ob][r] + cs.v[ob][r]) + ps.v[ob][r]);
        } else {
            frag16_zero(acc);
        }
        const float pix = hd.pix, piy = hd.piy, piz = hd.piz, pia = hd.pia;
        int j0 = tf.j0, j1 = tf.j1;
        float p0x = tf.p0x, p0y = tf.p0y, p0z = tf.p0z, p0a = tf.p0a;
#pragma unroll 1
        for (int k = ks; k < DRP_K; ++k) {
            if (__all(k >= cnt)) break;
            asm volatile("" ::: "memory");          // keep the packed-weight reads inside the loop
            const int jcur = j0;
            const float p1x = s[j1 * 3 + 0], p1y = s[j1 * 3 + 1], p1z = s[j1 * 3 + 2], p1a = at[j1];
            const int j2 = (k + 2 < cnt) ? (int)nb[min(k + 2, DRP_K - 1)] : i;
            float x[8];
            x[0] = pia; x[1] = p0a;
            x[2] = pix - p0x; x[3] = piy - p0y; x[4] = piz - p0z;
            x[5] = d; x[6] = 1.0f; x[7] = 0.0f;
            Frag16 sv;
            frag16_from_row((k < cnt) ? pj + (size_t)jcur * 128 + 64 : mw + R_SINK, g, sv);
            Frag16 a, c;
            FragB16 fb;
            frag16_zero(a);
            mfma16_layer8(wsp + T_RE0, x, g, a, lane);
            split16<true>(a, fb);
            frag16_from_row(rows + 0, g, c);
            mfma16_layer64(wsp + T_RE2, fb, c, lane);
            split16<true>(c, fb);
            frag16_from_row(rows + 64, g, a);
            mfma16_layer64(wsp + T_RE4, fb, a, lane);
            split16<true>(a, fb);
            c = bpr;
            mfma16_layer64(wsp + T_RPE, fb, c, lane);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc.v[ob][r] += relu1(c.v[ob][r] + sv.v[ob][r]);
            j0 = j1; j1 = j2;
            p0x = p1x; p0y = p1y; p0z = p1z; p0a = p1a;
        }
        // ---- node update on the aggregate still in registers
        asm volatile("" ::: "memory");
        {
            int q = 0;
            if (lane == 0) q = __hip_atomic_fetch_add(L.tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            li_next = __builtin_amdgcn_readfirstlane(q);
        }
        const bool more = li_next < wg_tiles;
        if (more) hd_next = tile_head(li_next);
        __builtin_amdgcn_sched_barrier(0);
        Frag16 e;
        {
            Frag16 cn;
            frag16_from_row(eff + row * 64, g, e);
            frag16_from_row(c_node + row * 64, g, cn);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) e.v[ob][r] += cn.v[ob][r];
        }
        FragB16x6 f6;
        split16_6(acc, f6);
        mfma16_layer64_6(L.w_agg, f6, e, lane);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) e.v[ob][r] = relu1(e.v[ob][r]);
        if (live) frag16_to_row(eff + row * 64, g, e);
        split16_6(e, f6);
        __builtin_amdgcn_sched_barrier(0);
        if (more) tf_next = tile_first(li_next, hd_next);
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST) {
            Frag16 p;
            frag16_zero(p);
            mfma16_layer64_6(L.w_x, f6, p, lane);
            if (live) frag16_to_row(proj_next + row * 128, g, p);
            frag16_zero(p);
            mfma16_layer64_6(L.w_x + 1536, f6, p, lane);
            if (live) frag16_to_row(proj_next + row * 128 + 64, g, p);
        } else {
            Frag16 hdn;
            frag16_from_row(L.rows_pr, g, hdn);
            mfma16_layer64_6(L.w_x, f6, hdn, lane);
            float out[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                Frag16 w;
                frag16_from_row(L.rows_pr + 64 + 64 * o, g, w);
                float p = 0.0f;
#pragma unroll
                for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p = fmaf(relu1(hdn.v[ob][r]), w.v[ob][r], p);
                p += __shfl_xor(p, 16, 64);
                out[o] = p + __shfl_xor(p, 32, 64);
            }
            if (g == 0 && live) {
                float* so = s_out + (size_t)b * out_stride + (size_t)i * 3;
#pragma unroll
                for (int o = 0; o < 3; ++o) so[o] = (out[o] + L.rows_pr[64 + 192 + o]) + s[i * 3 + o];
            }
        }
    }
}

// the three propagation steps of a rollout step in one launch, workgroup w owns samples [w*spw, (w+1)*spw)
__global__ void __launch_bounds__(64 * PROP16_WAVES)
km_prop3_t16(const uint16_t* __restrict__ sw, const uint16_t* __restrict__ sw6, const float* __restrict__ mw,
             const float* __restrict__ s_cur, int s_mod, size_t s_stride,
             const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
             const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt,
             float* __restrict__ proj_a, float* __restrict__ proj_b, const float* __restrict__ c_node,
             float* __restrict__ eff, int N, int B, int spw, float* __restrict__ s_out, size_t out_stride,
             const float* __restrict__ cself, const uint8_t* __restrict__ cself_ok) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wsp_f = lds;
    float* w6_f = wsp_f + T_TOTAL * 4;               // AGG | RPR | RPS | PR0
    float* rows = w6_f + T6_TOTAL * 4;               // b2,b4,b_rp,wd_rp | b_pr0, w_pr1[3], b_pr1
    lds_fill(wsp_f, reinterpret_cast<const float*>(sw), T_TOTAL * 4);
    lds_fill(w6_f, reinterpret_cast<const float*>(sw6), T6_TOTAL * 4);
    lds_fill(rows, mw + R_RE2_B, 256);
    lds_fill(rows + 256, mw + R_PR0_B, 260);
    int* tile_ctr = reinterpret_cast<int*>(rows + 516);
    if (threadIdx.x == 0) *tile_ctr = PROP16_WAVES;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * spw, nb = min(spw, B - b0);
    PropArgs A = {mw, s_cur, s_mod, s_stride, attr, attr_mod, dens, dens_mod, nbr_idx, nbr_cnt, proj_a, c_node, eff, eff,
                  N, B, proj_b, s_out, out_stride, cself, cself_ok, nullptr, nullptr};
    Prop16Lds L = {reinterpret_cast<const f16x8*>(wsp_f), reinterpret_cast<const bf16x8*>(w6_f) + T6_AGG,
                   reinterpret_cast<const bf16x8*>(w6_f) + T6_RPR, rows, rows + 256, tile_ctr};
#pragma unroll 1
    for (int p = 0; p < DRP_PSTEP; ++p) {
        if (p > 0) {
            __syncthreads();
            if (threadIdx.x == 0) *tile_ctr = PROP16_WAVES;
            __syncthreads();
        }
        A.proj = (p & 1) ? proj_b : proj_a;
        A.proj_next = (p & 1) ? proj_a : proj_b;
        if (p + 1 < DRP_PSTEP) {
            prop_tiles16<false>(A, L, b0, nb, lane, wave);
        } else {
            L.w_x = reinterpret_cast<const bf16x8*>(w6_f) + T6_PR0;
            prop_tiles16<true>(A, L, b0, nb, lane, wave);
        }
    }
}
#define KM_PROP3_T16_LDS ((size_t)(T_TOTAL * 4 + T6_TOTAL * 4 + 256 + 260 + 4) * sizeof(float))
