// Node-level stages of the reverse-mode pass on the fp32 matrix cores (rows f1 / f4).
// Each stage is a transposed dense product g_in = W^T g_out over B*N rows; computed like the
// forward chain of k_mlp_mfma.h, D[in][item] = sum_out W^T[in][out] G[out][item], with W^T as the
// packed A operand and the 32 rows of a tile on the lane columns (v_mfma_f32_32x32x2_f32: exact
// fp32 products, fp32 accumulation).  The VALU versions (k_backward.h: kb_update, kb_project,
// kb_predict, kb_node_encode) read their weights row by row from L2 and were latency bound:
// 930 us of a 2.2 ms planner iteration.
#pragma once
#include "k_backward.h"
#include "k_mlp_mfma.h"
#include "k_mlp_split.h"
#include "k_train.h"

// transposed 64x64 blocks, packed like M_* ([ob 2][s4 8][lane 64][c 4])
enum {
    MB_AGG = 0,                 // W_agg^T
    MB_RPR = MB_AGG + 4096,     // W_r^T
    MB_RPS = MB_RPR + 4096,     // W_s^T
    MB_PPE = MB_RPS + 4096,     // W_pe^T
    MB_PE2 = MB_PPE + 4096,     // particle encoder layer 2, transposed
    MB_PR0 = MB_PE2 + 4096,     // predictor layer 0, transposed
    RB_PE0 = MB_PR0 + 4096,     // particle encoder layer 0, columns 0..2 as three 64-rows (d / d s_delta)
    MB_RPE = RB_PE0 + 192,      // W_e^T            } the relation encoder's backward (kmb_edge_encode)
    MB_RE4 = MB_RPE + 4096,     // layer 4, transposed
    MB_RE2 = MB_RE4 + 4096,     // layer 2, transposed
    RB_RE0 = MB_RE2 + 4096,     // relation encoder layer 0, columns 2..4 as three 64-rows (d / d (s_r - s_s))
    MB_TOTAL = RB_RE0 + 192
};

inline void pack_mfma_bwd(const float* w, std::vector<float>& m) {
    m.assign(MB_TOTAL, 0.0f);
    auto PT64 = [&](int dst, int src, int ld, int col0) {      // A[i][k] = W[k][col0 + i]
        for (int ob = 0; ob < 2; ++ob)
            for (int s = 0; s < 32; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    m[dst + ((ob * 8 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] =
                        w[src + mfma_kidx(s, h) * ld + col0 + 32 * ob + i];
                }
    };
    PT64(MB_AGG, W_PP_W, 129, 64);
    PT64(MB_RPR, W_RP_W, 193, 64);
    PT64(MB_RPS, W_RP_W, 193, 128);
    PT64(MB_PPE, W_PP_W, 129, 0);
    PT64(MB_PE2, W_PE2_W, 64, 0);
    PT64(MB_PR0, W_PR0_W, 64, 0);
    for (int c = 0; c < 3; ++c)
        for (int o = 0; o < 64; ++o) m[RB_PE0 + c * 64 + o] = w[W_PE0_W + o * 5 + c];
    PT64(MB_RPE, W_RP_W, 193, 0);
    PT64(MB_RE4, W_RE4_W, 64, 0);
    PT64(MB_RE2, W_RE2_W, 64, 0);
    for (int c = 0; c < 3; ++c)
        for (int o = 0; o < 64; ++o) m[RB_RE0 + c * 64 + o] = w[W_RE0_W + o * 6 + 2 + c];
}

// One propagation step of the backward pass on the node rows:
//   PROJECT: g_eff += W_r^T g_proj[:, 0:64] + W_s^T g_proj[:, 64:128]          (kb_project)
//   UPDATE : g_z = g_eff . [eff_next > 0]; g_eff <- g_z; g_cnode (+)= g_z; g_agg = W_agg^T g_z   (kb_update)
// PROJECT of step p and UPDATE of step p-1 touch the same rows only: one launch does both.
template <bool PROJECT, bool UPDATE>
__global__ void __launch_bounds__(64 * MFMA_WAVES)
kmb_node_step(const float* __restrict__ mb, const float* g_eff_in, float* g_eff /* out; may be g_eff_in (row-local) */,
              const float* __restrict__ g_proj, const float* __restrict__ eff_next, float* __restrict__ g_cnode, int first,
              float* __restrict__ g_agg, int N, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wr = lds;
    float* ws = wr + (PROJECT ? 4096 : 0);
    float* wa = ws + (PROJECT ? 4096 : 0);
    if (PROJECT) { lds_fill(wr, mb + MB_RPR, 4096); lds_fill(ws, mb + MB_RPS, 4096); }
    if (UPDATE) lds_fill(wa, mb + MB_AGG, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tps = (N + 31) >> 5;
    const long ntiles = (long)B * tps;
    for (long gt = (long)blockIdx.x + (long)gridDim.x * wave; gt < ntiles; gt += (long)gridDim.x * MFMA_WAVES) {   // workgroup-cyclic first: few tiles spread one per CU
        const int b = (int)(gt / tps), t = (int)(gt - (long)b * tps);
        const int i = min(t * 32 + j, N - 1);
        const bool live = (t * 32 + j) < N;
        const size_t row = (size_t)b * N + i;
        Frag ge;
        frag_from_row(g_eff_in + row * 64, h, ge);
        if (PROJECT) {
            Frag g;
            frag_from_row(g_proj + row * 128, h, g);
            mfma_layer64<false>(reinterpret_cast<const float4*>(wr), g, ge, lane);
            frag_from_row(g_proj + row * 128 + 64, h, g);
            mfma_layer64<false>(reinterpret_cast<const float4*>(ws), g, ge, lane);
        }
        if (UPDATE) {
            Frag en, ga;
            frag_from_row(eff_next + row * 64, h, en);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ge.v[0][r] = en.v[0][r] > 0.0f ? ge.v[0][r] : 0.0f;
                ge.v[1][r] = en.v[1][r] > 0.0f ? ge.v[1][r] : 0.0f;
            }
            if (live) frag_to_row(g_eff + row * 64, h, ge);
            if (!first) {
                frag_from_row(g_cnode + row * 64, h, en);
#pragma unroll
                for (int r = 0; r < 16; ++r) { en.v[0][r] += ge.v[0][r]; en.v[1][r] += ge.v[1][r]; }
                if (live) frag_to_row(g_cnode + row * 64, h, en);
            } else if (live) {
                frag_to_row(g_cnode + row * 64, h, ge);
            }
            frag_zero(ga);
            mfma_layer64<false>(reinterpret_cast<const float4*>(wa), ge, ga, lane);
            if (live) frag_to_row(g_agg + row * 64, h, ga);
        } else if (live) {
            frag_to_row(g_eff + row * 64, h, ge);
        }
    }
}
#define KMB_STEP_LDS(PROJECT, UPDATE) ((size_t)(((PROJECT) ? 2 : 0) + ((UPDATE) ? 1 : 0)) * 4096 * sizeof(float))

// predictor backward (kb_predict): g_eff = W0^T ((W1^T g_out) . [W0 eff + b0 > 0]); optional dumps of
// relu(hidden) and of the hidden pre-activation gradient for the weight gradients
__global__ void __launch_bounds__(64 * MFMA_WAVES)
kmb_predict(const float* __restrict__ mw, const float* __restrict__ mb, const float* __restrict__ eff,
            const float* __restrict__ g_out, size_t g_stride, int N, int B, float* __restrict__ g_eff,
            float* __restrict__ dump_hact, float* __restrict__ dump_gh) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w0 = lds;                 // forward pack of predictor layer 0
    float* w0t = w0 + 4096;          // transposed
    float* rows = w0t + 4096;        // b_pr0 [64], w_pr1 [3][64]
    lds_fill(w0, mw + M_PR0, 4096);
    lds_fill(w0t, mb + MB_PR0, 4096);
    lds_fill(rows, mw + R_PR0_B, 256);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tps = (N + 31) >> 5;
    const long ntiles = (long)B * tps;
    for (long gt = (long)blockIdx.x + (long)gridDim.x * wave; gt < ntiles; gt += (long)gridDim.x * MFMA_WAVES) {   // workgroup-cyclic first: few tiles spread one per CU
        const int b = (int)(gt / tps), t = (int)(gt - (long)b * tps);
        const int i = min(t * 32 + j, N - 1);
        const bool live = (t * 32 + j) < N;
        const size_t row = (size_t)b * N + i;
        Frag x, hh, gh, ge;
        frag_from_row(eff + row * 64, h, x);
        frag_from_row(rows, h, hh);
        mfma_layer64<false>(reinterpret_cast<const float4*>(w0), x, hh, lane);
        const float* go = g_out + (size_t)b * g_stride + (size_t)i * 3;
        const float g0 = go[0], g1 = go[1], g2 = go[2];
        {
            Frag wx, wy, wz;
            frag_from_row(rows + 64, h, wx);
            frag_from_row(rows + 128, h, wy);
            frag_from_row(rows + 192, h, wz);
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g = wx.v[ob][r] * g0 + wy.v[ob][r] * g1 + wz.v[ob][r] * g2;
                    gh.v[ob][r] = hh.v[ob][r] > 0.0f ? g : 0.0f;
                }
        }
        frag_zero(ge);
        mfma_layer64<false>(reinterpret_cast<const float4*>(w0t), gh, ge, lane);
        if (live) {
            frag_to_row(g_eff + row * 64, h, ge);
            if (dump_hact != nullptr) {
                frag_relu(hh);
                frag_to_row(dump_hact + row * 64, h, hh);
                frag_to_row(dump_gh + row * 64, h, gh);
            }
        }
    }
}
#define KMB_PREDICT_LDS ((size_t)(2 * 4096 + 256) * sizeof(float))

// particle encoder backward (kb_node_encode): g_pe = g_eff0 + W_pe^T g_cnode, through
// relu(W2 relu(W1 x + b1) + b2) to the three impulse inputs; optional dumps for the weight gradients
__global__ void __launch_bounds__(64 * MFMA_WAVES)
kmb_node_encode(const float* __restrict__ mw, const float* __restrict__ mb, const float* __restrict__ s_delta,
                const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
                const float* __restrict__ pe, const float* __restrict__ g_eff0, const float* __restrict__ g_cnode,
                int N, int B, float* __restrict__ g_sdelta, float* __restrict__ dump_gpe, float* __restrict__ dump_a1,
                float* __restrict__ dump_gh1, float* __restrict__ dump_x) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w1 = lds;                 // forward pack of encoder layer 0 (K = 8, bias column)
    float* wpet = w1 + 512;          // W_pe^T
    float* w2t = wpet + 4096;        // layer 2, transposed
    float* rows = w2t + 4096;        // layer-0 columns 0..2: [3][64]
    lds_fill(w1, mw + M_PE0, 512);
    lds_fill(wpet, mb + MB_PPE, 4096);
    lds_fill(w2t, mb + MB_PE2, 4096);
    lds_fill(rows, mb + RB_PE0, 192);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tps = (N + 31) >> 5;
    const long ntiles = (long)B * tps;
    for (long gt = (long)blockIdx.x + (long)gridDim.x * wave; gt < ntiles; gt += (long)gridDim.x * MFMA_WAVES) {   // workgroup-cyclic first: few tiles spread one per CU
        const int b = (int)(gt / tps), t = (int)(gt - (long)b * tps);
        const int i = min(t * 32 + j, N - 1);
        const bool live = (t * 32 + j) < N;
        const size_t row = (size_t)b * N + i;
        const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
        const float* sd = s_delta + row * 3;
        const float at = attr[(size_t)(b % attr_mod) * N + i];
        float x[4];                  // inputs [sdx, sdy, sdz, a, d, 1, 0, 0]; this lane supplies index 2s + h
        if (h == 0) { x[0] = sd[0]; x[1] = sd[2]; x[2] = d; x[3] = 0.0f; }
        else { x[0] = sd[1]; x[1] = at; x[2] = 1.0f; x[3] = 0.0f; }
        Frag h1, gpe, g, gh;
        frag_zero(h1);
        mfma_layer8(reinterpret_cast<const float4*>(w1), x, h1, lane);
        frag_from_row(g_eff0 + row * 64, h, gpe);
        frag_from_row(g_cnode + row * 64, h, g);
        mfma_layer64<false>(reinterpret_cast<const float4*>(wpet), g, gpe, lane);
        frag_from_row(pe + row * 64, h, g);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gpe.v[0][r] = g.v[0][r] > 0.0f ? gpe.v[0][r] : 0.0f;
            gpe.v[1][r] = g.v[1][r] > 0.0f ? gpe.v[1][r] : 0.0f;
        }
        frag_zero(gh);
        mfma_layer64<false>(reinterpret_cast<const float4*>(w2t), gpe, gh, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gh.v[0][r] = h1.v[0][r] > 0.0f ? gh.v[0][r] : 0.0f;
            gh.v[1][r] = h1.v[1][r] > 0.0f ? gh.v[1][r] : 0.0f;
        }
        float out[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            Frag w;
            frag_from_row(rows + 64 * o, h, w);
            float p = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) p = fmaf(gh.v[0][r], w.v[0][r], p);
#pragma unroll
            for (int r = 0; r < 16; ++r) p = fmaf(gh.v[1][r], w.v[1][r], p);
            out[o] = p + __shfl_xor(p, 32, 64);
        }
        if (live && h == 0) {
            g_sdelta[row * 3 + 0] = out[0];
            g_sdelta[row * 3 + 1] = out[1];
            g_sdelta[row * 3 + 2] = out[2];
        }
        if (live && dump_gpe != nullptr) {
            frag_to_row(dump_gpe + row * 64, h, gpe);
            frag_to_row(dump_gh1 + row * 64, h, gh);
            frag_relu(h1);
            frag_to_row(dump_a1 + row * 64, h, h1);
            if (h == 0) {
                float* dx = dump_x + row * 8;
                dx[0] = sd[0]; dx[1] = sd[1]; dx[2] = sd[2]; dx[3] = at; dx[4] = d; dx[5] = 0.0f; dx[6] = 0.0f; dx[7] = 0.0f;
            }
        }
    }
}
// below this many 32-row tiles the chunked VALU row kernels of k_backward.h are faster (a workgroup's
// LDS fill of the packed weights is not amortised)
#define KMB_MIN_TILES 1          // round 3: the tiles are dealt workgroup-cyclically, so a handful of them (a training batch) runs one per CU
#define KMB_NODE_ENCODE_LDS ((size_t)(512 + 2 * 4096 + 192) * sizeof(float))

// acc += scale * v where the row's mask word has the element's bit (bit 31 - (16 ob + q)), + 0 elsewhere: one signed
// bit-field extract (0 / -1), one AND, one add per element -- inline assembly: written in C the optimiser turns the
// extract back into AND + compare + select and adds a copy per element (330 instructions per list entry instead of 96),
// and with the temporary inside the statement no 32 of them are alive at once.
template <int POS>
__device__ __forceinline__ float add_if_bit(float acc, float v, unsigned w) {
    float t;
    asm("v_bfe_i32 %1, %3, %4, 1\n\tv_and_b32 %1, %1, %2\n\tv_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(t) : "v"(v), "v"(w), "n"(POS));
    return acc;
}
template <int POS>
__device__ __forceinline__ float fma_if_bit(float acc, float v, unsigned w, float scale) {
    float t;
    asm("v_bfe_i32 %1, %3, %4, 1\n\tv_and_b32 %1, %1, %2\n\tv_fmac_f32 %0, %1, %5" : "+v"(acc), "=&v"(t) : "v"(v), "v"(w), "n"(POS), "v"(scale));
    return acc;
}
template <int E>
__device__ __forceinline__ void frag_add_masked_from(Frag& acc, const Frag& v, unsigned w) {
    if constexpr (E < 32) {
        acc.v[E >> 4][E & 15] = add_if_bit<31 - E>(acc.v[E >> 4][E & 15], v.v[E >> 4][E & 15], w);
        frag_add_masked_from<E + 1>(acc, v, w);
    }
}
__device__ __forceinline__ void frag_add_masked(Frag& acc, const Frag& v, unsigned w) { frag_add_masked_from<0>(acc, v, w); }
template <int E>
__device__ __forceinline__ void frag_fma_masked_from(Frag& acc, const Frag& v, unsigned w, float scale) {
    if constexpr (E < 32) {
        acc.v[E >> 4][E & 15] = fma_if_bit<31 - E>(acc.v[E >> 4][E & 15], v.v[E >> 4][E & 15], w, scale);
        frag_fma_masked_from<E + 1>(acc, v, w, scale);
    }
}
// The statements above are opaque to the compiler's hazard recognizer: it does not know that their last instruction is
// a vector write, and an MFMA reading that register needs two wait states after one (seen: the receiver term's element
// consumed by the very next v_mfma with one s_nop between -- stale operands, wrong gradients).  A fragment built by
// them goes through one VISIBLE vector instruction per element before the matrix cores read it: times an exact 1.0
// the optimiser cannot see through.
__device__ __forceinline__ void frag_settle(Frag& f) {
    float one = 1.0f;
    asm volatile("" : "+v"(one));
#pragma unroll
    for (int r = 0; r < 16; ++r) { f.v[0][r] *= one; f.v[1][r] *= one; }
}
// receiver term of a row: sum over its slots of (slot's mask) . g -- the SAME row g under every mask, so per element it is
// g times the number of slots whose mask has the bit: the ten mask words are added bit-sliced (four planes: counts up
// to 15), then pr = sum_b 2^b (plane_b . g): four masked terms instead of ten (each term exact, three roundings)
__device__ __forceinline__ void receiver_term(Frag& pr, const Frag& g, const unsigned (&wk)[DRP_K]) {
    unsigned c0 = 0u, c1 = 0u, c2 = 0u, c3 = 0u;
#pragma unroll
    for (int k = 0; k < DRP_K; ++k) {
        unsigned carry = wk[k], t;
        t = c0 & carry; c0 ^= carry; carry = t;
        t = c1 & carry; c1 ^= carry; carry = t;
        t = c2 & carry; c2 ^= carry; carry = t;
        c3 ^= carry;
    }
    frag_zero(pr);
    frag_add_masked(pr, g, c0);
    frag_fma_masked_from<0>(pr, g, c1, 2.0f);
    frag_fma_masked_from<0>(pr, g, c2, 4.0f);
    frag_fma_masked_from<0>(pr, g, c3, 8.0f);
    frag_settle(pr);
}

// ---- the whole node / edge part of one rollout step's backward pass in ONE launch (the GD planner) ------------
// The sequence above -- kmb_predict, then per propagation step kb_edge_terms between two kmb_node_step halves,
// then kmb_node_encode -- is nine launches whose intermediates (g_proj [B,N,128] written and read three times,
// g_eff, g_cnode) cross HBM each time: 0.75 ms of a 1.31 ms planner iteration at the demo shape, none of it
// arithmetic.  Here a workgroup owns whole samples, as km_prop3 does forward: the only cross-row dependency of the
// backward pass is the gather of g_agg rows over the REVERSED neighbour lists, and those are rows of the same
// sample, so __syncthreads() between phases is all the ordering it needs.
//   phase P : predictor backward and the update of propagation step 2          (row-local)   -> g_eff, g_cnode, g_agg[2]
//   phase p = 2, 1, 0 : receiver term (own masks) and sender term (reversed lists, masks, gathered g_agg[p] rows)
//             stay in registers and go straight into W_r^T / W_s^T; then the update of step p-1 (-> g_agg[p-1]),
//             or, for p = 0, the particle encoder's backward (-> g_s_delta)       -> g_proj never exists
// Tiles are 32 consecutive rows of the workgroup's row list (only its last tile has idle lanes), drawn on demand.
// Every sum runs in a fixed order (bit planes of the slot count, reversed-list order): gradients are bit-reproducible.
// fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout, as the launch-per-stage kernels.
#ifdef ROLLOUT_STAMPS
// Diagnostic build only (tools/bwd_stamps.py): 100 MHz wall stamps between the phases of a group, summed by wave 0 of
// every 32nd workgroup
__device__ unsigned long long g_bwd_stamps[16];
#define BWD_STAMP(q) do { if (roll_on) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
                                         atomicAdd(&g_bwd_stamps[q], now_ - roll_t); roll_t = now_; } } while (0)
#else
#define BWD_STAMP(q) do { } while (0)
#endif
#define KMB_FUSED_WAVES 8
#define KMB_ROWS_LD 68            // floats per 64-float row in LDS: 17 float4 -- consecutive rows start four banks apart
__device__ __forceinline__ void gagg_lds_read(const float* gl, int r, int h, Frag& f) { frag_from_row(gl + r * KMB_ROWS_LD, h, f); }
__device__ __forceinline__ void gagg_lds_write(float* gl, int r, int h, const Frag& f) { frag_to_row(gl + r * KMB_ROWS_LD, h, f); }
#define KMB_FUSED_LDS ((size_t)(7 * 4096 + 512 + 256 + 192 + 4) * sizeof(float))
// DUMP (the trainer): what the weight gradients read leaves the launch as well -- per propagation step the pre-activation
// gradient (ge[0..2]: steps 2, 1, 0; the chain reads ge[k] and writes ge[k + 1] instead of overwriting g_eff) and both edge
// terms (gp[p]: [rows][128] = receiver | sender), the predictor's hidden activation and its gradient, the particle encoder's
// (kmb_predict's / kmb_node_encode's dumps).  A training batch is a handful of samples: with `parts` > 1 a group of samples
// is shared by `parts` workgroups (tiles dealt statically: part + wave * parts, stride parts * waves) and the phases are
// separated by a barrier among those workgroups -- a counter in memory per group (`bar`, zeroed by the host; every wave's
// stores are performed at device scope before its workgroup arrives, every wave invalidates its L1 after the wait).  The
// host keeps the grid at or below the number of CUs (all workgroups resident -- each needs KMB_COOP_LDS, one per CU -- which
// ASSUMES the process has the device to itself: under a CU mask, or beside another process's LDS-heavy kernel, a group's
// workgroups may not all be resident); a wait that lasts two seconds gives up and sets bar_err instead of hanging the device.
// drp_train_step then moves nothing (k_adam reads the flag on the device) and runs the step again with one workgroup per
// group, for the rest of the context's life (capi_train.h).
struct KmbDump {
    float* hact; float* gh;          // predictor: relu(hidden), hidden pre-activation gradient        [rows][64]
    float* ge[3];                    // pre-activation gradient of propagation steps 2, 1, 0           [rows][64]
    float* gp[3];                    // edge terms of propagation step p                                [rows][128]
    float* gpe; float* a1n; float* gh1; float* xn;   // particle encoder (kmb_node_encode's dumps)
};
// both edge terms of one row by 16 lanes (lane q: features 4q .. 4q + 3), kb_edge_terms' layout with kmb_step_bwd's own
// arithmetic (the receiver term by the bit planes of the slot count, the sender term entry by entry in list order): the same
// bits as the wave-wide gather below, with 32 rows of a tile in flight over the workgroup's 512 threads instead of one wave's
// two-deep pipeline
__device__ __forceinline__ float plane_sum(int n, float g) {
    float acc = 0.0f + ((n & 1) ? g : 0.0f);
    acc = fmaf((n & 2) ? g : 0.0f, 2.0f, acc);
    acc = fmaf((n & 4) ? g : 0.0f, 4.0f, acc);
    return fmaf((n & 8) ? g : 0.0f, 8.0f, acc);
}
__device__ __forceinline__ void coop_edge_terms(const float* g_agg_p, const unsigned* __restrict__ mk /* the sample's slots */,
                                                const int* __restrict__ rv, int p0, int p1, int i, int cnt, size_t srow0, int q,
                                                float4& pr, float4& ps) {
    // Every load that does not depend on another is requested before the first wait: the list's first eight entries, the
    // row's own mask words and g_agg row; then per batch of eight entries their mask words and rows TOGETHER with the next
    // batch's entries -- one L2 round trip per eight entries instead of two per four
    constexpr int W = 8;
    const float4* ga = reinterpret_cast<const float4*>(g_agg_p);
    int e[W];
#pragma unroll
    for (int u = 0; u < W; ++u) e[u] = (p0 + u < p1) ? rv[p0 + u] : -1;
    const float4 gi = ga[(srow0 + i) * 16 + q];
    unsigned wk[DRP_K];
#pragma unroll
    for (int k = 0; k < DRP_K; ++k) wk[k] = (k < cnt) ? mk[((size_t)i * DRP_K + k) * 2 + (q & 1)] : 0u;
    const int sh = 28 - 16 * (q >> 3) - 4 * ((q >> 1) & 3);
    ps = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pp = p0; pp < p1; pp += W) {
        unsigned w[W];
        float4 v[W];
        int en[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int ee = e[u] < 0 ? 0 : e[u];
            w[u] = e[u] < 0 ? 0u : (mk[(size_t)ee * 2 + (q & 1)] >> sh) & 0xfu;
            v[u] = ga[(srow0 + ee / DRP_K) * 16 + q];
        }
#pragma unroll
        for (int u = 0; u < W; ++u) en[u] = (pp + W + u < p1) ? rv[pp + W + u] : -1;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            ps.x += (w[u] & 8u) ? v[u].x : 0.0f;
            ps.y += (w[u] & 4u) ? v[u].y : 0.0f;
            ps.z += (w[u] & 2u) ? v[u].z : 0.0f;
            ps.w += (w[u] & 1u) ? v[u].w : 0.0f;
            e[u] = en[u];
        }
    }
    int nx = 0, ny = 0, nz = 0, nw = 0;
#pragma unroll
    for (int k = 0; k < DRP_K; ++k) {
        const unsigned nib = (wk[k] >> sh) & 0xfu;
        nx += (nib >> 3) & 1; ny += (nib >> 2) & 1; nz += (nib >> 1) & 1; nw += nib & 1;
    }
    pr = make_float4(plane_sum(nx, gi.x), plane_sum(ny, gi.y), plane_sum(nz, gi.z), plane_sum(nw, gi.w));
}
#define KMB_COOP_SLOTS 2
#define KMB_COOP_LDS (KMB_FUSED_LDS + (size_t)KMB_COOP_SLOTS * 2 * 32 * KMB_ROWS_LD * sizeof(float))
// COOP (with DUMP, a handful of tiles per workgroup): the workgroup's tiles go round by round, KMB_COOP_SLOTS at a time -- all
// eight waves gather a tile's edge terms into LDS (coop_edge_terms), then one wave per tile runs the matrix chain on them
template <bool DUMP, bool COOP>
__global__ void __launch_bounds__(64 * KMB_FUSED_WAVES)
kmb_step_bwd(const float* __restrict__ mw, const float* __restrict__ mb,
             const float* __restrict__ eff_hist /* [4][B*N,64] */, const unsigned* __restrict__ mask_hist /* [3][B*N*10][2] */,
             const uint8_t* __restrict__ nbr_cnt, const int* __restrict__ rev_off, const int* __restrict__ rev,
             const float* __restrict__ g_out, size_t g_stride, const float* __restrict__ s_delta,
             const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
             int N, int B, int spw, float* g_eff, float* g_cnode,
             float* g_agg_hist /* [3][B*N,64] */, float* __restrict__ g_sdelta,
             KmbDump dump, int parts, unsigned* bar /* [groups], zero */, unsigned* bar_err) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w0 = lds;                 // predictor layer 0, forward pack
    float* w0t = w0 + 4096;          // transposed
    float* wagg = w0t + 4096;        // W_agg^T
    float* wr = wagg + 4096;         // W_r^T
    float* ws = wr + 4096;           // W_s^T
    float* wpet = ws + 4096;         // W_pe^T
    float* w2t = wpet + 4096;        // particle encoder layer 2, transposed
    float* w1 = w2t + 4096;          // particle encoder layer 0, forward pack (K = 8)
    float* rows_pr = w1 + 512;       // b_pr0 [64], w_pr1 [3][64]
    float* rows_pe = rows_pr + 256;  // encoder layer-0 columns 0..2
    int* ctr = reinterpret_cast<int*>(rows_pe + 192);
    float* et = rows_pe + 192 + 4;   // COOP: [slot][receiver | sender][32][KMB_ROWS_LD]
    lds_fill(w0, mw + M_PR0, 4096);
    lds_fill(w0t, mb + MB_PR0, 4096);
    lds_fill(wagg, mb + MB_AGG, 3 * 4096);          // MB_AGG, MB_RPR, MB_RPS are consecutive
    lds_fill(wpet, mb + MB_PPE, 2 * 4096);          // MB_PPE, MB_PE2
    lds_fill(w1, mw + M_PE0, 512);
    lds_fill(rows_pr, mw + R_PR0_B, 256);
    lds_fill(rows_pe, mb + RB_PE0, 192);
    if (threadIdx.x == 0) *ctr = KMB_FUSED_WAVES;
#ifdef ROLLOUT_STAMPS
    const bool roll_on = DUMP && threadIdx.x == 0;             // tools/train_stamps.py: wave 0 of every workgroup
    unsigned long long roll_t = roll_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
    __syncthreads();
    BWD_STAMP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int group = (int)blockIdx.x / parts, part = (int)blockIdx.x - group * parts;
    const int b0 = group * spw, nbw = min(spw, B - b0);
    const int wg_rows = (nbw > 0 ? nbw : 0) * N;
    const int wg_tiles = (wg_rows + 31) >> 5;
    const float inv_N = 1.0f / (float)N;
    const size_t bn64 = (size_t)B * N * 64;
    const size_t bnk2 = (size_t)B * N * DRP_K * 2;
    // one workgroup per group: tiles on demand (a counter in LDS); several: dealt statically, workgroup-cyclic first (a
    // handful of tiles: one per CU)
    const int first_tile = parts > 1 ? part + wave * parts : wave;
    auto next_tile = [&](int li) {
        if (parts > 1) return li + parts * KMB_FUSED_WAVES;
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(q);
    };
    unsigned arrivals = 0;
    auto phase_end = [&]() {
        if (parts > 1) {
            __threadfence();                               // this wave's rows are in memory, device-wide
            __syncthreads();
            BWD_STAMP(5);
            arrivals += (unsigned)parts;
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(bar + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(bar + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrivals) {
                    __builtin_amdgcn_s_sleep(4);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { *bar_err = 1u; break; }      // 2 s at 100 MHz
                }
            }
            BWD_STAMP(6);
            __syncthreads();
            __threadfence();                               // nothing of the other workgroups' rows from this CU's L1
            return;
        }
        __syncthreads();                                   // this phase's rows of the workgroup's samples are written
        if (threadIdx.x == 0) *ctr = KMB_FUSED_WAVES;
        __syncthreads();
    };
    // ---- phase P: predictor backward (kmb_predict) + update of the last propagation step
    {
        const float* eff3 = eff_hist + 3 * bn64;
        float* g_agg2 = g_agg_hist + 2 * bn64;
        for (int li = first_tile; li < wg_tiles; li = next_tile(li)) {
            const bool live = (li * 32 + j) < wg_rows;
            const int r = min(li * 32 + j, wg_rows - 1);
            int m, i;
            divmod_small(r, N, inv_N, m, i);
            const int b = b0 + m;
            const size_t row = (size_t)b0 * N + r;
            Frag x, hh, gh, ge;
            frag_from_row(eff3 + row * 64, h, x);
            frag_from_row(rows_pr, h, hh);
            mfma_layer64<false>(reinterpret_cast<const float4*>(w0), x, hh, lane);
            const float* go = g_out + (size_t)b * g_stride + (size_t)i * 3;
            const float g0 = go[0], g1 = go[1], g2 = go[2];
            {
                Frag wx, wy, wz;
                frag_from_row(rows_pr + 64, h, wx);
                frag_from_row(rows_pr + 128, h, wy);
                frag_from_row(rows_pr + 192, h, wz);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float g = wx.v[ob][q] * g0 + wy.v[ob][q] * g1 + wz.v[ob][q] * g2;
                        gh.v[ob][q] = hh.v[ob][q] > 0.0f ? g : 0.0f;
                    }
            }
            frag_zero(ge);
            mfma_layer64<false>(reinterpret_cast<const float4*>(w0t), gh, ge, lane);
            // update of step 2: g_z = g_eff . [eff3 > 0]; g_cnode = g_z; g_agg[2] = W_agg^T g_z
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                ge.v[0][q] = x.v[0][q] > 0.0f ? ge.v[0][q] : 0.0f;
                ge.v[1][q] = x.v[1][q] > 0.0f ? ge.v[1][q] : 0.0f;
            }
            if (live) {
                frag_to_row((DUMP ? dump.ge[0] : g_eff) + row * 64, h, ge);
                frag_to_row(g_cnode + row * 64, h, ge);
                if (DUMP) {
                    frag_to_row(dump.gh + row * 64, h, gh);
                    frag_relu(hh);
                    frag_to_row(dump.hact + row * 64, h, hh);
                }
            }
            Frag ga;
            frag_zero(ga);
            mfma_layer64<false>(reinterpret_cast<const float4*>(wagg), ge, ga, lane);
            if (live) frag_to_row(g_agg2 + row * 64, h, ga);
        }
    }
    // ---- phases p = 2, 1, 0
#pragma unroll 1
    for (int p = DRP_PSTEP - 1; p >= 0; --p) {
        BWD_STAMP(p == DRP_PSTEP - 1 ? 1 : 4);
        phase_end();
        BWD_STAMP(2);
        const float* g_agg_p = g_agg_hist + (size_t)p * bn64;
        const unsigned* mask_p = mask_hist + (size_t)p * bnk2;
        const float* ge_rd = DUMP ? dump.ge[DRP_PSTEP - 1 - p] : g_eff;
        float* ge_wr = DUMP ? dump.ge[p > 0 ? DRP_PSTEP - p : 0] : g_eff;
        for (int k0 = 0, li = first_tile;; k0 += KMB_COOP_SLOTS) {
            if (COOP) {
                // a round: the edge terms of up to KMB_COOP_SLOTS of the workgroup's tiles (part, part + parts, ...) into LDS
                if (part + k0 * parts >= wg_tiles) break;
                if (k0 > 0) __syncthreads();                          // the last round's chains have read their slots
#pragma unroll 1
                for (int sl = 0; sl < KMB_COOP_SLOTS; ++sl) {
                    const int lt = part + (k0 + sl) * parts;
                    if (lt >= wg_tiles) break;
                    const int rl = (int)threadIdx.x >> 4, q = (int)threadIdx.x & 15;
                    const bool live_g = (lt * 32 + rl) < wg_rows;
                    const int rr = min(lt * 32 + rl, wg_rows - 1);
                    int m, i;
                    divmod_small(rr, N, inv_N, m, i);
                    const int b = b0 + m;
                    const size_t srow0 = (size_t)b * N;
                    const int* ro = rev_off + (size_t)b * (N + 1);
                    float4 pr4, ps4;
                    coop_edge_terms(g_agg_p, mask_p + srow0 * DRP_K * 2, rev + srow0 * DRP_K, ro[i], ro[i + 1], i, nbr_cnt[srow0 + i],
                                    srow0, q, pr4, ps4);
                    float* e_r = et + (size_t)(sl * 2) * 32 * KMB_ROWS_LD + rl * KMB_ROWS_LD + q * 4;
                    *reinterpret_cast<float4*>(e_r) = pr4;
                    *reinterpret_cast<float4*>(e_r + 32 * KMB_ROWS_LD) = ps4;
                    if (DUMP && live_g) {
                        float* gp = dump.gp[p] + (srow0 + i) * 128 + q * 4;
                        *reinterpret_cast<float4*>(gp) = pr4;
                        *reinterpret_cast<float4*>(gp + 64) = ps4;
                    }
                }
                __syncthreads();
                BWD_STAMP(3);
                li = part + (k0 + wave) * parts;
                if (wave >= KMB_COOP_SLOTS || li >= wg_tiles) continue;
            } else {
                if (k0 > 0) li = next_tile(li);
                if (li >= wg_tiles) break;
            }
            const bool live = (li * 32 + j) < wg_rows;
            const int r = min(li * 32 + j, wg_rows - 1);
            int m, i;
            divmod_small(r, N, inv_N, m, i);
            const int b = b0 + m;
            const size_t row = (size_t)b0 * N + r;
            Frag pr, ps;
            if (COOP) {
                gagg_lds_read(et + (size_t)(wave * 2) * 32 * KMB_ROWS_LD, j, h, pr);
                gagg_lds_read(et + (size_t)(wave * 2 + 1) * 32 * KMB_ROWS_LD, j, h, ps);
            } else {
                const size_t srow0 = (size_t)b * N;                       // first row of the lane's sample
                // receiver term: g_agg[p][row] under the masks of the row's own slots, one addition per slot
                Frag gi;
                frag_from_row(g_agg_p + row * 64, h, gi);
                frag_zero(pr);
                frag_zero(ps);
                const int cnt = nbr_cnt[row];
                const int* ro = rev_off + (size_t)b * (N + 1);
                const int p0 = ro[i], p1 = ro[i + 1];
                const int* rv = rev + srow0 * DRP_K;
                const unsigned* mk = mask_p + srow0 * DRP_K * 2;          // the sample's slots, word h of a slot at [slot * 2 + h]
                int cmax = cnt, lmax = p1 - p0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    cmax = max(cmax, __shfl_xor(cmax, o, 64));
                    lmax = max(lmax, __shfl_xor(lmax, o, 64));
                }
                // the sender term's first entries are requested before the receiver term runs
                const int nrev = p1 - p0;
                int e0 = (0 < nrev) ? rv[p0] : 0;
                int e1 = (1 < nrev) ? rv[p0 + 1] : 0;
                {
                    // the row's ten mask words are independent loads: all in flight at once (a loop over k waited for each
                    // in turn -- ten L2 round trips per tile and phase)
                    unsigned wk[DRP_K];
#pragma unroll
                    for (int k = 0; k < DRP_K; ++k) wk[k] = (k < cnt) ? mk[((size_t)i * DRP_K + k) * 2 + h] : 0u;
                    receiver_term(pr, gi, wk);
                }
                // sender term over the reversed list (ascending receiver, then slot), software-pipelined two deep: the
                // mask word and the g_agg row of entry q0 + 1 and the index of entry q0 + 3 are requested before entry q0's
                // row is added (the additions keep the list's order)
                unsigned w_cur = (0 < nrev) ? mk[(size_t)e0 * 2 + h] : 0u;
                Frag v_cur;
                {
                    int er, ek;
                    divmod_small(e0, DRP_K, 0.1f, er, ek);
                    frag_from_row(g_agg_p + (srow0 + ((0 < nrev) ? er : i)) * 64, h, v_cur);
                }
                int e2 = (2 < nrev) ? rv[p0 + 2] : 0;
                for (int q0 = 0; q0 < lmax; ++q0) {
                    const bool on_next = q0 + 1 < nrev;
                    const unsigned w_nxt = on_next ? mk[(size_t)e1 * 2 + h] : 0u;
                    Frag v_nxt;
                    {
                        int er, ek;
                        divmod_small(e1, DRP_K, 0.1f, er, ek);
                        frag_from_row(g_agg_p + (srow0 + (on_next ? er : i)) * 64, h, v_nxt);
                    }
                    const int e3 = (q0 + 3 < nrev) ? rv[p0 + q0 + 3] : 0;
                    frag_add_masked(ps, v_cur, w_cur);
                    w_cur = w_nxt;
                    v_cur = v_nxt;
                    e1 = e2;
                    e2 = e3;
                }
                frag_settle(ps);
                if (DUMP && live) {
                    frag_to_row(dump.gp[p] + row * 128, h, pr);
                    frag_to_row(dump.gp[p] + row * 128 + 64, h, ps);
                }
            }
            // projection backward: g_eff += W_r^T (receiver term) + W_s^T (sender term)
            Frag ge;
            frag_from_row(ge_rd + row * 64, h, ge);
            mfma_layer64<false>(reinterpret_cast<const float4*>(wr), pr, ge, lane);
            mfma_layer64<false>(reinterpret_cast<const float4*>(ws), ps, ge, lane);
            if (p > 0) {
                // update of step p - 1: g_z = g_eff . [eff_p > 0]; g_eff <- g_z; g_cnode += g_z; g_agg[p-1] = W_agg^T g_z
                Frag en, ga;
                frag_from_row(eff_hist + (size_t)p * bn64 + row * 64, h, en);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    ge.v[0][q] = en.v[0][q] > 0.0f ? ge.v[0][q] : 0.0f;
                    ge.v[1][q] = en.v[1][q] > 0.0f ? ge.v[1][q] : 0.0f;
                }
                if (live) frag_to_row(ge_wr + row * 64, h, ge);
                frag_from_row(g_cnode + row * 64, h, en);
#pragma unroll
                for (int q = 0; q < 16; ++q) { en.v[0][q] += ge.v[0][q]; en.v[1][q] += ge.v[1][q]; }
                if (live) frag_to_row(g_cnode + row * 64, h, en);
                frag_zero(ga);
                mfma_layer64<false>(reinterpret_cast<const float4*>(wagg), ge, ga, lane);
                if (live) frag_to_row(g_agg_hist + (size_t)(p - 1) * bn64 + row * 64, h, ga);
            } else {
                // particle encoder backward (kmb_node_encode): g_pe = g_eff0 + W_pe^T g_cnode, through
                // relu(W2 relu(W1 x + b1) + b2) to the three impulse inputs
                int q_, bm;
                divmod_small(b, attr_mod, 1.0f / (float)attr_mod, q_, bm);
                const float d = dens[bm] / DRP_DENS_SCALE;
                const float* sd = s_delta + row * 3;
                const float at = attr[(size_t)bm * N + i];
                float x[4];
                if (h == 0) { x[0] = sd[0]; x[1] = sd[2]; x[2] = d; x[3] = 0.0f; }
                else { x[0] = sd[1]; x[1] = at; x[2] = 1.0f; x[3] = 0.0f; }
                Frag h1, g, gh;
                frag_zero(h1);
                mfma_layer8(reinterpret_cast<const float4*>(w1), x, h1, lane);
                frag_from_row(g_cnode + row * 64, h, g);
                mfma_layer64<false>(reinterpret_cast<const float4*>(wpet), g, ge, lane);
                frag_from_row(eff_hist + row * 64, h, g);                 // pe = effect after the encoder
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    ge.v[0][q] = g.v[0][q] > 0.0f ? ge.v[0][q] : 0.0f;
                    ge.v[1][q] = g.v[1][q] > 0.0f ? ge.v[1][q] : 0.0f;
                }
                frag_zero(gh);
                mfma_layer64<false>(reinterpret_cast<const float4*>(w2t), ge, gh, lane);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    gh.v[0][q] = h1.v[0][q] > 0.0f ? gh.v[0][q] : 0.0f;
                    gh.v[1][q] = h1.v[1][q] > 0.0f ? gh.v[1][q] : 0.0f;
                }
                float out[3];
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    Frag w;
                    frag_from_row(rows_pe + 64 * o, h, w);
                    float acc = 0.0f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc = fmaf(gh.v[0][q], w.v[0][q], acc);
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc = fmaf(gh.v[1][q], w.v[1][q], acc);
                    out[o] = acc + __shfl_xor(acc, 32, 64);
                }
                if (live && h == 0) {
                    g_sdelta[row * 3 + 0] = out[0];
                    g_sdelta[row * 3 + 1] = out[1];
                    g_sdelta[row * 3 + 2] = out[2];
                }
                if (DUMP && live) {
                    frag_to_row(dump.gpe + row * 64, h, ge);
                    frag_to_row(dump.gh1 + row * 64, h, gh);
                    frag_relu(h1);
                    frag_to_row(dump.a1n + row * 64, h, h1);
                    if (h == 0) {
                        float* dx = dump.xn + row * 8;
                        dx[0] = sd[0]; dx[1] = sd[1]; dx[2] = sd[2]; dx[3] = at; dx[4] = d; dx[5] = 0.0f; dx[6] = 0.0f; dx[7] = 0.0f;
                    }
                }
            }
        }
    }
#ifdef ROLLOUT_STAMPS
    BWD_STAMP(4);
    if (roll_on) atomicAdd(&g_bwd_stamps[15], 1ull);
#endif
}


// ---- the same pass with a wave keeping its 32 rows from the first phase to the last (piles of up to 256 particles) ------
// kmb_step_bwd above is latency: per tile and phase it re-reads g_eff, g_cnode and its own g_agg rows from memory, gathers
// the senders' g_agg rows from L2 (or behind it) and writes all of them back -- 0.9 GB per launch at the GD planner's
// demo shape, 46 % of the wave cycles waiting, 436 us.  Here a workgroup takes GROUPS of whole samples with at most 256
// rows together -- one tile of 32 rows per wave, eight waves -- and
//   * g_eff, g_cnode and the tile's own g_agg rows never leave the wave's registers between the phases;
//   * the group's g_agg rows of the current propagation step sit in LDS (68 KB: rows padded to 17 float4, so that a gather
//     of arbitrary rows spreads over the banks): the reversed-list gather is ds_read_b128, its mask words are requested
//     four entries ahead;
//   * g_agg goes to memory only when a later stage wants it (kmb_edge_encode of a horizon > 1; `g_agg_hist` may be null).
// LDS: W_agg^T, W_r^T, W_s^T resident; one 32-KB region holds the predictor's two matrices in phase P and the particle
// encoder's two (transposed) from then on, refilled per group.  Same sums in the same order as kmb_step_bwd: same bits.
#define KMB_ROWS_MAX 256
#define KMB_ROWS_LDS ((size_t)(3 * 1536 * 4 + KMB_ROWS_MAX * KMB_ROWS_LD + 512 + 256 + 192) * sizeof(float))
// one 64 x 64 layer on the six-product bf16 split: acc += W in, W packed as pack_split6 / kt_repack_split6_bwd
__device__ __forceinline__ void rows_layer(const bf16x8* __restrict__ wp, const Frag& in, Frag& acc, int lane) {
    FragB6 b;
    split_frag6(in, b);
    mfma_layer64_split6(wp, b, acc, lane);
}
__global__ void __launch_bounds__(64 * KMB_FUSED_WAVES)
kmb_rows_bwd(const float* __restrict__ mw, const float* __restrict__ mb, const uint16_t* __restrict__ sw6 /* pack_split6 */,
             const uint16_t* __restrict__ sb6 /* kt_repack_split6_bwd */,
             const float* __restrict__ eff_hist /* [4][B*N,64] */, const unsigned* __restrict__ mask_hist /* [3][B*N*10][2] */,
             const uint8_t* __restrict__ nbr_cnt, const int* __restrict__ rev_off, const int* __restrict__ rev,
             const float* __restrict__ g_out, size_t g_stride, const float* __restrict__ s_delta,
             const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
             int N, int B, int gps /* samples per group: gps * N <= 256 */,
             float* __restrict__ g_agg_hist /* [3][B*N,64], or null */, float* __restrict__ g_sdelta) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wres = lds;                               // W_agg^T, W_r^T, W_s^T in the split: 3 x 1536 bf16x8
    float* gl = wres + 3 * 1536 * 4;                 // the group's g_agg rows of the current propagation step; in phase P and in
                                                     // the encoder's part of phase 0 (no gathers) two more matrices instead
    float* w1 = gl + KMB_ROWS_MAX * KMB_ROWS_LD;     // particle encoder layer 0, forward pack (K = 8, fp32)
    float* rows_pr = w1 + 512;                       // b_pr0 [64], w_pr1 [3][64]
    float* rows_pe = rows_pr + 256;                  // encoder layer-0 columns 0..2
    const bf16x8* wagg = reinterpret_cast<const bf16x8*>(wres);
    const bf16x8* wr = wagg + 1536;
    const bf16x8* ws = wr + 1536;
    const bf16x8* wx0 = reinterpret_cast<const bf16x8*>(gl);
    const bf16x8* wx1 = wx0 + 1536;
    lds_fill(wres, reinterpret_cast<const float*>(sb6) + (size_t)SB6_AGG * 4, 3 * 1536 * 4);      // SB6_AGG, _RPR, _RPS consecutive
    lds_fill(w1, mw + M_PE0, 512);
    lds_fill(rows_pr, mw + R_PR0_B, 256);
    lds_fill(rows_pe, mb + RB_PE0, 192);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float inv_N = 1.0f / (float)N;
    const size_t bn64 = (size_t)B * N * 64;
    const size_t bnk2 = (size_t)B * N * DRP_K * 2;
    const int n_groups = (B + gps - 1) / gps;
#ifdef ROLLOUT_STAMPS
    const bool roll_on = threadIdx.x == 0 && (blockIdx.x & 31) == 0;
    unsigned long long roll_t = roll_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
#pragma unroll 1
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int b0 = grp * gps, nbw = min(gps, B - b0);
        const int grp_rows = nbw * N;
        __syncthreads();                                   // the group before is through with gl (the encoder's matrices)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));                      // the fills' per-thread addresses are recomputed per group, not kept (spilled) across it
        lds_fill(gl, reinterpret_cast<const float*>(sw6) + (size_t)S6_PR0 * 4, 1536 * 4, tid);                 // predictor layer 0, forward
        lds_fill(gl + 1536 * 4, reinterpret_cast<const float*>(sb6) + (size_t)SB6_PR0 * 4, 1536 * 4, tid);    // ... transposed
        __syncthreads();
        BWD_STAMP(0);
        const bool active = wave * 32 < grp_rows;          // wave-uniform
        const bool live = (wave * 32 + j) < grp_rows;
        const int r = min(wave * 32 + j, grp_rows - 1);
        int m, i;
        divmod_small(r, N, inv_N, m, i);
        const int b = b0 + m;
        const size_t row = (size_t)b0 * N + r;
        const size_t srow0 = (size_t)b * N;                // first row of the lane's sample
        const int lrow0 = m * N;                           // ... within the group
        const int cnt = live ? (int)nbr_cnt[row] : 0;
        const int* ro = rev_off + (size_t)b * (N + 1);
        const int p0 = ro[i];
        const int nrev = live ? ro[i + 1] - p0 : 0;
        const int* rv = rev + srow0 * DRP_K + p0;
        int lmax = nrev;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o, 64));
        Frag ge, gc, ga;
        // ---- phase P: predictor backward + update of the last propagation step
        if (active) {
            Frag x, hh, gh;
            frag_from_row(eff_hist + 3 * bn64 + row * 64, h, x);
            frag_from_row(rows_pr, h, hh);
            rows_layer(wx0, x, hh, lane);
            const float* go = g_out + (size_t)b * g_stride + (size_t)i * 3;
            const float g0 = go[0], g1 = go[1], g2 = go[2];
            {
                Frag wx, wy, wz;
                frag_from_row(rows_pr + 64, h, wx);
                frag_from_row(rows_pr + 128, h, wy);
                frag_from_row(rows_pr + 192, h, wz);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float g = wx.v[ob][q] * g0 + wy.v[ob][q] * g1 + wz.v[ob][q] * g2;
                        gh.v[ob][q] = hh.v[ob][q] > 0.0f ? g : 0.0f;
                    }
            }
            frag_zero(ge);
            rows_layer(wx1, gh, ge, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                ge.v[0][q] = x.v[0][q] > 0.0f ? ge.v[0][q] : 0.0f;
                ge.v[1][q] = x.v[1][q] > 0.0f ? ge.v[1][q] : 0.0f;
            }
            gc = ge;
            frag_zero(ga);
            rows_layer(wagg, ge, ga, lane);
            if (live && g_agg_hist != nullptr) frag_to_row(g_agg_hist + 2 * bn64 + row * 64, h, ga);
        }
        BWD_STAMP(1);
        __syncthreads();                                   // nobody reads the predictor's matrices any more: gl becomes rows
        if (active && live) gagg_lds_write(gl, r, h, ga);
        __syncthreads();
        BWD_STAMP(2);
        // ---- phases p = 2, 1, 0
        unsigned en_pos = 0u;                              // [eff_hist[p] > 0] of the lane's row, of the phase that ran last
#pragma unroll 1
        for (int p = DRP_PSTEP - 1; p >= 0; --p) {
            if (active) {
                const unsigned* mk = mask_hist + (size_t)p * bnk2 + srow0 * DRP_K * 2;      // the sample's slots of this step
                // the row of the effects this phase's ReLU mask comes from (eff_hist[p]: after step p - 1, or after the encoder)
                Frag en;
                frag_from_row(eff_hist + (size_t)p * bn64 + row * 64, h, en);
                // sender term's first entries and mask words
                int e0 = (0 < nrev) ? rv[0] : 0, e1 = (1 < nrev) ? rv[1] : 0, e2 = (2 < nrev) ? rv[2] : 0, e3 = (3 < nrev) ? rv[3] : 0;
                int e4 = (4 < nrev) ? rv[4] : 0, e5 = (5 < nrev) ? rv[5] : 0;
                unsigned w0 = (0 < nrev) ? mk[(size_t)e0 * 2 + h] : 0u, w1_ = (1 < nrev) ? mk[(size_t)e1 * 2 + h] : 0u;
                unsigned w2 = (2 < nrev) ? mk[(size_t)e2 * 2 + h] : 0u, w3 = (3 < nrev) ? mk[(size_t)e3 * 2 + h] : 0u;
                // receiver term: the tile's own g_agg rows (registers) under the masks of the row's own slots
                Frag pr, ps;
                frag_zero(ps);
                {
                    unsigned wk[DRP_K];
#pragma unroll
                    for (int k = 0; k < DRP_K; ++k) wk[k] = (k < cnt) ? mk[((size_t)i * DRP_K + k) * 2 + h] : 0u;
                    receiver_term(pr, ga, wk);
                }
                rows_layer(wr, pr, ge, lane);
                BWD_STAMP(3);
                en_pos = frag_positive_bits_any(en);      // all this phase wants of that row (requested a projection ago)
                // sender term over the reversed list (ascending receiver, then slot): rows from LDS one entry ahead, in two
                // buffers that change roles (no copies), mask words four entries ahead, list entries six
                Frag va, vb;
                frag_zero(va);
                frag_zero(vb);
                {
                    int er, ek;
                    divmod_small(e0, DRP_K, 0.1f, er, ek);
                    if (0 < nrev) gagg_lds_read(gl, lrow0 + er, h, va);
                }
#define KMB_ROWS_ENTRY(cur_, nxt_, q_) { \
                    { \
                        int er, ek; \
                        divmod_small(e1, DRP_K, 0.1f, er, ek); \
                        if ((q_) + 1 < nrev) gagg_lds_read(gl, lrow0 + er, h, nxt_); \
                    } \
                    const unsigned w4 = ((q_) + 4 < nrev) ? mk[(size_t)e4 * 2 + h] : 0u; \
                    const int e6 = ((q_) + 6 < nrev) ? rv[(q_) + 6] : 0; \
                    frag_add_masked(ps, cur_, w0); \
                    w0 = w1_; w1_ = w2; w2 = w3; w3 = w4; \
                    e1 = e2; e2 = e3; e3 = e4; e4 = e5; e5 = e6; }
                for (int q0 = 0; q0 < lmax; q0 += 2) {
                    KMB_ROWS_ENTRY(va, vb, q0)
                    KMB_ROWS_ENTRY(vb, va, q0 + 1)        /* past the longest list: no loads, mask words 0 */
                }
#undef KMB_ROWS_ENTRY
                frag_settle(ps);
                BWD_STAMP(4);
                rows_layer(ws, ps, ge, lane);
                if (p > 0) {
                    // update of step p - 1: g_z = g_eff . [eff_p > 0]; g_cnode += g_z; g_agg[p-1] = W_agg^T g_z
                    frag_keep_bits(ge, en_pos);
#pragma unroll
                    for (int q = 0; q < 16; ++q) { gc.v[0][q] += ge.v[0][q]; gc.v[1][q] += ge.v[1][q]; }
                    frag_zero(ga);
                    rows_layer(wagg, ge, ga, lane);
                    if (live && g_agg_hist != nullptr) frag_to_row(g_agg_hist + (size_t)(p - 1) * bn64 + row * 64, h, ga);
                }
            }
            BWD_STAMP(5);
            __syncthreads();                               // every gather of this step's rows is done
            if (p > 0) {
                if (active && live) gagg_lds_write(gl, r, h, ga);
            } else {
                asm volatile("" : "+v"(tid));
                lds_fill(gl, reinterpret_cast<const float*>(sb6) + (size_t)SB6_PPE * 4, 2 * 1536 * 4, tid);   // SB6_PPE, SB6_PE2 consecutive
            }
            __syncthreads();
            BWD_STAMP(6);
        }
        if (active) {
            // particle encoder backward: g_pe = g_eff0 + W_pe^T g_cnode, through relu(W2 relu(W1 x + b1) + b2) to the impulse
            int q_, bm;
            divmod_small(b, attr_mod, 1.0f / (float)attr_mod, q_, bm);
            const float d = dens[bm] / DRP_DENS_SCALE;
            const float* sd = s_delta + row * 3;
            const float at = attr[(size_t)bm * N + i];
            float x[4];
            if (h == 0) { x[0] = sd[0]; x[1] = sd[2]; x[2] = d; x[3] = 0.0f; }
            else { x[0] = sd[1]; x[1] = at; x[2] = 1.0f; x[3] = 0.0f; }
            Frag h1, gh;
            frag_zero(h1);
            mfma_layer8(reinterpret_cast<const float4*>(w1), x, h1, lane);
            rows_layer(wx0, gc, ge, lane);
            frag_keep_bits(ge, en_pos);                    // phase 0 read eff_hist[0]: the effects after the encoder
            frag_zero(gh);
            rows_layer(wx1, ge, gh, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                gh.v[0][q] = h1.v[0][q] > 0.0f ? gh.v[0][q] : 0.0f;
                gh.v[1][q] = h1.v[1][q] > 0.0f ? gh.v[1][q] : 0.0f;
            }
            float out[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                Frag w;
                frag_from_row(rows_pe + 64 * o, h, w);
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = fmaf(gh.v[0][q], w.v[0][q], acc);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = fmaf(gh.v[1][q], w.v[1][q], acc);
                out[o] = acc + __shfl_xor(acc, 32, 64);
            }
            if (live && h == 0) {
                g_sdelta[row * 3 + 0] = out[0];
                g_sdelta[row * 3 + 1] = out[1];
                g_sdelta[row * 3 + 2] = out[2];
            }
        }
        BWD_STAMP(7);
#ifdef ROLLOUT_STAMPS
        if (roll_on) atomicAdd(&g_bwd_stamps[15], 1ull);
#endif
    }
}

// ---- relation encoder backward on the matrix cores (horizons > 1 of the GD planner, training) -------------------
// What kb_edge_encode (k_backward.h) computes per edge slot, as the forward chain of km_edge_encode run both ways on
// tiles of 32 slots: the three Linear+ReLU layers forward (fp32 MFMA; only the SIGN of every pre-activation is kept:
// 96 bits per lane instead of three fragments), the gradient at c_edge rebuilt from the three propagation steps' ReLU
// masks and the receiver's g_agg rows (the mask words are in fragment order: word = half-wave, bit 31 - register),
// then W_e^T, layer 4^T, layer 2^T under the kept signs, and the dot with the three position columns of layer 0.
// The VALU kernel gave a wave one receiver and kept lane = feature: 5 x 16 KB of weights into LDS per FOUR receivers,
// 106 us per launch at the reference's training batch (4 x 300 particles), the largest single item of an iteration.
// Every sum has a fixed order (the MFMA's own k order): bit-reproducible.  Differences to kb_edge_encode: the
// receiver's own position gradient (the sum over its slots) is left to kb_gather_pos (add_recv), which reads
// gpos_edge anyway; the result agrees with the VALU kernel to fp32 rounding (another summation order), which the
// tests against the reference's autograd cover (tests/test_gpu_gd.py horizon 2, tests/test_gpu_train.py).
#define KMB_EDGE_ENCODE_LDS ((size_t)(512 + 5 * 4096 + 128 + 192) * sizeof(float))
__global__ void __launch_bounds__(64 * MFMA_WAVES)
kmb_edge_encode(const float* __restrict__ mw, const float* __restrict__ mb, const float* __restrict__ s_cur, int s_mod,
                size_t s_stride, const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
                const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt,
                const float* __restrict__ g_agg_hist /* [3][B*N,64] */, const unsigned* __restrict__ mask_hist /* [3][B*N*10][2] */,
                size_t bn, int N, int B, float* __restrict__ gpos_edge /* nullable: [B,N,10,4] */, KbEdgeDump dump) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w1 = lds;                 // forward: layer 0 (K = 8, bias column), layers 2 and 4
    float* w2 = w1 + 512;
    float* w4 = w2 + 4096;
    float* wet = w4 + 4096;          // backward: W_e^T, layer 4^T, layer 2^T
    float* w4t = wet + 4096;
    float* w2t = w4t + 4096;
    float* rows = w2t + 4096;        // b2, b4
    float* wxyz = rows + 128;        // layer 0, position columns
    lds_fill(w1, mw + M_RE0, 512);
    lds_fill(w2, mw + M_RE2, 2 * 4096);              // M_RE2, M_RE4 consecutive
    lds_fill(wet, mb + MB_RPE, 3 * 4096);            // MB_RPE, MB_RE4, MB_RE2 consecutive
    lds_fill(rows, mw + R_RE2_B, 128);               // R_RE2_B, R_RE4_B consecutive
    lds_fill(wxyz, mb + RB_RE0, 192);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int nslots = N * DRP_K;
    const int tps = (nslots + 31) >> 5;
    const long ntiles = (long)B * tps;
    const bool dumping = dump.re != nullptr;
    // tiles are dealt block-cyclically FIRST (tile = block + grid * (wave + 8 round)): a handful of tiles (a training
    // batch) spreads one wave per CU instead of eight waves on a few CUs
    for (long gt = (long)blockIdx.x + (long)gridDim.x * wave; gt < ntiles; gt += (long)gridDim.x * MFMA_WAVES) {
        const int b = (int)(gt / tps), t = (int)(gt - (long)b * tps);
        const bool live = (t * 32 + j) < nslots;
        const int slot = min(t * 32 + j, nslots - 1);
        const int i = slot / DRP_K, k = slot - i * DRP_K;
        const size_t nrow = (size_t)b * N + i;
        const size_t erow = nrow * DRP_K + k;
        const int cnt = nbr_cnt[nrow];
        const bool edge = k < cnt;
        const int jn = edge ? (int)nbr_idx[erow] : i;
        const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
        const float* at = attr + (size_t)(b % attr_mod) * N;
        const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
        const float ar = at[i], as = at[jn];
        const float dx = s[i * 3 + 0] - s[jn * 3 + 0], dy = s[i * 3 + 1] - s[jn * 3 + 1], dz = s[i * 3 + 2] - s[jn * 3 + 2];
        float x[4];
        if (h == 0) { x[0] = ar; x[1] = dx; x[2] = dz; x[3] = 1.0f; }
        else { x[0] = as; x[1] = dy; x[2] = d; x[3] = 0.0f; }
        if (dumping && live && h == 0) {
            float4* x0 = reinterpret_cast<float4*>(dump.x0 + erow * 8);
            x0[0] = make_float4(ar, as, dx, dy);
            x0[1] = make_float4(dz, d, 0.0f, 0.0f);
        }
        // ---- forward, keeping the signs
        Frag a, c;
        frag_zero(a);
        mfma_layer8(reinterpret_cast<const float4*>(w1), x, a, lane);
        const unsigned pos1 = frag_positive_bits_any(a);
        frag_relu(a);
        if (dumping && live) frag_to_row(dump.a1 + erow * 64, h, a);
        frag_from_row(rows + 0, h, c);
        mfma_layer64<false>(reinterpret_cast<const float4*>(w2), a, c, lane);
        const unsigned pos2 = frag_positive_bits_any(c);
        frag_relu(c);
        if (dumping && live) frag_to_row(dump.a2 + erow * 64, h, c);
        frag_from_row(rows + 64, h, a);
        mfma_layer64<false>(reinterpret_cast<const float4*>(w4), c, a, lane);
        const unsigned pos3 = frag_positive_bits_any(a);
        if (dumping && live) { frag_relu(a); frag_to_row(dump.re + erow * 64, h, a); }
        // ---- d loss / d c_edge of the slot: the three propagation steps share c_edge (zero for a padded slot)
        Frag g;
        frag_zero(g);
        if (edge) {
#pragma unroll
            for (int p = 0; p < DRP_PSTEP; ++p) {
                const unsigned m = mask_hist[((size_t)p * bn * DRP_K + erow) * 2 + h];
                Frag ga;
                frag_from_row(g_agg_hist + ((size_t)p * bn + nrow) * 64, h, ga);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((m >> (31 - (16 * ob + r))) & 1u) g.v[ob][r] += ga.v[ob][r];
            }
        }
        if (dumping && live) frag_to_row(dump.gce + erow * 64, h, g);
        // ---- backward through W_e and the three layers
        Frag tt;
        frag_zero(tt);
        mfma_layer64<false>(reinterpret_cast<const float4*>(wet), g, tt, lane);
        frag_keep_bits(tt, pos3);
        if (dumping && live) frag_to_row(dump.g3 + erow * 64, h, tt);
        frag_zero(g);
        mfma_layer64<false>(reinterpret_cast<const float4*>(w4t), tt, g, lane);
        frag_keep_bits(g, pos2);
        if (dumping && live) frag_to_row(dump.g2 + erow * 64, h, g);
        frag_zero(tt);
        mfma_layer64<false>(reinterpret_cast<const float4*>(w2t), g, tt, lane);
        frag_keep_bits(tt, pos1);
        if (dumping && live) frag_to_row(dump.g1 + erow * 64, h, tt);
        if (gpos_edge != nullptr) {
            float o3[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                Frag w;
                frag_from_row(wxyz + 64 * q, h, w);
                float pdot = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) pdot = fmaf(tt.v[0][r], w.v[0][r], pdot);
#pragma unroll
                for (int r = 0; r < 16; ++r) pdot = fmaf(tt.v[1][r], w.v[1][r], pdot);
                o3[q] = pdot + __shfl_xor(pdot, 32, 64);
            }
            if (h == 0 && live)
                *reinterpret_cast<float4*>(gpos_edge + erow * 4) = edge ? make_float4(o3[0], o3[1], o3[2], 0.0f)
                                                                       : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
}
