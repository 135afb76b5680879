// Reverse mode of the path for the gradient-descent planner:
//   loss = -sum_b reward_b    planners.py:743   ->   d loss / d pushes
// through config_reward_ptcl (env/flex_rewards.py:189-214), PropModuleDiffDen.forward
// (model/gnn_dyn.py:147-198; the graph masks are constants: autograd gives zero through
// topk / the radius test) and gen_s_delta (planners.py:211-257; the hard mask is constant,
// the soft mask and the projections are differentiated, the direction depends on the push).
//
// Any horizon: the reward is taken at the final step only (planners.py:436-438); going back
// through a step, the gradient w.r.t. its input positions is the residual's share, plus the
// relation encoder's (its inputs are position differences), plus gen_s_delta's dependence
// on the particle position.  fp32 VALU kernels (the node-level stages also exist on the matrix cores,
// k_backward_mfma.h) over what the fused forward pass (km_prop<., TAPE>) left in HBM: the effect after
// the encoder and after every propagation step, and the edges' ReLU masks.  The one scatter of the backward pass
// (gradient of the gathered sender rows) is turned into a gather over reversed neighbour lists
// (kb_reverse_lists, kb_edge_terms, kb_gather_pos): no atomics, reproducible sums.
#pragma once
#include "drp_common.h"
#include "k_reward.h"
#include "k_graph.h"
#include "k_mlp_valu.h"

// acc[r] += sum_k x[r][k] * W[k*ld + col0 + lane]   (W in torch [out][in] layout read as [k][lane]:
// the transposed product g_in = W^T g_out)
template <int IN, int R>
__device__ __forceinline__ void dense_bcast_ld(const float* __restrict__ W, int ld, int col0, const float (&x)[R],
                                               float (&acc)[R], int lane) {
#pragma unroll 8
    for (int k = 0; k < IN; ++k) {
        const float w = W[k * ld + col0 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fmaf(bcast_lane(x[r], k), w, acc[r]);
    }
}

#define KB_R 4

// The row kernels below run one workgroup per (sample, chunk of rows): `chunks` = 1 when the batch
// alone fills the chip (the planner's thousands of samples), more for the trainer's handful.
struct KbRange { int b, lo, hi; };
__device__ __forceinline__ KbRange kb_range(int N, int chunks) {
    KbRange r;
    r.b = blockIdx.x / chunks;
    const int ch = blockIdx.x - r.b * chunks;
    const int len = (((N + chunks - 1) / chunks) + KB_R - 1) & ~(KB_R - 1);
    r.lo = ch * len;
    r.hi = min(N, r.lo + len);
    return r;
}

// ---- reward backward: g_state[b,n,:] = d(-reward_b... ) see below ------------------------------
// loss = -sum_b reward_b, reward_b = -(r1 + r2)/N  =>  d loss / d r1 = d loss / d r2 = 1/N.
//   r1 = sum_n bilinear(G, pix_n)      -> grid_sample's gradient w.r.t. the grid (zero where the
//                                          border clamp is active, as torch does)
//   r2 = sum_m min_n |g_m - pix_n|     -> -(g_m - pix_n*) / dist to the arg-min particle n*
//   pix = (x fx / z + cx, y fy / z + cy)
// one workgroup per state row; gradient accumulated in LDS, then written.  dynamic LDS = KB_REWARD_LDS(N).
// reward_out (nullable): the reward itself too, in k_reward's arithmetic and reduction order (the same bits) --
// the planner's iteration then needs one launch for the reward and its gradient instead of two.
#define KB_REWARD_LDS(N) ((size_t)(2 * (N) + 8) * sizeof(float) + (size_t)(2 * (N)) * sizeof(long long))
__global__ void __launch_bounds__(256)
kb_reward(const float* __restrict__ state, size_t row_stride, int N, const float* __restrict__ G, int Hh, int Ww,
          const float* __restrict__ goal_coor, int M, DrpCam cam, int normalize, float* __restrict__ g_state,
          size_t g_stride, float* __restrict__ reward_out, float* __restrict__ reward_copy = nullptr /* pinned host memory */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* px = lds;
    float* py = lds + N;
    float* red = lds + 2 * N + 4 * N;                 // behind the two long long arrays
    float r1 = 0.0f, r2 = 0.0f;
    // d loss / d pixel, accumulated as 2^-40 fixed point: integer adds commute, so the many goal
    // points whose nearest particle is the same one can add in any order and the sum is the same
    // bits every run (fp32 LDS atomics were the last source of run-to-run differences)
    long long* gx = reinterpret_cast<long long*>(lds + 2 * N);
    long long* gy = gx + N;
    const float FIX = 1099511627776.0f, UNFIX = 1.0f / 1099511627776.0f;
    const float* s = state + (size_t)blockIdx.x * row_stride;
    const float scale = normalize ? 1.0f / (float)N : 1.0f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float x = s[n * 3 + 0], y = s[n * 3 + 1], z = s[n * 3 + 2];
        const float u = x * cam.fx / z + cam.cx, v = y * cam.fy / z + cam.cy;
        px[n] = u;
        py[n] = v;
        // bilinear sample gradient (align_corners = False, padding 'border')
        const float nx = u / (float)Hh * 2.0f - 1.0f, ny = v / (float)Hh * 2.0f - 1.0f;
        float ix = ((nx + 1.0f) * (float)Ww - 1.0f) / 2.0f, iy = ((ny + 1.0f) * (float)Hh - 1.0f) / 2.0f;
        // clip_coordinates_set_grad: gradient multiplier 0 outside [0, size-1]
        float mx = (float)Ww / (float)Hh, my = 1.0f;     // d ix / d u, d iy / d v
        if (ix <= 0.0f) { ix = 0.0f; mx = 0.0f; } else if (ix >= (float)(Ww - 1)) { ix = (float)(Ww - 1); mx = 0.0f; }
        if (iy <= 0.0f) { iy = 0.0f; my = 0.0f; } else if (iy >= (float)(Hh - 1)) { iy = (float)(Hh - 1); my = 0.0f; }
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float tx = ix - x0f, ty = iy - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = x0 + 1, y1 = y0 + 1;
        const float g00 = G[(size_t)y0 * Ww + x0];
        const float g01 = (x1 < Ww) ? G[(size_t)y0 * Ww + x1] : 0.0f;
        const float g10 = (y1 < Hh) ? G[(size_t)y1 * Ww + x0] : 0.0f;
        const float g11 = (x1 < Ww && y1 < Hh) ? G[(size_t)y1 * Ww + x1] : 0.0f;
        const float dgx = (g01 - g00) * (1.0f - ty) + (g11 - g10) * ty;
        const float dgy = (g10 - g00) * (1.0f - tx) + (g11 - g01) * tx;
        gx[n] = __float2ll_rn(scale * dgx * mx * FIX);
        gy[n] = __float2ll_rn(scale * dgy * my * FIX);
        if (reward_out != nullptr) {
            // k_reward's forward value: border clamp of the coordinates and of the upper taps
            const float fx_ = fminf(fmaxf(((nx + 1.0f) * (float)Ww - 1.0f) / 2.0f, 0.0f), (float)(Ww - 1));
            const float fy_ = fminf(fmaxf(((ny + 1.0f) * (float)Hh - 1.0f) / 2.0f, 0.0f), (float)(Hh - 1));
            const float fx0 = floorf(fx_), fy0 = floorf(fy_);
            const float ftx = fx_ - fx0, fty = fy_ - fy0;
            const int ax0 = (int)fx0, ay0 = (int)fy0;
            const int ax1 = min(ax0 + 1, Ww - 1), ay1 = min(ay0 + 1, Hh - 1);
            const float f00 = G[(size_t)ay0 * Ww + ax0], f01 = G[(size_t)ay0 * Ww + ax1];
            const float f10 = G[(size_t)ay1 * Ww + ax0], f11 = G[(size_t)ay1 * Ww + ax1];
            r1 += f00 * ((1.0f - ftx) * (1.0f - fty)) + f01 * (ftx * (1.0f - fty)) +
                  f10 * ((1.0f - ftx) * fty) + f11 * (ftx * fty);
        }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const float qx = goal_coor[m * 2 + 0], qy = goal_coor[m * 2 + 1];
        float best = __builtin_inff();
        int arg = 0;
        for (int n = 0; n < N; ++n) {
            const float dx = qx - px[n], dy = qy - py[n];
            const float d2 = dx * dx + dy * dy;
            if (d2 < best) { best = d2; arg = n; }       // first minimum, as torch.min
        }
        const float dist = drp_sqrt_rn(best);
        r2 += dist;
        // d |q - p| / d p = -(q - p) / dist
        atomicAdd(reinterpret_cast<unsigned long long*>(&gx[arg]), (unsigned long long)__float2ll_rn(-scale * (qx - px[arg]) / dist * FIX));
        atomicAdd(reinterpret_cast<unsigned long long*>(&gy[arg]), (unsigned long long)__float2ll_rn(-scale * (qy - py[arg]) / dist * FIX));
    }
    __syncthreads();
    float* g = g_state + (size_t)blockIdx.x * g_stride;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float x = s[n * 3 + 0], y = s[n * 3 + 1], z = s[n * 3 + 2];
        const float gxn = (float)((double)gx[n] * (double)UNFIX), gyn = (float)((double)gy[n] * (double)UNFIX);
        g[n * 3 + 0] = gxn * cam.fx / z;
        g[n * 3 + 1] = gyn * cam.fy / z;
        g[n * 3 + 2] = -(gxn * x * cam.fx + gyn * y * cam.fy) / (z * z);
    }
    if (reward_out != nullptr) {
        const float t1 = block_sum_256(r1, red);
        const float t2 = block_sum_256(r2, red);
        if (threadIdx.x == 0) {
            float r = t1 + t2;
            if (normalize) r = __fdiv_rn(r, (float)N);
            reward_out[blockIdx.x] = -r;
            if (reward_copy != nullptr) reward_copy[blockIdx.x] = -r;
        }
    }
}

// ---- predictor backward: g_eff = W0^T ((W1^T g_out) . [W0 eff + b0 > 0]) ------------------------
__global__ void __launch_bounds__(256)
kb_predict(const float* __restrict__ vw, const float* __restrict__ wraw, const float* __restrict__ eff,
           const float* __restrict__ g_out, size_t g_stride, int N, float* __restrict__ g_eff,
           float* __restrict__ dump_hact /* nullable [B*N,64]: relu(W0 eff + b0) */,
           float* __restrict__ dump_gh /* nullable [B*N,64]: gradient at the hidden pre-activation */, int chunks) {
    __shared__ float w0t[4096], w0[4096];
    lds_copy(w0t, vw + V_PR0_T, 4096);
    lds_copy(w0, wraw + W_PR0_W, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const KbRange rg = kb_range(N, chunks);
    const int b = rg.b;
    const float b0 = vw[V_PR0_B + lane];
    const float w1x = vw[V_PR1_W + lane], w1y = vw[V_PR1_W + 64 + lane], w1z = vw[V_PR1_W + 128 + lane];
    const float* go = g_out + (size_t)b * g_stride;
    for (int base = rg.lo + wave * KB_R; base < rg.hi; base += nwave * KB_R) {
        float x[KB_R], h[KB_R], gh[KB_R], ge[KB_R];
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            x[r] = eff[((size_t)b * N + min(base + r, N - 1)) * 64 + lane];
            h[r] = b0;
        }
        dense_bcast<64, KB_R>(w0t, x, h, lane);
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            const int i = min(base + r, N - 1);
            const float g = w1x * go[i * 3 + 0] + w1y * go[i * 3 + 1] + w1z * go[i * 3 + 2];
            gh[r] = (h[r] > 0.0f) ? g : 0.0f;
            ge[r] = 0.0f;
        }
        dense_bcast_ld<64, KB_R>(w0, 64, 0, gh, ge, lane);
#pragma unroll
        for (int r = 0; r < KB_R; ++r)
            if (base + r < N) {
                const size_t row = (size_t)b * N + base + r;
                g_eff[row * 64 + lane] = ge[r];
                if (dump_hact != nullptr) { dump_hact[row * 64 + lane] = fmaxf(h[r], 0.0f); dump_gh[row * 64 + lane] = gh[r]; }
            }
    }
}

// ---- node update backward: g_z = g_eff . [eff_next > 0]; g_cnode += g_z; g_agg = W_agg^T g_z;
//      g_eff <- g_z (the residual's share of the gradient w.r.t. the previous effect)
__global__ void __launch_bounds__(256)
kb_update(const float* __restrict__ wraw, const float* __restrict__ eff_next, float* __restrict__ g_eff,
          float* __restrict__ g_cnode, int first, int N, float* __restrict__ g_agg, int chunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const KbRange rg = kb_range(N, chunks);
    const int b = rg.b;
    const float* wpp = wraw + W_PP_W;
    for (int base = rg.lo + wave * KB_R; base < rg.hi; base += nwave * KB_R) {
        float gz[KB_R], ga[KB_R];
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            const size_t row = (size_t)b * N + min(base + r, N - 1);
            const float g = g_eff[row * 64 + lane];
            gz[r] = (eff_next[row * 64 + lane] > 0.0f) ? g : 0.0f;
            ga[r] = 0.0f;
        }
        dense_bcast_ld<64, KB_R>(wpp, 129, 64, gz, ga, lane);
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            if (base + r >= N) continue;
            const size_t row = (size_t)b * N + base + r;
            g_eff[row * 64 + lane] = gz[r];
            g_cnode[row * 64 + lane] = first ? gz[r] : g_cnode[row * 64 + lane] + gz[r];
            g_agg[row * 64 + lane] = ga[r];
        }
    }
}

// ---- reversed neighbour lists: for every sender j the edge slots (i*10 + k) it feeds, ascending --
// The backward pass of the sender gather is a scatter; with the lists reversed it becomes a
// gather again -- no atomics, a fixed summation order (the first version scattered with fp32 global
// atomics: 64 per edge, 61 % of a planner iteration, and not reproducible run to run).
// One workgroup per sample; out-degrees counted and scanned in LDS; the lists are filled and put
// in order in LDS too when they fit (dynamic LDS = 2*N ints, + 10*N ints with `in_lds`).
#define KB_REV_THREADS 1024
#define KB_REV_LDS(N, in_lds) ((size_t)((in_lds) ? 12 : 2) * (N) * sizeof(int))
#define KB_REV_LDS_MAX_N 3072
// T threads per sample: 1024, or 256 for samples of up to 512 particles (four times as many samples in flight per CU:
// the kernel is a chain of short LDS phases, 45 -> 15 us at 1500 x 100)
template <int T>
__global__ void __launch_bounds__(T)
kb_reverse_lists(const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt, int N,
                 int* __restrict__ rev_off /* [B][N+1] */, int* __restrict__ rev /* [B][N*10] */, int in_lds,
                 const int* __restrict__ n_real /* nullable [B]: receivers >= n_real[b] are padding whose
                                                   gradient is identically zero (training batches) */,
                 int n_real_mod = 0 /* > 0: the grid runs over several rollout steps' lists, sample = block % n_real_mod */) {
    extern __shared__ int s_rev[];
    const int b = blockIdx.x;
    const int n_recv = n_real ? n_real[n_real_mod > 0 ? b % n_real_mod : b] : N;
    reverse_lists<T>(nbr_idx + (size_t)b * N * DRP_K, nbr_cnt + (size_t)b * N, N, rev_off + (size_t)b * (N + 1),
                     rev + (size_t)b * N * DRP_K, in_lds, n_recv, s_rev);
}

// ---- aggregate backward.  The gradient of an edge's pre-activation is the receiver's g_agg row
//      under the edge's ReLU mask: g_u[i,k,:] = g_agg[i,:] . [relation effect of (i,k) > 0].  The
//      forward pass (km_prop<., TAPE>) leaves the 64 mask bits of every edge slot and propagation
//      step (8 B instead of a 256-B row); nothing else of the edge stage is kept.  Here: the
//      receiver term g_proj[i][0:64] = sum_k g_u and, over the reversed lists, the sender term
//      g_proj[j][64:128] (kb_edge_terms); kb_edge_encode and the weight gradients rebuild g_u the same way.
// Mask layout (k_mlp_split.h frag_positive_bits): two words per slot; feature f lives in word
// (f>>2)&1 at bit 31 - (16*(f>>5) + (f&3) + 4*((f&31)>>3)).  The four features 4q..4q+3 of a float4
// lane q are one nibble of word q&1.
__device__ __forceinline__ int kb_mask_word(int f) { return (f >> 2) & 1; }
__device__ __forceinline__ int kb_mask_bit(int f) { return 31 - (16 * (f >> 5) + (f & 3) + 4 * ((f & 31) >> 3)); }
__device__ __forceinline__ unsigned kb_mask_nibble(const unsigned* __restrict__ m2, int q) {
    return (m2[q & 1] >> (28 - 16 * (q >> 3) - 4 * ((q >> 1) & 3))) & 0xfu;      // bit 3 = component x ... bit 0 = w
}

// both edge terms of a node in one launch: g_proj[n][0:64] from the node's own slots (the same value
// g_agg[n] under 0/1 masks, added one by one as an edge loop would), g_proj[n][64:128] from the edges it
// feeds, in the order of the reversed lists (ascending receiver, then slot).
// Same 16-lanes-per-node layout as k_aggregate.
__global__ void __launch_bounds__(256)
kb_edge_terms(const float* __restrict__ g_agg, const unsigned* __restrict__ mask, const uint8_t* __restrict__ nbr_cnt,
              const int* __restrict__ rev_off, const int* __restrict__ rev, int N, float* __restrict__ g_proj, int chunks) {
    const KbRange rg = kb_range(N, chunks);
    const int b = rg.b;
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const float4* ga = reinterpret_cast<const float4*>(g_agg) + (size_t)b * N * 16;
    const unsigned* mk = mask + (size_t)b * N * DRP_K * 2;
    const int* ro = rev_off + (size_t)b * (N + 1);
    const int* rv = rev + (size_t)b * N * DRP_K;
    float* gp = g_proj + (size_t)b * N * 128;
    const uint8_t* nc = nbr_cnt + (size_t)b * N;
    for (int i = rg.lo + g; i < rg.hi; i += 16) {
        const int cnt = nc[i];
        const float4 gi = ga[(size_t)i * 16 + q];
        int nx = 0, ny = 0, nz = 0, nw = 0;
        for (int k = 0; k < cnt; ++k) {
            const unsigned nib = kb_mask_nibble(mk + ((size_t)i * DRP_K + k) * 2, q);
            nx += (nib >> 3) & 1; ny += (nib >> 2) & 1; nz += (nib >> 1) & 1; nw += nib & 1;
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < nx; ++t) acc.x += gi.x;
        for (int t = 0; t < ny; ++t) acc.y += gi.y;
        for (int t = 0; t < nz; ++t) acc.z += gi.z;
        for (int t = 0; t < nw; ++t) acc.w += gi.w;
        *reinterpret_cast<float4*>(gp + (size_t)i * 128 + q * 4) = acc;
        const int p0 = ro[i], p1 = ro[i + 1];
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = p0; p < p1; ++p) {
            const int e = rv[p];
            const unsigned nib = kb_mask_nibble(mk + (size_t)e * 2, q);
            const float4 v = ga[(size_t)(e / DRP_K) * 16 + q];
            acc.x += (nib & 8u) ? v.x : 0.0f;
            acc.y += (nib & 4u) ? v.y : 0.0f;
            acc.z += (nib & 2u) ? v.z : 0.0f;
            acc.w += (nib & 1u) ? v.w : 0.0f;
        }
        *reinterpret_cast<float4*>(gp + (size_t)i * 128 + 64 + q * 4) = acc;
    }
}

// ---- projection backward: g_eff += W_r^T g_proj[:, 0:64] + W_s^T g_proj[:, 64:128] -------------
__global__ void __launch_bounds__(256)
kb_project(const float* __restrict__ wraw, const float* __restrict__ g_proj, int N, float* __restrict__ g_eff,
           int chunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const KbRange rg = kb_range(N, chunks);
    const int b = rg.b;
    const float* wrp = wraw + W_RP_W;
    for (int base = rg.lo + wave * KB_R; base < rg.hi; base += nwave * KB_R) {
        float gr[KB_R], gs[KB_R], ge[KB_R];
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            const size_t row = (size_t)b * N + min(base + r, N - 1);
            gr[r] = g_proj[row * 128 + lane];
            gs[r] = g_proj[row * 128 + 64 + lane];
            ge[r] = g_eff[row * 64 + lane];
        }
        dense_bcast_ld<64, KB_R>(wrp, 193, 64, gr, ge, lane);
        dense_bcast_ld<64, KB_R>(wrp, 193, 128, gs, ge, lane);
#pragma unroll
        for (int r = 0; r < KB_R; ++r)
            if (base + r < N) g_eff[((size_t)b * N + base + r) * 64 + lane] = ge[r];
    }
}

// ---- particle encoder backward: g_pe = g_eff0 + W_pe^T g_cnode; through relu(W2 relu(W1 x + b1) + b2)
//      to the three impulse inputs: g_s_delta[b,n,0:3]
__global__ void __launch_bounds__(256)
kb_node_encode(const float* __restrict__ vw, const float* __restrict__ wraw, const float* __restrict__ s_delta,
               const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens, int dens_mod,
               const float* __restrict__ pe, const float* __restrict__ g_eff0, const float* __restrict__ g_cnode,
               int N, float* __restrict__ g_sdelta,
               float* __restrict__ dump_gpe /* nullable [B*N,64]: gradient at the encoder's output pre-activation */,
               float* __restrict__ dump_a1 /* [B*N,64]: relu(W1 x + b1) */,
               float* __restrict__ dump_gh1 /* [B*N,64]: gradient at the first layer's pre-activation */,
               float* __restrict__ dump_x /* [B*N,8]: the 5 encoder inputs */, int chunks) {
    __shared__ float w0t[5 * 64];
    lds_copy(w0t, vw + V_PE0_T, 5 * 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const KbRange rg = kb_range(N, chunks);
    const int b = rg.b;
    const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
    const float b0 = vw[V_PE0_B + lane];
    const float* sd = s_delta + (size_t)b * N * 3;
    const float* at = attr + (size_t)(b % attr_mod) * N;
    const float w1x = wraw[W_PE0_W + lane * 5 + 0], w1y = wraw[W_PE0_W + lane * 5 + 1], w1z = wraw[W_PE0_W + lane * 5 + 2];
    for (int base = rg.lo + wave * KB_R; base < rg.hi; base += nwave * KB_R) {
        float x[KB_R], h1[KB_R], gc[KB_R], gpe[KB_R], gh[KB_R];
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            const int i = min(base + r, N - 1);
            const size_t row = (size_t)b * N + i;
            float v = 0.0f;
            if (lane < 3) v = sd[i * 3 + lane];
            else if (lane == 3) v = at[i];
            else if (lane == 4) v = d;
            x[r] = v;
            h1[r] = b0;
            gc[r] = g_cnode[row * 64 + lane];
            gpe[r] = g_eff0[row * 64 + lane];
        }
        dense_bcast<5, KB_R>(w0t, x, h1, lane);                      // h1 pre-activation
        dense_bcast_ld<64, KB_R>(wraw + W_PP_W, 129, 0, gc, gpe, lane);   // + W_pe^T g_cnode
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            const size_t row = (size_t)b * N + min(base + r, N - 1);
            gpe[r] = (pe[row * 64 + lane] > 0.0f) ? gpe[r] : 0.0f;  // through the encoder's output ReLU
            gh[r] = 0.0f;
        }
        dense_bcast_ld<64, KB_R>(wraw + W_PE2_W, 64, 0, gpe, gh, lane);   // W2^T
#pragma unroll
        for (int r = 0; r < KB_R; ++r) {
            const float g = (h1[r] > 0.0f) ? gh[r] : 0.0f;
            const float ox = wave_sum(g * w1x), oy = wave_sum(g * w1y), oz = wave_sum(g * w1z);
            const int i = base + r;
            if (i < N && lane < 3) g_sdelta[((size_t)b * N + i) * 3 + lane] = (lane == 0) ? ox : (lane == 1) ? oy : oz;
            if (dump_gpe != nullptr && i < N) {
                const size_t row = (size_t)b * N + i;
                dump_gpe[row * 64 + lane] = gpe[r];
                dump_a1[row * 64 + lane] = fmaxf(h1[r], 0.0f);
                dump_gh1[row * 64 + lane] = g;
                if (lane < 8) dump_x[row * 8 + lane] = x[r];
            }
        }
    }
}

// ---- gen_s_delta backward: g_action[b, 0:4] = sum_n J_n^T g_s_delta[n], forward-mode over the four
//      push parameters (sx, sy, ex, ey); the hard mask is a constant (planners.py:248)
template <int ND>
struct Dual {
    float v, d[ND];
};
template <int ND> __device__ __forceinline__ Dual<ND> dconst(float v) {
    Dual<ND> r; r.v = v;
#pragma unroll
    for (int c = 0; c < ND; ++c) r.d[c] = 0.0f;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator+(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v + b.v;
#pragma unroll
    for (int c = 0; c < ND; ++c) r.d[c] = a.d[c] + b.d[c];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator-(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v - b.v;
#pragma unroll
    for (int c = 0; c < ND; ++c) r.d[c] = a.d[c] - b.d[c];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator*(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v * b.v;
#pragma unroll
    for (int c = 0; c < ND; ++c) r.d[c] = a.d[c] * b.v + a.v * b.d[c];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator/(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v / b.v;
    const float ib = 1.0f / b.v;
#pragma unroll
    for (int c = 0; c < ND; ++c) r.d[c] = (a.d[c] - r.v * b.d[c]) * ib;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> dsqrt(const Dual<ND>& a) {
    Dual<ND> r; r.v = sqrtf(a.v);
    const float k = 0.5f / r.v;
#pragma unroll
    for (int c = 0; c < ND; ++c) r.d[c] = a.d[c] * k;
    return r;
}

// ---- Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8) + the clip box ------------
//      planners.py:674, :743-746, :756-764
struct KbAdam {                 // the optimiser step of a row's pushes at the end of kb_sdelta (act == null: not in this launch)
    float* act; float* m; float* v; float* act_copy /* nullable: pinned host memory */;
    int n_row;                  // values per row: H * 4
    float step_size, bc2_sqrt, b1;
    float4 lo, hi;
};
__device__ __forceinline__ float adam_update(float a0, float g, float& m, float& v, float step_size, float bc2_sqrt, float b1,
                                             float l, float h) {
    const float b2 = 0.999f, eps = 1e-8f;
    const float mi = m + (g - m) * (1.0f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v * b2 + (1.0f - b2) * g * g;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float a = a0 - step_size * (mi / denom);
    return fminf(fmaxf(a, l), h);
}

// directions 0..3: the push (sx, sy, ex, ey); 4..6: the particle's own position (x, y, z).
// g_action[b, 0:4] = sum_n J_n^T g_s_delta[n];  g_pos[b, n, 0:3] += J_pos^T g_s_delta[n] (nullable)
__global__ void __launch_bounds__(256)
kb_sdelta(const float* __restrict__ s_cur, int s_mod, size_t s_stride, const float* actions /* may be adam.act: no restrict */,
          size_t act_stride, const float* __restrict__ g_sdelta, int N, DrpCam cam, float* g_action /* read back below */,
          size_t gact_stride, float* __restrict__ g_pos, size_t gpos_stride, KbAdam adam = KbAdam{}) {
    typedef Dual<7> D;
    __shared__ float red[4][4];
    const int b = blockIdx.x;
    // the push of this row and step, by value: with rollout step 0 the optimiser step at the end of this kernel overwrites
    // the very elements (adam.act is the actions buffer)
    const float act[4] = {actions[(size_t)b * act_stride + 0], actions[(size_t)b * act_stride + 1],
                          actions[(size_t)b * act_stride + 2], actions[(size_t)b * act_stride + 3]};
    const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
    const float* gs = g_sdelta + (size_t)b * N * 3;
    // camera-frame start / end as duals of (sx, sy, ex, ey): s3 = (sx, 0, -sy), e3 = (ex, 0, -ey)
    D sc[3], ec[3];
    const float igs = 1.0f / cam.gs;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float m0 = cam.m[r * 4 + 0], m2 = cam.m[r * 4 + 2], m3 = cam.m[r * 4 + 3];
        sc[r] = dconst<7>((m0 * act[0] - m2 * act[1] + m3) * igs);
        sc[r].d[0] = m0 * igs; sc[r].d[1] = -m2 * igs;
        ec[r] = dconst<7>((m0 * act[2] - m2 * act[3] + m3) * igs);
        ec[r].d[2] = m0 * igs; ec[r].d[3] = -m2 * igs;
    }
    const D vx = ec[0] - sc[0], vy = ec[1] - sc[1], vz = ec[2] - sc[2];
    const D len = dsqrt(vx * vx + vy * vy + vz * vz);
    const D dx = vx / len, dy = vy / len, dz = vz / len;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        D px = dconst<7>(s[n * 3 + 0]), py = dconst<7>(s[n * 3 + 1]), pz = dconst<7>(s[n * 3 + 2]);
        px.d[4] = 1.0f; py.d[5] = 1.0f; pz.d[6] = 1.0f;
        const D rx = px - sc[0], ry = py - sc[1], rz = pz - sc[2];
        const D v = ry * dx - rx * dy;                                  // (p - s) . ortho, ortho = (-dy, dx, 0)
        const D u = rx * dx + ry * dy + rz * dz;
        float g7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (u.v < len.v && u.v > 0.0f) {                                 // hard mask (constant)
            // soft = exp(-max(relu(-w - v), relu(v - w)) / 0.01)
            D pen = dconst<7>(0.0f);
            const float lo = -DRP_PUSHER_W - v.v, hi = v.v - DRP_PUSHER_W;
            if (lo > 0.0f && lo >= hi) pen = dconst<7>(-DRP_PUSHER_W) - v;
            else if (hi > 0.0f) pen = v - dconst<7>(DRP_PUSHER_W);
            const float e = expf(-pen.v / DRP_SOFT_SCALE);
            D soft = dconst<7>(e);
#pragma unroll
            for (int c = 0; c < 7; ++c) soft.d[c] = -e * pen.d[c] / DRP_SOFT_SCALE;
            const D te = (ec[0] - px) * dx + (ec[1] - py) * dy + (ec[2] - pz) * dz;
            const D base = te * soft;
            const D ox = base * dx, oy = base * dy, oz = base * dz;
            const float g0 = gs[n * 3 + 0], g1 = gs[n * 3 + 1], g2 = gs[n * 3 + 2];
#pragma unroll
            for (int c = 0; c < 7; ++c) g7[c] = g0 * ox.d[c] + g1 * oy.d[c] + g2 * oz.d[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += g7[c];
        if (g_pos != nullptr) {
            float* gp = g_pos + (size_t)b * gpos_stride + (size_t)n * 3;
            gp[0] += g7[4]; gp[1] += g7[5]; gp[2] += g7[6];
        }
    }
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float t = wave_sum(acc[c]);
        if ((threadIdx.x & 63) == 0) red[wave][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float t = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w][threadIdx.x];
        g_action[(size_t)b * gact_stride + threadIdx.x] = t;
    }
    // The row's optimiser step, in the launch that completes its gradient (rollout step 0's: the later steps' launches ran
    // before it, and with step 0 `g_action + b * gact_stride` is the start of the row's H * 4 gradients): thread x owns
    // value x of the row -- for this step's four values the sum it has just stored itself.
    if (adam.act != nullptr) {
        // (value x < 4 is the sum thread x has just stored itself; the others come from the later steps' launches)
        for (int x = threadIdx.x; x < adam.n_row; x += blockDim.x) {         // any horizon: a row's H * 4 values over the block
            const size_t i = (size_t)b * adam.n_row + x;
            const int c = x & 3;
            const float l = (c == 0) ? adam.lo.x : (c == 1) ? adam.lo.y : (c == 2) ? adam.lo.z : adam.lo.w;
            const float h = (c == 0) ? adam.hi.x : (c == 1) ? adam.hi.y : (c == 2) ? adam.hi.z : adam.hi.w;
            float mi = adam.m[i], vi = adam.v[i];
            const float a = adam_update(adam.act[i], g_action[(size_t)b * gact_stride + x], mi, vi, adam.step_size,
                                        adam.bc2_sqrt, adam.b1, l, h);
            adam.m[i] = mi;
            adam.v[i] = vi;
            adam.act[i] = a;
            if (adam.act_copy != nullptr) adam.act_copy[i] = a;
        }
    }
}

// what the weight-gradient pass of the training path needs from kb_edge_encode, per edge slot
// (row = (b*N + i)*10 + k; every slot is written, padded ones with zero gradients)
struct KbEdgeDump {
    float* re;    // [rows,64] relation encoding relu(h3)
    float* a2;    // [rows,64] relu(h2)
    float* a1;    // [rows,64] relu(h1)
    float* x0;    // [rows,8]  the 6 encoder inputs
    float* gce;   // [rows,64] gradient at c_edge (= W_e . re + ...), rebuilt from the masks
    float* g3;    // [rows,64] gradients at the three pre-activations
    float* g2;
    float* g1;
};

// ---- relation encoder backward (horizons > 1, training): the gradient at c_edge (rebuilt from the
//      propagation steps' masks and g_agg rows) -> through W_e and the three
//      Linear+ReLU layers (forward recomputed per slot) to the position-difference inputs
//      x[2:5] = s_r - s_s (gnn_dyn.py:179-180):  g_pos[recv] += g,  g_pos[send] -= g  (atomics)
// One wave = the slots of one receiver, as k_edge_encode.
#define KB_EDGE_ENCODE_LDS ((size_t)(6 * 64 + 5 * 4096) * sizeof(float))
__global__ void __launch_bounds__(256)
kb_edge_encode(const float* __restrict__ vw, const float* __restrict__ wraw, const float* __restrict__ s_cur, int s_mod,
               size_t s_stride, const float* __restrict__ attr, int attr_mod, const float* __restrict__ dens,
               int dens_mod, const int16_t* __restrict__ nbr_idx, const uint8_t* __restrict__ nbr_cnt,
               const float* __restrict__ g_agg_hist /* [3][B*N,64]: g_agg of the three propagation steps */,
               const unsigned* __restrict__ mask_hist /* [3][B*N*10][2] */, size_t bn,
               int N, float* __restrict__ g_pos /* nullable */, size_t gpos_stride,
               float* __restrict__ gpos_edge /* [B,N,10,4]: the slot's gradient w.r.t. s_r - s_s, for kb_gather_pos */,
               KbEdgeDump dump, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w0 = lds;               // [6][64] forward packs
    float* w2 = w0 + 6 * 64;
    float* w4 = w2 + 4096;
    float* bwe = w4 + 4096;        // backward (torch [out][in]) copies: W_e, RE4, RE2 -- read 64 rows per
    float* bw4 = bwe + 4096;       // layer and slot pass; from L2 their latency was the kernel's time
    float* bw2 = bw4 + 4096;
    lds_copy(w0, vw + V_RE0_T, 6 * 64);
    lds_copy(w2, vw + V_RE2_T, 4096);
    lds_copy(w4, vw + V_RE4_T, 4096);
    for (int t = threadIdx.x; t < 4096; t += blockDim.x) bwe[t] = wraw[W_RP_W + (t >> 6) * 193 + (t & 63)];
    lds_copy(bw4, wraw + W_RE4_W, 4096);
    lds_copy(bw2, wraw + W_RE2_W, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const KbRange rg = kb_range(N, chunks);
    const int b = rg.b;
    const float d = dens[b % dens_mod] / DRP_DENS_SCALE;
    const float b0 = vw[V_RE0_B + lane], b2 = vw[V_RE2_B + lane], b4 = vw[V_RE4_B + lane];
    const float* s = s_cur + (size_t)(b % s_mod) * s_stride;
    const float* at = attr + (size_t)(b % attr_mod) * N;
    float* gp = g_pos ? g_pos + (size_t)b * gpos_stride : nullptr;
    const bool dumping = dump.re != nullptr;
    const int mbit = kb_mask_bit(lane), mword = kb_mask_word(lane);
    const float wx = wraw[W_RE0_W + lane * 6 + 2], wy = wraw[W_RE0_W + lane * 6 + 3], wz = wraw[W_RE0_W + lane * 6 + 4];
    constexpr int R = 5;           // two passes of five slots keep the register count moderate
    for (int i = rg.lo + wave; i < rg.hi; i += nwave) {
        const int cnt = nbr_cnt[(size_t)b * N + i];
        const int16_t* nb = nbr_idx + ((size_t)b * N + i) * DRP_K;
        const float ar = at[i];
        const float sr = (lane >= 2 && lane < 5) ? s[i * 3 + lane - 2] : 0.0f;
        float recv_sum = 0.0f;       // lanes 0..2: sum over this receiver's slots, in slot order
        float ga3[DRP_PSTEP];
#pragma unroll
        for (int p = 0; p < DRP_PSTEP; ++p) ga3[p] = g_agg_hist[((size_t)p * bn + (size_t)b * N + i) * 64 + lane];
        for (int k0 = 0; k0 < (dumping ? DRP_K : cnt); k0 += R) {
            float x[R], h1[R], h2[R], h3[R], g[R], t[R];
            int js[R];
            const size_t row0 = ((size_t)b * N + i) * DRP_K + k0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                js[r] = (k0 + r < cnt) ? (int)nb[k0 + r] : i;
                float v = 0.0f;
                if (lane == 0) v = ar;
                else if (lane == 1) v = at[js[r]];
                else if (lane < 5) v = sr - s[js[r] * 3 + lane - 2];
                else if (lane == 5) v = d;
                x[r] = v;
                h1[r] = b0;
                if (dumping && lane < 8) dump.x0[(row0 + r) * 8 + lane] = v;
            }
            dense_bcast<6, R>(w0, x, h1, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) { x[r] = fmaxf(h1[r], 0.0f); h2[r] = b2; }
            if (dumping)
#pragma unroll
                for (int r = 0; r < R; ++r) dump.a1[(row0 + r) * 64 + lane] = x[r];
            dense_bcast<64, R>(w2, x, h2, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) { x[r] = fmaxf(h2[r], 0.0f); h3[r] = b4; }
            if (dumping)
#pragma unroll
                for (int r = 0; r < R; ++r) dump.a2[(row0 + r) * 64 + lane] = x[r];
            dense_bcast<64, R>(w4, x, h3, lane);
            if (dumping)
#pragma unroll
                for (int r = 0; r < R; ++r) dump.re[(row0 + r) * 64 + lane] = fmaxf(h3[r], 0.0f);
            // backward
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // d loss / d c_edge of the slot: the three propagation steps share c_edge
                float gv = 0.0f;
                if (k0 + r < cnt) {
                    const size_t e = ((size_t)b * N + i) * DRP_K + k0 + r;
#pragma unroll
                    for (int p = 0; p < DRP_PSTEP; ++p)
                        if ((mask_hist[((size_t)p * bn * DRP_K + e) * 2 + mword] >> mbit) & 1u) gv += ga3[p];
                }
                g[r] = gv;
                t[r] = 0.0f;
            }
            if (dumping)
#pragma unroll
                for (int r = 0; r < R; ++r) dump.gce[(row0 + r) * 64 + lane] = g[r];
            dense_bcast_ld<64, R>(bwe, 64, 0, g, t, lane);                        // W_e^T
#pragma unroll
            for (int r = 0; r < R; ++r) { g[r] = (h3[r] > 0.0f) ? t[r] : 0.0f; t[r] = 0.0f; }
            if (dumping)
#pragma unroll
                for (int r = 0; r < R; ++r) dump.g3[(row0 + r) * 64 + lane] = g[r];
            dense_bcast_ld<64, R>(bw4, 64, 0, g, t, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) { g[r] = (h2[r] > 0.0f) ? t[r] : 0.0f; t[r] = 0.0f; }
            if (dumping)
#pragma unroll
                for (int r = 0; r < R; ++r) dump.g2[(row0 + r) * 64 + lane] = g[r];
            dense_bcast_ld<64, R>(bw2, 64, 0, g, t, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float gh = (h1[r] > 0.0f) ? t[r] : 0.0f;
                if (dumping) dump.g1[(row0 + r) * 64 + lane] = gh;
                if (gp == nullptr) continue;
                const float ox = wave_sum(gh * wx), oy = wave_sum(gh * wy), oz = wave_sum(gh * wz);
                if (lane < 4) {
                    const float v = (k0 + r < cnt && lane < 3) ? ((lane == 0) ? ox : (lane == 1) ? oy : oz) : 0.0f;
                    recv_sum += v;
                    if (k0 + r < DRP_K) gpos_edge[(((size_t)b * N + i) * DRP_K + k0 + r) * 4 + lane] = v;
                }
            }
        }
        if (gp != nullptr && lane < 3) gp[(size_t)i * 3 + lane] += recv_sum;     // this wave is the only writer of row i
    }
}

// sender part of the relation encoder's position gradient: g_pos[j] -= sum over the edges j feeds,
// in the order of the reversed lists (no atomics)
__global__ void __launch_bounds__(256)
kb_gather_pos(const float* __restrict__ gpos_edge, const int* __restrict__ rev_off, const int* __restrict__ rev, int N,
              float* __restrict__ g_pos, size_t gpos_stride, int add_recv = 0, const uint8_t* __restrict__ nbr_cnt = nullptr,
              const float* __restrict__ g_add = nullptr /* nullable, laid out as g_pos: added first (the residual's share of the
                                                           next step's gradient -- the trainer's kt_add folded in) */) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int* ro = rev_off + (size_t)b * (N + 1);
    const int* rv = rev + (size_t)b * N * DRP_K;
    const float4* ge = reinterpret_cast<const float4*>(gpos_edge) + (size_t)b * N * DRP_K;
    float* gp = g_pos + (size_t)b * gpos_stride + (size_t)j * 3;
    float g0 = gp[0], g1 = gp[1], g2 = gp[2];
    if (g_add != nullptr) {
        const float* ga = g_add + (size_t)b * gpos_stride + (size_t)j * 3;
        g0 += ga[0]; g1 += ga[1]; g2 += ga[2];
    }
    if (add_recv) {
        // receiver part (kmb_edge_encode leaves it here): the sum over the node's own slots, in slot order
        const int cnt = nbr_cnt[(size_t)b * N + j];
        float rx = 0.0f, ry = 0.0f, rz = 0.0f;
        for (int k = 0; k < cnt; ++k) {
            const float4 v = ge[(size_t)j * DRP_K + k];
            rx += v.x; ry += v.y; rz += v.z;
        }
        g0 += rx; g1 += ry; g2 += rz;
    }
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int p = ro[j]; p < ro[j + 1]; ++p) {
        const float4 v = ge[rv[p]];
        ax += v.x; ay += v.y; az += v.z;
    }
    gp[0] = g0 - ax; gp[1] = g1 - ay; gp[2] = g2 - az;
}

// the same step as a launch of its own (the trainer's weights; a planner row whose gradient needs no kb_sdelta launch)
__global__ void k_adam(float* __restrict__ act, const float* __restrict__ grad, float* __restrict__ m,
                       float* __restrict__ v, int n, float step_size, float bc2_sqrt, float4 lo, float4 hi,
                       float b1 = 0.9f, float* __restrict__ act_copy = nullptr /* pinned host memory: the updated values once more */,
                       const unsigned* __restrict__ skip = nullptr /* not null and set: the gradient is not to be trusted, nothing moves */,
                       unsigned* __restrict__ skip_copy = nullptr /* pinned host memory: what `skip` held */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && skip != nullptr && skip_copy != nullptr) *skip_copy = *skip;
    if (i >= n) return;
    if (skip != nullptr && *skip != 0u) return;
    const int c = i & 3;
    const float l = (c == 0) ? lo.x : (c == 1) ? lo.y : (c == 2) ? lo.z : lo.w;
    const float h = (c == 0) ? hi.x : (c == 1) ? hi.y : (c == 2) ? hi.z : hi.w;
    float mi = m[i], vi = v[i];
    const float a = adam_update(act[i], grad[i], mi, vi, step_size, bc2_sqrt, b1, l, h);
    m[i] = mi;
    v[i] = vi;
    act[i] = a;
    if (act_copy != nullptr) act_copy[i] = a;
}
