// Training on the same kernels: train/train_gnn_dyn.py:159-189
//   loss = sum_t sum_b mse(s_pred_t[b, :n_b], s_nxt_t[b, :n_b]) / (n_rollout * B),   s_cur <- s_pred
//   loss.backward(); Adam(lr, betas=(beta1, 0.999)).step()
// The state gradients run through the reverse-mode kernels of k_backward.h (graph masks constant,
// s_delta is data); this file adds the loss, the weight gradients and the optimiser.
// Weight gradients are plain fp32 outer-product sums over the rows the backward kernels leave in
// HBM: lane = output feature keeps one row of dW in registers, the input row is broadcast lane by
// lane (v_readlane), waves stride over the rows, partial sums meet in LDS, then one partial per
// workgroup in HBM, added up by a second launch in a fixed order (no atomics: deterministic).
#pragma once
#include "drp_common.h"
#include "k_mlp_split.h"

// d loss / d s_pred_t and the loss itself, every rollout step in one launch (grid B x H; blockIdx.y = t: the states of
// step t sit t * N * 3 floats into a sample's block, on both sides).  particle_nums[b] real particles per sample, the rest
// of the N rows are padding (zero rows at the origin, never connected to a real particle).
__global__ void __launch_bounds__(256)
kt_mse_grad(const float* __restrict__ s_pred, size_t pred_stride, const float* __restrict__ s_nxt, size_t nxt_stride,
            const int* __restrict__ particle_nums, int N, float scale, float* __restrict__ g_out /* [H][B][N][3] */,
            double* __restrict__ loss /* [H][B] */, float* __restrict__ zero /* nullable */, size_t n_zero) {
    __shared__ double s_w[4];
    const int b = blockIdx.x, t = blockIdx.y, B = gridDim.x;
    // with a backward pass to follow: the gradient blob (and the counters behind it) start at zero -- here instead of a
    // memset of their own (two fill launches)
    if (zero != nullptr)
        for (size_t e = ((size_t)t * B + b) * 256 + threadIdx.x; e < n_zero; e += (size_t)gridDim.x * gridDim.y * 256) zero[e] = 0.0f;
    const int nb = particle_nums[b];
    const float* p = s_pred + (size_t)b * pred_stride + (size_t)t * N * 3;
    const float* q = s_nxt + (size_t)b * nxt_stride + (size_t)t * N * 3;
    float* g = g_out + ((size_t)t * B + b) * N * 3;
    const float inv = scale / (float)(3 * nb);
    double acc = 0.0;
    for (int i = threadIdx.x; i < N * 3; i += 256) {
        float gv = 0.0f;
        if (i < nb * 3) {
            const float d = p[i] - q[i];
            gv = 2.0f * d * inv;
            acc += (double)d * (double)d;
        }
        g[i] = gv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[(size_t)t * B + b] = (s_w[0] + s_w[1] + s_w[2] + s_w[3]) * (double)inv;   // one slot per (step, sample)
}

// A training batch arrives as ONE copy (pinned staging -> arena: states | impulses | attributes | densities | particle counts,
// the caller's layouts).  The states and the counts are read where they land; this launch puts the rest where the step
// kernels expect it: the impulses step-major ([H][B][N][3]: a step's slice is the s_delta of that step, and the tape's),
// a_cur = attrs[:, 0] (train/train_gnn_dyn.py:173), the densities.
__global__ void __launch_bounds__(256)
kt_unpack_inputs(const float* __restrict__ sdelta_in /* [B][H][N][3] */, const float* __restrict__ attrs_in /* [B][H+1][N] */,
                 const float* __restrict__ dens_in, int B, int H, int N, float* __restrict__ sdelta_out /* [H][B][N][3] */,
                 float* __restrict__ attr_out /* [B][N] */, float* __restrict__ dens_out,
                 const float4* __restrict__ arena_src = nullptr /* not null: the whole staged batch (pinned HOST memory the device
                                                                   reads over the bus: the inputs above point into it too) ... */,
                 float4* __restrict__ arena_dst = nullptr /* ... copied to the device arena the later kernels read, n16 float4 */,
                 size_t n16 = 0) {
    // the upload as part of this launch: a copy engine's transfer in front of it costs ~80 us of hand-over between the engines
    // on a stream that is otherwise kernels (tools/train_trace.sh), for 90 KB
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n16; e += (size_t)gridDim.x * 256) arena_dst[e] = arena_src[e];
    const size_t n3 = (size_t)N * 3, total = (size_t)B * H * n3;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t bt = e / n3, r = e - bt * n3;
        const int b = (int)(bt / H), t = (int)(bt - (size_t)b * H);
        sdelta_out[((size_t)t * B + b) * n3 + r] = sdelta_in[e];
        if (e < (size_t)B * N) {
            const int ab = (int)(e / N), ai = (int)(e - (size_t)ab * N);
            attr_out[e] = attrs_in[(size_t)ab * (H + 1) * N + ai];
        }
        if (e < (size_t)B) dens_out[e] = dens_in[e];
    }
}

__global__ void kt_add(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// dW[lane * lane_stride + k * k_stride] += sum_rows g[row][lane] * x[row][k]      k < IN
// db[lane]                              += sum_rows g[row][lane]                   (nullable)
// dwd[lane * lane_stride]               += sum_rows g[row][lane] * dens[row / rows_per_sample] / 5000   (nullable)
// Two launches, no atomics, fixed summation order: kt_wgrad leaves one partial [IN+2][64] per
// workgroup in `part`, kt_wgrad_reduce adds them up in workgroup order.
template <int IN>
__global__ void __launch_bounds__(256)
kt_wgrad(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx, long M, float* __restrict__ part,
         const float* __restrict__ dens, int dens_mod, long rows_per_sample) {
    extern __shared__ float s_part[];                  // [3][IN + 2][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) acc[k] = 0.0f;
    float accb = 0.0f, accd = 0.0f;
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        const float gv = g[row * ldg + lane];
        const float xv = (IN == 64 || lane < IN) ? x[row * ldx + lane] : 0.0f;
#pragma unroll
        for (int k = 0; k < IN; ++k) acc[k] = fmaf(gv, bcast_lane(xv, k), acc[k]);
        accb += gv;
        if (dens != nullptr) accd = fmaf(gv, dens[(row / rows_per_sample) % dens_mod] / DRP_DENS_SCALE, accd);
    }
    if (wave > 0) {
        float* dst = s_part + (size_t)(wave - 1) * (IN + 2) * 64;
#pragma unroll
        for (int k = 0; k < IN; ++k) dst[k * 64 + lane] = acc[k];
        dst[IN * 64 + lane] = accb;
        dst[(IN + 1) * 64 + lane] = accd;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float* src = s_part + (size_t)w * (IN + 2) * 64;
#pragma unroll
            for (int k = 0; k < IN; ++k) acc[k] += src[k * 64 + lane];
            accb += src[IN * 64 + lane];
            accd += src[(IN + 1) * 64 + lane];
        }
        float* out = part + (size_t)blockIdx.x * (IN + 2) * 64;
#pragma unroll
        for (int k = 0; k < IN; ++k) out[k * 64 + lane] = acc[k];
        out[IN * 64 + lane] = accb;
        out[(IN + 1) * 64 + lane] = accd;
    }
}
#define KT_WGRAD_MAX_BLOCKS 128
#define KT_WGRAD_LDS(IN) ((size_t)3 * ((IN) + 2) * 64 * sizeof(float))

template <int IN>
__global__ void __launch_bounds__(256)
kt_wgrad_reduce(const float* __restrict__ part, int nblocks, float* __restrict__ dW, int lane_stride, int k_stride,
                float* __restrict__ db, float* __restrict__ dwd) {
    __shared__ float s_w[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.x;      // k in [0, IN + 2)
    // wave w adds the partials w, w+4, ... (four independent loads in flight), then the four wave
    // sums are added in wave order: a fixed order for a given number of partials
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
    int b = wave;
    for (; b + 12 < nblocks; b += 16) {
        t0 += part[((size_t)b * (IN + 2) + k) * 64 + lane];
        t1 += part[((size_t)(b + 4) * (IN + 2) + k) * 64 + lane];
        t2 += part[((size_t)(b + 8) * (IN + 2) + k) * 64 + lane];
        t3 += part[((size_t)(b + 12) * (IN + 2) + k) * 64 + lane];
    }
    for (; b < nblocks; b += 4) t0 += part[((size_t)b * (IN + 2) + k) * 64 + lane];
    s_w[wave][lane] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (wave != 0) return;
    const float t = (s_w[0][lane] + s_w[1][lane]) + (s_w[2][lane] + s_w[3][lane]);
    if (k < IN) dW[(size_t)lane * lane_stride + (size_t)k * k_stride] += t;
    else if (k == IN) { if (db != nullptr) db[lane] += t; }
    else if (dwd != nullptr) dwd[(size_t)lane * lane_stride] += t;
}

// ---- several outer-product sums per launch ------------------------------------------------------------------
// A training iteration needs 17 weight gradients per rollout step, each a 20-us kernel over ~1 000 - 12 000 rows
// plus its reduction: 34 launches per step, a fifth of the iteration in launch overheads and tails.  The jobs
// whose inputs exist at the same point of the stream (the three of a propagation step, the encoder's, the relation
// encoder's four ...) go through ONE pair of launches: blockIdx.y selects the job, the arithmetic, the row -> (block,
// wave) assignment and the order of every sum are kt_wgrad<IN>'s / kt_wgrad_reduce<IN>'s, so the gradients keep
// their bits.  IN is a run-time number here (rows of x shorter than 64 are read as zero beyond IN).
#define KT_WGRAD_MAX_JOBS 6
struct WgradJob {
    const float* g; const float* x; float* dW; float* db; float* dwd; const float* dens; float* part;
    long M, rows_per_sample;
    int ldg, ldx, lane_stride, k_stride, dens_mod, in, blocks;
};
struct WgradJobs {
    WgradJob j[KT_WGRAD_MAX_JOBS];
};

__device__ __forceinline__ void wgrad_valu_body(const WgradJob& q, float* s_part /* [3][66][64] */) {
    if ((int)blockIdx.x >= q.blocks) return;
    const int IN = q.in;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ g = q.g;
    const float* __restrict__ x = q.x;
    const float* __restrict__ dens = q.dwd ? q.dens : nullptr;
    float acc[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) acc[k] = 0.0f;
    float accb = 0.0f, accd = 0.0f;
    for (long row = (long)blockIdx.x * 4 + wave; row < q.M; row += (long)q.blocks * 4) {
        const float gv = g[row * q.ldg + lane];
        const float xv = (lane < IN) ? x[row * q.ldx + lane] : 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k) acc[k] = fmaf(gv, bcast_lane(xv, k), acc[k]);
        accb += gv;
        if (dens != nullptr) accd = fmaf(gv, dens[(row / q.rows_per_sample) % q.dens_mod] / DRP_DENS_SCALE, accd);
    }
    if (wave > 0) {
        float* dst = s_part + (size_t)(wave - 1) * 66 * 64;
#pragma unroll
        for (int k = 0; k < 64; ++k) dst[k * 64 + lane] = acc[k];
        dst[64 * 64 + lane] = accb;
        dst[65 * 64 + lane] = accd;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float* src = s_part + (size_t)w * 66 * 64;
#pragma unroll
            for (int k = 0; k < 64; ++k) acc[k] += src[k * 64 + lane];
            accb += src[64 * 64 + lane];
            accd += src[65 * 64 + lane];
        }
        float* out = q.part + (size_t)blockIdx.x * 66 * 64;
#pragma unroll
        for (int k = 0; k < 64; ++k) out[k * 64 + lane] = acc[k];
        out[64 * 64 + lane] = accb;
        out[65 * 64 + lane] = accd;
    }
}
__global__ void __launch_bounds__(256)
kt_wgrad_multi(WgradJobs J) {
    extern __shared__ float s_part[];
    wgrad_valu_body(J.j[blockIdx.y], s_part);
}
// the jobs of a whole training iteration, in device memory: blockIdx.y walks `order[base ...]` (jobs of one size)
__global__ void __launch_bounds__(256)
kt_wgrad_list(const WgradJob* __restrict__ jobs, const int* __restrict__ order, int base) {
    extern __shared__ float s_part[];
    const WgradJob q = jobs[order[base + blockIdx.y]];
    wgrad_valu_body(q, s_part);
}
#define KT_WGRAD_MULTI_LDS ((size_t)3 * 66 * 64 * sizeof(float))

__global__ void __launch_bounds__(256)
kt_wgrad_reduce_multi(WgradJobs J) {
    __shared__ float s_w[4][64];
    const WgradJob& q = J.j[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.x;      // k: 0..63 columns, 64 bias, 65 density column
    if (k < 64 && k >= q.in) return;
    const float* __restrict__ part = q.part;
    const int nblocks = q.blocks;
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
    int b = wave;
    for (; b + 12 < nblocks; b += 16) {
        t0 += part[((size_t)b * 66 + k) * 64 + lane];
        t1 += part[((size_t)(b + 4) * 66 + k) * 64 + lane];
        t2 += part[((size_t)(b + 8) * 66 + k) * 64 + lane];
        t3 += part[((size_t)(b + 12) * 66 + k) * 64 + lane];
    }
    for (; b < nblocks; b += 4) t0 += part[((size_t)b * 66 + k) * 64 + lane];
    s_w[wave][lane] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (wave != 0) return;
    const float t = (s_w[0][lane] + s_w[1][lane]) + (s_w[2][lane] + s_w[3][lane]);
    if (k < 64) q.dW[(size_t)lane * q.lane_stride + (size_t)k * q.k_stride] += t;
    else if (k == 64) { if (q.db != nullptr) q.db[lane] += t; }
    else if (q.dwd != nullptr) q.dwd[(size_t)lane * q.lane_stride] += t;
}

// ---- the same outer-product sums on the fp32 matrix cores ----------------------------------------------------------
// dW^T[k][o] = sum_rows x[row][k] g[row][o] is a GEMM whose contraction runs over the ROWS: v_mfma_f32_32x32x2_f32 with
// A = x^T (32 input features x 2 rows), B = g (2 rows x 32 output features), so the accumulator's lane columns are
// consecutive output features -- the [k][o] layout of the partials kt_wgrad_reduce_multi adds up.  A wave takes row pairs
// (2p, 2p + 1), p = (block * 4 + wave) + 4 * blocks * t; bias and density column ride along as per-lane sums of the B operand.
// Every sum has a fixed order (the wave's row pairs, the four waves in wave order, the blocks in block order): bit-reproducible,
// as the VALU kernel -- in another order, so the last bits differ from it (the tests compare with the reference's autograd).
__device__ __forceinline__ void wgrad_mfma_body(const WgradJob& q, float* s_part /* [3][66][64] */) {
    if ((int)blockIdx.x >= q.blocks) return;
    const int IN = q.in;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float* __restrict__ g = q.g;
    const float* __restrict__ x = q.x;
    const float* __restrict__ dens = q.dwd ? q.dens : nullptr;
    const bool wide = IN > 32;                         // input features 32 .. 63 exist
    f32x16 acc[2][2];                                  // [kb: input block][ob: output block]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float accb[2] = {0.0f, 0.0f}, accd[2] = {0.0f, 0.0f};
    const long npairs = (q.M + 1) >> 1;
    const long step = (long)q.blocks * 4;
    // the operands of the next row pair are requested before this pair's MFMAs: a wave's pairs are a chain of L2 round trips otherwise
    auto fetch = [&](long p, float& g0, float& g1, float& x0, float& x1, float& dv) {
        const long row = 2 * p + h;
        const bool on = p < npairs && row < q.M;
        g0 = on ? g[row * q.ldg + j] : 0.0f;
        g1 = on ? g[row * q.ldg + 32 + j] : 0.0f;
        x0 = (on && j < IN) ? x[row * q.ldx + j] : 0.0f;
        x1 = (on && wide && 32 + j < IN) ? x[row * q.ldx + 32 + j] : 0.0f;
        dv = (dens != nullptr && on) ? dens[(row / q.rows_per_sample) % q.dens_mod] / DRP_DENS_SCALE : 0.0f;
    };
    // a wave's pairs are a chain of L2 round trips unless several are in flight: the operands of the next WG_AHEAD pairs are
    // requested before this pair's MFMAs (one pair ahead: 1.5 us a pair at 750 rows per block; the sums keep their order)
    constexpr int WG_AHEAD = 4;
    float g0[WG_AHEAD], g1[WG_AHEAD], x0[WG_AHEAD], x1[WG_AHEAD], dv[WG_AHEAD];
    long p = (long)blockIdx.x * 4 + wave;
#pragma unroll
    for (int a = 0; a < WG_AHEAD; ++a) fetch(p + a * step, g0[a], g1[a], x0[a], x1[a], dv[a]);
    for (; p < npairs; p += WG_AHEAD * step) {
#pragma unroll
        for (int a = 0; a < WG_AHEAD; ++a) {
            if (p + a * step < npairs) {               // wave-uniform
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[a], g0[a], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[a], g1[a], acc[0][1], 0, 0, 0);
                if (wide) {
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[a], g0[a], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[a], g1[a], acc[1][1], 0, 0, 0);
                }
                accb[0] += g0[a]; accb[1] += g1[a];
                accd[0] = fmaf(g0[a], dv[a], accd[0]); accd[1] = fmaf(g1[a], dv[a], accd[1]);
            }
            fetch(p + (a + WG_AHEAD) * step, g0[a], g1[a], x0[a], x1[a], dv[a]);
        }
    }
    // the two half-waves hold the even and the odd rows' bias sums: even + odd, in that order
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const float ob_ = __shfl_xor(accb[ob], 32, 64), od_ = __shfl_xor(accd[ob], 32, 64);
        accb[ob] = (h == 0) ? accb[ob] + ob_ : ob_ + accb[ob];
        accd[ob] = (h == 0) ? accd[ob] + od_ : od_ + accd[ob];
    }
    // accumulator register r of acc[kb][ob] in lane (j, h) = dW^T[k = 32 kb + (r & 3) + 8 (r >> 2) + 4 h][o = 32 ob + j]
    auto put = [&](float* dst) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[(32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * ob + j] = acc[kb][ob][r];
        if (h == 0) {
            dst[64 * 64 + j] = accb[0]; dst[64 * 64 + 32 + j] = accb[1];
            dst[65 * 64 + j] = accd[0]; dst[65 * 64 + 32 + j] = accd[1];
        }
    };
    if (wave > 0) put(s_part + (size_t)(wave - 1) * 66 * 64);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float* src = s_part + (size_t)w * 66 * 64;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[kb][ob][r] += src[(32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * ob + j];
            accb[0] += src[64 * 64 + j]; accb[1] += src[64 * 64 + 32 + j];
            accd[0] += src[65 * 64 + j]; accd[1] += src[65 * 64 + 32 + j];
        }
        put(q.part + (size_t)blockIdx.x * 66 * 64);
    }
}

__global__ void __launch_bounds__(256)
kt_wgrad_mfma_multi(WgradJobs J) {
    extern __shared__ float s_part[];
    wgrad_mfma_body(J.j[blockIdx.y], s_part);
}
__global__ void __launch_bounds__(256)
kt_wgrad_mfma_list(const WgradJob* __restrict__ jobs, const int* __restrict__ order, int base) {
    extern __shared__ float s_part[];
    const WgradJob q = jobs[order[base + blockIdx.y]];
    wgrad_mfma_body(q, s_part);
}

// The reduction for a whole iteration's jobs: blockIdx.y = a TARGET (one dW), whose jobs -- the same matrix's gradient from
// every rollout step and propagation step -- are added in list order, each job's partials as kt_wgrad_reduce_multi adds
// them: what a sequence of kt_wgrad_reduce_multi launches in that order leaves in a zeroed dW, bit for bit.
__global__ void __launch_bounds__(256)
kt_wgrad_reduce_lists(const WgradJob* __restrict__ jobs, const int* __restrict__ tgt_off, const int* __restrict__ tgt_jobs) {
    __shared__ float s_w[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.x;      // k: 0..63 columns, 64 bias, 65 density column
    const int e0 = tgt_off[blockIdx.y], e1 = tgt_off[blockIdx.y + 1];
    const WgradJob q0 = jobs[tgt_jobs[e0]];
    if (k < 64 && k >= q0.in) return;
    if (k == 64 && q0.db == nullptr) return;
    if (k == 65 && q0.dwd == nullptr) return;
    float tot = 0.0f;
    for (int e = e0; e < e1; ++e) {
        const WgradJob q = jobs[tgt_jobs[e]];
        const float* __restrict__ part = q.part;
        const int nblocks = q.blocks;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        int b = wave;
        for (; b + 12 < nblocks; b += 16) {
            t0 += part[((size_t)b * 66 + k) * 64 + lane];
            t1 += part[((size_t)(b + 4) * 66 + k) * 64 + lane];
            t2 += part[((size_t)(b + 8) * 66 + k) * 64 + lane];
            t3 += part[((size_t)(b + 12) * 66 + k) * 64 + lane];
        }
        for (; b < nblocks; b += 4) t0 += part[((size_t)b * 66 + k) * 64 + lane];
        __syncthreads();                               // the previous job's sums have been read
        s_w[wave][lane] = (t0 + t1) + (t2 + t3);
        __syncthreads();
        if (wave == 0) tot += (s_w[0][lane] + s_w[1][lane]) + (s_w[2][lane] + s_w[3][lane]);
    }
    if (wave != 0) return;
    if (k < 64) q0.dW[(size_t)lane * q0.lane_stride + (size_t)k * q0.k_stride] += tot;
    else if (k == 64) q0.db[lane] += tot;
    else q0.dwd[(size_t)lane * q0.lane_stride] += tot;
}

// column sums of a [M,3] gradient (bias of the predictor's last layer): ONE workgroup, every partial sum in a fixed
// order (a strided pass per thread, a wave reduction, the waves' sums in wave order) -- the first version added the
// waves' sums of 16 workgroups with fp32 atomics, the one place where two runs of the same training step could differ
// in the last bit (and, through Adam, drift apart by 1e-9 per step).
__global__ void __launch_bounds__(1024) kt_colsum3(const float* __restrict__ g, long M, float* __restrict__ out) {
    __shared__ float s_w[16][3];
    float a[3] = {0.f, 0.f, 0.f};
    for (long r = threadIdx.x; r < M; r += 1024) {
        a[0] += g[r * 3]; a[1] += g[r * 3 + 1]; a[2] += g[r * 3 + 2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = wave_sum(a[c]);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += s_w[w][threadIdx.x];
        out[threadIdx.x] += t;
    }
}


// ---- re-packing the updated weights on the device -------------------------------------------------------------
// After an optimiser step the engines' packed copies of the blob have to follow (w_valu, w_mfma, w_mfma_bwd: plain
// re-arrangements; w_split, w_split6: re-arrangements of fp16 / bf16 split terms).  Round 2 fetched the blob and ran the
// host packers again (five uploads, 0.4 of an iteration's 3.6 ms at the reference's batch of 4).  The plain copies are
// GATHERS: the index map of a packer is what it makes of the probe blob w[i] = i + 1 (0 -> the constant 0, -1e30 ->
// the sink row), built once from the host packer itself -- so the device copy cannot drift from it.  The split copies
// repeat pack_split / pack_split6 element by element (same rounding: RNE to fp16 by the conversion instruction, RNE to
// bf16 by the integer rule of host_bf16_rne; residuals in fp32); tests/test_gpu_train.py compares all five with the
// host packers byte by byte.
__device__ __forceinline__ void repack_gather_block(const float* __restrict__ w, const int* __restrict__ map, float* __restrict__ dst, int n, int blk) {
    const int i = blk * 256 + threadIdx.x;
    if (i >= n) return;
    const int m = map[i];
    dst[i] = m > 0 ? w[m - 1] : (m == 0 ? 0.0f : -1e30f);
}
__global__ void kt_repack_gather(const float* __restrict__ w, const int* __restrict__ map, float* __restrict__ dst, int n) {
    repack_gather_block(w, map, dst, n, (int)blockIdx.x);
}

__device__ __forceinline__ uint16_t dev_f16_rne(float f) {
    const _Float16 h = (_Float16)f;                  // v_cvt_f16_f32, round to nearest even
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float dev_f16_to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ uint16_t dev_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float dev_bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ int dev_split_feature(int s, int h, int jj) {
    const int r = 8 * (s & 1) + jj;
    return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// pack_split (k_mlp_split.h) on the device: grid.x = 4 jobs (RE0, RE2, RE4, RPE) x 16 blocks of 256 threads;
// one thread per (ob, s, lane, jj) of a 64x64 matrix (4096), per (ob, lane, jj) of the first layer (1024)
__global__ void __launch_bounds__(256)
kt_repack_split(const float* __restrict__ w, int shift, uint16_t* __restrict__ out,
                const int* __restrict__ shift_dev = nullptr /* nullable: the shift kt_repack_all derived from these weights */,
                int* __restrict__ shift_copy = nullptr /* pinned host memory: the shift used */) {
    if (shift_dev != nullptr) shift = *shift_dev;
    if (shift_copy != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *shift_copy = shift;
    const int job = blockIdx.x >> 4, e = (blockIdx.x & 15) * 256 + threadIdx.x;
    if (job == 0) {
        if (e < 1024) {
            const int jj = e & 7, lane = (e >> 3) & 63, ob = e >> 9;
            const int i = lane & 31, h = lane >> 5, o = 32 * ob + i;
            float v = 0.0f;
            if (h == 0 && jj < 6) v = w[W_RE0_W + o * 6 + jj];
            else if (h == 0 && jj == 6) v = w[W_RE0_B + o];
            v = ldexpf(v, shift);
            const uint16_t hq = dev_f16_rne(v);
            const float hi = dev_f16_to_f32(hq);
            out[(size_t)(S_RE0 + (0 * 2 + ob) * 64 + lane) * 8 + jj] = hq;
            out[(size_t)(S_RE0 + (1 * 2 + ob) * 64 + lane) * 8 + jj] = dev_f16_rne(v - hi);
        } else if (e < 1024 + 64) {
            // the chain's bias rows: 2^k b2, 2^k b4, b_rp, wd_rp (floats behind the fragments)
            const int o = e - 1024;
            float* srows = reinterpret_cast<float*>(out + (size_t)S_ROWS * 8);
            srows[o] = ldexpf(w[W_RE2_B + o], shift);
            srows[64 + o] = ldexpf(w[W_RE4_B + o], shift);
            srows[128 + o] = w[W_RP_B + o];
            srows[192 + o] = w[W_RP_W + o * 193 + 192];
        }
        return;
    }
    const int dst = job == 1 ? S_RE2 : (job == 2 ? S_RE4 : S_RPE);
    const int src = job == 1 ? W_RE2_W : (job == 2 ? W_RE4_W : W_RP_W);
    const int ld = job == 3 ? 193 : 64;
    const int jj = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 3, ob = e >> 11;
    const int i = lane & 31, h = lane >> 5;
    const float v = w[src + (32 * ob + i) * ld + dev_split_feature(s, h, jj)];
    const uint16_t hq = dev_f16_rne(v);
    const float hi = dev_f16_to_f32(hq);
    out[(size_t)(dst + ((0 * 2 + ob) * 4 + s) * 64 + lane) * 8 + jj] = hq;
    out[(size_t)(dst + ((1 * 2 + ob) * 4 + s) * 64 + lane) * 8 + jj] = dev_f16_rne(v - hi);
}

// The TRANSPOSED 64 x 64 layers of the backward pass in the six-product bf16 split (kmb_rows_bwd): the layout of
// pack_split6 with A[i][k] = W[k][col0 + i].  Packed on the device only, from the raw blob, at load time and after every
// optimiser step: grid.x = 6 jobs (AGG, RPR, RPS, PR0, PPE, PE2) x 16 blocks.
enum {                        // units of bf16x8, [part 3][ob 2][s 4][lane 64] each
    SB6_AGG = 0,
    SB6_RPR = SB6_AGG + 1536,
    SB6_RPS = SB6_RPR + 1536,
    SB6_PR0 = SB6_RPS + 1536,
    SB6_PPE = SB6_PR0 + 1536,
    SB6_PE2 = SB6_PPE + 1536,
    SB6_TOTAL = SB6_PE2 + 1536
};
__device__ __forceinline__ void repack_split6_bwd_block(const float* __restrict__ w, uint16_t* __restrict__ out, int blk) {
    const int job = blk >> 4, e = (blk & 15) * 256 + threadIdx.x;
    const int dsts[6] = {SB6_AGG, SB6_RPR, SB6_RPS, SB6_PR0, SB6_PPE, SB6_PE2};
    const int srcs[6] = {W_PP_W, W_RP_W, W_RP_W, W_PR0_W, W_PP_W, W_PE2_W};
    const int lds_[6] = {129, 193, 193, 64, 129, 64};
    const int cols[6] = {64, 64, 128, 0, 0, 0};
    const int dst = dsts[job], src = srcs[job], ld = lds_[job], col0 = cols[job];
    const int jj = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 3, ob = e >> 11;
    const int i = lane & 31, h = lane >> 5;
    float v = w[src + dev_split_feature(s, h, jj) * ld + col0 + 32 * ob + i];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const uint16_t q = dev_bf16_rne(v);
        out[((size_t)dst + ((part * 2 + ob) * 4 + s) * 64 + lane) * 8 + jj] = q;
        v -= dev_bf16_to_f32(q);
    }
}

__global__ void __launch_bounds__(256)
kt_repack_split6_bwd(const float* __restrict__ w, uint16_t* __restrict__ out) { repack_split6_bwd_block(w, out, (int)blockIdx.x); }

// pack_split6 on the device: grid.x = 7 jobs (AGG, RPR, RPS, PR0, PE2, PPE, PE0) x 16 blocks
__device__ __forceinline__ void repack_split6_block(const float* __restrict__ w, uint16_t* __restrict__ out, int blk) {
    const int job = blk >> 4, e = (blk & 15) * 256 + threadIdx.x;
    if (job == 6) {
        if (e >= 1024) return;
        const int jj = e & 7, lane = (e >> 3) & 63, ob = e >> 9;
        const int i = lane & 31, h = lane >> 5, o = 32 * ob + i;
        float v = 0.0f;
        if (h == 0 && jj < 5) v = w[W_PE0_W + o * 5 + jj];
        else if (h == 0 && jj == 5) v = w[W_PE0_B + o];
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            const uint16_t q = dev_bf16_rne(v);
            out[((size_t)S6_PE0 + (part * 2 + ob) * 64 + lane) * 8 + jj] = q;
            v -= dev_bf16_to_f32(q);
        }
        return;
    }
    const int dsts[6] = {S6_AGG, S6_RPR, S6_RPS, S6_PR0, S6_PE2, S6_PPE};
    const int srcs[6] = {W_PP_W, W_RP_W, W_RP_W, W_PR0_W, W_PE2_W, W_PP_W};
    const int lds_[6] = {129, 193, 193, 64, 64, 129};
    const int cols[6] = {64, 64, 128, 0, 0, 0};
    const int dst = dsts[job], src = srcs[job], ld = lds_[job], col0 = cols[job];
    const int jj = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 3, ob = e >> 11;
    const int i = lane & 31, h = lane >> 5;
    float v = w[src + (32 * ob + i) * ld + col0 + dev_split_feature(s, h, jj)];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const uint16_t q = dev_bf16_rne(v);
        out[((size_t)dst + ((part * 2 + ob) * 4 + s) * 64 + lane) * 8 + jj] = q;
        v -= dev_bf16_to_f32(q);
    }
}
__global__ void __launch_bounds__(256)
kt_repack_split6(const float* __restrict__ w, uint16_t* __restrict__ out) { repack_split6_block(w, out, (int)blockIdx.x); }

// The range shift of the split relation encoder from the weights, on the device: split_range_init + split_range_bound
// (k_mlp_split.h) operation by operation -- the same float sums in the same order, the same double products and maxima,
// range_shift_of's exponent read -- so that host and device agree on k (the host still compares, capi_train.h).
// One workgroup of 256 threads; thread o < 64 owns row o.
__device__ __forceinline__ void split_range_shift_block(const float* __restrict__ w, int forced, int* __restrict__ shift_out) {
    __shared__ double s_h[64];
    __shared__ float s_rs2[64], s_b2[64], s_rs4[64], s_b4[64];
    const int o = threadIdx.x;
    if (o < 64) {
        const float* w1 = w + W_RE0_W + o * 6;
        const float a1 = fabsf(w1[0]) + fabsf(w1[1]);
        const float d1 = fabsf(w1[2]) + fabsf(w1[3]) + fabsf(w1[4]);
        const float m1 = fabsf(w1[5]);
        const float b1 = fabsf(w[W_RE0_B + o]);
        float s2 = 0, s4 = 0;
        for (int k = 0; k < 64; ++k) {
            s2 += fabsf(w[W_RE2_W + o * 64 + k]);
            s4 += fabsf(w[W_RE4_W + o * 64 + k]);
        }
        s_rs2[o] = s2; s_b2[o] = fabsf(w[W_RE2_B + o]);
        s_rs4[o] = s4; s_b4[o] = fabsf(w[W_RE4_B + o]);
        s_h[o] = a1 * SPLIT_ENV_ATTR + d1 * SPLIT_ENV_DELTA + m1 * SPLIT_ENV_DENS + b1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double h1 = 0, h2 = 0, h3 = 0;
        for (int q = 0; q < 64; ++q) h1 = fmax(h1, s_h[q]);
        for (int q = 0; q < 64; ++q) h2 = fmax(h2, s_rs2[q] * h1 + s_b2[q]);
        for (int q = 0; q < 64; ++q) h3 = fmax(h3, s_rs4[q] * h2 + s_b4[q]);
        const double bound = fmax(fmax(h1, h2), fmax(h3, 1.0));
        *shift_out = forced != 0x7fffffff ? forced : range_shift_of(bound);
    }
}

// Everything that follows an optimiser step except the shifted fp16 fragments, in ONE launch: the three gathered copies,
// the two bf16 splits, the range shift (the last block).  kt_repack_split then reads the shift from memory.
struct RepackAll {
    const int* map_v; float* dst_v; int n_v;
    const int* map_m; float* dst_m; int n_m;
    const int* map_mb; float* dst_mb; int n_mb;
    uint16_t* out6; uint16_t* out6b;
    int forced_shift; int* shift_out;
};
#define KT_REPACK_ALL_BLOCKS(nv, nm, nmb) (((nv) + 255) / 256 + ((nm) + 255) / 256 + ((nmb) + 255) / 256 + 7 * 16 + 6 * 16 + 1)
__global__ void __launch_bounds__(256)
kt_repack_all(const float* __restrict__ w, RepackAll a) {
    int blk = (int)blockIdx.x;
    const int bv = (a.n_v + 255) / 256, bm = (a.n_m + 255) / 256, bmb = (a.n_mb + 255) / 256;
    if (blk < bv) { repack_gather_block(w, a.map_v, a.dst_v, a.n_v, blk); return; }
    blk -= bv;
    if (blk < bm) { repack_gather_block(w, a.map_m, a.dst_m, a.n_m, blk); return; }
    blk -= bm;
    if (blk < bmb) { repack_gather_block(w, a.map_mb, a.dst_mb, a.n_mb, blk); return; }
    blk -= bmb;
    if (blk < 7 * 16) { repack_split6_block(w, a.out6, blk); return; }
    blk -= 7 * 16;
    if (blk < 6 * 16) { repack_split6_bwd_block(w, a.out6b, blk); return; }
    split_range_shift_block(w, a.forced_shift, a.shift_out);
}
