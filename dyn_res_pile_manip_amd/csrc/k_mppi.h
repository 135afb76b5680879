// Sampling-MPC bookkeeping on the device: action sampling and the softmax-weighted
// update, written so the sample axis can be sharded over GPUs.
//
//   sample_action_sequences   planners.py:69-190  (noise_type 'normal', 'uniform', 'total_rand')
//   optimize_action           planners.py:549-561
//
// Both are dead code in the reference (nothing calls them; the live planner is gradient
// descent) -- they are implemented from their definitions.
#pragma once
#include "drp_common.h"

// ---- Philox4x32-10 ------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32(uint32_t (&c)[4], uint64_t key) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;   // (0,1]
    const float u2 = (float)b * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    n0 = r * c;
    n1 = r * s;
}

// One thread per sample: temporally filtered Gaussian residual added to the nominal
// sequence, clipped to the action box; written to all n_batch rows of the sample
// (row = sample * n_batch + batch, planners.py:661-662).
//   noise: null -> Philox draws keyed by (seed; global sample, t, iteration),
//          else [n_sample,H,4] draws from the host (standard normal / U(-1,1) / U[0,1) by noise_type).
//   noise_type (planners.py:116-135,169-175): 0 'normal' N(0, sigma); 1 'uniform' U(-sigma, sigma);
//          2 'total_rand': no residual, the push is drawn uniformly from the clip box.
__global__ void k_mppi_sample(const double* __restrict__ nominal, const float* __restrict__ noise,
                              int n_sample, int n_batch, int H, double sigma, double beta, float4 lo,
                              float4 hi, uint64_t seed, uint64_t sample_offset, uint64_t iteration,
                              int noise_type, float* __restrict__ actions) {
    // four threads per sample, one per push component: the temporal filter runs along t only (the four lanes of a
    // sample evaluate the same Philox block -- one block yields the draws of all four components)
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = tid >> 2, c = tid & 3;
    if (s >= n_sample) return;
    const float lo_[4] = {lo.x, lo.y, lo.z, lo.w}, hi_[4] = {hi.x, hi.y, hi.z, hi.w};
    const double lc = (double)lo_[c], hc = (double)hi_[c];
    double resid = 0.0;
    for (int t = 0; t < H; ++t) {
        float n;
        if (noise != nullptr) {
            n = noise[((size_t)s * H + t) * 4 + c];
        } else {
            const uint64_t gs = sample_offset + (uint64_t)s;
            uint32_t ctr[4] = {(uint32_t)gs, (uint32_t)(gs >> 32), (uint32_t)t, (uint32_t)iteration};
            philox4x32(ctr, seed);
            if (noise_type == 0) {
                float n0, n1;
                box_muller(ctr[c & 2], ctr[(c & 2) + 1], n0, n1);
                n = (c & 1) ? n1 : n0;
            } else {
                const float u = (float)(ctr[c] >> 8) * 5.9604644775390625e-08f;      // [0,1), 24 bits
                n = (noise_type == 1) ? 2.0f * u - 1.0f : u;
            }
        }
        const double z = (noise_type == 2) ? 0.0 : sigma * (double)n;
        resid = beta * z + resid * (1.0 - beta);
        double a = nominal[t * 4 + c] + resid;
        a = fmin(fmax(a, lc), hc);
        if (noise_type == 2) a = lc + (double)n * (hc - lc);
        for (int j = 0; j < n_batch; ++j)
            actions[(((size_t)s * n_batch + j) * H + t) * 4 + c] = (float)a;
    }
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

__device__ __forceinline__ double block_max_d(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, red[w]);
    return t;
}

// Per-rank partial record of the softmax-weighted mean (layout documented in drp.h):
//   [0] m = max_s lambda r_s   [1] Z = sum_s exp(lambda r_s - m)   [2..2+4H) A = sum_s w_s act_s
//   [2+4H] sum r   [3+4H] sum r^2   [4+4H] max r   [5+4H] argmax (global sample index)
// r_s = mean over the n_batch columns of the sample's final-step reward.
// grid = 4H + 1 blocks of 256 threads; block j < 4H reduces A[j], the last block the rest.
__global__ void __launch_bounds__(256)
k_mppi_partials(const float* __restrict__ reward, int reward_stride, const float* __restrict__ actions,
                int n_sample, int n_batch, int H, double lambda, uint64_t sample_offset,
                double* __restrict__ out) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const int HJ = 4 * H;
    double mloc = -__builtin_inf();
    for (int s = threadIdx.x; s < n_sample; s += blockDim.x) {
        double r = 0.0;
        for (int c = 0; c < n_batch; ++c) r += (double)reward[((size_t)s * n_batch + c) * reward_stride];
        r /= (double)n_batch;
        mloc = fmax(mloc, lambda * r);
    }
    const double m = block_max_d(mloc, red);
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    double rmax = -__builtin_inf();
    int amax = 0;
    for (int s = threadIdx.x; s < n_sample; s += blockDim.x) {
        double r = 0.0;
        for (int c = 0; c < n_batch; ++c) r += (double)reward[((size_t)s * n_batch + c) * reward_stride];
        r /= (double)n_batch;
        const double w = exp(lambda * r - m);
        if (j < HJ) {
            acc0 += w * (double)actions[((size_t)s * n_batch) * HJ + j];
        } else {
            acc0 += w;
            acc1 += r;
            acc2 += r * r;
            if (r > rmax) { rmax = r; amax = s; }
        }
    }
    const double t0 = block_sum_d(acc0, red);
    if (j < HJ) {
        if (threadIdx.x == 0) out[2 + j] = t0;
        return;
    }
    const double t1 = block_sum_d(acc1, red);
    const double t2 = block_sum_d(acc2, red);
    const double gmax = block_max_d(rmax, red);
    // lowest sample index attaining the max (deterministic)
    double cand = (rmax == gmax) ? (double)amax : 1e300;
    cand = -block_max_d(-cand, red);
    if (threadIdx.x == 0) {
        out[0] = m;
        out[1] = t0;
        out[2 + HJ] = t1;
        out[3 + HJ] = t2;
        out[4 + HJ] = gmax;
        out[5 + HJ] = cand + (double)sample_offset;
    }
}

// Combine n_ranks partial records into the new nominal sequence (and global stats).  Rank g's record
// starts at partials + g * rank_stride (>= 6 + 4H doubles: the elite form of the exchange packs a rank's
// statistics record and its elite block into one all-gathered message).
//   stats_out: [0] mean r  [1] unbiased std r  [2] max r  [3] argmax  [4] Z  [5] m
__global__ void k_mppi_update(const double* __restrict__ partials, int n_ranks, int rank_stride, int H,
                              double n_sample_total, double* __restrict__ nominal,
                              double* __restrict__ stats_out) {
    const int HJ = 4 * H, REC = rank_stride;
    double m = -__builtin_inf();
    for (int g = 0; g < n_ranks; ++g) m = fmax(m, partials[(size_t)g * REC]);
    double Z = 0.0;
    for (int g = 0; g < n_ranks; ++g) Z += partials[(size_t)g * REC + 1] * exp(partials[(size_t)g * REC] - m);
    for (int j = threadIdx.x; j < HJ; j += blockDim.x) {
        double a = 0.0;
        for (int g = 0; g < n_ranks; ++g)
            a += partials[(size_t)g * REC + 2 + j] * exp(partials[(size_t)g * REC] - m);
        nominal[j] = a / Z;
    }
    if (threadIdx.x == 0) {
        double s1 = 0.0, s2 = 0.0, rmax = -__builtin_inf(), arg = 0.0;
        for (int g = 0; g < n_ranks; ++g) {
            const double* p = partials + (size_t)g * REC;
            s1 += p[2 + HJ];
            s2 += p[3 + HJ];
            if (p[4 + HJ] > rmax) { rmax = p[4 + HJ]; arg = p[5 + HJ]; }
        }
        const double mean = s1 / n_sample_total;
        const double var = (n_sample_total > 1.0) ? fmax((s2 - s1 * mean) / (n_sample_total - 1.0), 0.0) : 0.0;
        stats_out[0] = mean;
        stats_out[1] = sqrt(var);
        stats_out[2] = rmax;
        stats_out[3] = arg;
        stats_out[4] = Z;
        stats_out[5] = m;
    }
}


// ---- elite update (cross-entropy-method style) ------------------------------------------------
// Not in the reference (planners.py has neither CEM nor a working sampling planner; SURVEY.md section 8e
// names it as the other form of the one exchange): the new nominal sequence is the MEAN of the k best
// samples' sequences.  Order: higher reward first, ties to the lower global sample index, so every rank
// count gives the same elite.  A rank contributes its k best as records
//     [reward, global sample index, act[4H]]            (2 + 4H doubles each, best first;
//                                                        reward = -inf pads a rank with fewer samples)
// and the combine picks the k best of all ranks' records.  Selection = k rounds of "best key after the last
// pick" over <= a few thousand keys: no sort, no scratch.
__device__ __forceinline__ bool elite_before(double ra, double ia, double rb, double ib) {
    return ra > rb || (ra == rb && ia < ib);
}

// best (r, i) over a wave; every lane returns it, and `owner` = a lane that holds it
__device__ __forceinline__ void wave_best(double& r, double& i, int& owner) {
    owner = threadIdx.x & 63;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double r2 = __shfl_xor(r, o, 64), i2 = __shfl_xor(i, o, 64);
        const int w2 = __shfl_xor(owner, o, 64);
        if (elite_before(r2, i2, r, i)) { r = r2; i = i2; owner = w2; }
    }
}

// k picks out of n keys (key[q] = reward, idx[q] = global index or < 0 for padding) held in LDS, by ONE wave:
// every lane caches the best of its strided share, a round is one shuffle reduction, and only the lane that owned
// the pick rescans its share.  The picked positions go to pick[0..k) (or -1), in order.  Destroys key[].
__device__ __forceinline__ void wave_select(double* key, const double* idx, int n, int k, int* pick) {
    const int lane = threadIdx.x & 63;
    auto rescan = [&](double& br, double& bi, int& bq) {
        br = -__builtin_inf(); bi = 1e300; bq = -1;
        for (int q = lane; q < n; q += 64) {
            const double r = key[q], gi = idx[q];
            if (gi >= 0.0 && r == r && elite_before(r, gi, br, bi)) { br = r; bi = gi; bq = q; }
        }
    };
    double br, bi;
    int bq;
    rescan(br, bi, bq);
    for (int e = 0; e < k; ++e) {
        double r = br, i = bi;
        int owner;
        wave_best(r, i, owner);
        const bool found = i < 1e299;
        const int q = __shfl(bq, owner, 64);
        if (lane == 0) pick[e] = found ? q : -1;
        if (found && lane == owner) {
            key[bq] = __builtin_nan("");         // taken
            rescan(br, bi, bq);
        }
    }
}

// all keys best first, by the whole workgroup: bitonic network over n2 = 2^m >= n entries in LDS (padding: idx < 0).
// src[] travels with the keys (the entry's original position).  ~m(m+1)/2 barriers instead of k dependent rounds.
__device__ __forceinline__ bool elite_before_pad(double ra, double ia, double rb, double ib) {
    // padding (idx < 0) and NaN rewards sort last
    const bool va = ia >= 0.0 && ra == ra, vb = ib >= 0.0 && rb == rb;
    if (va != vb) return va;
    if (!va) return false;
    return elite_before(ra, ia, rb, ib);
}
__device__ __forceinline__ void block_sort_best_first(double* key, double* idx, int* src, int n2) {
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const int a = 2 * t - (t & (stride - 1)), b = a + stride;
                const bool up = (a & size) == 0;
                const double ra = key[a], ia = idx[a], rb = key[b], ib = idx[b];
                const bool swap = up ? elite_before_pad(rb, ib, ra, ia) : elite_before_pad(ra, ia, rb, ib);
                if (swap) {
                    key[a] = rb; idx[a] = ib; key[b] = ra; idx[b] = ia;
                    const int sa = src[a]; src[a] = src[b]; src[b] = sa;
                }
            }
        }
    __syncthreads();
}

// this rank's k best samples as records [reward, global index, act[4H]]; dynamic LDS: 2 * n_sample doubles + k ints
__global__ void __launch_bounds__(256)
k_elite_local(const float* __restrict__ reward, int reward_stride, const float* __restrict__ actions,
              int n_sample, int n_batch, int H, int k, uint64_t sample_offset, int n2 /* 2^m >= n_sample: sort path; 0: k rounds */,
              double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double el_lds[];
    const int cap = n2 > 0 ? n2 : n_sample;
    double* key = el_lds;
    double* idx = key + cap;
    int* pick = reinterpret_cast<int*>(idx + cap);
    const int HJ = 4 * H, REC = 2 + HJ;
    for (int s = threadIdx.x; s < n_sample; s += blockDim.x) {
        double r = 0.0;
        for (int c = 0; c < n_batch; ++c) r += (double)reward[((size_t)s * n_batch + c) * reward_stride];
        key[s] = r / (double)n_batch;
        idx[s] = (double)s + (double)sample_offset;
    }
    if (n2 > 0) {
        // sort path: LDS holds n2 keys, indices and positions (the host picks it when they fit)
        int* src = reinterpret_cast<int*>(idx + n2);
        for (int s = threadIdx.x; s < n2; s += blockDim.x) {
            src[s] = s;
            if (s >= n_sample) { key[s] = 0.0; idx[s] = -1.0; }
        }
        block_sort_best_first(key, idx, src, n2);
        for (int e = threadIdx.x; e < k; e += blockDim.x) {
            const bool ok = e < n2 && idx[e] >= 0.0 && key[e] == key[e];
            out[(size_t)e * REC] = ok ? key[e] : -__builtin_inf();
            out[(size_t)e * REC + 1] = ok ? idx[e] : -1.0;
        }
        for (int t = threadIdx.x; t < k * HJ; t += blockDim.x) {
            const int e = t / HJ, j = t - e * HJ;
            const bool ok = e < n2 && idx[e] >= 0.0 && key[e] == key[e];
            out[(size_t)e * REC + 2 + j] = ok ? (double)actions[((size_t)src[e] * n_batch) * HJ + j] : 0.0;
        }
        return;
    }
    __syncthreads();
    // the selection consumes key[]: keep the rewards of the picks
    if (threadIdx.x < 64) wave_select(key, idx, n_sample, k, pick);
    __syncthreads();
    for (int e = threadIdx.x; e < k; e += blockDim.x) {
        const int s = pick[e];
        double r = 0.0;
        if (s >= 0) {
            for (int c = 0; c < n_batch; ++c) r += (double)reward[((size_t)s * n_batch + c) * reward_stride];
            r /= (double)n_batch;
        }
        out[(size_t)e * REC] = (s >= 0) ? r : -__builtin_inf();
        out[(size_t)e * REC + 1] = (s >= 0) ? idx[s] : -1.0;
    }
    for (int t = threadIdx.x; t < k * HJ; t += blockDim.x) {       // all picks' sequences at once
        const int e = t / HJ, j = t - e * HJ, s = pick[e];
        out[(size_t)e * REC + 2 + j] = (s >= 0) ? (double)actions[((size_t)s * n_batch) * HJ + j] : 0.0;
    }
}

// records [n_ranks][k][2+4H] (rank g's block at recs + g * rank_stride doubles) -> nominal = mean of the k best
// sequences (summed in pick order);
// elite_out: [0] elite size, [1] worst elite reward.  dynamic LDS: max(2 * n_ranks * k, k * 4H) doubles + k ints
__global__ void __launch_bounds__(256)
k_elite_update(const double* __restrict__ recs, int n_ranks, int rank_stride, int k, int H, int n2 /* 2^m >= n_ranks * k: sort path; 0: k rounds */,
               double* __restrict__ nominal, double* __restrict__ elite_out) {
    extern __shared__ __attribute__((aligned(16))) double el_lds[];
    const int HJ = 4 * H, REC = 2 + HJ, total = n_ranks * k;
    auto rec_at = [&](int q) { return recs + (size_t)(q / k) * rank_stride + (size_t)(q % k) * REC; };
    double* key = el_lds;
    double* idx = key + (n2 > 0 ? n2 : total);
    int* pick = reinterpret_cast<int*>(el_lds + max(2 * total, k * HJ));     // behind whichever use of el_lds is larger
    for (int q = threadIdx.x; q < total; q += blockDim.x) {
        key[q] = rec_at(q)[0];
        idx[q] = rec_at(q)[1];
    }
    if (n2 > 0) {
        int* src = reinterpret_cast<int*>(el_lds + 2 * n2);
        for (int q = threadIdx.x; q < n2; q += blockDim.x) {
            src[q] = q;
            if (q >= total) { key[q] = 0.0; idx[q] = -1.0; }
        }
        block_sort_best_first(key, idx, src, n2);
        int taken = 0;
        for (int e = 0; e < k && e < n2; ++e) taken += (idx[e] >= 0.0 && key[e] == key[e]) ? 1 : 0;
        double* seq = el_lds + 2 * n2 + (n2 + 1) / 2;             // behind keys, indices and positions
        for (int t = threadIdx.x; t < taken * HJ; t += blockDim.x) {
            const int e = t / HJ, j = t - e * HJ;
            seq[t] = rec_at(src[e])[2 + j];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < HJ; j += blockDim.x) {
            double acc = 0.0;
            for (int e = 0; e < taken; ++e) acc += seq[e * HJ + j];
            if (taken > 0) nominal[j] = acc / (double)taken;
        }
        if (threadIdx.x == 0 && elite_out != nullptr) {
            elite_out[0] = (double)taken;
            elite_out[1] = (taken > 0) ? key[taken - 1] : 0.0;
        }
        return;
    }
    __syncthreads();
    if (threadIdx.x < 64) wave_select(key, idx, total, k, pick);
    __syncthreads();
    int taken = 0;
    for (int e = 0; e < k; ++e) taken += (pick[e] >= 0) ? 1 : 0;
    // the picked sequences into LDS (all loads in flight at once), then one thread per action sums them in pick order
    double* seq = el_lds;                            // key[] and idx[] are dead
    __syncthreads();
    for (int t = threadIdx.x; t < k * HJ; t += blockDim.x) {
        const int e = t / HJ, j = t - e * HJ;
        seq[t] = (pick[e] >= 0) ? rec_at(pick[e])[2 + j] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HJ; j += blockDim.x) {
        double acc = 0.0;
        for (int e = 0; e < taken; ++e) acc += seq[e * HJ + j];
        if (taken > 0) nominal[j] = acc / (double)taken;
    }
    if (threadIdx.x == 0 && elite_out != nullptr) {
        elite_out[0] = (double)taken;
        elite_out[1] = (taken > 0) ? rec_at(pick[taken - 1])[0] : 0.0;
    }
}
