// Particle extraction from a depth image: env/flex_env.py:933-951
//   depth2fgpcd (utils.py:491-506) -> open3d voxel_down_sample (utils.py:533-544)
//   -> dgl farthest_point_sampler + particle_r (utils.py:423-436) -> recenter (utils.py:468-477)
// The reference does this in float64 numpy (float32 only inside the sampler and for the gathered
// samples); the kernels keep those types and the same evaluation order, so the outputs are
// bit-identical to the restatement in oracle/particles.py.  No FMA contraction (-ffp-contract=off).
#pragma once
#include "drp_common.h"

#define PX_BLOCK 256
#define PX_PER_THREAD 4
#define PX_TILE (PX_BLOCK * PX_PER_THREAD)

struct PxGrid {            // written by k_px_bounds, read by the host and by the voxel kernels
    double vmin[3];        // voxel_min_bound = min_bound - voxel/2
    int dims[3];
    int n;                 // number of foreground points
    long long cells;
};

__device__ __forceinline__ bool px_fg(const float* __restrict__ depth, const uint8_t* __restrict__ mask, float gs,
                                      float thr, size_t i, float& d) {
    d = __fdiv_rn(depth[i], gs);                      // env/flex_env.py:941 (float32 / scale)
    const bool m = mask ? (mask[i] != 0) : (d < thr); // :945 depth < 0.599/0.8 (float32 compare)
    return m && d > 0.0f;                             // utils.py:496
}

// exclusive scan of one int per thread over a PX_BLOCK-thread block; total in `total`
__device__ __forceinline__ int px_block_scan(int v, int& total) {
    __shared__ int s_w[PX_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < PX_BLOCK / 64; ++w) {
        if (w < wave) base += s_w[w];
        total += s_w[w];
    }
    return base + inc - v;
}

__global__ void __launch_bounds__(PX_BLOCK)
k_px_count(const float* __restrict__ depth, const uint8_t* __restrict__ mask, float gs, float thr, size_t npix,
           unsigned long long* __restrict__ blk_cnt) {
    const size_t base = (size_t)blockIdx.x * PX_TILE + (size_t)threadIdx.x * PX_PER_THREAD;
    int c = 0;
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q) {
        float d;
        if (base + q < npix && px_fg(depth, mask, gs, thr, base + q, d)) ++c;
    }
    int total;
    (void)px_block_scan(c, total);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = (unsigned long long)total;
}

// single block: exclusive scan of n 64-bit values (two packed 32-bit sums never carry into each
// other below 2^32 items); total written to out[n]
__global__ void __launch_bounds__(1024)
k_px_scan_u64(const unsigned long long* __restrict__ in, int n, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const unsigned long long v = i < n ? in[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned long long pre = s_carry, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) pre += s_w[w];
            tot += s_w[w];
        }
        if (i < n) out[i] = pre + inc - v;
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0) out[n] = s_carry;
}

// row-major compaction of the foreground pixels into camera-frame float64 points
// (utils.py:498-505) + per-block bounds
__global__ void __launch_bounds__(PX_BLOCK)
k_px_compact(const float* __restrict__ depth, const uint8_t* __restrict__ mask, float gs, float thr, int w,
             size_t npix, double fx, double fy, double cx, double cy,
             const unsigned long long* __restrict__ blk_off, double* __restrict__ pcd,
             double* __restrict__ blk_min, double* __restrict__ blk_max) {
    __shared__ double s_red[PX_BLOCK / 64][6];
    const size_t base = (size_t)blockIdx.x * PX_TILE + (size_t)threadIdx.x * PX_PER_THREAD;
    float dv[PX_PER_THREAD];
    bool fg[PX_PER_THREAD];
    int c = 0;
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q) {
        fg[q] = base + q < npix && px_fg(depth, mask, gs, thr, base + q, dv[q]);
        c += fg[q] ? 1 : 0;
    }
    int total;
    size_t pos = (size_t)blk_off[blockIdx.x] + (size_t)px_block_scan(c, total);
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q) {
        if (!fg[q]) continue;
        const size_t i = base + q;
        const int py = (int)(i / (size_t)w), px = (int)(i - (size_t)py * w);
        const double d = (double)dv[q];
        double p[3];
        p[0] = (((double)px - cx) * d) / fx;
        p[1] = (((double)py - cy) * d) / fy;
        p[2] = d;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            pcd[pos * 3 + a] = p[a];
            mn[a] = fmin(mn[a], p[a]);
            mx[a] = fmax(mx[a], p[a]);
        }
        ++pos;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fmin(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmax(mx[a], __shfl_xor(mx[a], off, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int a = 0; a < 3; ++a) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a0 = s_red[0][threadIdx.x], b0 = s_red[0][3 + threadIdx.x];
        for (int wv = 1; wv < PX_BLOCK / 64; ++wv) {
            a0 = fmin(a0, s_red[wv][threadIdx.x]);
            b0 = fmax(b0, s_red[wv][3 + threadIdx.x]);
        }
        blk_min[(size_t)blockIdx.x * 3 + threadIdx.x] = a0;
        blk_max[(size_t)blockIdx.x * 3 + threadIdx.x] = b0;
    }
}

// bounds of a point list that did not come from k_px_compact (stand-alone downsample_pcd)
__global__ void __launch_bounds__(PX_BLOCK)
k_px_point_bounds(const double* __restrict__ pcd, int n, double* __restrict__ blk_min, double* __restrict__ blk_max) {
    __shared__ double s_red[PX_BLOCK / 64][6];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int q = 0; q < PX_PER_THREAD; ++q) {
        const size_t i = (size_t)blockIdx.x * PX_TILE + (size_t)q * PX_BLOCK + threadIdx.x;
        if (i < (size_t)n)
            for (int a = 0; a < 3; ++a) {
                mn[a] = fmin(mn[a], pcd[i * 3 + a]);
                mx[a] = fmax(mx[a], pcd[i * 3 + a]);
            }
    }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fmin(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmax(mx[a], __shfl_xor(mx[a], off, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int a = 0; a < 3; ++a) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a0 = s_red[0][threadIdx.x], b0 = s_red[0][3 + threadIdx.x];
        for (int wv = 1; wv < PX_BLOCK / 64; ++wv) {
            a0 = fmin(a0, s_red[wv][threadIdx.x]);
            b0 = fmax(b0, s_red[wv][3 + threadIdx.x]);
        }
        blk_min[(size_t)blockIdx.x * 3 + threadIdx.x] = a0;
        blk_max[(size_t)blockIdx.x * 3 + threadIdx.x] = b0;
    }
}

// open3d PointCloud::VoxelDownSample: voxel_min_bound = min_bound - voxel/2; the grid extent
// follows from the largest voxel index any point can take
__global__ void k_px_bounds(const double* __restrict__ blk_min, const double* __restrict__ blk_max, int nblk, int n,
                            double voxel, PxGrid* __restrict__ g) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        double mn = INFINITY, mx = -INFINITY;
        for (int b = 0; b < nblk; ++b) {
            mn = fmin(mn, blk_min[(size_t)b * 3 + a]);
            mx = fmax(mx, blk_max[(size_t)b * 3 + a]);
        }
        const double vmin = mn - voxel * 0.5;
        g->vmin[a] = vmin;
        const double top = floor((mx - vmin) / voxel);
        const int d = n > 0 ? (top < 16777215.0 ? (int)top + 1 : 16777216) : 0;
        g->dims[a] = d;
        cells *= d;
    }
    g->n = n;
    g->cells = cells;
}

__device__ __forceinline__ long long px_cell_key(const double* __restrict__ p, const PxGrid* __restrict__ g, double voxel) {
    int id[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) id[a] = (int)floor((p[a] - g->vmin[a]) / voxel);
    return ((long long)id[0] * g->dims[1] + id[1]) * (long long)g->dims[2] + id[2];
}

__global__ void __launch_bounds__(256)
k_px_cell_count(const double* __restrict__ pcd, int n, double voxel, const PxGrid* __restrict__ g,
                int* __restrict__ keys, int* __restrict__ cell_cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int key = (int)px_cell_key(pcd + (size_t)i * 3, g, voxel);
    keys[i] = key;
    atomicAdd(&cell_cnt[key], 1);
}

__device__ __forceinline__ unsigned long long px_pack(int cnt) {
    return cnt ? ((1ull << 32) | (unsigned long long)(unsigned)cnt) : 0ull;
}

__global__ void __launch_bounds__(PX_BLOCK)
k_px_cell_blocksum(const int* __restrict__ cell_cnt, long long cells, unsigned long long* __restrict__ blk_sum) {
    __shared__ unsigned long long s_w[PX_BLOCK / 64];
    const long long base = (long long)blockIdx.x * PX_TILE + (long long)threadIdx.x * PX_PER_THREAD;
    unsigned long long v = 0;
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q)
        if (base + q < cells) v += px_pack(cell_cnt[base + q]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < PX_BLOCK / 64; ++w) t += s_w[w];
        blk_sum[blockIdx.x] = t;
    }
}

// per-cell exclusive prefix: high word = output slot of the voxel, low word = start of its point list
__global__ void __launch_bounds__(PX_BLOCK)
k_px_cell_offsets(const int* __restrict__ cell_cnt, long long cells, const unsigned long long* __restrict__ blk_off,
                  unsigned long long* __restrict__ cell_off) {
    __shared__ unsigned long long s_w[PX_BLOCK / 64];
    const long long base = (long long)blockIdx.x * PX_TILE + (long long)threadIdx.x * PX_PER_THREAD;
    unsigned long long pv[PX_PER_THREAD];
    unsigned long long v = 0;
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q) {
        pv[q] = base + q < cells ? px_pack(cell_cnt[base + q]) : 0ull;
        v += pv[q];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned long long pre = blk_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) pre += s_w[w];
    pre += inc - v;
#pragma unroll
    for (int q = 0; q < PX_PER_THREAD; ++q) {
        if (base + q < cells) cell_off[base + q] = pre;
        pre += pv[q];
    }
}

__global__ void __launch_bounds__(256)
k_px_cell_fill(const int* __restrict__ keys, int n, const unsigned long long* __restrict__ cell_off,
               int* __restrict__ cell_fill, int* __restrict__ list) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int key = keys[i];
    const unsigned lo = (unsigned)(cell_off[key] & 0xffffffffull);
    list[lo + (unsigned)atomicAdd(&cell_fill[key], 1)] = i;
}

// AccumulatedPoint: points of a voxel summed in index order, divided by their count
__global__ void __launch_bounds__(256)
k_px_voxel_mean(const int* __restrict__ cell_cnt, const unsigned long long* __restrict__ cell_off,
                int* __restrict__ list, const double* __restrict__ pcd, long long cells,
                double* __restrict__ down, float* __restrict__ down32) {
    const long long cell = (long long)blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    const int cnt = cell_cnt[cell];
    if (cnt == 0) return;
    const unsigned long long off = cell_off[cell];
    int* seg = list + (unsigned)(off & 0xffffffffull);
    for (int a = 1; a < cnt; ++a) {                    // the atomics filled the segment in any order
        const int v = seg[a];
        int b = a - 1;
        while (b >= 0 && seg[b] > v) { seg[b + 1] = seg[b]; --b; }
        seg[b + 1] = v;
    }
    double acc[3] = {0.0, 0.0, 0.0};
    for (int a = 0; a < cnt; ++a) {
        const double* p = pcd + (size_t)seg[a] * 3;
        acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
    }
    const size_t slot = (size_t)(off >> 32);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double m = acc[a] / (double)cnt;
        down[slot * 3 + a] = m;
        down32[slot * 3 + a] = (float)m;               // utils.py:426 .float()
    }
}

__global__ void k_px_to_f32(const double* __restrict__ in, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

__device__ __forceinline__ unsigned long long px_mix(unsigned long long z) {   // splitmix64
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// dgl.geometry.farthest_point_sampler (src/geometry/cpu/geometry_op_impl.cc), one workgroup per
// cloud of the batch: float32 squared distances summed x,y,z; running minimum; first maximum.
__global__ void __launch_bounds__(1024)
k_px_fps(const float* __restrict__ pts, int m, int npoints, const int* __restrict__ init_idx, unsigned long long seed,
         float* __restrict__ dist_all, int* __restrict__ chosen_all, float* __restrict__ out_pts) {
    __shared__ float sval[16];
    __shared__ int sidx[16];
    __shared__ int s_last;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* dist = dist_all + (size_t)b * m;
    int* chosen = chosen_all + (size_t)b * npoints;
    int last = init_idx ? init_idx[b] : (int)(px_mix(seed + (unsigned long long)b) % (unsigned long long)m);
    if (tid == 0) chosen[0] = last;
    for (int it = 0; it + 1 < npoints; ++it) {
        const float lx = pts[(size_t)last * 3], ly = pts[(size_t)last * 3 + 1], lz = pts[(size_t)last * 3 + 2];
        float best = -1.0f;
        int arg = 0;
        for (int i = tid; i < m; i += 1024) {
            const float dx = pts[(size_t)i * 3] - lx, dy = pts[(size_t)i * 3 + 1] - ly, dz = pts[(size_t)i * 3 + 2] - lz;
            const float one = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            float nd = one;
            if (it > 0) { const float od = dist[i]; nd = od > one ? one : od; }
            dist[i] = nd;
            if (nd > best) { best = nd; arg = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(arg, off, 64);
            if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
        }
        __syncthreads();
        if (lane == 0) { sval[wave] = best; sidx[wave] = arg; }
        __syncthreads();
        if (tid == 0) {
            float bv = sval[0];
            int bi = sidx[0];
            for (int w = 1; w < 16; ++w)
                if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
            s_last = bi;
            chosen[it + 1] = bi;
        }
        __syncthreads();
        last = s_last;
    }
    __syncthreads();
    for (int i = tid; i < npoints * 3; i += 1024) out_pts[(size_t)b * npoints * 3 + i] = pts[(size_t)chosen[i / 3] * 3 + i % 3];
}

__device__ __forceinline__ double px_sq(const double* __restrict__ p, const float* __restrict__ q) {
    const double dx = p[0] - (double)q[0], dy = p[1] - (double)q[1], dz = p[2] - (double)q[2];
    return (dx * dx + dy * dy) + dz * dz;              // np.linalg.norm: sequential sum of 3 squares
}

// particle_r = max over cloud points of the distance to the nearest sample (utils.py:433-435)
// and the recentering radius min(0.02, 0.5 * particle_r) (env/flex_env.py:949)
__global__ void __launch_bounds__(1024)
k_px_radius(const double* __restrict__ pcd, int m, const float* __restrict__ samples, int npoints,
            double* __restrict__ r_out, double* __restrict__ rr_out) {
    __shared__ double s_w[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sp = samples + (size_t)b * npoints * 3;
    double mx = 0.0;
    for (int i = tid; i < m; i += 1024) {
        double mn = INFINITY;
        for (int j = 0; j < npoints; ++j) mn = fmin(mn, px_sq(pcd + (size_t)i * 3, sp + (size_t)j * 3));
        mx = fmax(mx, mn);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) s_w[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) mx = fmax(mx, s_w[w]);
        const double r = sqrt(mx);
        r_out[b] = r;
        if (rr_out) rr_out[b] = fmin(0.02, 0.5 * r);
    }
}

// recenter (utils.py:468-477): one wavefront per sample; cloud points with |p - sample| < r are
// summed in index order, the mean is rounded to float32 (zeros_like(sampled_pcd))
__global__ void __launch_bounds__(256)
k_px_recenter(const double* __restrict__ pcd, int m, const float* __restrict__ samples, int npoints, int batch,
              const double* __restrict__ radius, float* __restrict__ out32, double* __restrict__ out64) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wid >= batch * npoints) return;
    const int b = wid / npoints;
    const float* q = samples + (size_t)wid * 3;
    const double r = radius[b];
    double acc[3] = {0.0, 0.0, 0.0};
    int cnt = 0;
    for (int base = 0; base < m; base += 64) {
        const int i = base + lane;
        bool in = false;
        if (i < m) in = sqrt(px_sq(pcd + (size_t)i * 3, q)) < r;
        unsigned long long mask = __ballot(in);
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const double* p = pcd + (size_t)(base + l) * 3;
            acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
            ++cnt;
        }
    }
    if (lane < 3) {
        const double sum = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : acc[2]);
        const double mean = sum / (double)cnt;   // 0/0 = NaN when nothing is within r, as numpy
        const float f = (float)mean;
        if (out32) out32[(size_t)wid * 3 + lane] = f;
        if (out64) out64[(size_t)wid * 3 + lane] = (double)f;
    }
}

// utils.py:438-449 fps_rad: farthest-point sampling of a float64 cloud until every point is within
// `radius` of a sample (the dataset's particle sampler, dataset/dataset_gnn_dyn.py:99).  One
// workgroup; distances as np.linalg.norm evaluates them in float64, first maximum as np.argmax.
__global__ void __launch_bounds__(1024)
k_px_fps_rad(const double* __restrict__ pcd, int n, double radius, int init_idx, int cap, double* __restrict__ dist,
             int* __restrict__ chosen, int* __restrict__ count_out) {
    __shared__ double sval[16];
    __shared__ int sidx[16];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int last = init_idx, count = 1;
    if (tid == 0) chosen[0] = init_idx;
    for (int it = 0;; ++it) {
        const double lx = pcd[(size_t)last * 3], ly = pcd[(size_t)last * 3 + 1], lz = pcd[(size_t)last * 3 + 2];
        double best = -1.0;
        int arg = 0x7fffffff;
        for (int i = tid; i < n; i += 1024) {
            const double dx = pcd[(size_t)i * 3] - lx, dy = pcd[(size_t)i * 3 + 1] - ly, dz = pcd[(size_t)i * 3 + 2] - lz;
            double nd = sqrt((dx * dx + dy * dy) + dz * dz);
            if (it > 0) nd = fmin(dist[i], nd);
            dist[i] = nd;
            if (nd > best) { best = nd; arg = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(arg, off, 64);
            if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
        }
        __syncthreads();
        if (lane == 0) { sval[wave] = best; sidx[wave] = arg; }
        __syncthreads();
        if (tid == 0) {
            double bv = sval[0];
            int bi = sidx[0];
            for (int w = 1; w < 16; ++w)
                if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
            s_last = (bv > radius && count < cap) ? bi : -1;      // while dist.max() > radius: append
            if (s_last >= 0) chosen[count] = bi;
        }
        __syncthreads();
        last = s_last;
        if (last < 0) break;
        ++count;
    }
    if (tid == 0) *count_out = count;
}
