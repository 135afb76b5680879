// Micro-benchmark of aggregate-kernel variants (development tool, not part of the library).
// hipcc --offload-arch=gfx950 -O3 -I../dyn_res_pile_manip_amd/csrc agg_bench.hip -o agg_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "k_aggregate.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// (b) stream only: no sender gather
__global__ void __launch_bounds__(512)
v_stream(const float* __restrict__ c_edge, const float* __restrict__ proj, const uint8_t* __restrict__ nbr_cnt, int N, float* __restrict__ agg) {
    const int b = blockIdx.x, q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const float4* ce = reinterpret_cast<const float4*>(c_edge) + (size_t)b * N * DRP_K * 16;
    const float4* pj = reinterpret_cast<const float4*>(proj) + (size_t)b * N * 32;
    float4* out = reinterpret_cast<float4*>(agg) + (size_t)b * N * 16;
    for (int i = g; i < N; i += 32) {
        const float4 pr = pj[(size_t)i * 32 + q];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 c[DRP_K];
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) c[k] = ce[((size_t)i * DRP_K + k) * 16 + q];
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) {
            acc.x += fmaxf(c[k].x + pr.x, 0.0f); acc.y += fmaxf(c[k].y + pr.y, 0.0f);
            acc.z += fmaxf(c[k].z + pr.z, 0.0f); acc.w += fmaxf(c[k].w + pr.w, 0.0f);
        }
        out[(size_t)i * 16 + q] = acc;
    }
}

// (c) lds gather, indices fetched once per receiver by lanes q<10 and broadcast; flat grid over
// receiver groups is impossible with LDS staging, so still block per sample; THREADS templated
template <int THREADS, bool PIPE>
__global__ void __launch_bounds__(THREADS)
v_lds2(const float* __restrict__ c_edge, const float* __restrict__ proj, const int16_t* __restrict__ nbr_idx,
       const uint8_t* __restrict__ nbr_cnt, int N, float* __restrict__ agg) {
    extern __shared__ __attribute__((aligned(16))) float4 ps[];
    const int b = blockIdx.x, q = threadIdx.x & 15, g = threadIdx.x >> 4;
    constexpr int G = THREADS / 16;
    const float4* ce = reinterpret_cast<const float4*>(c_edge) + (size_t)b * N * DRP_K * 16;
    const float4* pj = reinterpret_cast<const float4*>(proj) + (size_t)b * N * 32;
    float4* out = reinterpret_cast<float4*>(agg) + (size_t)b * N * 16;
    const int16_t* nb = nbr_idx + (size_t)b * N * DRP_K;
    const uint8_t* nc = nbr_cnt + (size_t)b * N;
    // issue the first receiver's stream loads before staging
    float4 c[DRP_K];
    int i = g;
    if (PIPE && i < N) {
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) c[k] = ce[((size_t)i * DRP_K + k) * 16 + q];
    }
    for (int idx = threadIdx.x; idx < N * 16; idx += THREADS)
        ps[idx] = pj[(size_t)(idx >> 4) * 32 + 16 + (idx & 15)];
    __syncthreads();
    for (; i < N; i += G) {
        const int cnt = nc[i];
        const int myj = (q < DRP_K) ? (int)nb[i * DRP_K + q] : 0;
        const float4 pr = pj[(size_t)i * 32 + q];
        if (!PIPE) {
#pragma unroll
            for (int k = 0; k < DRP_K; ++k) c[k] = ce[((size_t)i * DRP_K + k) * 16 + q];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < DRP_K; ++k) {
            int j = __shfl(myj, k, 16);
            j = (k < cnt) ? j : i;
            const float4 s = ps[j * 16 + q];
            if (k < cnt) {
                acc.x += fmaxf((c[k].x + pr.x) + s.x, 0.0f); acc.y += fmaxf((c[k].y + pr.y) + s.y, 0.0f);
                acc.z += fmaxf((c[k].z + pr.z) + s.z, 0.0f); acc.w += fmaxf((c[k].w + pr.w) + s.w, 0.0f);
            }
        }
        if (PIPE) {
            const int in = i + G;
            if (in < N) {
#pragma unroll
                for (int k = 0; k < DRP_K; ++k) c[k] = ce[((size_t)in * DRP_K + k) * 16 + q];
            }
        }
        out[(size_t)i * 16 + q] = acc;
    }
}

int main() {
    const int B = 1024, N = 300;
    const size_t bn = (size_t)B * N;
    float *c_edge, *proj, *agg; int16_t* idx; uint8_t* cnt;
    CK(hipMalloc(&c_edge, bn * 10 * 64 * 4)); CK(hipMalloc(&proj, bn * 128 * 4)); CK(hipMalloc(&agg, bn * 64 * 4));
    CK(hipMalloc(&idx, bn * 10 * 2)); CK(hipMalloc(&cnt, bn));
    std::vector<int16_t> hidx(bn * 10); std::vector<uint8_t> hcnt(bn, 10);
    for (size_t e = 0; e < bn * 10; ++e) hidx[e] = rand() % N;
    CK(hipMemcpy(idx, hidx.data(), bn * 20, hipMemcpyHostToDevice)); CK(hipMemcpy(cnt, hcnt.data(), bn, hipMemcpyHostToDevice));
    CK(hipMemset(c_edge, 0, bn * 10 * 64 * 4)); CK(hipMemset(proj, 0, bn * 128 * 4));
    CK(hipFuncSetAttribute((const void*)k_aggregate_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 600 * 256));
    CK(hipFuncSetAttribute((const void*)v_lds2<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 600 * 256));
    CK(hipFuncSetAttribute((const void*)v_lds2<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 600 * 256));
    CK(hipFuncSetAttribute((const void*)v_lds2<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 600 * 256));
    CK(hipFuncSetAttribute((const void*)v_lds2<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 600 * 256));
    CK(hipFuncSetAttribute((const void*)v_lds2<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 600 * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms;
        }
        CK(hipGetLastError());
        printf("%-28s best %.4f ms  mean %.4f ms  compulsory 1.022 GB -> %.2f TB/s\n", name, best, tot / 10, 1.022e9 / (best * 1e-3) / 1e12);
    };
    timeit("k_aggregate (global gather)", [&] { hipLaunchKernelGGL(k_aggregate, dim3(B), dim3(256), 0, 0, c_edge, proj, idx, cnt, N, agg, 1); });
    timeit("k_aggregate_lds", [&] { hipLaunchKernelGGL(k_aggregate_lds, dim3(B), dim3(512), N * 256, 0, c_edge, proj, idx, cnt, N, agg); });
    timeit("v_stream (no gather)", [&] { hipLaunchKernelGGL(v_stream, dim3(B), dim3(512), 0, 0, c_edge, proj, cnt, N, agg); });
    timeit("v_lds2<512,nopipe>", [&] { hipLaunchKernelGGL((v_lds2<512, false>), dim3(B), dim3(512), N * 256, 0, c_edge, proj, idx, cnt, N, agg); });
    timeit("v_lds2<512,pipe>", [&] { hipLaunchKernelGGL((v_lds2<512, true>), dim3(B), dim3(512), N * 256, 0, c_edge, proj, idx, cnt, N, agg); });
    timeit("v_lds2<1024,nopipe>", [&] { hipLaunchKernelGGL((v_lds2<1024, false>), dim3(B), dim3(1024), N * 256, 0, c_edge, proj, idx, cnt, N, agg); });
    timeit("v_lds2<1024,pipe>", [&] { hipLaunchKernelGGL((v_lds2<1024, true>), dim3(B), dim3(1024), N * 256, 0, c_edge, proj, idx, cnt, N, agg); });
    timeit("v_lds2<256,pipe>", [&] { hipLaunchKernelGGL((v_lds2<256, true>), dim3(B), dim3(256), N * 256, 0, c_edge, proj, idx, cnt, N, agg); });
    // plain copy of the same bytes for reference
    timeit("hipMemcpy D2D 786MB", [&] { (void)hipMemcpyAsync(c_edge, c_edge + bn * 5 * 64, bn * 5 * 64 * 4, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
