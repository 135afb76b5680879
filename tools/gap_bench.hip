// What fits in an MFMA gap on gfx950: shader cycles (s_memtime) per v_mfma_f32_32x32x16_f16 when K copies of
// one other instruction are issued after every MFMA, one or two waves per SIMD.  Operands are independent
// (four accumulators round robin, fillers write scratch registers), so this prices ISSUE, not latency.
//   hipcc --offload-arch=gfx950 -O3 tools/gap_bench.hip -o tools/gap_bench.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F_NONE, F_FMA, F_PKRTZ, F_MIXLO, F_MIXHI, F_MIX32, F_PKMAXI16, F_MAXI32, F_PKADD, F_ADD, F_DSREAD, F_MOV, F_NOP, F_CNT };
static const char* names[F_CNT] = {"(none)", "v_fma_f32", "v_cvt_pkrtz_f16_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_fma_mix_f32",
                                    "v_pk_max_i16", "v_max_i32", "v_pk_add_f32", "v_add_f32", "ds_read_b128", "v_mov_b32", "s_nop 0"};

template <int F>
__device__ __forceinline__ void filler(float& t0, float& t1, unsigned& u0, float x0, float x1, f32x2& p0, const f32x2& p1, f32x4& l, unsigned lds_addr) {
    if (F == F_FMA) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(t0) : "v"(x0), "v"(x1));
    if (F == F_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u0) : "v"(x0), "v"(x1));
    if (F == F_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "+v"(u0) : "v"(x0), "v"(x1));
    if (F == F_MIXHI) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(u0) : "v"(x0), "v"(x1));
    if (F == F_MIX32) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(x0), "v"(x1));
    if (F == F_PKMAXI16) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(u0) : "v"(x0));
    if (F == F_MAXI32) asm volatile("v_max_i32 %0, %1, 0" : "=v"(u0) : "v"(x0));
    if (F == F_PKADD) asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(p0) : "v"(p1));
    if (F == F_ADD) asm volatile("v_add_f32 %0, %1, %2" : "=v"(t0) : "v"(x0), "v"(x1));
    if (F == F_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"(lds_addr));
    if (F == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(t0) : "v"(x0));
    if (F == F_NOP) asm volatile("s_nop 0");
}

template <int F, int K>
__global__ void __launch_bounds__(512) kgap(unsigned long long* clk, float* sink, int iters) {
    __shared__ f32x4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)(0.01f * (lane + j)); B[j] = (_Float16)(0.02f * (lane - j)); }
    float t0 = 0, t1 = 0, x0 = 0.5f * lane, x1 = 1.0f + lane;
    unsigned u0 = 0;
    f32x2 p0 = {0, 0}, p1 = {x0, x1};
    f32x4 l = {0, 0, 0, 0};
    const unsigned lds_addr = (unsigned)(size_t)(&lds[(threadIdx.x * 1) & 2047]);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(A), "v"(B));
#pragma unroll
            for (int k = 0; k < K; ++k) filler<F>(t0, t1, u0, x0, x1, p0, p1, l, lds_addr);
        }
        if (F == F_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;
    float keep = t0 + t1 + p0[0] + l[0] + (float)u0;
    for (int a = 0; a < 4; ++a) keep += acc[a][0];
    if (keep == 123.456f) sink[0] = keep;
}

template <int F, int K>
void run(unsigned long long* dclk, float* sink) {
    const int iters = 2000;
    for (int threads = 256; threads <= 512; threads += 256) {
        const int nw = 256 * threads / 64;
        hipLaunchKernelGGL((kgap<F, K>), dim3(256), dim3(threads), 0, 0, dclk, sink, 100);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kgap<F, K>), dim3(256), dim3(threads), 0, 0, dclk, sink, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nw);
        (void)hipMemcpy(h.data(), dclk, nw * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[nw / 2] / (iters * 8.0);
        printf("%-22s x%d  waves/SIMD %d: %6.1f shader cycles per MFMA per wave  (%.2f GHz)\n", names[F], K, threads / 256, cyc,
               (double)h[nw / 2] / (ms * 1e6));
    }
}

int main() {
    unsigned long long* dclk; float* sink;
    (void)hipMalloc(&dclk, 8 * 4096); (void)hipMalloc(&sink, 4);
    run<F_NONE, 0>(dclk, sink);
    run<F_FMA, 4>(dclk, sink); run<F_FMA, 6>(dclk, sink); run<F_FMA, 8>(dclk, sink); run<F_FMA, 12>(dclk, sink);
    run<F_PKRTZ, 4>(dclk, sink); run<F_PKRTZ, 8>(dclk, sink);
    run<F_MIXLO, 4>(dclk, sink); run<F_MIXLO, 8>(dclk, sink);
    run<F_MIXHI, 4>(dclk, sink); run<F_MIXHI, 8>(dclk, sink);
    run<F_MIX32, 4>(dclk, sink); run<F_MIX32, 8>(dclk, sink);
    run<F_PKMAXI16, 4>(dclk, sink); run<F_PKMAXI16, 8>(dclk, sink);
    run<F_MAXI32, 4>(dclk, sink); run<F_MAXI32, 8>(dclk, sink);
    run<F_PKADD, 2>(dclk, sink); run<F_PKADD, 4>(dclk, sink); run<F_PKADD, 8>(dclk, sink);
    run<F_ADD, 4>(dclk, sink); run<F_ADD, 8>(dclk, sink);
    run<F_MOV, 8>(dclk, sink);
    run<F_NOP, 4>(dclk, sink);
    run<F_DSREAD, 1>(dclk, sink); run<F_DSREAD, 2>(dclk, sink); run<F_DSREAD, 3>(dclk, sink);
    return 0;
}
