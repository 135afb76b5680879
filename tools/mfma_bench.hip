// Micro-benchmark: issue behaviour of v_mfma_f32_32x32x16_bf16 on gfx950 next to vector ALU work.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bench.hip -o gpurun_out/mfma_bench && gpurun_out/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed * 0.5f + i); }
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = seed; c1[i] = -seed; }
    float v[24];
    for (int i = 0; i < 24; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {           // 12 dependent MFMAs on one accumulator
#pragma unroll
            for (int q = 0; q < 12; ++q) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {    // 12 MFMAs alternating two accumulators
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
        } else if (MODE == 2) {    // 72 independent VALU fmas
#pragma unroll
            for (int q = 0; q < 72; ++q) v[q % 24] = fmaf(v[q % 24], 1.0001f, 0.5f);
        } else if (MODE == 3) {    // 12 dependent MFMAs interleaved with 72 independent VALU fmas (1 : 6)
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 6; ++u) v[(q * 6 + u) % 24] = fmaf(v[(q * 6 + u) % 24], 1.0001f, 0.5f);
            }
        } else if (MODE == 4) {    // MFMA chain, then VALU that depends on its result, then MFMA depending on the VALU
#pragma unroll
            for (int q = 0; q < 12; ++q) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = fmaf(c0[u], 1.0001f, v[u]);
#pragma unroll
            for (int u = 0; u < 56; ++u) v[u % 24] = fmaf(v[u % 24], 1.0001f, 0.5f);
            for (int i = 0; i < 8; ++i) b[i] = (__bf16)v[i];
        } else if (MODE == 6) {    // 72 v_dot2c_f32_bf16 on independent accumulators
#pragma unroll
            for (int q = 0; q < 72; ++q)
                asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[q % 24]) : "v"(__float_as_uint(v[(q + 7) % 24])), "v"(0xBF800000u));
        } else if (MODE == 7) {    // 36 v_cvt_pk_bf16_f32 (two values each)
#pragma unroll
            for (int q = 0; q < 36; ++q) {
                unsigned r;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[q % 24]), "v"(v[(q + 1) % 24]));
                v[q % 24] = __uint_as_float(r & 0xffff0000u);
            }
        } else if (MODE == 5) {    // 72 v_cvt_pk_bf16_f32-style conversions
#pragma unroll
            for (int q = 0; q < 72; ++q) { __bf16 t = (__bf16)v[q % 24]; v[q % 24] += (float)t; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 24; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += (float)b[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* d) {
    const int iters = 20000;
    const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU, waves spread over the 4 SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / iters;       // cycles per loop iteration at 2.4 GHz
    printf("%-58s waves/SIMD %d: %8.1f cycles per iteration (per SIMD)\n", name, waves_per_simd, cyc);
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    for (int w = 1; w <= 2; ++w) {
        run<0>("12 dependent MFMA 32x32x16 bf16, one accumulator", w, d);
        run<1>("12 MFMA alternating two accumulators", w, d);
        run<2>("72 independent v_fma_f32", w, d);
        run<3>("12 dependent MFMA interleaved with 72 independent v_fma", w, d);
        run<4>("12 MFMA -> 72 dependent v_fma -> next MFMA operand", w, d);
        run<5>("72 f32->bf16->f32 round trips", w, d);
        run<6>("72 v_dot2c_f32_bf16", w, d);
        run<7>("36 v_cvt_pk_bf16_f32 + 36 v_and", w, d);
    }
    return 0;
}
