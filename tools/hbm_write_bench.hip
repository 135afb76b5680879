// Streaming-store ceiling of the chip, to price km_node_encode_split (315 MB of fp32 rows per launch):
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_write_bench.hip -o tools/hbm_write_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_fill(float4* __restrict__ p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(v, v, v, v);
}
// the accumulator-layout pattern of the encoder: a wave writes 32 rows x 32 B per instruction (8 instructions per 256-B row)
__global__ void __launch_bounds__(512) k_rows(float* __restrict__ p, size_t rows, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    for (size_t t = (size_t)blockIdx.x * 8 + wave; t * 32 < rows; t += (size_t)gridDim.x * 8) {
        float* row = p + (t * 32 + j) * 64;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(row + 32 * ob + 8 * g + 4 * h) = make_float4(v, v, v, v);
    }
}
int main() {
    const size_t bytes = 315ull << 20;
    float* d;
    (void)hipMalloc(&d, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                for (int it = 0; it < 10; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, (float4*)d, bytes / 16, 1.0f + it);
                    else hipLaunchKernelGGL(k_rows, dim3(grid / 2 > 0 ? grid / 2 : 1), dim3(512), 0, 0, d, bytes / 256, 1.0f + it);
                }
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) printf("%s grid %5d: %.2f TB/s (%.1f us per 315 MB)\n", mode ? "rows  " : "stream", grid, bytes * 10.0 / (ms * 1e-3) / 1e12, ms * 100.0);
            }
        }
    }
    return 0;
}
