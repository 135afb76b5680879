// What clock does a kernel see when only ONE workgroup is resident (the single-workgroup farthest-point sampler) compared with a
// full chip?  s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz: their ratio over a spin loop is the clock.
// build: hipcc --offload-arch=gfx950 -O3 -o clock_probe tools/clock_probe.hip ; run: ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, int iters) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 1e-9f;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    unsigned long long h[3];
    for (int grid : {1, 1, 32, 256, 2048, 1, 1}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(grid), dim3(512), 0, 0, d, 2000000);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("grid %5d: %llu shader cycles in %llu x 10 ns -> %.0f MHz\n", grid, h[0], h[1], (double)h[0] / ((double)h[1] * 1e-2));
        }
    }
    return 0;
}
