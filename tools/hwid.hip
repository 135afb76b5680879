// Where do workgroups and waves land?  Every wave records XCC_ID and HW_ID (SE, CU, SIMD) for its (block, wave).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid tools/hwid.hip && /tmp/hwid [blocks] [threads] [lds_bytes]
// Prints, for the first 64 blocks: block -> XCC, SE, CU, and the SIMD of each of its waves; then how many distinct
// (XCC, SE) pairs the blocks with id = k mod 32 use (1 for every k = the dispatch is static in both).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

__global__ void k_hwid(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that later ones cannot simply reuse its slot
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) lds[0] = a;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, threads = argc > 2 ? atoi(argv[2]) : 512;
    const int lds = argc > 3 ? atoi(argv[3]) : 150 * 1024;
    const int wpb = threads / 64;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 2 * blocks * wpb);
    hipFuncSetAttribute((const void*)k_hwid, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_hwid, dim3(blocks), dim3(threads), lds, 0, d, 20000);
    std::vector<unsigned> h(2 * blocks * wpb);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    auto se = [](unsigned hw) { return (hw >> 13) & 7; };
    auto cu = [](unsigned hw) { return (hw >> 8) & 15; };
    auto simd = [](unsigned hw) { return (hw >> 4) & 3; };
    for (int b = 0; b < blocks && b < 64; ++b) {
        printf("block %3d: xcc %u se %u cu %2u  simd of waves:", b, h[2 * b * wpb + 1] & 15, se(h[2 * b * wpb]), cu(h[2 * b * wpb]));
        for (int w = 0; w < wpb; ++w) printf(" %u", simd(h[2 * (b * wpb + w)]));
        printf("\n");
    }
    int worst = 0;
    for (int k = 0; k < 32; ++k) {
        std::set<unsigned> s;
        for (int b = k; b < blocks; b += 32) s.insert(((h[2 * b * wpb + 1] & 15) << 4) | se(h[2 * b * wpb]));
        if ((int)s.size() > worst) worst = (int)s.size();
    }
    printf("distinct (xcc, se) pairs among blocks of one id mod 32: at most %d\n", worst);
    int worst8 = 0;
    for (int k = 0; k < 8; ++k) {
        std::set<unsigned> s;
        for (int b = k; b < blocks; b += 8) s.insert(h[2 * b * wpb + 1] & 15);
        if ((int)s.size() > worst8) worst8 = (int)s.size();
    }
    printf("distinct xcc among blocks of one id mod 8: at most %d\n", worst8);
    return 0;
}
