// Does v_dot2c_f32_bf16 give the exact residual x - bf16(x)?  (candidate for the split of k_mlp_split.h)
//   hipcc --offload-arch=gfx950 -O3 tools/dot2c_test.hip -o tools/dot2c_test.bin && tools/dot2c_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k(const float* in, float* ref, float* got, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const float a = in[i], b = in[i + 1];
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;
    ref[i] = a - (float)ha;
    ref[i + 1] = b - (float)hb;
    const bf16x2 pk = __builtin_convertvector((f32x2){a, b}, bf16x2);
    unsigned pku;
    memcpy(&pku, &pk, 4);
    float la = a, lb = b;
    const unsigned m0 = 0x0000BF80u, m1 = 0xBF800000u;      // (-1, 0) and (0, -1) as bf16 pairs
    asm volatile("v_dot2c_f32_bf16 %0, %2, %3\n\tv_dot2c_f32_bf16 %1, %2, %4\n\ts_nop 2"
                 : "+v"(la), "+v"(lb) : "v"(pku), "v"(m0), "v"(m1));
    got[i] = la;
    got[i + 1] = lb;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 60 - 40;
        h[i] = ldexpf((float)rand() / RAND_MAX * 2.0f - 1.0f, e);
        if (i % 97 == 0) h[i] = 0.0f;
        if (i % 101 == 0) h[i] = ldexpf(1.0f, -130 + rand() % 10);     // denormals
    }
    float *d, *r, *g;
    hipMalloc(&d, n * 4); hipMalloc(&r, n * 4); hipMalloc(&g, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, d, r, g, n);
    std::vector<float> hr(n), hg(n);
    hipMemcpy(hr.data(), r, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hg.data(), g, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_normal = 0;
    for (int i = 0; i < n; ++i)
        if (memcmp(&hr[i], &hg[i], 4) != 0) {
            ++bad;
            if (fabsf(h[i]) > 1e-30f) { if (bad_normal < 5) printf("x=%.9g ref=%.9g got=%.9g\n", h[i], hr[i], hg[i]); ++bad_normal; }
        }
    printf("mismatches: %ld of %d (%ld with |x| > 1e-30)\n", bad, n, bad_normal);
    return 0;
}
