// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal operands?  (needed by an fp16 split of fp32 values:
// the residual x - half(x) of an O(0.01..1) activation is an fp16 subnormal)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float bval, float aval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float tests[][2] = {{1.0f, 1.0f}, {1.0f, 3.0e-6f}, {1.0f, 5.96e-8f}, {3.0e-6f, 1.0f}, {2.0e-5f, 2.0e-5f}, {0.5f, 4.0e-5f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[1], t[0]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const float a = (float)(_Float16)t[0], b = (float)(_Float16)t[1];
        printf("a=%.4g b=%.4g (as fp16: %.6g, %.6g): mfma sum of 16 products = %.8g, expected %.8g\n", t[0], t[1], a, b, h, 16.0f * a * b);
    }
    return 0;
}
