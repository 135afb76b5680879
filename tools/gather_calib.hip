// FETCH_SIZE calibration on a GATHER: random 256-B rows of a table far larger than the Infinity Cache, each row read once
// with 16-B-per-lane loads (sixteen lanes a row, four rows a wave instruction) -- the access pattern of km_prop3's sender-row
// gathers.  The byte count is known (rows x 256); run under `rocprofv3 --pmc FETCH_SIZE` and compare (tools/profile_r04.sh
// writes the ratio to profiles/r04_gather_calibration.txt).  A second kernel streams the same table in order (the pattern
// the guide's "FETCH_SIZE reports half of a wide coalesced read" was measured on) for the pair of numbers.
//   hipcc --offload-arch=gfx950 -O3 -o gather_calib tools/gather_calib.hip && ./gather_calib [table MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_gather_rows(const float4* __restrict__ table, const unsigned* __restrict__ rows, size_t n_rows, float* __restrict__ out) {
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;     // one 16-lane group per row
    const int l = threadIdx.x & 15;
    float acc = 0.0f;
    for (size_t r = g; r < n_rows; r += ((size_t)gridDim.x * blockDim.x) >> 4) {
        const float4 v = table[(size_t)rows[r] * 16 + l];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;          // keeps the loads
}

__global__ void k_stream_rows(const float4* __restrict__ table, size_t n_vec, float* __restrict__ out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = table[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1024;
    const size_t bytes = mib << 20, n_rows = bytes / 256;
    float4* table = nullptr; unsigned* rows = nullptr; float* out = nullptr;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&rows, n_rows * 4) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(table, 0, bytes);
    std::vector<unsigned> perm(n_rows);
    for (size_t i = 0; i < n_rows; ++i) perm[i] = (unsigned)i;
    unsigned long long s = 88172645463325252ull;
    for (size_t i = n_rows - 1; i > 0; --i) {      // Fisher-Yates with xorshift64: every row exactly once, in random order
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const size_t j = (size_t)(s % (i + 1));
        const unsigned t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    (void)hipMemcpy(rows, perm.data(), n_rows * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather_rows, dim3(256 * 8), dim3(256), 0, 0, table, rows, n_rows, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather_rows: %zu rows x 256 B = %.1f MiB in %.3f ms = %.2f TB/s\n", n_rows, bytes / 1048576.0, ms, bytes / ms / 1e9);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream_rows, dim3(256 * 8), dim3(256), 0, 0, table, bytes / 16, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_stream_rows: %.1f MiB in %.3f ms = %.2f TB/s\n", bytes / 1048576.0, ms, bytes / ms / 1e9);
    }
    printf("KNOWN_BYTES_PER_LAUNCH %zu\n", bytes);
    return 0;
}
