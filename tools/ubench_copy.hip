// Device-to-device copy rate by access width (4, 8, 16 bytes per lane) and by shape (contiguous, or 1 KiB row
// segments 16 KiB apart as in the column passes).  Prints read+write TB/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o tools/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T>
__global__ void __launch_bounds__(256) copy_flat(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) dst[i] = src[i];
}
// column-pass shape: thread j of a 256-column chunk reads ROWS values 4096 elements apart, writes them the same way
template <int ROWS>
__global__ void __launch_bounds__(256) copy_columns(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
    const size_t row = blockIdx.x / 16, j = (blockIdx.x % 16) * 256 + threadIdx.x;
    const uint32_t* s = src + row * (4096 * ROWS) + j;
    uint32_t* d = dst + row * (4096 * ROWS) + j;
    uint32_t x[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) x[r] = s[(size_t)r * 4096];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) d[(size_t)r * 4096] = x[r];
}

template <typename F>
int timeit(const char* name, size_t bytes, F launch) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 40; ++i) launch();   // past the clock transient
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %.2f TB/s (read+write)\n", name, 2.0 * bytes * reps / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    for (int grid : {4096, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "flat 4 B/lane grid %d", grid);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL(copy_flat<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, bytes / 4); });
        snprintf(nm, 64, "flat 8 B/lane grid %d", grid);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL(copy_flat<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)a, (uint2*)b, bytes / 8); });
        snprintf(nm, 64, "flat 16 B/lane grid %d", grid);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL(copy_flat<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16); });
    }
    timeit("columns 64 rows x 4 B (1 GiB)", bytes, [&] { hipLaunchKernelGGL(copy_columns<64>, dim3((unsigned)(bytes / 4 / (4096 * 64) * 16)), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b); });
    timeit("columns 32 rows x 4 B (1 GiB)", bytes, [&] { hipLaunchKernelGGL(copy_columns<32>, dim3((unsigned)(bytes / 4 / (4096 * 32) * 16)), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b); });
    CHECK(hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice));
    timeit("hipMemcpyAsync D2D", bytes, [&] { hipError_t e = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); (void)e; });
    return 0;
}
