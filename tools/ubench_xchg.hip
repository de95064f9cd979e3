// Exchange microbenchmark: the 16 x 16 transposition between two radix-16 rounds of a transform (16 threads trade their 16
// registers so that thread i ends up with register i of everybody), (a) through LDS the way the 4096-point core does it
// (16 writes, barrier, 16 reads, conflict-free swizzle), (b) inside the wavefront with ds_bpermute shuffles and selects
// (four butterfly steps of 16 shuffles each: a register-indexed exchange has to go through log2(16) rounds because a lane
// cannot pick a register by a run-time index), (c) the same with DPP row rotations where the pattern allows it.
// Prints time per exchange per wave.   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_xchg.hip -o tools/ubench_xchg
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;

__device__ __forceinline__ void work(uint32_t (&v)[16], uint32_t c) {   // a little arithmetic between exchanges, so that nothing folds away
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = v[i] * 2654435761u + c + i;
}

// (a) LDS: slot (row, col) -> row * 16 + (col ^ row): writes by rows, reads by columns, both conflict-free
__global__ void __launch_bounds__(256) k_lds(uint32_t* out, uint32_t seed) {
    __shared__ uint32_t lds[4096];
    const int t = threadIdx.x, hi = t >> 4, lo = t & 15;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) {
        work(v, (uint32_t)it);
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[hi * 256 + lo * 16 + (i ^ lo)] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = lds[hi * 256 + i * 16 + (lo ^ i)];
        __syncthreads();
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

// (b) shuffles: transpose of a 16 x 16 block held by 16 consecutive lanes, butterfly steps d = 8, 4, 2, 1:
// lanes l and l^d swap the registers whose index has bit d different from the lane's bit d
__global__ void __launch_bounds__(256) k_shfl(uint32_t* out, uint32_t seed) {
    const int t = threadIdx.x, lo = t & 15;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) {
        work(v, (uint32_t)it);
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            const bool up = (lo & d) != 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((i & d) == 0) {
                    // pair (i, i+d): a lane with bit d clear keeps v[i] and trades v[i+d]; the partner the other way round
                    const uint32_t give = up ? v[i] : v[i + d];
                    const uint32_t got = (uint32_t)__shfl_xor((int)give, d, 64);
                    if (up) v[i] = got; else v[i + d] = got;
                }
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

template <class K>
int run(const char* name, K kern, uint32_t* d_out) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_launch = ms / reps;
    // waves per SIMD = blocks * 4 waves / (256 CUs * 4 SIMDs) = 8; each does ITERS exchanges
    const double ns_per_exchange_per_wave_per_simd = per_launch * 1e6 / ITERS / 8.0;
    printf("%-28s %.3f ms per launch  %.1f ns per exchange+work per wave per SIMD\n", name, per_launch, ns_per_exchange_per_wave_per_simd);
    return 0;
}

__global__ void __launch_bounds__(256) k_none(uint32_t* out, uint32_t seed) {
    const int t = threadIdx.x;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) work(v, (uint32_t)it);
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

int main() {
    uint32_t* d_out = nullptr;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    if (run("work only", k_none, d_out)) return 1;
    if (run("LDS exchange (core's way)", k_lds, d_out)) return 1;
    if (run("wave shuffles (bpermute)", k_shfl, d_out)) return 1;
    CHECK(hipFree(d_out));
    return 0;
}
