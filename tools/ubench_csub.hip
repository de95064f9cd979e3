// Conditional-subtract formulations: time per wave-op per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V>
__device__ __forceinline__ uint32_t csub(uint32_t x, uint32_t m) {
    if (V == 0) { uint32_t t = x - m; return t < x ? t : x; }            // sub + min
    if (V == 1) {                                                         // sub with borrow-out + select
        uint32_t t, r;
        asm volatile("v_sub_co_u32 %0, vcc, %2, %3\n\tv_cndmask_b32 %1, %0, %2, vcc" : "=&v"(t), "=v"(r) : "v"(x), "v"(m) : "vcc");
        return r;
    }
    if (V == 2) return x >= m ? x - m : x;
    if (V == 3) { int32_t t = (int32_t)(x - m); return (uint32_t)t + (m & (uint32_t)(t >> 31)); }
    if (V == 6) { uint32_t t; bool b = __builtin_sub_overflow(x, m, &t); return b ? x : t; }
    return x;
}

template <int V>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t m, int iters) {
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x * 16 + i) * 2654435761u % m;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {   // add pairs then conditional subtract: 1 add + 1 csub per element
            uint32_t s = x[i] + x[i + 1];
            uint32_t d = x[i] - x[i + 1] + m;
            x[i] = csub<V>(s, m);
            x[i + 1] = csub<V>(d, m);
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V>
int run(const char* name, int wps, uint32_t* d_out) {
    int blocks = 256 * wps; const int iters = 4000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, 1996488706u, iters);
    CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, 1996488706u, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double sec = ms * 1e-3 / reps;
    double pairs = (double)iters * 8 * wps;  // per SIMD: (add, sub+add, 2 csub) groups
    printf("%-28s waves/SIMD=%d  ns per (add, sub, add, 2 csub) group per SIMD = %.3f\n", name, wps, sec * 1e9 / pairs);
    return 0;
}

int main() {
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    for (int w : {4, 8}) {
        run<0>("sub+min", w, d_out);
        run<1>("sub_co+cndmask (vcc)", w, d_out);
        run<2>("select (compiler)", w, d_out);
        run<3>("asr mask", w, d_out);
        run<6>("sub_overflow builtin", w, d_out);
        run<5>("no csub (3 ops only)", w, d_out);
    }
    return 0;
}
