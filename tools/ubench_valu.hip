// Instruction-throughput microbenchmark for the integer ops a Montgomery/Shoup
// modular multiply is built from. Prints cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;   // independent accumulators

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, uint32_t m) {
    uint32_t a[UNROLL];
    uint64_t a64[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + threadIdx.x * 7 + i; a64[i] = a[i]; d[i] = (double)a[i]; }
    uint32_t b = m | 1u;
    double db = (double)b * 1e-9;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a64[i]) : "v"(a[i]), "v"(b) : "vcc");
            if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(b));
            if (OP == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(a64[i]) : "v"(a[i]), "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 11) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 13) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            if (OP == 14) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
            if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if (OP == 16) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r += a[i] + (uint32_t)a64[i] + (uint32_t)(a64[i] >> 32) + (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, int waves_per_simd, double clk_ghz, uint32_t* d_out) {
    // one CU-filling configuration: blocks of 256 threads (4 waves -> 1 per SIMD)
    int blocks = 256 * waves_per_simd;  // 256 CUs
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 998244353u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 998244353u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double sec = ms * 1e-3 / reps;
    double insts_per_simd = (double)ITERS * UNROLL * waves_per_simd;  // wave-instructions per SIMD
    double cyc = sec * clk_ghz * 1e9 / insts_per_simd;
    printf("%-22s waves/SIMD=%d  time=%.3f ms  cycles/wave-inst/SIMD=%.2f (at %.2f GHz)\n", name, waves_per_simd, sec * 1e3, cyc, clk_ghz);
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    double clk = prop.clockRate * 1e-6;  // kHz -> GHz
    printf("device %s CUs=%d clock=%.3f GHz\n", prop.name, prop.multiProcessorCount, clk);
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    for (int w : {1, 2, 4}) {
        run<4>("v_add_u32", w, clk, d_out);
        run<11>("v_sub_u32", w, clk, d_out);
        run<5>("v_min_u32", w, clk, d_out);
        run<12>("v_add3_u32", w, clk, d_out);
        run<15>("v_cndmask_b32", w, clk, d_out);
        run<10>("v_fma_f32", w, clk, d_out);
        run<0>("v_mul_lo_u32", w, clk, d_out);
        run<8>("v_mul_lo_u32(sgpr)", w, clk, d_out);
        run<1>("v_mul_hi_u32", w, clk, d_out);
        run<2>("v_mad_u64_u32(acc)", w, clk, d_out);
        run<9>("v_mad_u64_u32(0)", w, clk, d_out);
        run<3>("v_mul_u32_u24", w, clk, d_out);
        run<16>("v_mul_hi_u32_u24", w, clk, d_out);
        run<7>("v_mad_u32_u24", w, clk, d_out);
        run<6>("v_fma_f64", w, clk, d_out);
        run<14>("v_mul_f64", w, clk, d_out);
        run<13>("v_cvt_f64_u32", w, clk, d_out);
    }
    hipFree(d_out);
    return 0;
}
