// Butterfly throughput of the two 64-bit field flavours (ntt_amd/csrc/ntt_field64.h) as the compiler schedules them: a
// 16-point register transform of full butterflies (sum reduced, difference multiplied by a twiddle) repeated in a loop.
// Reports ns per wave-butterfly per SIMD and the butterflies per second of the whole chip (1024 SIMDs, 64 lanes).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ntt_amd/csrc tools/ubench_bfly64.hip -o tools/ubench_bfly64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#include "ntt_field64.h"

using namespace nttamd;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class F>
__global__ void __launch_bounds__(256) k(uint64_t* out, typename F::Consts fc, const uint64_t* tw, int iters) {
    const F f(fc);
    typename F::V x[16];
    for (int i = 0; i < 16; ++i) x[i] = f.from_mem(tw[8 + ((threadIdx.x + i) & 7)]);
    typename F::W w[8];
    for (int i = 0; i < 8; ++i) w[i] = f.tw_from_mem(tw[i]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int h = 8 >> s;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((i & h) == 0) f.dif(x[i], x[i + h], w[(i & (h - 1)) << s]);
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 16; ++i) r ^= f.to_mem(x[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class F>
int run(const char* name, int wps, uint64_t* d_out, const typename F::Consts& fc, const uint64_t* d_tw) {
    const int blocks = 256 * wps, iters = 1000, reps = 10;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256), 0, 0, d_out, fc, d_tw, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256), 0, 0, d_out, fc, d_tw, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double sec = ms * 1e-3 / reps, bf_per_simd = (double)iters * 32 * wps;
    printf("%-34s waves/SIMD=%d  ns/wave-butterfly/SIMD=%.3f  (=%.1f cyc @2.4GHz)  chip: %.3e butterflies/s\n", name, wps,
           sec * 1e9 / bf_per_simd, sec * 2.4e9 / bf_per_simd, 1024.0 * 64.0 * bf_per_simd / sec);
    return 0;
}

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((unsigned __int128)a * b % p); }

int main() {
    const uint64_t p = 1337006139375617ull;       // 19 * 2^46 + 1
    // double-precision flavour: balanced twiddles as doubles, data words value + p
    FieldDConsts fd{};
    fd.p = (double)p; fd.u = 1.0 / (double)p;
    uint64_t h_tw[16];
    for (int i = 0; i < 8; ++i) {
        const int64_t v = (int64_t)(mulmod(0x9E3779B97F4A7C15ull % p, 7919 * (i + 1), p)) - (int64_t)(p / 2);   // balanced
        const double d = (double)v;
        h_tw[i] = __builtin_bit_cast(uint64_t, d);
    }
    for (int i = 0; i < 8; ++i) h_tw[8 + i] = p / 2 + mulmod(12345 + i, 0xD1B54A32D192ED03ull % p, p);            // words in [p/2, 3p/2)
    uint64_t *d_out, *d_tw;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 8)); CHECK(hipMalloc(&d_tw, sizeof(h_tw)));
    CHECK(hipMemcpy(d_tw, h_tw, sizeof(h_tw), hipMemcpyHostToDevice));
    for (int w : {2, 4, 5, 8}) if (run<FieldD>("FieldD (exact fp64) dif butterfly", w, d_out, fd, d_tw)) return 1;
    // integer flavour: Montgomery R = 2^64
    Field64Consts fi{};
    fi.p = p;
    uint64_t inv = p;                             // Newton: p^-1 mod 2^64
    for (int i = 0; i < 6; ++i) inv *= 2 - p * inv;
    fi.np = 0 - inv;
    fi.r1 = (uint64_t)(((unsigned __int128)1 << 64) % p);
    fi.r2 = mulmod(fi.r1, fi.r1, p);
    for (int i = 0; i < 8; ++i) h_tw[i] = mulmod(7919 * (i + 1), fi.r1, p);
    for (int i = 0; i < 8; ++i) h_tw[8 + i] = mulmod(12345 + i, 0xD1B54A32D192ED03ull % p, p);
    CHECK(hipMemcpy(d_tw, h_tw, sizeof(h_tw), hipMemcpyHostToDevice));
    for (int w : {2, 4, 8}) if (run<Field64>("Field64 (Montgomery 2^64) dif butterfly", w, d_out, fi, d_tw)) return 1;
    return 0;
}
