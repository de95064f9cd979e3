// Butterfly-throughput microbenchmark: times candidate modular butterfly
// formulations as the compiler actually schedules them (16-point register
// transform repeated in a loop).  Reports cycles per wave-butterfly per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct F { uint32_t p, p2, ninv; };  // ninv = -p^-1 mod 2^32

// ---- variant 0: REDC via 64-bit C arithmetic, cond-sub via min ----
__device__ __forceinline__ uint32_t redc0(uint64_t T, const F& f) {
    uint32_t m = (uint32_t)T * f.ninv;
    return (uint32_t)((T + (uint64_t)m * f.p) >> 32);          // [0,2p)
}
__device__ __forceinline__ uint32_t csub_min(uint32_t x, uint32_t m) { uint32_t t = x - m; return t < x ? t : x; }  // min(x, x-m)
__device__ __forceinline__ uint32_t csub_sel(uint32_t x, uint32_t m) { return x >= m ? x - m : x; }
__device__ __forceinline__ uint32_t csub_asr(uint32_t x, uint32_t m) { int32_t t = (int32_t)(x - m); return (uint32_t)t + (m & (uint32_t)(t >> 31)); }

// 24 x 24 -> 48-bit product, low 32 bits and bits 32..47 (the masks fold into v_mul_u32_u24 / v_mul_hi_u32_u24, which read
// only the low 24 bits of their operands)
__device__ __forceinline__ uint32_t mul24_lo(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
__device__ __forceinline__ uint32_t mul24_hi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)) >> 32); }

template <int V>
__device__ __forceinline__ void bfly(uint32_t& u, uint32_t& v, uint32_t w, uint32_t w2, const F& f) {
    // DIF (Gentleman-Sande): u' = u+v mod, v' = (u-v)*w
    uint32_t s = u + v;
    uint32_t d = u - v + f.p2;
    if (V == 0) { u = csub_min(s, f.p2); v = redc0((uint64_t)d * w, f); }
    if (V == 1) { u = csub_sel(s, f.p2); v = redc0((uint64_t)d * w, f); }
    if (V == 2) { u = csub_asr(s, f.p2); v = redc0((uint64_t)d * w, f); }
    if (V == 3) {  // Shoup: w2 = floor(w*2^32/p)
        u = csub_min(s, f.p2);
        uint32_t q = __umulhi(d, w2);
        v = d * w - q * f.p;
    }
    if (V == 4) {  // Montgomery, precomputed w2 = w*pinv (positive inverse), subtractive form
        u = csub_min(s, f.p2);
        uint32_t m = d * w2;
        uint32_t hi = __umulhi(d, w);
        v = hi - __umulhi(m, f.p) + f.p;
    }
    if (V == 5) {  // no reduction of the sum at all (lower bound: mult cost only)
        u = s; v = redc0((uint64_t)d * w, f);
    }
    if (V == 6) {  // no multiply (add/sub/cond-sub cost only)
        u = csub_min(s, f.p2); v = csub_min(d, f.p2);
    }
    if (V == 7) {  // moduli below 2^23: Montgomery with R = 2^24 on the 24-bit multipliers (v_mul_u32_u24 / v_mul_hi_u32_u24)
        // f.ninv holds -p^-1 mod 2^24 here.  The difference is brought below 2^24 first (the multipliers take 24 bits).
        u = csub_min(s, f.p2);
        const uint32_t dr = csub_min(d, f.p2);
        const uint32_t t_lo = mul24_lo(dr, w);                 // bits 0..31 of the 48-bit product
        const uint32_t t_hi = mul24_hi(dr, w);               // bits 32..47
        const uint32_t m = mul24_lo(t_lo, f.ninv);             // only its low 24 bits are used below
        const uint32_t q_lo = mul24_lo(m, f.p), q_hi = mul24_hi(m, f.p);
        // (T + m p) >> 24: the low 24 bits cancel, a carry comes out of them exactly when they are not zero
        const uint32_t a = __builtin_amdgcn_alignbit(t_hi, t_lo, 24), b = __builtin_amdgcn_alignbit(q_hi, q_lo, 24);
        v = a + b + ((t_lo & 0xFFFFFFu) != 0 ? 1u : 0u);
    }
}

template <int V>
__global__ void __launch_bounds__(256) k(uint32_t* out, const uint32_t* tw, F f, int iters) {
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x * 16 + i) * 2654435761u % f.p;
    uint32_t w[8], w2[8];
    for (int i = 0; i < 8; ++i) { w[i] = tw[i]; w2[i] = tw[8 + i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int h = 8 >> s;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((i & h) == 0) {
                    const int e = (i & (h - 1)) << s;
                    bfly<V>(x[i], x[i + h], w[e], w2[e], f);
                }
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V>
int run(const char* name, int wps, uint32_t* d_out, uint32_t* d_tw, F f) {
    int blocks = 256 * wps;
    const int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_tw, f, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_tw, f, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double sec = ms * 1e-3 / reps;
    double bf_per_simd = (double)iters * 32 * wps;
    printf("%-28s waves/SIMD=%d time=%.3f ms  ns/wave-butterfly/SIMD=%.3f  (=%.2f cyc @2.4GHz, %.2f @2.0GHz)\n", name, wps, sec * 1e3,
           sec * 1e9 / bf_per_simd, sec * 2.4e9 / bf_per_simd, sec * 2.0e9 / bf_per_simd);
    return 0;
}

int main() {
    F f; f.p = 998244353u; f.p2 = 2 * f.p; f.ninv = 998244351u;
    uint32_t h_tw[16];
    for (int i = 0; i < 16; ++i) h_tw[i] = (i * 7919u + 12345u) % f.p;
    uint32_t *d_out, *d_tw;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4)); CHECK(hipMalloc(&d_tw, 64));
    CHECK(hipMemcpy(d_tw, h_tw, 64, hipMemcpyHostToDevice));
    for (int w : {2, 4, 8}) {
        run<0>("redc64 + csub_min", w, d_out, d_tw, f);
        run<1>("redc64 + csub_sel", w, d_out, d_tw, f);
        run<2>("redc64 + csub_asr", w, d_out, d_tw, f);
        run<3>("shoup + csub_min", w, d_out, d_tw, f);
        run<4>("mont-sub(w*pinv) + csub_min", w, d_out, d_tw, f);
        run<5>("redc64, no csub", w, d_out, d_tw, f);
        run<6>("no mult, 2 csub", w, d_out, d_tw, f);
    }
    {   // the 23-bit modulus of the reference's first data sets, R = 2^24
        F g; g.p = 7340033u; g.p2 = 2 * g.p;
        uint32_t inv = g.p;                                   // Newton: p^-1 mod 2^32, then negated and cut to 24 bits
        for (int i = 0; i < 5; ++i) inv *= 2u - g.p * inv;
        g.ninv = (0u - inv) & 0xFFFFFFu;
        uint32_t h2[16];
        for (int i = 0; i < 16; ++i) h2[i] = (i * 7919u + 12345u) % g.p;
        CHECK(hipMemcpy(d_tw, h2, 64, hipMemcpyHostToDevice));
        for (int w : {2, 4, 8}) {
            run<0>("p=7340033 redc64 + csub_min", w, d_out, d_tw, g);
            run<7>("p=7340033 u24 Montgomery", w, d_out, d_tw, g);
        }
    }
    return 0;
}
