// Kernels over the 64-bit prime field (ntt_field64.h): transforms of any power-of-two
// length along the middle axis of an [outer][len][inner] array, the building blocks of the
// four-step product of one very long polynomial pair (single GPU or sharded over ranks with
// an all-to-all transpose between the two transform axes).
//
// Forward = in-place decimation in frequency (natural order in, bit-reversed order out),
// inverse = the mirror (bit-reversed in, natural out, unscaled).  A length-2^l transform is a
// chain of strided radix-2^q register passes (q <= 4) from the largest block down; when
// inner == 1 the last 12 levels run inside LDS on contiguous 4096-value blocks.
//
// Replaces, for moduli that are themselves NTT-friendly, the reference's replicate-and-
// all-reduce distributed transform (main_mpi_t.cc:139-291) and extends the serial transform
// (main_naive.cc:96-137) beyond the 2-adicity of any 32-bit prime.
#pragma once
#include "ntt_field64.h"

namespace nttamd {

constexpr int LOCAL64_LOG = 12;  // log2 of the largest LDS-resident block (4096 x 8 B = 32 KiB)

// one strided pass: blocks of length Lb = cq << Q; thread = (block, column jq < cq, inner index c);
// the 2^Q rows jq + cq*r are transformed in registers, then multiplied by w_Lb^(jq * bitrev_Q(r)).
struct Pass64 {
    uint64_t cq;        // columns per block
    uint64_t inner;     // contiguous inner extent S
    uint64_t n_threads; // blocks * cq * inner
    uint64_t tw_stride; // (table root order) / Lb : exponent scale to the table's root
    uint64_t tw_mask;   // table root order - 1
};

template <int Q, bool INVERSE>
__global__ void __launch_bounds__(256) k64_strided(uint64_t* data, Pass64 g, Field64Consts fc, Twiddle64 tw) {
    constexpr int R = 1 << Q;
    const Field64 f(fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= g.n_threads) return;
    const uint64_t c = tid % g.inner;
    const uint64_t rest = tid / g.inner;
    const uint64_t jq = rest % g.cq, blk = rest / g.cq;
    uint64_t* base = data + ((blk * g.cq * R + jq) * g.inner + c);
    const uint64_t step = g.cq * g.inner;
    uint64_t x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = base[(uint64_t)r * step];
    if (!INVERSE) {
        // DIF stages, compile-time twiddles w_{2^Q}^e = w16^(e * 16/2^Q)
        static_for<0, Q>([&](auto S) {
            constexpr int s = decltype(S)::value;
            constexpr int h = R >> (s + 1);
            static_for<0, R>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr ((i & h) == 0) {
                    constexpr int e = ((i & (h - 1)) << s) << (4 - Q);
                    if constexpr (e == 0) f.dif1(x[i], x[i + h]);
                    else f.dif(x[i], x[i + h], fc.root16F[e]);
                }
            });
        });
        static_for<1, R>([&](auto P) {
            constexpr int pos = decltype(P)::value;
            constexpr int k = bitrev(pos, Q);
            const uint64_t e = (jq * (uint64_t)k * g.tw_stride) & g.tw_mask;
            x[pos] = f.mulw(x[pos], twiddle_pow(tw, f, e));
        });
    } else {
        static_for<1, R>([&](auto P) {
            constexpr int pos = decltype(P)::value;
            constexpr int k = bitrev(pos, Q);
            const uint64_t e = (0 - jq * (uint64_t)k * g.tw_stride) & g.tw_mask;  // negative exponent
            x[pos] = f.mulw(x[pos], twiddle_pow(tw, f, e));
        });
        static_for<0, Q>([&](auto S) {
            constexpr int s = Q - 1 - decltype(S)::value;
            constexpr int h = R >> (s + 1);
            static_for<0, R>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr ((i & h) == 0) {
                    constexpr int e = ((i & (h - 1)) << s) << (4 - Q);
                    if constexpr (e == 0) f.dit1(x[i], x[i + h]);
                    else f.dit(x[i], x[i + h], fc.root16I[e]);
                }
            });
        });
    }
#pragma unroll
    for (int r = 0; r < R; ++r) base[(uint64_t)r * step] = INVERSE ? x[r] : f.csub(x[r], f.B);
}

// contiguous blocks of 2^m <= 4096 values: all m stages in LDS.  Stage twiddles are powers of the
// 2^m-th root, i.e. table exponents that are multiples of (root order)/2^m -- one lookup each.
template <bool INVERSE>
__global__ void __launch_bounds__(256) k64_local(uint64_t* data, int m, uint64_t n_blocks, Field64Consts fc, Twiddle64 tw) {
    __shared__ uint64_t lds[1 << LOCAL64_LOG];
    const Field64 f(fc);
    const int N = 1 << m, half = N >> 1, t = threadIdx.x, T = blockDim.x;
    const uint64_t scale = (uint64_t)1 << (tw.logtab - m);
    const uint64_t mask = ((uint64_t)1 << tw.logtab) - 1;
    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        uint64_t* src = data + blk * (uint64_t)N;
        for (int i = t; i < N; i += T) lds[i] = src[i];
        __syncthreads();
        if (!INVERSE) {
            for (int s = 0; s < m; ++s) {
                const int h = half >> s;
                for (int k = t; k < half; k += T) {
                    const int j = k & (h - 1), i = ((k - j) << 1) + j;
                    const uint64_t e = ((uint64_t)(j << s) * scale) & mask;
                    uint64_t u = lds[i], v = lds[i + h];
                    if (e == 0) f.dif1(u, v); else f.dif(u, v, twiddle_pow(tw, f, e));
                    lds[i] = u; lds[i + h] = v;
                }
                __syncthreads();
            }
        } else {
            for (int s = m - 1; s >= 0; --s) {
                const int h = half >> s;
                for (int k = t; k < half; k += T) {
                    const int j = k & (h - 1), i = ((k - j) << 1) + j;
                    const uint64_t e = (0 - (uint64_t)(j << s) * scale) & mask;
                    uint64_t u = lds[i], v = lds[i + h];
                    if (e == 0) f.dit1(u, v); else f.dit(u, v, twiddle_pow(tw, f, e));
                    lds[i] = u; lds[i + h] = v;
                }
                __syncthreads();
            }
        }
        for (int i = t; i < N; i += T) src[i] = lds[i];
        __syncthreads();
    }
}

// dst[row][i] = i < n ? src[row][i] mod p : 0   (zero padding + reduction of arbitrary 64-bit input)
__global__ void __launch_bounds__(256) k64_load(const uint64_t* src, size_t src_stride, uint64_t n, uint64_t* dst, uint64_t len,
                                                uint64_t rows, Field64Consts fc) {
    const Field64 f(fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= rows * len) return;
    const uint64_t row = tid / len, i = tid % len;
    dst[tid] = i < n ? f.reduce(src[row * src_stride + i], fc.r2) : 0;
}

// a[i] = a[i] * b[i] * R^-1  (values < 4p in, [0,2p) out)
__global__ void __launch_bounds__(256) k64_pointwise(uint64_t* a, const uint64_t* b, uint64_t count, Field64Consts fc) {
    const Field64 f(fc);
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) a[i] = f.mulw(f.csub(a[i], f.B), f.csub(b[i], f.B));
}

// data[r][c] *= w^(+-(row0 + bitrev?(r)) * (col0 + c)) for a rows x cols tile of the four-step
// matrix; w = root of order 2^log_total.  ROWS_BITREV: row r of the tile holds frequency
// bitrev(row0 + r, row_bits) (the layout a forward transform along the row axis leaves).
__global__ void __launch_bounds__(256) k64_twiddle(uint64_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0,
                                                   int row_bits, int rows_bitrev, int inverse, uint64_t tw_stride,
                                                   uint64_t tw_mask, Field64Consts fc, Twiddle64 tw) {
    const Field64 f(fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= rows * cols) return;
    const uint64_t r = tid / cols, c = tid % cols;
    uint64_t k = row0 + r;
    if (rows_bitrev) {
        uint64_t rev = 0;
        for (int b = 0; b < row_bits; ++b) rev |= ((k >> b) & 1ull) << (row_bits - 1 - b);
        k = rev;
    }
    uint64_t e = (k * (col0 + c) * tw_stride) & tw_mask;
    if (inverse) e = (0 - e) & tw_mask;
    data[tid] = f.mulw(f.csub(data[tid], f.B), twiddle_pow(tw, f, e));
}

// in-place bit-reversal permutation of every row of 2^bits values (the reference's reverse(), main_naive.cc:82-95):
// thread i swaps with bitrev(i) when i < bitrev(i)
__global__ void __launch_bounds__(256) k64_bitrev(uint64_t* data, int bits, uint64_t rows) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t len = (uint64_t)1 << bits;
    if (tid >= rows * len) return;
    const uint64_t row = tid >> bits, i = tid & (len - 1);
    uint64_t rev = 0;
    for (int b = 0; b < bits; ++b) rev |= ((i >> b) & 1ull) << (bits - 1 - b);
    if (i < rev) {
        uint64_t* r = data + row * len;
        const uint64_t t = r[i];
        r[i] = r[rev];
        r[rev] = t;
    }
}

// out[row][i] = canonical(data[row][i] * scale) for i < n_out   (scale_m = Montgomery form of the scale)
__global__ void __launch_bounds__(256) k64_finalize(const uint64_t* data, uint64_t len, uint64_t* out, size_t out_stride,
                                                    uint64_t n_out, uint64_t rows, uint64_t scale_m, Field64Consts fc) {
    const Field64 f(fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= rows * n_out) return;
    const uint64_t row = tid / n_out, i = tid % n_out;
    out[row * out_stride + i] = f.csub(f.mulw(f.csub(data[row * len + i], f.B), scale_m), f.p);
}

// ---------------------------------------------------------------------------
// k64_garner: recombination of residues modulo two or three 59-bit NTT primes into the product coefficient
// modulo an arbitrary 64-bit p (mixed-radix / Garner form, all arithmetic in 64-bit words):
//   x = t1 + q1*(t2 + q2*t3),  t1 = r1,  t2 = (r2 - t1)/q1 mod q2,  t3 = ((r3 - t1)/q1 - t2)/q2 mod q3,
//   out = (t1 + (q1 mod p)*t2 + (q1*q2 mod p)*t3) mod p.
// The same integer the reference's four-prime sum reconstructs (main_naive.cc:216-223), for operand sizes and
// moduli whose exact product coefficients no longer fit below the product of its four 31-bit primes.
// ---------------------------------------------------------------------------
struct WideCrtConsts {
    int k;                 // number of primes in use (2 or 3)
    uint64_t q[3], nq[3];  // primes, -q^-1 mod 2^64
    uint64_t inv1_m2;      // Montgomery form (mod q2) of q1^-1
    uint64_t inv1_m3;      // Montgomery form (mod q3) of q1^-1
    uint64_t inv2_m3;      // Montgomery form (mod q3) of q2^-1
    uint64_t p, d, v;      // final modulus, p << shift, reciprocal of d (as in the four-prime merge)
    int shift;
    uint64_t q1_p, q12_p;  // q1 mod p, q1*q2 mod p
};

__device__ __forceinline__ uint64_t rem128(uint64_t u1, uint64_t u0, uint64_t d, uint64_t v, int shift) {
    // (u1:u0) mod (d >> shift) for u1 < (d >> shift); d normalised (Moeller-Granlund 2-by-1 division)
    if (shift) { u1 = (u1 << shift) | (u0 >> (64 - shift)); u0 <<= shift; }
    uint64_t q0 = v * u1, q1 = mulhi64(v, u1);
    q0 += u0;
    q1 += u1 + (q0 < u0 ? 1u : 0u);
    q1 += 1;
    uint64_t r = u0 - q1 * d;
    if (r > q0) r += d;
    if (r >= d) r -= d;
    return r >> shift;
}

__global__ void __launch_bounds__(256) k64_garner(const uint64_t* res, size_t res_stride, uint64_t* out, size_t count,
                                                  size_t rows, size_t out_stride, size_t row_stride, WideCrtConsts c) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    auto mont = [](uint64_t x, uint64_t wm, uint64_t q, uint64_t nq) -> uint64_t {  // x*w mod q, canonical
        const uint64_t lo = x * wm, hi = mulhi64(x, wm);
        const uint64_t m = lo * nq;
        const uint64_t r = hi + mulhi64(m, q) + (lo != 0 ? 1u : 0u);
        return r >= q ? r - q : r;
    };
    auto csub = [](uint64_t x, uint64_t q) -> uint64_t { return x >= q ? x - q : x; };
    auto modp = [&](uint64_t x) -> uint64_t { return rem128(0, x, c.d, c.v, c.shift); };
    auto mulp = [&](uint64_t a, uint64_t b) -> uint64_t { return rem128(mulhi64(a, b), a * b, c.d, c.v, c.shift); };  // a, b < p
    auto addp = [&](uint64_t a, uint64_t b) -> uint64_t { const uint64_t s = a + b; return (s < a || s >= c.p) ? s - c.p : s; };
    for (size_t row = blockIdx.y; row < rows; row += gridDim.y) {   // grid.y is capped at the launch limit
        const uint64_t r1 = res[row * row_stride + i], r2 = res[res_stride + row * row_stride + i];
        const uint64_t t1 = r1;
        const uint64_t t2 = mont(r2 - csub(t1, c.q[1]) + c.q[1], c.inv1_m2, c.q[1], c.nq[1]);
        uint64_t t3 = 0;
        if (c.k == 3) {
            const uint64_t r3 = res[2 * res_stride + row * row_stride + i];
            const uint64_t u = mont(r3 - csub(t1, c.q[2]) + c.q[2], c.inv1_m3, c.q[2], c.nq[2]);
            t3 = mont(u - csub(t2, c.q[2]) + c.q[2], c.inv2_m3, c.q[2], c.nq[2]);
        }
        uint64_t acc = addp(modp(t1), mulp(c.q1_p, modp(t2)));
        if (c.k == 3) acc = addp(acc, mulp(c.q12_p, modp(t3)));
        out[row * out_stride + i] = acc;
    }
}

}  // namespace nttamd
