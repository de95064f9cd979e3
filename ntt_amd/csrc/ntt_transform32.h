// Stand-alone transforms in natural order on 32-bit residues: the reference's NTT(a, n, invert, p, 3)
// (main_naive.cc:96-137) -- bit-reversal permutation (main_naive.cc:82-95), radix-2 butterfly stages, and for the
// inverse the n^-1 scaling (main_naive.cc:131-136) -- for odd primes p < 2^31 and n = 2 .. 2^18 with n | p-1.
//
// The products of ntt_kernels.h never need natural order (their spectra stay scrambled); a transform that promises it
// has to move every value across the array once.  That permutation is never a pass of scattered accesses here:
//
//   n <= 4096          k_ntt_small16   4096/n transforms per workgroup, radix-16 register rounds with LDS exchanges (the
//                                      rounds of k_small16); the bit reversal happens inside LDS, loads and stores are
//                                      contiguous.  k_ntt_small2: radix-2 stages, for n < 32 or 64 not dividing p-1.
//   n = 4096 * 2^Q     forward: column pass (k_fwd_strided, or k_ntt_outer2 when Q = 1) -> k_ntt_core_fwd (4096-point
//   (Q = 1..6)         transform of every row, its outputs put in natural order inside LDS) -> k_ntt_transpose (rows of
//                      4096 in, natural order out: a tiled transposition through LDS, both sides contiguous);
//                      inverse: the mirror -- k_ntt_transpose, k_ntt_core_inv, k_inv_strided / k_ntt_outer2.
//
// Three passes of 8 bytes per value; the 64-bit-field route that served these moduli before moves 80-96.
#pragma once
#include "ntt_kernels.h"

namespace nttamd {

NTT_HD uint32_t brev32(uint32_t x) {
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    return x;
}
// bit reversal of the low `bits` (>= 1) bits of x
NTT_HD int brev_low(int x, int bits) { return (int)(brev32((uint32_t)x) >> (32 - bits)); }

// LDS slot of natural position K < 4096 in the staging buffers below: bits 5-9 folded into the bank bits, so that
// both sides of a bit-reversing exchange -- consecutive K on one side, K with consecutive HIGH bits on the other -- touch
// 32 different banks per half-wave (two-way conflicts at most for the shortest transforms).
NTT_HD int nat_slot(int K) { return K ^ ((K >> 5) & 31); }

// any value of the caller's array -> [0, B): 32-bit words are reduced by one multiplication with the Montgomery form of 1
// (they may be anything below 2^32), 64-bit words by reduce64
template <bool LZ>
__device__ __forceinline__ uint32_t reduce_any(uint32_t x, const Field<LZ>& f, const FieldConsts& fc) { return f.mulw(x, fc.r1); }
template <bool LZ>
__device__ __forceinline__ uint32_t reduce_any(uint64_t x, const Field<LZ>& f, const FieldConsts& fc) { return f.reduce64(x, fc.r1, fc.r2); }
template <bool LZ, typename T>
__device__ __forceinline__ uint32_t load_any(const T* src, size_t idx, const Field<LZ>& f, const FieldConsts& fc) {
    return reduce_any<LZ>(src[idx], f, fc);
}

// ---------------------------------------------------------------------------
// 4096-point forward core of ONE array (core_forward2 of ntt_kernels.h transforms two at once): in a[i] = element
// t + 256 i, out a[i] = position e = 16 t + i of the in-place decimation-in-frequency result, i.e. X[bitrev_12(e)].
// ---------------------------------------------------------------------------
template <bool LZ>
__device__ __forceinline__ void core_forward1(uint32_t (&a)[16], uint32_t* lds, const Field<LZ>& f, const FieldConsts& fc,
                                              const CoreTables& tb, int t) {
    uint32_t tw[16];
    const int hi = t >> 4, lo = t & 15;
    dif_reg<4, LZ, false, true>(a, f, fc.rootF, fc.rootFn);
    load_tw16(tw, tb.tw1F, t);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        if (pos) a[pos] = f.mulw(a[pos], tw[pos]);
        lds[core_slot(pos, hi, lo)] = a[pos];
    }
    __syncthreads();
    load_tw16(tw, tb.tw2F, lo);
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = lds[core_slot(hi, i, lo)];
    dif_reg<4, LZ, false, true>(a, f, fc.rootF, fc.rootFn);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        if (pos) a[pos] = f.mulw(a[pos], tw[pos]);
        lds[core_slot(hi, pos, lo)] = a[pos];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = lds[core_slot(hi, lo, i)];
    dif_reg<4, LZ>(a, f, fc.rootF, fc.rootFn);
}

// natural index (within a 4096-point transform) of position e = 16 t + i
__device__ __forceinline__ int core_natural_index(int t, int i) { return (bitrev(i, 4) << 8) | brev_low(t, 8); }

// Rows of 4096 values, in place: row -> its 4096-point forward transform in natural order, values in [0, B).
template <bool LZ>
__global__ void __launch_bounds__(CORE_T) k_ntt_core_fwd(uint32_t* data, FieldConsts fc, CoreTables tb) {
    __shared__ uint32_t lds[CORE_N];
    const Field<LZ> f(fc);
    const int t = threadIdx.x;
    uint32_t* row = data + (size_t)blockIdx.x * CORE_N;
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = row[t + 256 * i];
    core_forward1<LZ>(a, lds, f, fc, tb, t);
    __syncthreads();                       // the last round's reads are done before the buffer is reused for the reordering
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[nat_slot(core_natural_index(t, i))] = a[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) row[t + 256 * i] = lds[nat_slot(t + 256 * i)];
}

// Rows of 4096 values in natural order, in place: row -> 4096 * c * (its inverse transform), c the constant the table
// tw1I of `tb` carries (n^-1 for the stand-alone transform), in the inverse side's lazy range.
template <bool LZ>
__global__ void __launch_bounds__(CORE_T) k_ntt_core_inv(uint32_t* data, FieldConsts fc, CoreTables tb) {
    __shared__ uint32_t lds[CORE_N];
    const Field<LZ> f(fc);
    const int t = threadIdx.x;
    uint32_t* row = data + (size_t)blockIdx.x * CORE_N;
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[nat_slot(t + 256 * i)] = row[t + 256 * i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = lds[nat_slot(core_natural_index(t, i))];
    __syncthreads();                       // core_inverse writes the slots of its own layout next
    core_inverse<LZ>(a, lds, f, fc, tb, t);
#pragma unroll
    for (int i = 0; i < 16; ++i) row[t + 256 * i] = a[i];
}

// ---------------------------------------------------------------------------
// k_ntt_transpose: the move between the row layout of the passes and natural order, L = 4096 << Q per transform.
//   row layout:  value (pos, k2) at pos * 4096 + k2, pos < R = 2^Q
//   natural:     the same value at k2 * R + bitrev_Q(pos)
// A workgroup moves the 4096 values of 4096/R consecutive k2 (all R rows) through LDS; reads and writes are contiguous
// runs of at least 256 bytes.  Forward: row layout (32-bit work array) -> natural (caller's array, canonical).
// Inverse: natural (caller's array, any values, reduced here) -> row layout.
// ---------------------------------------------------------------------------
template <bool LZ, bool INVERSE, typename T>
__global__ void __launch_bounds__(256) k_ntt_transpose(uint32_t* work, T* natural, int Q, FieldConsts fc) {
    __shared__ uint32_t lds[CORE_N];
    const Field<LZ> f(fc);
    const int t = threadIdx.x;
    const int R = 1 << Q, logTK = 12 - Q, TK = 1 << logTK;
    const size_t tr = blockIdx.x >> Q;                       // transform (row of the batch)
    const int k2_0 = (int)(blockIdx.x & (R - 1)) << logTK;   // first k2 of this tile
    uint32_t* w = work + tr * ((size_t)CORE_N << Q);
    T* nat = natural + tr * ((size_t)CORE_N << Q) + ((size_t)k2_0 << Q);
    const int sh = Q > 5 ? Q : 5;
    auto slot = [&](int idx) -> int { return idx ^ ((idx >> sh) & 31); };
    if (!INVERSE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i, pos = idx >> logTK, kk = idx & (TK - 1);
            lds[slot((kk << Q) | brev_low(pos, Q))] = f.canon_fwd(w[(size_t)pos * CORE_N + k2_0 + kk]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int idx = t + 256 * i; nat[idx] = (T)lds[slot(idx)]; }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int idx = t + 256 * i; lds[slot(idx)] = load_any<LZ>(nat, (size_t)idx, f, fc); }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i, pos = idx >> logTK, kk = idx & (TK - 1);
            w[(size_t)pos * CORE_N + k2_0 + kk] = lds[slot((kk << Q) | brev_low(pos, Q))];
        }
    }
}

// ---------------------------------------------------------------------------
// k_ntt_outer2: the column pass of n = 8192 = 2 x 4096 (the register column passes of ntt_kernels.h start at four
// rows).  Forward: (x[j], x[j + 4096]) -> row 0: x[j] + x[j + 4096], row 1: (x[j] - x[j + 4096]) * w^j.  Inverse: the
// mirror, canonical.  ow = [w8192^j | w8192^-j], j < 4096, Montgomery forms.
// ---------------------------------------------------------------------------
template <bool LZ, bool INVERSE, typename T>
__global__ void __launch_bounds__(256) k_ntt_outer2(uint32_t* work, T* natural, const uint32_t* ow, FieldConsts fc) {
    const Field<LZ> f(fc);
    const int j = (int)(blockIdx.x & 15) * 256 + threadIdx.x;
    const size_t tr = blockIdx.x >> 4;
    uint32_t* w = work + tr * (size_t)(2 * CORE_N);
    T* nat = natural + tr * (size_t)(2 * CORE_N);
    if (!INVERSE) {
        uint32_t u = load_any<LZ>(nat, (size_t)j, f, fc), v = load_any<LZ>(nat, (size_t)(j + CORE_N), f, fc);
        const uint32_t wm = ow[j];
        f.dif(u, v, wm, f.p - wm);
        w[j] = u; w[j + CORE_N] = v;
    } else {
        uint32_t x = w[j], y = w[j + CORE_N];
        f.dit(x, y, ow[CORE_N + j]);
        nat[j] = (T)f.canon_inv(x); nat[j + CORE_N] = (T)f.canon_inv(y);
    }
}

// ---------------------------------------------------------------------------
// k_ntt_small16: transforms of length n = 2^lg, 32 <= n <= 4096, 4096/n of them per workgroup, in place in the caller's
// array (rows of n values, `count` of them).  Same rounds as k_small16 of ntt_kernels.h on one array; the bit reversal
// is an exchange inside LDS (nat_slot), so that the global loads and stores are contiguous.
//   forward: natural in -> in-place decimation in frequency (position e holds X[bitrev(e)]) -> reorder -> natural out
//   inverse: natural in -> reorder (position e takes a[bitrev(e)]) -> decimation in time -> natural out, times scale_m
//            (the Montgomery form of n^-1)
// ---------------------------------------------------------------------------
#ifndef NTT_TSMALL16_DIRECT_FROM
#define NTT_TSMALL16_DIRECT_FROM 6
#endif
template <bool LZ, bool INVERSE, typename T>
__global__ void __launch_bounds__(CORE_T) k_ntt_small16(T* data, int lg, size_t count, uint32_t scale_m, FieldConsts fc,
                                                        RoundTables32 rt) {
    __shared__ uint32_t la[CORE_N];
    __shared__ uint32_t lb[CORE_N];      // staging in natural order
    const Field<LZ> f(fc);
    const int t = threadIdx.x;
    const int ql = ((lg - 1) & 3) + 1;
    const int mask = (1 << lg) - 1;
    const size_t first = (size_t)blockIdx.x << (12 - lg);     // first transform of this workgroup
    const size_t total = count << lg;                          // values in the array
    T* base = data + (first << lg);
    const size_t avail = total - (first << lg);                // values from `base` on (the last workgroup may be short)
    uint32_t a[16];
    // position e = 16 t + i of a thread <-> natural position inside the workgroup's 4096 values
    auto natural_of = [&](int e) -> int { return (e & ~mask) | brev_low(e & mask, lg); };
    const bool direct = lg >= NTT_TSMALL16_DIRECT_FROM;
    if (!INVERSE) {
        // One radix-16 round over blocks of 2^bl values: twiddle, then back to the same slots.
        auto finish_round = [&](int bl, int e0, int sub, int j) {
            dif_reg<4, LZ, false, true>(a, f, fc.rootF, fc.rootFn);
            uint32_t tw[16];
            load_tw16(tw, rt.f[bl], j);
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                if (pos) a[pos] = f.mulw(a[pos], tw[pos]);
                la[small_slot(e0 + pos * sub)] = a[pos];
            }
            __syncthreads();
        };
        int bl = lg;
        T raw[16];      // all loads requested before the first use, none conditional (see k_small16)
        if (direct) {
            // first round straight from the caller's array: no staging pass and no barrier before it
            const int sub = 1 << (bl - 4);
            const int blk = t >> (bl - 4), j = t & (sub - 1);
            const int e0 = (blk << bl) + j;
            const bool valid = (size_t)e0 < avail;      // transforms are whole: one answer for all 16 values
#pragma unroll
            for (int k = 0; k < 16; ++k) raw[k] = base[valid ? e0 + k * sub : 0];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t v = reduce_any<LZ>(raw[k], f, fc);
                a[k] = valid ? v : 0u;
            }
            finish_round(bl, e0, sub, j);
            bl -= 4;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int e = t + CORE_T * i; raw[i] = base[(size_t)e < avail ? e : 0]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int e = t + CORE_T * i;
                NTT_OPAQUE(e);
                const uint32_t v = reduce_any<LZ>(raw[i], f, fc);
                la[small_slot(e)] = (size_t)e < avail ? v : 0u;
            }
            __syncthreads();
        }
        for (; bl > 4; bl -= 4) {
            const int sub = 1 << (bl - 4);
            const int blk = t >> (bl - 4), j = t & (sub - 1);
            const int e0 = (blk << bl) + j;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = la[small_slot(e0 + k * sub)];
            finish_round(bl, e0, sub, j);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = la[small_slot((t << 4) + i)];
        switch (ql) {
            case 1: dif_blocks32<1, LZ>(a, f, fc); break;
            case 2: dif_blocks32<2, LZ>(a, f, fc); break;
            case 3: dif_blocks32<3, LZ>(a, f, fc); break;
            default: dif_blocks32<4, LZ>(a, f, fc); break;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) lb[nat_slot(natural_of((t << 4) + i))] = f.canon_fwd(a[i]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int e = t + CORE_T * i;
            NTT_OPAQUE(e);
            if ((size_t)e < avail) base[e] = (T)lb[nat_slot(e)];
        }
    } else {
        T raw[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int e = t + CORE_T * i; raw[i] = base[(size_t)e < avail ? e : 0]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int e = t + CORE_T * i;
            NTT_OPAQUE(e);
            const uint32_t v = reduce_any<LZ>(raw[i], f, fc);
            lb[nat_slot(e)] = (size_t)e < avail ? v : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = lb[nat_slot(natural_of((t << 4) + i))];
        switch (ql) {
            case 1: dit_blocks32<1, LZ>(a, f, fc); break;
            case 2: dit_blocks32<2, LZ>(a, f, fc); break;
            case 3: dit_blocks32<3, LZ>(a, f, fc); break;
            default: dit_blocks32<4, LZ>(a, f, fc); break;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) la[small_slot((t << 4) + i)] = a[i];
        for (int bl = ql + 4; bl <= lg; bl += 4) {
            __syncthreads();
            const int sub = 1 << (bl - 4);
            const int blk = t >> (bl - 4), j = t & (sub - 1);
            const int e0 = (blk << bl) + j;
            uint32_t tw[16];
            load_tw16(tw, rt.i[bl], j);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t v = la[small_slot(e0 + k * sub)];
                a[k] = k ? f.mulw(v, tw[k]) : (LZ ? csub(v, f.B) : v);
            }
            dit_reg<4, LZ, true>(a, f, fc.rootI);
            if (direct && bl + 4 > lg) {
                // last round straight to the caller's array
                if ((size_t)e0 < avail) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) base[e0 + k * sub] = (T)f.canon_fwd(f.mulw(a[k], scale_m));
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) la[small_slot(e0 + k * sub)] = a[k];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int e = t + CORE_T * i;
            NTT_OPAQUE(e);
            if ((size_t)e < avail) base[e] = (T)f.canon_fwd(f.mulw(la[small_slot(e)], scale_m));
        }
    }
}

// ---------------------------------------------------------------------------
// k_ntt_small2: any n = 2^lg <= 4096 (and any NTT-friendly p): one transform per workgroup, the reference's own
// structure -- bit-reversal permutation, then radix-2 decimation-in-time stages (main_naive.cc:103-128) -- in LDS with
// table twiddles.  tw = Montgomery forms of w_n^e (forward) or w_n^-e (inverse), e < n/2.
// ---------------------------------------------------------------------------
template <bool LZ, bool INVERSE, typename T>
__global__ void __launch_bounds__(256) k_ntt_small2(T* data, int lg, const uint32_t* tw, uint32_t scale_m, FieldConsts fc) {
    __shared__ uint32_t la[CORE_N];
    const Field<LZ> f(fc);
    const int n = 1 << lg, half = n >> 1;
    const int t = threadIdx.x, TT = blockDim.x;
    T* row = data + ((size_t)blockIdx.x << lg);
    for (int i = t; i < n; i += TT) la[brev_low(i, lg)] = load_any<LZ>(row, (size_t)i, f, fc);
    __syncthreads();
    for (int s = lg - 1; s >= 0; --s) {          // blocks of 2 h values, h = 1, 2, ... n/2
        const int h = half >> s;
        for (int k = t; k < half; k += TT) {
            const int j = k & (h - 1);
            const int i = ((k - j) << 1) + j;
            f.dit(la[i], la[i + h], tw[j << s]);
        }
        __syncthreads();
    }
    for (int i = t; i < n; i += TT)
        row[i] = (T)(INVERSE ? f.canon_fwd(f.mulw(la[i], scale_m)) : f.canon_inv(la[i]));
}

}  // namespace nttamd
