// Device kernels of the polynomial-multiplication path (gfx950).
//
// A product of two length-n polynomials uses transform length L (power of two
// >= 2n-1).  The spectrum is never materialised in natural order: the forward
// transforms are decimation-in-frequency (natural in, digit-scrambled out),
// the pointwise product is taken slot by slot, and the inverse transform is
// the exact mirror (decimation-in-time), so the reference's bit-reversal
// permutation (main_naive.cc:82-95) disappears while the outputs stay
// identical (canonical residues of the same product).
//
//   L <= 16, or 64 does   k_small       one workgroup per product, radix-2 stages in LDS
//     not divide p-1
//   32 <= L <= 4096       k_small16     4096/L products per workgroup, radix-16 register rounds with one LDS exchange
//                                       between rounds; L = 32 and 64 finish inside a quad of lanes (DPP exchanges)
//   L == 8192             k_fused8192   one workgroup per product: outer radix-2 stage in registers + two 4096-point
//                                       LDS-resident cores
//   L == 4096 * 2^Q       k_fwd_strided (radix-2^Q columns in registers, twiddle, store)
//     (Q = 2..6)          k_core_mul    (4096-point forward of A and B, pointwise product, 4096-point inverse, all in
//                                        one workgroup)
//                         k_inv_strided (twiddle, radix-2^Q inverse columns, canonical store)
//   four-prime CRT route  (moduli above 2^31 under the reference's route policy, and moduli that are not NTT-friendly)
//     L >= 2^14           k_fwd_strided_crt (64-bit operands read once, residue step and forward column pass of all
//                         four primes) -> k_core_mul_crt (the four cores, one launch) -> k_inv_strided_crt (inverse
//                         column pass of all four primes and the recombination; the residues never reach memory)
//     shorter             the single-kernel paths above once per prime on 64-bit inputs, then k_crt_merge
//
// Reference loops these replace: zero-pad/copy main_naive.cc:147-153, butterfly
// stages :106-128, pointwise product :163-165, n^-1 scaling :131-136, copy-out
// :171-173, residue step :206-209, CRT merge :216-223.
#pragma once
#include "ntt_field.h"
#include "ntt_field64.h"

namespace nttamd {

// Compiler-only barrier for memory operations: keeps loads issued before it ahead of the arithmetic
// that follows and stores after it behind, which pins the software pipelining written below
// (hipcc otherwise either hoists every load to the top, costing occupancy, or serialises
// load -> wait -> use).  Emits no instruction.
#define NTT_FENCE() asm volatile("" ::: "memory")

// Diagnostic builds only (tools/ablate.sh): -DNTT_ABLATE_BARRIER drops the workgroup barriers of the
// core, -DNTT_ABLATE_GLOAD replaces its global data loads by register values.  Results are then
// wrong by construction; only the timing is read.
#ifdef NTT_ABLATE_BARRIER
#define NTT_CORE_SYNC() NTT_FENCE()
#else
#define NTT_CORE_SYNC() __syncthreads()
#endif
// -DNTT_ABLATE_LDS: the core's LDS exchanges become register pass-throughs
#ifdef NTT_ABLATE_LDS
#define NTT_LDS_ST(arr, idx, val) asm volatile("" :: "v"(val))
#define NTT_LDS_LD(arr, idx, cur) (cur)
#else
#define NTT_LDS_ST(arr, idx, val) ((arr)[idx] = (val))
#define NTT_LDS_LD(arr, idx, cur) ((arr)[idx])
#endif

constexpr int CORE_N = 4096;   // points per LDS-resident core transform
constexpr int CORE_T = 256;    // threads per core workgroup (16 points each)

// ---------------------------------------------------------------------------
// loads / stores with on-the-fly width conversion
// ---------------------------------------------------------------------------
template <bool LZ>
__device__ __forceinline__ uint32_t load_coeff(const uint32_t* src, size_t idx, const Field<LZ>&, const FieldConsts&) {
    return src[idx];
}
template <bool LZ>
__device__ __forceinline__ uint32_t load_coeff(const uint64_t* src, size_t idx, const Field<LZ>& f, const FieldConsts& fc) {
    return f.reduce64(src[idx], fc.r1, fc.r2);
}

// the same on a word that has already been loaded
template <bool LZ>
__device__ __forceinline__ uint32_t reduce_coeff(uint32_t x, const Field<LZ>&, const FieldConsts&) { return x; }
template <bool LZ>
__device__ __forceinline__ uint32_t reduce_coeff(uint64_t x, const Field<LZ>& f, const FieldConsts& fc) { return f.reduce64(x, fc.r1, fc.r2); }

// Coefficient idx of a row that holds only `limit` (>= 1) values, zero beyond.  The address is always a valid one and the
// zero is selected afterwards: a conditional load would be compiled as a branch per element, which ends the basic
// block and makes every load wait for the one before it.  Used by the memory-bound column pass; the single-kernel
// paths are bound by arithmetic and run 4 % faster with the plain conditional.
template <bool LZ, typename TIn>
__device__ __forceinline__ uint32_t load_coeff_padded(const TIn* src, int idx, int limit, const Field<LZ>& f, const FieldConsts& fc) {
    const bool inside = idx < limit;
    const uint32_t v = load_coeff<LZ>(src, (size_t)(inside ? idx : 0), f, fc);
    return inside ? v : 0u;
}

__device__ __forceinline__ void load_tw16(uint32_t (&tw)[16], const uint32_t* base, int row) {
    const uint4* q = reinterpret_cast<const uint4*>(base) + (size_t)row * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint4 v = q[k];
        tw[4 * k + 0] = v.x; tw[4 * k + 1] = v.y; tw[4 * k + 2] = v.z; tw[4 * k + 3] = v.w;
    }
}

// ---------------------------------------------------------------------------
// 4096-point core.  Element e = d2*256 + d1*16 + d0 (base-16 digits).
//   round 1: thread (d1,d0) holds the 16 values over d2      (stride 256)
//   round 2: thread (d2,d0) holds the 16 values over d1      (stride 16)
//   round 3: thread (d2,d1) holds the 16 values over d0      (contiguous)
// LDS slot of (d2,d1,d0) is swizzled so that all three access patterns hit 32
// distinct banks per half-wave.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int core_slot(int d2, int d1, int d0) {
    return (d2 << 8) | ((d1 ^ (d2 & 1)) << 4) | (d0 ^ d1);
}

struct CoreTables {
    const uint32_t* tw1F;  // [256][16]  Montgomery form of w4096^( t * bitrev4(pos))
    const uint32_t* tw2F;  // [16][16]   Montgomery form of w256^ ( c * bitrev4(pos))
    const uint32_t* tw1I;  // [256][16]  Montgomery form of w4096^(-t * bitrev4(pos)) * scale
    const uint32_t* tw2I;  // [16][16]   Montgomery form of w256^ (-c * bitrev4(pos))
};

// Forward transform of two arrays at once.  In: a[i], b[i] = element t + 256*i, values in [0,B).
// Out: a[i], b[i] = spectrum slot (d2 = t>>4, d1 = t&15, d0 = i), values in [0,B).
template <bool LZ>
__device__ __forceinline__ void core_forward2(uint32_t (&a)[16], uint32_t (&b)[16], uint32_t* ldsA, uint32_t* ldsB,
                                              const Field<LZ>& f, const FieldConsts& fc, const CoreTables& tb, int t) {
    uint32_t tw[16];
    const int hi = t >> 4, lo = t & 15;
    // round 1
    dif_reg<4, LZ, false, true>(a, f, fc.rootF, fc.rootFn);
    dif_reg<4, LZ, false, true>(b, f, fc.rootF, fc.rootFn);
    load_tw16(tw, tb.tw1F, t);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        if (pos) { a[pos] = f.mulw(a[pos], tw[pos]); b[pos] = f.mulw(b[pos], tw[pos]); }
        const int s = core_slot(pos, hi, lo);
        NTT_LDS_ST(ldsA, s, a[pos]); NTT_LDS_ST(ldsB, s, b[pos]);
    }
    NTT_CORE_SYNC();
    load_tw16(tw, tb.tw2F, lo);
    // round 2
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int s = core_slot(hi, i, lo); a[i] = NTT_LDS_LD(ldsA, s, a[i]); b[i] = NTT_LDS_LD(ldsB, s, b[i]); }
    dif_reg<4, LZ, false, true>(a, f, fc.rootF, fc.rootFn);
    dif_reg<4, LZ, false, true>(b, f, fc.rootF, fc.rootFn);
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        if (pos) { a[pos] = f.mulw(a[pos], tw[pos]); b[pos] = f.mulw(b[pos], tw[pos]); }
        const int s = core_slot(hi, pos, lo);
        NTT_LDS_ST(ldsA, s, a[pos]); NTT_LDS_ST(ldsB, s, b[pos]);
    }
    NTT_CORE_SYNC();
    // round 3
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int s = core_slot(hi, lo, i); a[i] = NTT_LDS_LD(ldsA, s, a[i]); b[i] = NTT_LDS_LD(ldsB, s, b[i]); }
    dif_reg<4, LZ>(a, f, fc.rootF, fc.rootFn);
    dif_reg<4, LZ>(b, f, fc.rootF, fc.rootFn);
}

// Pointwise product of the two spectra followed by the inverse transform (mirror of
// core_forward2).  Out: a[i] = element t + 256*i of 4096 * scale * (cyclic product) * R^-1,
// LZ: in [0,4p), !LZ: in [0,p).
template <bool LZ>
__device__ __forceinline__ void core_pointwise(uint32_t (&a)[16], const uint32_t (&b)[16], const Field<LZ>& f) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f.mulpw(a[i], b[i]);
}

template <bool LZ>
__device__ __forceinline__ void core_inverse(uint32_t (&a)[16], uint32_t* ldsA, const Field<LZ>& f, const FieldConsts& fc,
                                             const CoreTables& tb, int t) {
    uint32_t tw[16];
    const int hi = t >> 4, lo = t & 15;
    // inverse of round 3 (same thread owns the same slots: no barrier before the write)
    dit_reg<4, LZ, true>(a, f, fc.rootI);
#pragma unroll
    for (int i = 0; i < 16; ++i) NTT_LDS_ST(ldsA, core_slot(hi, lo, i), a[i]);
    NTT_FENCE();
    load_tw16(tw, tb.tw2I, lo);  // next round's twiddles travel while the workgroup synchronises
    NTT_CORE_SYNC();
    // inverse of round 2
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint32_t v = NTT_LDS_LD(ldsA, core_slot(hi, i, lo), a[i]);
        a[i] = i ? f.mulw(v, tw[i]) : (LZ ? csub(v, f.B) : v);
    }
    dit_reg<4, LZ, true>(a, f, fc.rootI);
#pragma unroll
    for (int i = 0; i < 16; ++i) NTT_LDS_ST(ldsA, core_slot(hi, i, lo), a[i]);
    NTT_FENCE();
    load_tw16(tw, tb.tw1I, t);
    NTT_CORE_SYNC();
    // inverse of round 1 (every slot is multiplied: the table carries the scale factor)
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f.mulw(NTT_LDS_LD(ldsA, core_slot(i, hi, lo), a[i]), tw[i]);
    dit_reg<4, LZ, true>(a, f, fc.rootI);
}

// ---------------------------------------------------------------------------
// k_core_mul: middle kernel of the multi-pass path.  Block (product, k1) of tmpA/tmpB
// (4096 values each) -> same block of tmpC.  tmpC may alias tmpA (a block is fully read before it
// is written, and other blocks are other addresses).
// ---------------------------------------------------------------------------
// One 4096-block: pa/pb/pc point at the block's first value (pc may equal pa).
template <bool LZ>
__device__ __forceinline__ void core_block(const uint32_t* pa, const uint32_t* pb, uint32_t* pc, uint32_t* lds,
                                           const FieldConsts& fc, const CoreTables& tb, int t) {
    const Field<LZ> f(fc);
    uint32_t a[16], b[16];
#ifdef NTT_ABLATE_GLOAD
    for (int i = 0; i < 16; ++i) { a[i] = (uint32_t)(size_t)pa * 2654435761u + i + t; b[i] = a[i] ^ 12345u; }
#else
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = pa[t + 256 * i]; b[i] = pb[t + 256 * i]; }
#endif
    core_forward2<LZ>(a, b, lds, lds + CORE_N, f, fc, tb, t);
    core_pointwise<LZ>(a, b, f);
    core_inverse<LZ>(a, lds, f, fc, tb, t);
#pragma unroll
    for (int i = 0; i < 16; ++i) pc[t + 256 * i] = a[i];
}

template <bool LZ>
__global__ void __launch_bounds__(CORE_T) k_core_mul(const uint32_t* tmpA, const uint32_t* tmpB, uint32_t* tmpC,
                                                     FieldConsts fc, CoreTables tb) {
#ifdef NTT_CORE_LDS_PAD   // diagnostic: extra LDS lowers the number of resident workgroups
    __shared__ uint32_t lds[2 * CORE_N + NTT_CORE_LDS_PAD];
#else
    __shared__ uint32_t lds[2 * CORE_N];
#endif
    const size_t base = (size_t)blockIdx.x * CORE_N;
    core_block<LZ>(tmpA + base, tmpB + base, tmpC + base, lds, fc, tb, threadIdx.x);
}

// The fused core of the four CRT primes in one launch: blockIdx.y picks the prime (its arrays, constants and tables), so the
// four cores of a product batch share one grid -- one ramp-up and one tail instead of four.  gridDim.x is a multiple of
// eight, so a 4096-point row still runs on XCD (row mod 8) whichever prime it belongs to.
struct CrtCoreArgs {
    const uint32_t* A[4];
    const uint32_t* B[4];
    uint32_t* C[4];
    FieldConsts fc[4];
    CoreTables tb[4];
};

// One copy of the core per prime, each reading its constants at fixed offsets of the argument block (indexed by a run-time
// prime number they would be fetched once and kept in vector registers: 165 instead of 119, one wavefront per SIMD fewer).
// The second prime of the basis is the one above 2^30 (values kept fully reduced), as in the column passes below.
template <int K>
__device__ __forceinline__ void core_block_of_prime(const CrtCoreArgs& a, uint32_t* lds) {
    const size_t base = (size_t)blockIdx.x * CORE_N;
    core_block<K != 1>(a.A[K] + base, a.B[K] + base, a.C[K] + base, lds, a.fc[K], a.tb[K], threadIdx.x);
}

__global__ void __launch_bounds__(CORE_T) k_core_mul_crt(CrtCoreArgs a) {
    __shared__ uint32_t lds[2 * CORE_N];
    switch (blockIdx.y) {
        case 0: core_block_of_prime<0>(a, lds); break;
        case 1: core_block_of_prime<1>(a, lds); break;
        case 2: core_block_of_prime<2>(a, lds); break;
        default: core_block_of_prime<3>(a, lds); break;
    }
}

// ---------------------------------------------------------------------------
// Strided passes.  fwd_column: first pass of a length Lb = C * 2^Q transform for one column j:
// registers = the 2^Q rows j + C*r of src.  PADDED: the source row has only n_valid <= Lb/2
// values, the rest are implicit zeros that are never read; FULL: n_valid is exactly Lb/2 (no
// per-element bound check).  Output slot pos*C + j holds (row-transform output bitrev_Q(pos))
// * w_Lb^(j*bitrev_Q(pos)).  Twiddle table layout: [(pos>>2)][j][pos&3], so a thread's 4
// consecutive positions are one 16-byte load and neighbouring threads are contiguous.
// ---------------------------------------------------------------------------
// load(r, f): the reduced coefficient of row r of this column (rows below R/2 only when PADDED)
template <int Q, bool LZ, bool PADDED, class Load>
__device__ __forceinline__ void fwd_column_from(Load&& load, uint32_t* dst, int C, const uint32_t* T1, const FieldConsts& fc, int j) {
    constexpr int R = 1 << Q;
    const Field<LZ> f(fc);
    uint32_t x[R];
    constexpr int G = R / 4;              // 16-byte twiddle groups (4 positions each)
    constexpr int SG = G < 4 ? G : 4;     // groups per pipeline stage
    constexpr int NS = G / SG;
    const uint4* Tq = reinterpret_cast<const uint4*>(T1) + j;
    uint4 wb[2][SG];
#pragma unroll
    for (int r = 0; r < (PADDED ? R / 2 : R); ++r) x[r] = load(r, f);
#pragma unroll
    for (int k = 0; k < SG; ++k) wb[0][k] = Tq[(size_t)k * C];
    NTT_FENCE();
#ifndef NTT_ABLATE_STRIDED_MATH   // diagnostic: memory traffic of the pass without its butterflies
    dif_reg<Q, LZ, PADDED, true>(x, f, fc.rootF, fc.rootFn);
#else
    if (PADDED) { for (int r = R / 2; r < R; ++r) x[r] = x[r - R / 2] + 1u; }
#endif
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        if (st + 1 < NS) {
#pragma unroll
            for (int k = 0; k < SG; ++k) wb[(st + 1) & 1][k] = Tq[(size_t)((st + 1) * SG + k) * C];
        }
        NTT_FENCE();
#pragma unroll
        for (int k = 0; k < SG; ++k) {
            const int g = st * SG + k;
            const uint4 w = wb[st & 1][k];
#ifndef NTT_ABLATE_STRIDED_MATH
            if (g) x[4 * g] = f.mulw(x[4 * g], w.x);
            x[4 * g + 1] = f.mulw(x[4 * g + 1], w.y);
            x[4 * g + 2] = f.mulw(x[4 * g + 2], w.z);
            x[4 * g + 3] = f.mulw(x[4 * g + 3], w.w);
#else
            x[4 * g] += w.x; x[4 * g + 1] += w.y; x[4 * g + 2] += w.z; x[4 * g + 3] += w.w;
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[(size_t)(4 * g + q) * C + j] = x[4 * g + q];
        }
        NTT_FENCE();
    }
}

template <int Q, bool LZ, bool PADDED, bool FULL, typename TIn>
__device__ __forceinline__ void fwd_column(const TIn* src, int n_valid, uint32_t* dst, int C, const uint32_t* T1,
                                           const FieldConsts& fc, int j) {
    fwd_column_from<Q, LZ, PADDED>([&](int r, const Field<LZ>& f) {
        const int idx = j + C * r;
        return (!PADDED || FULL) ? load_coeff<LZ>(src, (size_t)idx, f, fc) : load_coeff_padded<LZ>(src, idx, n_valid, f, fc);
    }, dst, C, T1, fc, j);
}

// inv_column_regs: mirror of fwd_column up to the registers: x[r] = value of row r of column j after the twiddles and the
// radix-2^Q inverse, in the inverse side's lazy range.
template <int Q, bool LZ>
__device__ __forceinline__ void inv_column_regs(const uint32_t* src, int C, const uint32_t* T3, const FieldConsts& fc, int j,
                                                uint32_t (&x)[1 << Q]) {
    constexpr int R = 1 << Q;
    const Field<LZ> f(fc);
    constexpr int G = R / 4;
    constexpr int SG = G < 4 ? G : 4;
    constexpr int NS = G / SG;
    const uint4* Tq = reinterpret_cast<const uint4*>(T3) + j;
    uint4 wb[2][SG];
#pragma unroll
    for (int k = 0; k < SG; ++k) wb[0][k] = Tq[(size_t)k * C];
#pragma unroll
    for (int pos = 0; pos < R; ++pos) x[pos] = src[(size_t)pos * C + j];
    NTT_FENCE();
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        if (st + 1 < NS) {
#pragma unroll
            for (int k = 0; k < SG; ++k) wb[(st + 1) & 1][k] = Tq[(size_t)((st + 1) * SG + k) * C];
        }
        NTT_FENCE();
#pragma unroll
        for (int k = 0; k < SG; ++k) {
            const int g = st * SG + k;
            const uint4 w = wb[st & 1][k];
#ifndef NTT_ABLATE_STRIDED_MATH
            x[4 * g] = g ? f.mulw(x[4 * g], w.x) : (LZ ? csub(x[0], f.B) : x[0]);
            x[4 * g + 1] = f.mulw(x[4 * g + 1], w.y);
            x[4 * g + 2] = f.mulw(x[4 * g + 2], w.z);
            x[4 * g + 3] = f.mulw(x[4 * g + 3], w.w);
#else
            x[4 * g] += w.x; x[4 * g + 1] += w.y; x[4 * g + 2] += w.z; x[4 * g + 3] += w.w;
#endif
        }
        NTT_FENCE();
    }
#ifndef NTT_ABLATE_STRIDED_MATH
    dit_reg<Q, LZ, true>(x, f, fc.rootI);
#endif
}

// inv_column: the inverse column pass of one column.  FINAL: results are reduced to [0,p) and only indices < n_out
// are stored; otherwise the block is written back in place layout (uint32, lazy range).
template <int Q, bool LZ, bool FINAL, bool FULL, typename TOut>
__device__ __forceinline__ void inv_column(const uint32_t* src, TOut* dst, int n_out, int C, const uint32_t* T3,
                                           const FieldConsts& fc, int j) {
    constexpr int R = 1 << Q;
    const Field<LZ> f(fc);
    uint32_t x[R];
    inv_column_regs<Q, LZ>(src, C, T3, fc, j, x);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int idx = j + C * r;
        if (FINAL) {
            // FULL: n_out = C*2^Q - 1, only the very last index is dropped
            if ((FULL && r < R - 1) || idx < n_out) dst[idx] = (TOut)f.canon_inv(x[r]);
        } else {
            dst[idx] = (TOut)x[r];
        }
    }
}

// Grid: x = (C/256) * rows, z = 2 (A and B).  MINW: waves per SIMD the register allocation must allow.
template <int Q, bool LZ, bool PADDED, bool FULL, int MINW, typename TIn>
__global__ void __launch_bounds__(256, MINW) k_fwd_strided(const TIn* inA, const TIn* inB, size_t a_stride, size_t b_stride,
                                                     int n_valid, uint32_t* outA, uint32_t* outB, int C,
                                                     const uint32_t* T1, FieldConsts fc) {
    const int chunks = C >> 8;
    const int j = (int)(blockIdx.x % chunks) * 256 + threadIdx.x;
    const size_t row = blockIdx.x / chunks;
    const TIn* src = (blockIdx.z ? inB + row * b_stride : inA + row * a_stride);
    uint32_t* dst = (blockIdx.z ? outB : outA) + row * ((size_t)C << Q);
    fwd_column<Q, LZ, PADDED, FULL, TIn>(src, n_valid, dst, C, T1, fc, j);
}

template <int Q, bool LZ, bool FINAL, bool FULL, int MINW, typename TOut>
__global__ void __launch_bounds__(256, MINW) k_inv_strided(const uint32_t* in, TOut* out, size_t out_stride, int n_out, int C,
                                                     const uint32_t* T3, FieldConsts fc) {
    const int chunks = C >> 8;
    const int j = (int)(blockIdx.x % chunks) * 256 + threadIdx.x;
    const size_t row = blockIdx.x / chunks;
    inv_column<Q, LZ, FINAL, FULL, TOut>(in + row * ((size_t)C << Q), out + row * out_stride, n_out, C, T3, fc, j);
}

// ---------------------------------------------------------------------------
// k_fused8192: whole product for L = 8192 (n <= 4096) in one workgroup.  The zero-padded
// outer radix-2 stage splits the transform into an "even" 4096-point problem on x[j] and an
// "odd" one on x[j] * w8192^j; both run through the core, and the last inverse stage
// recombines them in registers.  ow = [owF[4096] | owI[4096]] Montgomery forms of w8192^(+-j).
// ---------------------------------------------------------------------------
template <bool LZ, typename TIn, typename TOut>
__global__ void __launch_bounds__(CORE_T) k_fused8192(const TIn* A, const TIn* Bp, TOut* out, size_t a_stride,
                                                      size_t b_stride, size_t out_stride, int n, int n_out,
                                                      const uint32_t* ow, FieldConsts fc, CoreTables tb) {
    // third quarter of the LDS: the even half's result waits there while the odd half is computed (each thread
    // re-reads only what it wrote).  Held in registers instead it cost a wave of occupancy (188 -> 160 VGPRs, +8 %).
    __shared__ uint32_t lds[3 * CORE_N];
    uint32_t* park = lds + 2 * CORE_N;
    const Field<LZ> f(fc);
    const int t = threadIdx.x;
    const TIn* sa = A + (size_t)blockIdx.x * a_stride;
    const TIn* sb = Bp + (size_t)blockIdx.x * b_stride;
    uint32_t a[16], b[16];
    // Every group of loads below is requested as a whole before its first value is used, and no load is conditional
    // (padding reads the row's first word and is replaced by zero afterwards): a load inside a condition together with
    // its first use -- a reduction, a twiddle multiply -- makes the thread wait for each load in turn.
    // even half
    {
        TIn ra[16], rb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int idx = t + 256 * i; ra[i] = sa[idx < n ? idx : 0]; rb[i] = sb[idx < n ? idx : 0]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i;
            const uint32_t va = reduce_coeff<LZ>(ra[i], f, fc), vb = reduce_coeff<LZ>(rb[i], f, fc);
            a[i] = idx < n ? va : 0u; b[i] = idx < n ? vb : 0u;
        }
    }
    core_forward2<LZ>(a, b, lds, lds + CORE_N, f, fc, tb, t);
    core_pointwise<LZ>(a, b, f);
    core_inverse<LZ>(a, lds, f, fc, tb, t);
#pragma unroll
    for (int i = 0; i < 16; ++i) park[t + 256 * i] = a[i];
    __syncthreads();  // the odd half reuses the LDS the even half may still be reading
    // odd half
    {
        TIn ra[16], rb[16];
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i;
            ra[i] = sa[idx < n ? idx : 0]; rb[i] = sb[idx < n ? idx : 0]; w[i] = ow[idx];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i;
            const uint32_t va = f.mulw(reduce_coeff<LZ>(ra[i], f, fc), w[i]), vb = f.mulw(reduce_coeff<LZ>(rb[i], f, fc), w[i]);
            a[i] = idx < n ? va : 0u; b[i] = idx < n ? vb : 0u;
        }
    }
    core_forward2<LZ>(a, b, lds, lds + CORE_N, f, fc, tb, t);
    core_pointwise<LZ>(a, b, f);
    core_inverse<LZ>(a, lds, f, fc, tb, t);
    TOut* dst = out + (size_t)blockIdx.x * out_stride;
    {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = ow[CORE_N + t + 256 * i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i;
            uint32_t x = park[idx], y = a[i];
            f.dit(x, y, w[i]);
            if (idx < n_out) dst[idx] = (TOut)f.canon_inv(x);
            if (idx + CORE_N < n_out) dst[idx + CORE_N] = (TOut)f.canon_inv(y);
        }
    }
}

// ---------------------------------------------------------------------------
// k_small: any L = 2^logL <= 4096, one workgroup per product, radix-2 stages in LDS with a
// barrier per stage.  twF/twI = Montgomery forms of wL^(+-e), e < L/2; scale_m = Montgomery
// form of L^-1 * R.
// ---------------------------------------------------------------------------
template <bool LZ, typename TIn, typename TOut>
__global__ void __launch_bounds__(256) k_small(const TIn* A, const TIn* Bp, TOut* out, size_t a_stride, size_t b_stride,
                                               size_t out_stride, int n, int n_out, int logL, const uint32_t* twF,
                                               const uint32_t* twI, uint32_t scale_m, FieldConsts fc) {
    __shared__ uint32_t la[CORE_N];
    __shared__ uint32_t lb[CORE_N];
    const Field<LZ> f(fc);
    const int L = 1 << logL, half = L >> 1;
    const int t = threadIdx.x, T = blockDim.x;
    const TIn* sa = A + (size_t)blockIdx.x * a_stride;
    const TIn* sb = Bp + (size_t)blockIdx.x * b_stride;
    for (int i = t; i < L; i += T) {
        la[i] = i < n ? load_coeff<LZ>(sa, (size_t)i, f, fc) : 0u;
        lb[i] = i < n ? load_coeff<LZ>(sb, (size_t)i, f, fc) : 0u;
    }
    __syncthreads();
    for (int s = 0; s < logL; ++s) {
        const int h = half >> s;
        for (int k = t; k < half; k += T) {
            const int j = k & (h - 1);
            const int i = ((k - j) << 1) + j;
            const uint32_t w = twF[j << s], wn = f.p - w;
            f.dif(la[i], la[i + h], w, wn);
            f.dif(lb[i], lb[i + h], w, wn);
        }
        __syncthreads();
    }
    for (int i = t; i < L; i += T) la[i] = f.mulpw(la[i], lb[i]);
    __syncthreads();
    for (int s = logL - 1; s >= 0; --s) {
        const int h = half >> s;
        for (int k = t; k < half; k += T) {
            const int j = k & (h - 1);
            const int i = ((k - j) << 1) + j;
            f.dit(la[i], la[i + h], twI[j << s]);
        }
        __syncthreads();
    }
    TOut* dst = out + (size_t)blockIdx.x * out_stride;
    for (int i = t; i < n_out; i += T) dst[i] = (TOut)f.canon_fwd(f.mulw(la[i], scale_m));
}

// ---------------------------------------------------------------------------
// k_small16: whole products for L = 2^logL, 32 <= L <= 4096, when 64 | p-1.  A workgroup of 256 threads holds
// 4096 / L products in LDS (4096 values of A and of B) and runs radix-16 register rounds with one LDS exchange
// between rounds, like the 4096-point core, for every length:
//   round over blocks of 2^b values (b = logL, logL-4, ... > 4): thread t takes values e0 + k * 2^(b-4), k < 16
//   last round (blocks of 2^ql <= 16 values): values 16 t + i.
// The pointwise product and the first inverse round stay in the registers of the last forward round.
// rt.f[b][j*16 + pos] = Montgomery form of w_{2^b}^(j * bitrev4(pos)), rt.i[b] the inverse powers.
// ---------------------------------------------------------------------------
struct RoundTables32 {
    const uint32_t* f[13];
    const uint32_t* i[13];
};

// 2^Q-point transforms on the 16 >> Q consecutive blocks of a 16-entry array
template <int Q, bool LZ>
__device__ __forceinline__ void dif_blocks32(uint32_t (&x)[16], const Field<LZ>& f, const FieldConsts& fc) {
    static_for<0, (16 >> Q)>([&](auto Bk) {
        constexpr int off = decltype(Bk)::value << Q;
        uint32_t y[1 << Q];
        static_for<0, (1 << Q)>([&](auto I) { y[decltype(I)::value] = x[off + decltype(I)::value]; });
        dif_reg<Q, LZ>(y, f, fc.rootF, fc.rootFn);
        static_for<0, (1 << Q)>([&](auto I) { x[off + decltype(I)::value] = y[decltype(I)::value]; });
    });
}
template <int Q, bool LZ>
__device__ __forceinline__ void dit_blocks32(uint32_t (&x)[16], const Field<LZ>& f, const FieldConsts& fc) {
    static_for<0, (16 >> Q)>([&](auto Bk) {
        constexpr int off = decltype(Bk)::value << Q;
        uint32_t y[1 << Q];
        static_for<0, (1 << Q)>([&](auto I) { y[decltype(I)::value] = x[off + decltype(I)::value]; });
        dit_reg<Q, LZ, true>(y, f, fc.rootI);
        static_for<0, (1 << Q)>([&](auto I) { x[off + decltype(I)::value] = y[decltype(I)::value]; });
    });
}

__device__ __forceinline__ int small_slot(int e) { return core_slot(e >> 8, (e >> 4) & 15, e & 15); }

// Register exchanges inside a quad of lanes with DPP permutes (no LDS, no barrier): lanes l and l ^ D trade one register
// of the pair (x0, x1) -- the lane whose bit D is clear gives x1 and takes the partner's x0, the other the other way round.
template <int D>
__device__ __forceinline__ void quad_pair_swap(uint32_t& x0, uint32_t& x1, int lane) {
    [[maybe_unused]] constexpr int ctrl = D == 2 ? 0x4E : 0xB1;              // quad_perm [2,3,0,1] / [1,0,3,2]
    const bool up = (lane & D) != 0;
    const uint32_t give = up ? x0 : x1;
    const uint32_t got = NTT_QUAD_PERM(give, ctrl);
    x0 = up ? got : x0;
    x1 = up ? x1 : got;
}
// 4 x 4 transposition among the four lanes of a quad: r[4g + c] of lane q <-> r[4g + q] of lane c, for the four groups g
__device__ __forceinline__ void quad_transpose16(uint32_t (&r)[16], int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        quad_pair_swap<1>(r[4 * g + 0], r[4 * g + 1], lane); quad_pair_swap<1>(r[4 * g + 2], r[4 * g + 3], lane);
        quad_pair_swap<2>(r[4 * g + 0], r[4 * g + 2], lane); quad_pair_swap<2>(r[4 * g + 1], r[4 * g + 3], lane);
    }
}
// 2 x 2 transposition between lanes l and l ^ 1: r[2m + c] of lane q <-> r[2m + q] of lane c
__device__ __forceinline__ void pair_transpose16(uint32_t (&r)[16], int lane) {
#pragma unroll
    for (int m = 0; m < 8; ++m) quad_pair_swap<1>(r[2 * m], r[2 * m + 1], lane);
}

#ifndef NTT_SMALL16_WAVES
#define NTT_SMALL16_WAVES 1
#endif
#ifndef NTT_SMALL16_DIRECT_FROM
#define NTT_SMALL16_DIRECT_FROM 6
#endif
template <bool LZ, typename TIn, typename TOut>
__global__ void __launch_bounds__(CORE_T, NTT_SMALL16_WAVES) k_small16(const TIn* A, const TIn* Bp, TOut* out, size_t a_stride, size_t b_stride,
                                                    size_t out_stride, int n, int n_out, int logL, size_t batch,
                                                    uint32_t scale_m, FieldConsts fc, RoundTables32 rt) {
    __shared__ uint32_t la[CORE_N];
    __shared__ uint32_t lb[CORE_N];
    const Field<LZ> f(fc);
    const int t = threadIdx.x;
    const int ql = ((logL - 1) & 3) + 1;
    const size_t first = (size_t)blockIdx.x << (12 - logL);     // first product of this workgroup
    uint32_t a[16], b[16];
    // From transform length 64 on, the first forward round takes its values straight from the operands and the last
    // inverse round stores straight to the result (no staging copy through LDS, two barriers fewer, and the padding half
    // of every transform is neither read nor run through its first butterfly level): n = 64 ... 2048 +26 ... +37 %,
    // n = 17 ... 32 +30 ... +46 %.  At length 32 the runs a wavefront reads would be 8 bytes long (n = 16: 13 % slower than
    // the staged copy), so the shortest length keeps the staging.
    const bool direct = logL >= NTT_SMALL16_DIRECT_FROM;
    // first round from the operands: values j + k * sub of product `first + blk`, k < 8 -- the upper half of a transform
    // (k >= 8) is padding for every product (n <= L/2) and is not read
    auto load_first_round = [&](int sub, int blk, int j) {
        const size_t prod = first + (size_t)blk;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = j + k * sub;
            const bool inside = prod < batch && idx < n;
            a[k] = load_coeff<LZ>(A, inside ? prod * a_stride + (size_t)idx : 0, f, fc);
            b[k] = load_coeff<LZ>(Bp, inside ? prod * b_stride + (size_t)idx : 0, f, fc);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool inside = prod < batch && j + k * sub < n;
            a[k] = inside ? a[k] : 0u; b[k] = inside ? b[k] : 0u;
        }
    };
    // last round to the result: its outputs are coefficients j + k * sub of product `first + blk`
    auto store_last_round = [&](int sub, int blk, int j) {
        const size_t prod = first + (size_t)blk;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = j + k * sub;
            if (prod < batch && idx < n_out) out[prod * out_stride + idx] = (TOut)f.canon_fwd(f.mulw(a[k], scale_m));
        }
    };
    if (!direct) {
    // operands -> LDS, zero padded (values beyond n, products beyond the batch).  All thirty-two loads of a thread are
    // requested before the first value is used, and none is conditional (padding reads the operand's first word and is
    // replaced by zero afterwards): loads inside a condition, each followed by its LDS write, waited for one another --
    // sixteen memory round trips in a row, which held every length at 1.4 TB/s.
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int e = t + CORE_T * i;
        NTT_OPAQUE(e);          // or sixteen sets of product number / index / slot stay in registers until the final store
        const size_t prod = first + (size_t)(e >> logL);
        const int idx = e & ((1 << logL) - 1);
        const bool inside = prod < batch && idx < n;
        a[i] = load_coeff<LZ>(A, inside ? prod * a_stride + (size_t)idx : 0, f, fc);
        b[i] = load_coeff<LZ>(Bp, inside ? prod * b_stride + (size_t)idx : 0, f, fc);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int e = t + CORE_T * i;
        NTT_OPAQUE(e);
        const size_t prod = first + (size_t)(e >> logL);
        const int idx = e & ((1 << logL) - 1);
        const bool inside = prod < batch && idx < n;
        const int s = small_slot(e);
        la[s] = inside ? a[i] : 0u;
        lb[s] = inside ? b[i] : 0u;
    }
    __syncthreads();
    }
    if (logL <= 6) {
        // Transform lengths 32 and 64: after one register round over the four high levels a product lives in two or four
        // neighbouring lanes, and the remaining one or two levels, the pointwise product and their inverses run on
        // registers exchanged inside the quad (DPP permutes) instead of two more round trips through LDS
        // (n = 17 .. 32: 3.28 -> 3.49 G products/s).
        const int lb2 = logL - 4, sub = 1 << lb2;          // lanes per product
        const int blk = t >> lb2, j = t & (sub - 1);
        const int e0 = (blk << logL) + j;
        uint32_t tw[16];
        if (direct) {
            load_first_round(sub, blk, j);
            dif_reg<4, LZ, true, true>(a, f, fc.rootF, fc.rootFn);
            dif_reg<4, LZ, true, true>(b, f, fc.rootF, fc.rootFn);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int s = small_slot(e0 + k * sub); a[k] = la[s]; b[k] = lb[s]; }
            dif_reg<4, LZ, false, true>(a, f, fc.rootF, fc.rootFn);
            dif_reg<4, LZ, false, true>(b, f, fc.rootF, fc.rootFn);
        }
        load_tw16(tw, rt.f[logL], j);
#pragma unroll
        for (int pos = 1; pos < 16; ++pos) { a[pos] = f.mulw(a[pos], tw[pos]); b[pos] = f.mulw(b[pos], tw[pos]); }
        if (logL == 6) {
            quad_transpose16(a, t); quad_transpose16(b, t);
            dif_blocks32<2, LZ>(a, f, fc); dif_blocks32<2, LZ>(b, f, fc);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = f.mulpw(a[i], b[i]);
            dit_blocks32<2, LZ>(a, f, fc);
            quad_transpose16(a, t);
        } else {
            pair_transpose16(a, t); pair_transpose16(b, t);
            dif_blocks32<1, LZ>(a, f, fc); dif_blocks32<1, LZ>(b, f, fc);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = f.mulpw(a[i], b[i]);
            dit_blocks32<1, LZ>(a, f, fc);
            pair_transpose16(a, t);
        }
        load_tw16(tw, rt.i[logL], j);
#pragma unroll
        for (int k = 1; k < 16; ++k) a[k] = f.mulw(a[k], tw[k]);
        if (LZ) a[0] = csub(a[0], f.B);
        dit_reg<4, LZ, true>(a, f, fc.rootI);
        if (direct) {
            store_last_round(sub, blk, j);
        } else {
            __syncthreads();     // every thread has taken its operands before the results overwrite them
#pragma unroll
            for (int k = 0; k < 16; ++k) la[small_slot(e0 + k * sub)] = a[k];
        }
    } else {
    // forward rounds
    int bl0 = logL;
    if (direct) {
        // first round from the operands: rows of the upper half (k >= 8) are padding for every product (n <= L/2)
        const int sub = 1 << (logL - 4);
        const int blk = t >> (logL - 4), j = t & (sub - 1);
        const int e0 = (blk << logL) + j;
        load_first_round(sub, blk, j);
        dif_reg<4, LZ, true, true>(a, f, fc.rootF, fc.rootFn);
        dif_reg<4, LZ, true, true>(b, f, fc.rootF, fc.rootFn);
        uint32_t tw[16];
        load_tw16(tw, rt.f[logL], j);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            if (pos) { a[pos] = f.mulw(a[pos], tw[pos]); b[pos] = f.mulw(b[pos], tw[pos]); }
            const int s = small_slot(e0 + pos * sub);
            la[s] = a[pos]; lb[s] = b[pos];
        }
        __syncthreads();
        bl0 = logL - 4;
    }
    for (int bl = bl0; bl > 4; bl -= 4) {
        const int sub = 1 << (bl - 4);
        const int blk = t >> (bl - 4), j = t & (sub - 1);
        const int e0 = (blk << bl) + j;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int s = small_slot(e0 + k * sub); a[k] = la[s]; b[k] = lb[s]; }
        dif_reg<4, LZ, false, true>(a, f, fc.rootF, fc.rootFn);
        dif_reg<4, LZ, false, true>(b, f, fc.rootF, fc.rootFn);
        uint32_t tw[16];
        load_tw16(tw, rt.f[bl], j);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            if (pos) { a[pos] = f.mulw(a[pos], tw[pos]); b[pos] = f.mulw(b[pos], tw[pos]); }
            const int s = small_slot(e0 + pos * sub);
            la[s] = a[pos]; lb[s] = b[pos];
        }
        __syncthreads();
    }
    // last forward round, pointwise product, first inverse round: all on values 16 t + i
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int s = small_slot((t << 4) + i); a[i] = la[s]; b[i] = lb[s]; }
    switch (ql) {
        case 1: dif_blocks32<1, LZ>(a, f, fc); dif_blocks32<1, LZ>(b, f, fc); break;
        case 2: dif_blocks32<2, LZ>(a, f, fc); dif_blocks32<2, LZ>(b, f, fc); break;
        case 3: dif_blocks32<3, LZ>(a, f, fc); dif_blocks32<3, LZ>(b, f, fc); break;
        default: dif_blocks32<4, LZ>(a, f, fc); dif_blocks32<4, LZ>(b, f, fc); break;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f.mulpw(a[i], b[i]);
    switch (ql) {
        case 1: dit_blocks32<1, LZ>(a, f, fc); break;
        case 2: dit_blocks32<2, LZ>(a, f, fc); break;
        case 3: dit_blocks32<3, LZ>(a, f, fc); break;
        default: dit_blocks32<4, LZ>(a, f, fc); break;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) la[small_slot((t << 4) + i)] = a[i];
    // remaining inverse rounds
    for (int bl = ql + 4; bl <= logL; bl += 4) {
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
        if (direct && bl == logL) {
            store_last_round(sub, blk, j);       // scaled, canonical, straight out
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) la[small_slot(e0 + k * sub)] = a[k];
        }
    }
    }
    if (direct) return;
    __syncthreads();
    // scale by L^-1 (and the factor the pointwise product left), canonical, first 2n-1 coefficients of every product
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int e = t + CORE_T * i;
        NTT_OPAQUE(e);
        const size_t prod = first + (size_t)(e >> logL);
        const int idx = e & ((1 << logL) - 1);
        if (prod < batch && idx < n_out) out[prod * out_stride + idx] = (TOut)f.canon_fwd(f.mulw(la[small_slot(e)], scale_m));
    }
}

// ---------------------------------------------------------------------------
// CRT recombination of one coefficient from its four canonical residues s[0..3] (moduli in ascending order: s[i] belongs
// to prime src[i] of the basis):
// the value x in [0, M) with x = r_i mod q_i, then x mod p.  The reference forms S = sum_i M_i * ((r_i * M_i^-1) mod q_i),
// S mod M and then mod p with 128-bit divisions (main_naive.cc:216-223, main_mpi_CRT.cc:512-522); x is unique, so any exact
// way to it gives the same bits.  Here: mixed-radix (Garner) digits with the primes taken in ascending order m0 < m1 < m2 <
// m3 -- six modular multiplies in 32-bit Montgomery arithmetic, x = v0 + m0 (v1 + m1 (v2 + m2 v3)) -- and then
//   p <= 1.5e15:  x mod p = v0 + (m0 mod p) v1 + (m0 m1 mod p) v2 + (m0 m1 m2 mod p) v3 in exact double-precision
//                 arithmetic (the multiply of ntt_field64.h: |v_i| < 2^31 times a balanced constant, quotient below 2^31);
//   larger p:     x as a 128-bit integer by Horner's rule, then a reciprocal-based 128-by-64 remainder (Moeller-Granlund).
// ---------------------------------------------------------------------------
struct CrtConsts {
    uint32_t m[4];        // the four NTT primes in ascending order
    uint32_t nm[4];       // -m^-1 mod 2^32
    int src[4];           // m[i] is prime src[i] of the basis
    uint32_t k1;          // Montgomery form mod m1 of m0^-1
    uint32_t c2, k2;      // mod m2: m0, (m0 m1)^-1
    uint32_t c31, c32, k3;  // mod m3: m0, m0 m1, (m0 m1 m2)^-1
    int fp;               // p <= 1.5e15: the double-precision tail
    double pd, ud;        // p, 1/p
    double d1, d2, d3;    // m0, m0 m1, m0 m1 m2 mod p, balanced
    uint64_t p;           // final modulus
    uint64_t d;           // p << shift (top bit set)
    uint64_t v;           // floor((2^128-1)/d) - 2^64
    int shift;
};

// (u1:u0) mod (d >> shift) for u1 < (d >> shift); d normalised (Moeller-Granlund 2-by-1 division)
__device__ __forceinline__ uint64_t rem_2by1(uint64_t u1, uint64_t u0, uint64_t d, uint64_t v, int shift) {
    if (shift) { u1 = (u1 << shift) | (u0 >> (64 - shift)); u0 <<= shift; }
    uint64_t q0 = v * u1, q1 = __umul64hi(v, u1);
    q0 += u0;
    q1 += u1 + (q0 < u0 ? 1u : 0u);
    q1 += 1;
    uint64_t r = u0 - q1 * d;
    if (r > q0) r += d;
    if (r >= d) r -= d;
    return r >> shift;
}

__device__ __forceinline__ uint64_t crt_merge_value(const uint32_t (&s)[4], const CrtConsts& c) {
    // Garner digits.  Every difference is kept non-negative by adding multiples of the modulus; the primes are below
    // 2^31 and ascending, so the sums stay below 2^32 (3 m3 < 3.7e9).
    const uint32_t s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
    const uint32_t v0 = s0;
    const uint32_t v1 = csub(redc_lazy((uint64_t)(s1 + c.m[1] - v0) * c.k1, c.m[1], c.nm[1]), c.m[1]);
    const uint32_t u2 = csub(redc_lazy((uint64_t)v1 * c.c2, c.m[2], c.nm[2]), c.m[2]);
    const uint32_t v2 = csub(redc_lazy((uint64_t)(s2 + 2 * c.m[2] - v0 - u2) * c.k2, c.m[2], c.nm[2]), c.m[2]);
    const uint32_t u3 = csub(redc_lazy((uint64_t)v1 * c.c31 + (uint64_t)v2 * c.c32, c.m[3], c.nm[3]), c.m[3]);
    const uint32_t v3 = csub(redc_lazy((uint64_t)(s3 + 2 * c.m[3] - v0 - u3) * c.k3, c.m[3], c.nm[3]), c.m[3]);
    if (c.fp) {
        FieldDConsts fdc{};
        fdc.p = c.pd; fdc.u = c.ud;
        const FieldD f(fdc);
        const double acc = (double)v0 + f.mulw((double)v1, c.d1) + f.mulw((double)v2, c.d2) + f.mulw((double)v3, c.d3);
        return f.canon(acc);
    }
    // x = ((v3 m2 + v2) m1 + v1) m0 + v0 < M < 2^119
    const uint64_t t1 = (uint64_t)v3 * c.m[2] + v2;
    uint64_t lo = t1 * c.m[1], hi = __umul64hi(t1, (uint64_t)c.m[1]);
    lo += v1; hi += lo < v1 ? 1u : 0u;
    uint64_t x_lo = lo * c.m[0], x_hi = __umul64hi(lo, (uint64_t)c.m[0]) + hi * c.m[0];
    x_lo += v0; x_hi += x_lo < v0 ? 1u : 0u;
    const uint64_t r1 = rem_2by1(0, x_hi, c.d, c.v, c.shift);
    return rem_2by1(r1, x_lo, c.d, c.v, c.shift);
}

__global__ void __launch_bounds__(256) k_crt_merge(const uint32_t* res, size_t res_stride, uint64_t* out, size_t count,
                                                   size_t rows, size_t out_stride, size_t row_stride, CrtConsts c) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    for (size_t row = blockIdx.y; row < rows; row += gridDim.y) {   // grid.y is capped at the launch limit
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = res[(size_t)c.src[k] * res_stride + row * row_stride + i];
        out[row * out_stride + i] = crt_merge_value(r, c);
    }
}

// ---------------------------------------------------------------------------
// k_inv_strided_crt: the inverse column pass of all four CRT primes and the recombination in one kernel, so that the
// residues never travel through HBM (the reference's merge loop, main_naive.cc:216-223, reads them back from four arrays).
// A workgroup of 4 * CW threads takes CW adjacent columns of one product; its threads [k*CW, (k+1)*CW) run the column
// pass of prime k (a whole number of wavefronts per prime, so the prime is uniform within a wavefront) and leave the
// canonical residues of PH rows at a time in LDS; then every thread recombines PH/4 rows of its column and stores the
// 64-bit coefficients.  src[k]: the core's output of prime k, rows of C << Q words; T3[k], fc[k]: that prime's tables.
// ---------------------------------------------------------------------------
// k_fwd_strided_crt: the forward column pass of all four CRT primes in one launch.  A workgroup of 4 * CW threads takes
// CW adjacent columns of one operand row; its threads [k*CW, (k+1)*CW) reduce the 64-bit coefficients modulo prime k
// (the reference's residue step, main_naive.cc:206-209) and run that prime's column pass.  The workgroup brings the operand
// words of its columns into LDS once; every prime's group reads them from there.
struct CrtFwdArgs {
    const uint32_t* T1[4];
    uint32_t* outA[4];
    uint32_t* outB[4];
    FieldConsts fc[4];
};

template <int Q, int CW, bool FULL, int MINW>
__global__ void __launch_bounds__(4 * CW, MINW) k_fwd_strided_crt(const uint64_t* inA, const uint64_t* inB, size_t a_stride,
                                                                  size_t b_stride, int n_valid, int C, CrtFwdArgs a) {
    static_assert(CW % 64 == 0, "a prime owns whole wavefronts");
    constexpr int R = 1 << Q, RH = R / 2;       // rows of the column, rows that hold coefficients (the rest is padding)
    __shared__ uint64_t raw[RH][CW];
    const int k = NTT_UNIFORM((int)threadIdx.x / CW);
    const int lane = (int)threadIdx.x % CW;
    const int chunks = C / CW;
    const int j0 = (int)(blockIdx.x % chunks) * CW;
    const size_t row = blockIdx.x / chunks;
    const uint64_t* src = blockIdx.z ? inB + row * b_stride : inA + row * a_stride;
    uint32_t* dst = (blockIdx.z ? a.outB[k] : a.outA[k]) + row * ((size_t)C << Q);
    // the operand words of these columns, once for the four primes (zero beyond the operand's length)
#pragma unroll
    for (int e = (int)threadIdx.x; e < RH * CW; e += 4 * CW) {
        const int r = e / CW, col = e % CW;
        const int idx = j0 + col + C * r;
        const bool inside = FULL || idx < n_valid;
        const uint64_t v = src[inside ? idx : 0];
        raw[r][col] = inside ? v : 0;
    }
    __syncthreads();
    // the second prime of the basis is the one above 2^30 (values kept fully reduced); the others run in the lazy ranges
    if (k == 1) {
        const FieldConsts& fc = a.fc[1];
        fwd_column_from<Q, false, true>([&](int r, const Field<false>& f) { return f.reduce64(raw[r][lane], fc.r1, fc.r2); },
                                        dst, C, a.T1[1], fc, j0 + lane);
    } else {
        const FieldConsts& fc = a.fc[k];
        fwd_column_from<Q, true, true>([&](int r, const Field<true>& f) { return f.reduce64(raw[r][lane], fc.r1, fc.r2); },
                                       dst, C, a.T1[k], fc, j0 + lane);
    }
}

struct CrtInvArgs {
    const uint32_t* src[4];
    const uint32_t* T3[4];
    FieldConsts fc[4];
};

template <int Q, int CW, bool FULL>
__global__ void __launch_bounds__(4 * CW) k_inv_strided_crt(CrtInvArgs a, uint64_t* out, size_t out_stride, int n_out, int C,
                                                            CrtConsts c) {
    constexpr int R = 1 << Q;
    constexpr int PH = R < 16 ? R : 16;          // rows per exchange phase
    static_assert(CW % 64 == 0 && PH % 4 == 0, "a prime owns whole wavefronts; every thread recombines PH / 4 rows");
    __shared__ uint32_t xch[4][PH][CW];
    const int k = NTT_UNIFORM((int)threadIdx.x / CW);
    const int lane = (int)threadIdx.x % CW;
    const int chunks = C / CW;
    const int j = (int)(blockIdx.x % chunks) * CW + lane;
    const size_t row = blockIdx.x / chunks;
    uint32_t x[R];
    // the second prime of the basis is the one above 2^30 (values kept fully reduced); the others run in the lazy ranges
    if (k == 1) {
        inv_column_regs<Q, false>(a.src[1] + row * ((size_t)C << Q), C, a.T3[1], a.fc[1], j, x);
    } else {
        const Field<true> f(a.fc[k]);
        inv_column_regs<Q, true>(a.src[k] + row * ((size_t)C << Q), C, a.T3[k], a.fc[k], j, x);
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = f.canon_inv(x[r]);
    }
    uint64_t* dst = out + row * out_stride;
#pragma unroll
    for (int ph = 0; ph < R / PH; ++ph) {
        if (ph) __syncthreads();                  // the previous phase's residues have been read
#pragma unroll
        for (int i = 0; i < PH; ++i) xch[k][i][lane] = x[ph * PH + i];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < PH / 4; ++m) {
            const int i = k * (PH / 4) + m;
            const uint32_t r[4] = {xch[c.src[0]][i][lane], xch[c.src[1]][i][lane], xch[c.src[2]][i][lane], xch[c.src[3]][i][lane]};
            const int idx = j + C * (ph * PH + i);
            if ((FULL && ph * PH + i < R - 1) || idx < n_out) dst[idx] = crt_merge_value(r, c);
        }
    }
}

}  // namespace nttamd
