// Kernels over the 64-bit prime fields (ntt_field64.h): transforms of any power-of-two length along the middle
// axis of an [outer][len][inner] array, the building blocks of the product of one very long polynomial pair
// (single GPU, or sharded over ranks with an all-to-all transpose between the two transform axes).
//
// Forward = in-place decimation in frequency (natural order in, bit-reversed order out), inverse = the mirror
// (bit-reversed in, natural out, unscaled).  A length-2^l transform is a chain of passes over memory, from the
// largest block down; every pass multiplies by the twiddles that join it to the next one:
//
//   k64_tile     a workgroup of 512 threads holds a tile of R = 2^5..2^9 rows x W = 8192/R adjacent columns in
//                LDS (64 KiB) and runs all log2 R levels on it: radix-16 rounds in registers, one LDS exchange
//                between rounds.  Rows are `step` elements apart in memory, columns adjacent (>= 128 bytes).
//   k64_local    contiguous blocks of up to 4096 values (the last 12 levels of an axis with unit stride): copied
//                into LDS, radix-16 rounds, copied out; optionally the pointwise product of two spectra on the
//                way in (the first pass of the inverse transform of a product).
//   k64_strided  register-only radix 2..16 pass for shapes the tiles do not fit (odd inner extents, tiny axes).
//   k64_small    radix-2 stages in LDS for short contiguous axes in arrays smaller than one k64_local chunk.
//
// A product never runs its element-wise steps as passes of their own.  The outermost pass of a transform (k64_tile or
// k64_strided) reads the caller's operands -- any 64-bit words, reduced on the fly, the zero padding never read -- and,
// on the way back, writes the scaled canonical coefficients; and the innermost passes of both operands, the pointwise
// product and the innermost pass of the inverse are one kernel:
//
//   k64_core_mul   chunk c of A and of B: 12 forward levels each, product, 12 inverse levels (transform lengths > 4096)
//   k64_short_mul  whole products of transform length <= 4096, 4096 / length of them per workgroup
//
// Every kernel is a template over the field (Field64: Montgomery integers for p < 2^60; FieldD: exact
// double-precision arithmetic for p <= 1.5e15, about four times fewer issue cycles per butterfly).
//
// Replaces, for moduli that are themselves NTT-friendly, the reference's replicate-and-all-reduce distributed
// transform (main_mpi_t.cc:139-291) and extends the serial transform (main_naive.cc:96-137) beyond the 2-adicity
// of any 32-bit prime.
#pragma once
#include "ntt_field64.h"

namespace nttamd {

constexpr int LOCAL64_LOG = 12;              // log2 of a k64_local chunk (4096 x 8 B = 32 KiB of LDS)
constexpr int LOCAL64_N = 1 << LOCAL64_LOG;
constexpr int LOCAL64_T = 256;               // threads of k64_local: 16 values each
#ifndef NTT_TILE64_LOG
#define NTT_TILE64_LOG 13                    // diagnostic builds may choose 12 (4096-word tiles, 256 threads)
#endif
constexpr int TILE64_LOG = NTT_TILE64_LOG;   // log2 of a k64_tile tile (8192 x 8 B = 64 KiB of LDS)
constexpr int TILE64_N = 1 << TILE64_LOG;
constexpr int TILE64_T = TILE64_N / 16;      // threads of k64_tile: 16 values each
constexpr int TILE64_MIN_LOGR = 5, TILE64_MAX_LOGR = TILE64_LOG - 4;   // at least 16 adjacent columns (128 bytes) per row

// Twiddles between the radix-16 rounds inside a tile or chunk: for block length Bk = 2^b (b = 5..12),
// f[b][j*16 + pos] = w_Bk^(j * bitrev4(pos)), j < Bk/16 (i[b]: the inverse powers), in the field's twiddle form.
struct RoundTables64 {
    const uint64_t* f[14];
    const uint64_t* i[14];
};

// One pass of q levels over blocks of 2^q rows.  Memory: block b starts at b * block_stride; its rows are `step` elements
// apart; a row holds `columns` adjacent "fused columns" x.  Afterwards (forward) / beforehand (inverse) row k of column
// x is multiplied by w^(+-jq(x) * bitrev_q(k) * tw_stride), w the plan's root, where the twiddle index of a column is
//   jq(x) = ((x >> hi_shift) << hi_lshift) + (lo(x) & lo_mask) + jq_off,   lo(x) = x / inner_div or x >> lo_shift.
// A plain transform along the middle axis of [outer][len][inner] has lo = x / inner and nothing else; the passes of a
// four-step slab (columns col0.. of a wider matrix) and of the split layout an all-to-all leaves use the other terms.
struct Pass64 {
    uint64_t columns;       // fused columns per block row
    uint64_t step;          // elements between rows
    uint64_t block_stride;  // elements between blocks
    uint64_t n_blocks;
    uint64_t inner_div;     // 0: lo(x) = x >> lo_shift; else lo(x) = x / inner_div (extents that are not powers of two)
    int lo_shift, hi_shift, hi_lshift;
    uint64_t lo_mask, jq_off;
    uint64_t tw_stride;     // exponent scale to the table's root
    uint64_t tw_mask;       // table root order - 1
    int q;                  // log2 rows
    uint64_t tiles_per_block;   // k64_tile: columns / W
    uint64_t src_valid;     // first pass of a product reading caller data: elements of a source row beyond this index are zero
    // first / last pass of a batch of products (always the outermost pass: one block = one product's work row): block b
    // reads its operand row at src + b * src_stride and writes its result row at out + b * out_stride; the position of an
    // element in the row is its offset inside the block
    uint64_t src_stride, out_stride;
};

// timing experiments only: -DNTT_ABLATE64_MATH compiles the arithmetic out of k64_tile and k64_local (wrong results), which
// leaves the memory floor of their access patterns
// compiler-only barrier: keeps the loads of a later phase from being hoisted over the arithmetic of an earlier one (the
// registers they would occupy cost more in occupancy than the early issue gains)
#define NTT_FENCE64() asm volatile("" ::: "memory")

#ifdef NTT_ABLATE64_MATH
#define NTT_MATH64(...)
#else
#define NTT_MATH64(...) __VA_ARGS__
#endif

NTT_HD uint64_t pass_jq(const Pass64& g, uint64_t x) {
    const uint64_t lo = g.inner_div ? x / g.inner_div : x >> g.lo_shift;
    return ((x >> g.hi_shift) << g.hi_lshift) + (lo & g.lo_mask) + g.jq_off;
}

// ---------------------------------------------------------------------------
// register transforms on 2^Q consecutive entries of a 16-entry array, entries OFF .. OFF + 2^Q - 1
// (fixed roots w16^e from the field constants; DIF: natural in, bit-reversed out; DIT: the mirror)
// ---------------------------------------------------------------------------
// LAZY: the level of trivial butterflies next to the multiplies that follow (forward: the last level) or precede
// (inverse: the first level) the transform does not reduce its results (F::dif1_lazy / F::dit1_lazy state the ranges).
template <class F, int Q, int OFF, bool LAZY = false>
NTT_HD void dif_sub(typename F::V (&x)[16], const F& f, const typename F::W (&root)[8]) {
    constexpr int N = 1 << Q;
    static_for<0, Q>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int h = N >> (s + 1);
        static_for<0, N>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr ((i & h) == 0) {
                constexpr int e = ((i & (h - 1)) << s) << (4 - Q);
                if constexpr (e == 0 && LAZY && h == 1) f.dif1_lazy(x[OFF + i], x[OFF + i + h]);
                else if constexpr (e == 0) f.dif1(x[OFF + i], x[OFF + i + h]);
                else f.dif(x[OFF + i], x[OFF + i + h], root[e]);
            }
        });
    });
}
template <class F, int Q, int OFF, bool LAZY = false>
NTT_HD void dit_sub(typename F::V (&x)[16], const F& f, const typename F::W (&root)[8]) {
    constexpr int N = 1 << Q;
    static_for<0, Q>([&](auto S) {
        constexpr int s = Q - 1 - decltype(S)::value;
        constexpr int h = N >> (s + 1);
        static_for<0, N>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr ((i & h) == 0) {
                constexpr int e = ((i & (h - 1)) << s) << (4 - Q);
                if constexpr (e == 0 && LAZY && h == 1) f.dit1_lazy(x[OFF + i], x[OFF + i + h]);
                else if constexpr (e == 0) f.dit1(x[OFF + i], x[OFF + i + h]);
                else f.dit(x[OFF + i], x[OFF + i + h], root[e]);
            }
        });
    });
}
// the 16 entries as 16 >> Q independent blocks of 2^Q
template <class F, int Q, bool LAZY = false>
NTT_HD void dif_blocks(typename F::V (&x)[16], const F& f, const typename F::W (&root)[8]) {
    static_for<0, (16 >> Q)>([&](auto Bk) { dif_sub<F, Q, (decltype(Bk)::value << Q), LAZY>(x, f, root); });
}
template <class F, int Q, bool LAZY = false>
NTT_HD void dit_blocks(typename F::V (&x)[16], const F& f, const typename F::W (&root)[8]) {
    static_for<0, (16 >> Q)>([&](auto Bk) { dit_sub<F, Q, (decltype(Bk)::value << Q), LAZY>(x, f, root); });
}
template <class F, bool LAZY = false>
NTT_HD void dif_blocks_rt(int q, typename F::V (&x)[16], const F& f, const typename F::W (&root)[8]) {
    switch (q) {
        case 1: dif_blocks<F, 1, LAZY>(x, f, root); break;
        case 2: dif_blocks<F, 2, LAZY>(x, f, root); break;
        case 3: dif_blocks<F, 3, LAZY>(x, f, root); break;
        default: dif_blocks<F, 4, LAZY>(x, f, root); break;
    }
}
template <class F, bool LAZY = false>
NTT_HD void dit_blocks_rt(int q, typename F::V (&x)[16], const F& f, const typename F::W (&root)[8]) {
    switch (q) {
        case 1: dit_blocks<F, 1, LAZY>(x, f, root); break;
        case 2: dit_blocks<F, 2, LAZY>(x, f, root); break;
        case 3: dit_blocks<F, 3, LAZY>(x, f, root); break;
        default: dit_blocks<F, 4, LAZY>(x, f, root); break;
    }
}

template <class F>
NTT_HD void load_roots(typename F::W (&rF)[8], typename F::W (&rI)[8], const typename F::Consts& c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { rF[e] = c.root16F[e]; rI[e] = c.root16I[e]; }
}

NTT_HD uint64_t bitrev64(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1ull) << (bits - 1 - b);
    return r;
}

// ---------------------------------------------------------------------------
// k64_tile: R = 2^q rows (q = 5..9) x W = 8192 >> q adjacent columns per workgroup.
//   thread t: column x = t % W, group grp = t / W (grp < R/16).
//   round over blocks of 2^b rows (b = q, q-4, ... > 4): rows blk*2^b + j + m*2^(b-4), m < 16, (blk, j) from grp
//   last round (blocks of 2^ql <= 16 rows): rows 16 grp + i, i < 16.
// SRC_RAW (forward only): the first round reads caller data (any 64-bit words, rows of g.src_valid elements, zero
//   beyond) from `src` instead of `data`.
// FINAL (inverse only): the last round multiplies by `scale`, reduces to [0,p) and writes the first n_out elements of
//   every row to `out` instead of `data`.
// gridDim.y = 2 (forward only): the same pass over a second array (data2, src2, src_stride2) in the same launch -- the two
//   operands of a product; one ramp-up and one tail instead of two.
// ---------------------------------------------------------------------------
template <class F, bool INVERSE, bool SRC_RAW, bool FINAL>
__global__ void __launch_bounds__(TILE64_T, 4) k64_tile(uint64_t* data, const uint64_t* src, uint64_t* out, uint64_t n_out,
                                                     typename F::W scale, Pass64 g, typename F::Consts fc, Twiddle64 tw,
                                                     RoundTables64 rt, uint64_t* data2, const uint64_t* src2, uint64_t src_stride2) {
    typedef typename F::V V;
    typedef typename F::W W;
    __shared__ V lds[TILE64_N];
    // forward passes of the two operands of a product in one launch: blockIdx.y = 1 works on the second array
    uint64_t src_stride = g.src_stride;
    if (!INVERSE && blockIdx.y) { data = data2; src = src2; src_stride = src_stride2; }
    const F f(fc);
    W rF[8], rI[8];
    load_roots<F>(rF, rI, fc);
    const int q = g.q;
    const int logW = TILE64_LOG - q;
    const int t = threadIdx.x;
    const int x = t & ((1 << logW) - 1), grp = t >> logW;
    const uint64_t tile = blockIdx.x;
    const uint64_t blk_g = tile / g.tiles_per_block, xt = tile % g.tiles_per_block;
    const uint64_t step = g.step;                               // elements between rows
    const uint64_t col = (xt << logW) + (uint64_t)x;            // fused column inside the block
    const uint64_t origin = blk_g * g.block_stride + col;       // element offset of row 0 of this column
    const uint64_t jq = pass_jq(g, col);
    const int ql = ((q - 1) & 3) + 1;                           // levels of the last (smallest) round, 1..4
    // LDS slot of (row, this thread's column).  64-bit LDS accesses are served 32 lanes at a time: with 16 columns per
    // row two rows share an access, and in the rounds whose rows lie 16 apart they would land on the same banks, so
    // row bit 4 is folded into slot bit 4 there.
    const int fold = logW == 4 ? 1 : 0;
    auto tile_slot = [&](int row) -> int { return ((row << logW) + x) ^ ((((row >> 4) & 1) & fold) << 4); };
    V v[16];

    // Pass twiddles of the 16 rows 16*grp + i that a thread holds in the last forward / first inverse round: row r holds
    // frequency k = bitrev_q(r) = bitrev4(i) << (q-4) | bitrev_{q-4}(grp), so the exponents jq * k * tw_stride are
    // e0 + bitrev4(i) * de with two per-thread constants.  Only w^e0 and w^de are looked up (lanes hold different
    // columns, so every lookup is a gather over the table); the sixteen twiddles w^(e0 + c*de) are then built by
    // multiplication: three squarings of w^de and one multiply per twiddle.
    // Memory requests go out as early as possible and in batches: the thread's sixteen data words first, then the four table
    // words of the pass twiddles and the round twiddles of the first round; the chain of multiplications that builds the
    // pass twiddles runs while the data is on its way, and every later round asks for its twiddle row before the barrier
    // in front of it.  Compiler-only fences pin that order.
    W ptw[16];   // indexed by c = bitrev4(i)
    uint64_t ptw_word[4];
    auto request_pass_twiddles = [&]() {
        const uint64_t k0 = bitrev64((uint64_t)grp, q - 4);
        uint64_t e0 = (jq * k0 * g.tw_stride) & g.tw_mask, de = ((jq << (q - 4)) * g.tw_stride) & g.tw_mask;
        if (INVERSE) { e0 = (0 - e0) & g.tw_mask; de = (0 - de) & g.tw_mask; }
        const uint64_t lo_mask = ((uint64_t)1 << tw.t0) - 1;
        ptw_word[0] = tw.w1[de >> tw.t0]; ptw_word[1] = tw.w0[de & lo_mask];
        ptw_word[2] = tw.w1[e0 >> tw.t0]; ptw_word[3] = tw.w0[e0 & lo_mask];
    };
    auto build_pass_twiddles = [&]() {
        const W s1 = f.twmul(f.tw_from_mem(ptw_word[0]), f.tw_from_mem(ptw_word[1]));
        const W s2 = f.twmul(s1, s1), s4 = f.twmul(s2, s2), s8 = f.twmul(s4, s4);
        ptw[0] = f.twmul(f.tw_from_mem(ptw_word[2]), f.tw_from_mem(ptw_word[3]));
        static_for<1, 16>([&](auto C) {
            constexpr int c = decltype(C)::value;
            constexpr int low = c & -c;                       // lowest set bit: ptw[c] = ptw[c - low] * w^(low * de)
            ptw[c] = f.twmul(ptw[c - low], low == 1 ? s1 : low == 2 ? s2 : low == 4 ? s4 : s8);
        });
    };
    auto pass_twiddle = [&](auto I) -> W { return ptw[bitrev(decltype(I)::value, 4)]; };
    uint64_t rtw[16];   // round twiddles of the coming round (table words)
    auto request_round_twiddles = [&](const uint64_t* table, int j) {
        const uint64_t* T = table + ((size_t)j << 4);
#pragma unroll
        for (int pos = 1; pos < 16; ++pos) rtw[pos] = T[pos];
    };

    if (!INVERSE) {
        uint64_t word[16];
        bool inside[16];
        const bool upper_empty = SRC_RAW && g.src_valid <= (step << (q - 1));   // uniform: no column has a word in rows >= R/2
        if (q > 4) {      // first round reads memory: rows row0 + m * sub
            const int b = q, sub = 1 << (b - 4);
            const int blk = grp >> (b - 4), j = grp & (sub - 1);
            const int row0 = (blk << b) + j;
            uint64_t idx = origin + (uint64_t)row0 * step;
            const uint64_t dstep = (uint64_t)sub * step;
            const uint64_t* src_row = SRC_RAW ? src + blk_g * src_stride : nullptr;   // uniform over the workgroup
            uint64_t pos = col + (uint64_t)row0 * step;                                   // offset inside the block
            if (SRC_RAW) {
                // The address is always a valid one and the zero is selected afterwards: a conditional load is compiled
                // as a branch per element, which ends the basic block and makes every load wait for the one before it.
                // Rows from the middle of the block on are padding for every column when the operand is at most half the
                // block long (always, when the block is a product's whole transform): they are not even requested then.
#pragma unroll
                for (int m = 0; m < 16; ++m, pos += dstep) {
                    inside[m] = pos < g.src_valid;
                    if (m < 8 || !upper_empty) word[m] = src_row[inside[m] ? pos : 0];
                }
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m, idx += dstep) word[m] = data[idx];
            }
            request_pass_twiddles();      // used after the last round only: the table words wait in eight registers, the
            request_round_twiddles(rt.f[b], j);   // sixteen twiddles are built there
            NTT_FENCE64();
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (SRC_RAW) {
                    if (m < 8 || !upper_empty) { const V r = f.from_raw(word[m]); v[m] = inside[m] ? r : V(0); }
                } else {
                    v[m] = f.from_mem(word[m]);
                }
            }
        }
        for (int b = q; b > 4; b -= 4) {         // rounds over blocks of 2^b rows
            const int sub = 1 << (b - 4);
            const int blk = grp >> (b - 4), j = grp & (sub - 1);
            const int row0 = (blk << b) + j;
            if (b < q) {
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = lds[tile_slot(row0 + m * sub)];
            }
            if (SRC_RAW && upper_empty && b == q) {
                // first level against zeros: (u, 0) -> (u, u * w16^i); then the two 8-point halves
                NTT_MATH64(static_for<1, 8>([&](auto I) { constexpr int i = decltype(I)::value; v[8 + i] = f.twiddled(v[i], rF[i]); });
                           v[8] = v[0];
                           dif_sub<F, 3, 0, true>(v, f, rF); dif_sub<F, 3, 8, true>(v, f, rF); v[0] = f.settle(v[0]));
            } else {
                NTT_MATH64(dif_blocks<F, 4, true>(v, f, rF); v[0] = f.settle(v[0]));
            }
#pragma unroll
            for (int pos = 1; pos < 16; ++pos) NTT_MATH64(v[pos] = f.twiddled(v[pos], f.tw_from_mem(rtw[pos])));
#pragma unroll
            for (int m = 0; m < 16; ++m) lds[tile_slot(row0 + m * sub)] = v[m];
            if (b - 4 > 4) {                     // the next round's twiddle row travels while the workgroup synchronises
                NTT_FENCE64();
                request_round_twiddles(rt.f[b - 4], grp & ((1 << (b - 8)) - 1));
                NTT_FENCE64();
            }
            __syncthreads();
        }
        // last round: rows 16*grp + i, blocks of 2^ql rows; then the pass twiddle and the store
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = lds[tile_slot((grp << 4) + i)];
        NTT_MATH64(dif_blocks_rt<F>(ql, v, f, rF));
        build_pass_twiddles();
        uint64_t idx = origin + (uint64_t)(grp << 4) * step;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            NTT_MATH64(v[i] = f.twiddled(v[i], pass_twiddle(I)));
            data[idx + (uint64_t)i * step] = f.to_mem(v[i]);
        });
    } else {
        // first round of the inverse = mirror of the last forward round
        uint64_t idx = origin + (uint64_t)(grp << 4) * step;
        uint64_t word[16];
#pragma unroll
        for (int i = 0; i < 16; ++i, idx += step) word[i] = data[idx];
        request_pass_twiddles();
        if (ql + 4 <= q) request_round_twiddles(rt.i[ql + 4], grp & ((1 << ql) - 1));
        NTT_FENCE64();
        build_pass_twiddles();
        NTT_FENCE64();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = f.from_mem(word[i]);
        static_for<0, 16>([&](auto I) { constexpr int i = decltype(I)::value; NTT_MATH64(v[i] = f.twiddled(v[i], pass_twiddle(I))); });
        NTT_MATH64(dit_blocks_rt<F, true>(ql, v, f, rI));
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[tile_slot((grp << 4) + i)] = v[i];
        __syncthreads();
        for (int b = ql + 4; b <= q; b += 4) {
            const int sub = 1 << (b - 4);
            const int blk = grp >> (b - 4), j = grp & (sub - 1);
            const int row0 = (blk << b) + j;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const V r = lds[tile_slot(row0 + m * sub)];
                v[m] = r;
                NTT_MATH64(v[m] = m ? f.twiddled(r, f.tw_from_mem(rtw[m])) : f.settle(r));
            }
            NTT_MATH64(dit_blocks<F, 4, true>(v, f, rI));
            if (b < q) {
#pragma unroll
                for (int m = 0; m < 16; ++m) lds[tile_slot(row0 + m * sub)] = v[m];
                NTT_FENCE64();
                request_round_twiddles(rt.i[b + 4], grp & ((1 << b) - 1));
                NTT_FENCE64();
                __syncthreads();
            } else {
                uint64_t o = origin + (uint64_t)row0 * step;
                const uint64_t dstep = (uint64_t)sub * step;
                uint64_t* out_row = FINAL ? out + blk_g * g.out_stride : nullptr;              // uniform over the workgroup
                uint64_t pos = col + (uint64_t)row0 * step;
#pragma unroll
                for (int m = 0; m < 16; ++m, o += dstep, pos += dstep) {
                    if (FINAL) {
                        if (pos < n_out) out_row[pos] = f.canon(f.twiddled(v[m], scale));
                    } else {
                        data[o] = f.to_mem(v[m]);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k64_local: chunks of 4096 consecutive values holding 4096 >> m complete blocks of 2^m values each (m = 1..12);
// all m levels inside LDS.  LDS slot of element e is e with bits 4-7 XORed into bits 0-3 and bit 8 into bit 4, which
// gives every group of 32 consecutive lanes 32 different 8-byte slots modulo 32 in the three access patterns of a
// full chunk (copy in/out by consecutive threads, rounds with a stride of 256 and of 16, 16 consecutive values per
// thread): 64-bit LDS accesses are served 32 lanes at a time.
// MUL (inverse only): the chunk starts as the pointwise product of `data` and `other`.
// FINAL (inverse only): multiply by `scale`, canonical, elements below n_out to `out`.
// ---------------------------------------------------------------------------
NTT_HD int local_slot(int e) { return e ^ ((e >> 4) & 15) ^ (((e >> 8) & 1) << 4); }

template <class F, bool INVERSE, bool MUL, bool FINAL>
__global__ void __launch_bounds__(LOCAL64_T) k64_local(uint64_t* data, const uint64_t* other, uint64_t* out, uint64_t n_out,
                                                       typename F::W scale, int m, uint64_t n_chunks,
                                                       typename F::Consts fc, RoundTables64 rt) {
    typedef typename F::V V;
    typedef typename F::W W;
    __shared__ V lds[LOCAL64_N];
    const F f(fc);
    W rF[8], rI[8];
    load_roots<F>(rF, rI, fc);
    const int t = threadIdx.x;
    const int ql = ((m - 1) & 3) + 1;
    V v[16];
    // Full-length chunks (m = 12): the first forward round reads memory in the order its threads need anyway (values
    // t + 256 k) and the last inverse round writes in it, so the copy through LDS is skipped on that side.
    const bool direct = m == LOCAL64_LOG;
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        uint64_t* base = data + chunk * (uint64_t)LOCAL64_N;
        const uint64_t* base_b = MUL ? other + chunk * (uint64_t)LOCAL64_N : nullptr;
        auto fetch = [&](int e) -> V {
            V r = f.from_mem(base[e]);
            if (MUL) r = f.mulpw(r, f.from_mem(base_b[e]));
            return r;
        };
        auto emit = [&](int e, V r) {
            if (FINAL) {
                const uint64_t idx = chunk * (uint64_t)LOCAL64_N + e;
                if (idx < n_out) out[idx] = f.canon(f.twiddled(r, scale));
            } else {
                base[e] = f.to_mem(r);
            }
        };
        if (INVERSE || !direct) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int e = t + LOCAL64_T * i; lds[local_slot(e)] = fetch(e); }
            __syncthreads();
        }
        if (!INVERSE) {
            for (int b = m; b > 4; b -= 4) {
                const int sub = 1 << (b - 4);
                const int blk = t >> (b - 4), j = t & (sub - 1);
                const int e0 = (blk << b) + j;
                if (direct && b == m) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = fetch(e0 + k * sub);
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = lds[local_slot(e0 + k * sub)];
                }
                NTT_MATH64(dif_blocks<F, 4, true>(v, f, rF); v[0] = f.settle(v[0]));
                const uint64_t* T = rt.f[b] + ((size_t)j << 4);
#pragma unroll
                for (int pos = 1; pos < 16; ++pos) NTT_MATH64(v[pos] = f.twiddled(v[pos], f.tw_from_mem(T[pos])));
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[local_slot(e0 + k * sub)] = v[k];
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lds[local_slot((t << 4) + i)];
            NTT_MATH64(dif_blocks_rt<F, true>(ql, v, f, rF));      // unreduced results: the store reduces
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[local_slot((t << 4) + i)] = v[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int e = t + LOCAL64_T * i; emit(e, lds[local_slot(e)]); }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lds[local_slot((t << 4) + i)];
            NTT_MATH64(dit_blocks_rt<F, true>(ql, v, f, rI));
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[local_slot((t << 4) + i)] = v[i];
            bool stored = false;
            for (int b = ql + 4; b <= m; b += 4) {
                __syncthreads();
                const int sub = 1 << (b - 4);
                const int blk = t >> (b - 4), j = t & (sub - 1);
                const int e0 = (blk << b) + j;
                const uint64_t* T = rt.i[b] + ((size_t)j << 4);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const V r = lds[local_slot(e0 + k * sub)];
                    v[k] = r;
                    NTT_MATH64(v[k] = k ? f.twiddled(r, f.tw_from_mem(T[k])) : f.settle(r));
                }
                NTT_MATH64(dit_blocks<F, 4, true>(v, f, rI));
                if (direct && b == m) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) emit(e0 + k * sub, v[k]);
                    stored = true;
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) lds[local_slot(e0 + k * sub)] = v[k];
                }
            }
            if (!stored) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 16; ++i) { const int e = t + LOCAL64_T * i; emit(e, lds[local_slot(e)]); }
            }
        }
        __syncthreads();   // the next chunk overwrites the LDS
    }
}

// ---------------------------------------------------------------------------
// k64_core_mul: the middle of a product whose transforms end in full chunks (12 levels on 4096 consecutive values):
// forward levels of chunk c of `data`, the same for `other`, pointwise product and the inverse levels, result back
// into `data` -- one read of each spectrum-to-be and one write instead of three round trips through memory.  The first
// forward round reads memory and the last inverse round writes it in the order their threads need anyway; A's
// transformed values wait in registers while B goes through the same LDS buffer, and the product enters the inverse
// from the registers it was formed in.  The butterflies here keep their reductions (no LAZY levels): with A's values
// parked the kernel sits at 125 registers = four waves per SIMD, the unreduced variants need 145-196 and measured
// the same or slower (160 us against 160-175 us for 64 products of 2^18 points).
// ---------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(LOCAL64_T) k64_core_mul(uint64_t* data, const uint64_t* other, uint64_t n_chunks,
                                                          typename F::Consts fc, RoundTables64 rt) {
    typedef typename F::V V;
    typedef typename F::W W;
    __shared__ V lds[LOCAL64_N];
    const F f(fc);
    W rF[8], rI[8];
    load_roots<F>(rF, rI, fc);
    const int t = threadIdx.x;
    constexpr int m = LOCAL64_LOG;
    V v[16], a[16];
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        uint64_t* base = data + chunk * (uint64_t)LOCAL64_N;
        const uint64_t* base_b = other + chunk * (uint64_t)LOCAL64_N;
        auto forward = [&](const uint64_t* src) {
            for (int b = m; b > 4; b -= 4) {
                NTT_FENCE64();
                const int sub = 1 << (b - 4);
                const int blk = t >> (b - 4), j = t & (sub - 1);
                const int e0 = (blk << b) + j;
                if (b == m) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = f.from_mem(src[e0 + k * sub]);
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = lds[local_slot(e0 + k * sub)];
                }
                NTT_MATH64(dif_blocks<F, 4>(v, f, rF));
                const uint64_t* T = rt.f[b] + ((size_t)j << 4);
#pragma unroll
                for (int pos = 1; pos < 16; ++pos) NTT_MATH64(v[pos] = f.twiddled(v[pos], f.tw_from_mem(T[pos])));
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[local_slot(e0 + k * sub)] = v[k];
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lds[local_slot((t << 4) + i)];
            NTT_MATH64(dif_blocks<F, 4>(v, f, rF));
        };
        forward(base);
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = v[i];
        NTT_FENCE64();
        __syncthreads();            // every thread has taken its sixteen values before the next first round overwrites them
        forward(base_b);
        NTT_FENCE64();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = f.mulpw(a[i], v[i]);
        NTT_MATH64(dit_blocks<F, 4>(v, f, rI));
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[local_slot((t << 4) + i)] = v[i];   // the thread's own slots: no barrier needed before
        for (int b = 8; b <= m; b += 4) {
            __syncthreads();
            NTT_FENCE64();
            const int sub = 1 << (b - 4);
            const int blk = t >> (b - 4), j = t & (sub - 1);
            const int e0 = (blk << b) + j;
            const uint64_t* T = rt.i[b] + ((size_t)j << 4);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const V r = lds[local_slot(e0 + k * sub)];
                v[k] = r;
                NTT_MATH64(if (k) v[k] = f.twiddled(r, f.tw_from_mem(T[k])));
            }
            NTT_MATH64(dit_blocks<F, 4>(v, f, rI));
            if (b == m) {
#pragma unroll
                for (int k = 0; k < 16; ++k) base[e0 + k * sub] = f.to_mem(v[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[local_slot(e0 + k * sub)] = v[k];
            }
        }
        __syncthreads();   // the next chunk overwrites the LDS
    }
}

// ---------------------------------------------------------------------------
// k64_short_mul: whole products of transform length 2^m <= 4096 in one kernel, 4096 >> m of them per workgroup: operands
// from caller memory (any words, rows of n values, zero padded to the transform length), forward levels of A and of B
// in the same LDS buffer (A's values wait in registers), pointwise product, inverse levels, scaled canonical results
// (n_out per row) to caller memory.  Same rounds as k64_local for chunks of several blocks.
// ---------------------------------------------------------------------------
// DIRECT (transform lengths from 64 on): the first round reads the operands itself -- words j + k * sub of product
// row0 + blk, k < 8: the upper half of every transform is padding (n <= length / 2), so it is neither read nor converted
// and the first butterfly level is a multiply -- and the last inverse round writes the results itself; otherwise both ends
// are staged through LDS (the runs a wavefront would read are too short).  A template parameter, not a run-time switch,
// and compiled for three waves per SIMD: at four the double-precision flavour spilled 40 registers and lost a quarter
// of its rate; at three it gains 10 ... 49 % over the staged form.
template <class F, bool DIRECT>
__global__ void __launch_bounds__(LOCAL64_T, DIRECT ? 3 : F::SHORT_MUL_WAVES) k64_short_mul(const uint64_t* a, uint64_t a_stride, const uint64_t* b, uint64_t b_stride,
                                                           uint64_t* out, uint64_t out_stride, uint64_t n, uint64_t n_out, uint64_t rows,
                                                           int m, typename F::W scale, typename F::Consts fc, RoundTables64 rt) {
    typedef typename F::V V;
    typedef typename F::W W;
    __shared__ V lds[LOCAL64_N];
    const F f(fc);
    W rF[8], rI[8];
    load_roots<F>(rF, rI, fc);
    const int t = threadIdx.x;
    const int ql = ((m - 1) & 3) + 1;
    const uint64_t pos_mask = ((uint64_t)1 << m) - 1;
    const uint64_t per_chunk = (uint64_t)LOCAL64_N >> m;                 // products per workgroup
    const uint64_t n_chunks = (rows + per_chunk - 1) / per_chunk;
    V v[16], va[16];
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t row0 = chunk * per_chunk;
        auto forward = [&](const uint64_t* src, uint64_t stride) {
            NTT_FENCE64();
            int first_bb = m;
            if constexpr (DIRECT) {
                const int sub = 1 << (m - 4);
                const int blk = t >> (m - 4), j = t & (sub - 1);
                const int e0 = (blk << m) + j;
                const uint64_t row = row0 + (uint64_t)blk;
                for (int k0 = 0; k0 < 8; k0 += 4) {      // four loads in flight at a time: the conversions need registers of their own
                    uint64_t word[4];
                    NTT_FENCE64();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint64_t pos = (uint64_t)(j + (k0 + k) * sub);
                        word[k] = src[(row < rows && pos < n) ? row * stride + pos : 0];
                    }
                    NTT_FENCE64();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const V r = f.from_raw(word[k]);
                        v[k0 + k] = (row < rows && (uint64_t)(j + (k0 + k) * sub) < n) ? r : V(0);
                    }
                }
                static_for<1, 8>([&](auto I) { constexpr int i = decltype(I)::value; v[8 + i] = f.twiddled(v[i], rF[i]); });
                v[8] = v[0];
                dif_sub<F, 3, 0>(v, f, rF); dif_sub<F, 3, 8>(v, f, rF);
                const uint64_t* T = rt.f[m] + ((size_t)j << 4);
#pragma unroll
                for (int pos = 1; pos < 16; ++pos) v[pos] = f.twiddled(v[pos], f.tw_from_mem(T[pos]));
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[local_slot(e0 + k * sub)] = v[k];
                __syncthreads();
                first_bb = m - 4;
            } else {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {     // four loads in flight at a time: the conversions need registers of their own
                int e = t + LOCAL64_T * i;
                NTT_OPAQUE(e);
                const uint64_t row = row0 + ((uint64_t)e >> m), pos = (uint64_t)e & pos_mask;
                // (conditional loads, i.e. a branch per element: measured 8-10 % faster here than loading unconditionally and
                // selecting the zero afterwards -- this kernel is bound by vector issue, not by the latency of its loads)
                lds[local_slot(e)] = (row < rows && pos < n) ? f.from_raw(src[row * stride + pos]) : V(0);
            }
            __syncthreads();
            }
            for (int bb = first_bb; bb > 4; bb -= 4) {
                NTT_FENCE64();
                const int sub = 1 << (bb - 4);
                const int blk = t >> (bb - 4), j = t & (sub - 1);
                const int e0 = (blk << bb) + j;
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = lds[local_slot(e0 + k * sub)];
                dif_blocks<F, 4>(v, f, rF);
                const uint64_t* T = rt.f[bb] + ((size_t)j << 4);
#pragma unroll
                for (int pos = 1; pos < 16; ++pos) v[pos] = f.twiddled(v[pos], f.tw_from_mem(T[pos]));
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[local_slot(e0 + k * sub)] = v[k];
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lds[local_slot((t << 4) + i)];
            dif_blocks_rt<F>(ql, v, f, rF);
        };
        forward(a, a_stride);
#pragma unroll
        for (int i = 0; i < 16; ++i) va[i] = v[i];
        NTT_FENCE64();
        __syncthreads();            // every thread has taken its sixteen values before the buffer is filled again
        forward(b, b_stride);
        NTT_FENCE64();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = f.mulpw(va[i], v[i]);
        dit_blocks_rt<F>(ql, v, f, rI);
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[local_slot((t << 4) + i)] = v[i];   // the thread's own slots
        for (int bb = ql + 4; bb <= m; bb += 4) {
            __syncthreads();
            NTT_FENCE64();
            const int sub = 1 << (bb - 4);
            const int blk = t >> (bb - 4), j = t & (sub - 1);
            const int e0 = (blk << bb) + j;
            const uint64_t* T = rt.i[bb] + ((size_t)j << 4);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const V r = lds[local_slot(e0 + k * sub)];
                v[k] = k ? f.twiddled(r, f.tw_from_mem(T[k])) : r;
            }
            dit_blocks<F, 4>(v, f, rI);
            if (DIRECT && bb == m) {
                const uint64_t row = row0 + (uint64_t)blk;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint64_t pos = (uint64_t)(j + k * sub);
                    if (row < rows && pos < n_out) out[row * out_stride + pos] = f.canon(f.twiddled(v[k], scale));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[local_slot(e0 + k * sub)] = v[k];
            }
        }
        if constexpr (!DIRECT) {
        __syncthreads();
        NTT_FENCE64();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            int e = t + LOCAL64_T * i;
            NTT_OPAQUE(e);
            const uint64_t row = row0 + ((uint64_t)e >> m), pos = (uint64_t)e & pos_mask;
            if (row < rows && pos < n_out) out[row * out_stride + pos] = f.canon(f.twiddled(lds[local_slot(e)], scale));
        }
        }
        __syncthreads();   // the next chunk overwrites the LDS
    }
}

// ---------------------------------------------------------------------------
// k64_strided: register-only pass, radix 2^Q (Q <= 4), one thread per column; any inner extent.
// gridDim.y = 2 (forward only): the same pass over a second array in the same launch, as in k64_tile.
// SRC_RAW (forward) / FINAL (inverse): as in k64_tile -- caller words in, scaled canonical values out, rows of a batch
// addressed through the block index and the strides.
// ---------------------------------------------------------------------------
template <class F, int Q, bool INVERSE, bool SRC_RAW = false, bool FINAL = false>
__global__ void __launch_bounds__(256) k64_strided(uint64_t* data, const uint64_t* src, uint64_t* out, uint64_t n_out,
                                                   typename F::W scale, Pass64 g, typename F::Consts fc, Twiddle64 tw,
                                                   uint64_t* data2, const uint64_t* src2, uint64_t src_stride2) {
    typedef typename F::V V;
    typedef typename F::W W;
    constexpr int R = 1 << Q;
    // forward passes of the two operands of a product in one launch: blockIdx.y = 1 works on the second array
    uint64_t src_stride = g.src_stride;
    if (!INVERSE && blockIdx.y) { data = data2; src = src2; src_stride = src_stride2; }
    const F f(fc);
    W rF[8], rI[8];
    load_roots<F>(rF, rI, fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= g.n_blocks * g.columns) return;
    const uint64_t col = tid % g.columns, blk = tid / g.columns;
    const uint64_t jq = pass_jq(g, col);
    const uint64_t origin = blk * g.block_stride + col;
    uint64_t* base = data + origin;
    const uint64_t step = g.step;
    V x[16];
    // Everything this thread reads -- its column and the table words of its twiddles -- is requested before anything is
    // used, and nothing is read conditionally (a conditional load is compiled as a branch, which ends the basic block and
    // makes every load wait for the one before it): an operand row's padding is replaced by zero after the load, and a
    // twiddle with exponent zero is looked up like any other (both table words are then the constant one).
    if (SRC_RAW) {
        uint64_t word[R];
        bool inside[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t pos = col + (uint64_t)r * step;
            inside[r] = pos < g.src_valid;
            word[r] = src[blk * src_stride + (inside[r] ? pos : 0)];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const V v = f.from_raw(word[r]);
            x[r] = inside[r] ? v : V(0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = f.from_mem(base[(uint64_t)r * step]);
    }
    uint64_t t1[R], t0[R];
    {
        const uint64_t lo_mask = ((uint64_t)1 << tw.t0) - 1;
        static_for<1, R>([&](auto P) {
            constexpr int pos = decltype(P)::value;
            constexpr int k = bitrev(pos, Q);
            uint64_t e = jq * (uint64_t)k * g.tw_stride;
            if (INVERSE) e = 0 - e;
            e &= g.tw_mask;
            t1[pos] = tw.w1[e >> tw.t0];
            t0[pos] = tw.w0[e & lo_mask];
        });
    }
    auto pass_twiddle = [&](int pos) -> W { return f.twmul(f.tw_from_mem(t1[pos]), f.tw_from_mem(t0[pos])); };
    if (!INVERSE) {
        dif_sub<F, Q, 0>(x, f, rF);
        static_for<1, R>([&](auto P) { constexpr int pos = decltype(P)::value; x[pos] = f.twiddled(x[pos], pass_twiddle(pos)); });
    } else {
        static_for<1, R>([&](auto P) { constexpr int pos = decltype(P)::value; x[pos] = f.twiddled(x[pos], pass_twiddle(pos)); });
        dit_sub<F, Q, 0>(x, f, rI);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (FINAL) {
            const uint64_t pos = col + (uint64_t)r * step;
            if (pos < n_out) out[blk * g.out_stride + pos] = f.canon(f.twiddled(x[r], scale));
        } else {
            base[(uint64_t)r * step] = f.to_mem(x[r]);
        }
    }
}

// contiguous blocks of 2^m <= 4096 values, radix-2 stages in LDS with a barrier per stage: the fall-back for arrays
// that do not fill whole k64_local chunks.
template <class F, bool INVERSE>
__global__ void __launch_bounds__(256) k64_small(uint64_t* data, int m, uint64_t n_blocks, typename F::Consts fc, Twiddle64 tw) {
    typedef typename F::V V;
    __shared__ V lds[LOCAL64_N];
    const F f(fc);
    const int N = 1 << m, half = N >> 1, t = threadIdx.x, T = blockDim.x;
    const uint64_t scale = (uint64_t)1 << (tw.logtab - m);
    const uint64_t mask = ((uint64_t)1 << tw.logtab) - 1;
    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        uint64_t* src = data + blk * (uint64_t)N;
        for (int i = t; i < N; i += T) lds[i] = f.from_mem(src[i]);
        __syncthreads();
        if (!INVERSE) {
            for (int s = 0; s < m; ++s) {
                const int h = half >> s;
                for (int k = t; k < half; k += T) {
                    const int j = k & (h - 1), i = ((k - j) << 1) + j;
                    const uint64_t e = ((uint64_t)(j << s) * scale) & mask;
                    V u = lds[i], v = lds[i + h];
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
                    V u = lds[i], v = lds[i + h];
                    if (e == 0) f.dit1(u, v); else f.dit(u, v, twiddle_pow(tw, f, e));
                    lds[i] = u; lds[i + h] = v;
                }
                __syncthreads();
            }
        }
        for (int i = t; i < N; i += T) src[i] = f.to_mem(lds[i]);
        __syncthreads();
    }
}

// dst[row][i] = i < n ? src[row][i] mod p : 0   (zero padding + reduction of arbitrary 64-bit input; canonical words)
template <class F>
__global__ void __launch_bounds__(256) k64_load(const uint64_t* src, size_t src_stride, uint64_t n, uint64_t* dst, uint64_t len,
                                                uint64_t rows, typename F::Consts fc) {
    const F f(fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= rows * len) return;
    const uint64_t row = tid / len, i = tid % len;
    dst[tid] = i < n ? f.canon(f.from_raw(src[row * src_stride + i])) : 0;
}

// a[i] = a[i] * b[i] (times the field's pointwise factor: 2^-64 for Field64, 1 for FieldD)
template <class F>
__global__ void __launch_bounds__(256) k64_pointwise(uint64_t* a, const uint64_t* b, uint64_t count, typename F::Consts fc) {
    const F f(fc);
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) a[i] = f.to_mem(f.mulpw(f.from_mem(a[i]), f.from_mem(b[i])));
}

// data[r][c] *= w^(+-(row0 + bitrev?(r)) * (col0 + c)) for a rows x cols tile of the four-step
// matrix; w = root of order 2^log_total.  ROWS_BITREV: row r of the tile holds frequency
// bitrev(row0 + r, row_bits) (the layout a forward transform along the row axis leaves).
template <class F>
__global__ void __launch_bounds__(256) k64_twiddle(uint64_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0,
                                                   int row_bits, int rows_bitrev, int inverse, uint64_t tw_stride,
                                                   uint64_t tw_mask, typename F::Consts fc, Twiddle64 tw) {
    const F f(fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= rows * cols) return;
    const uint64_t r = tid / cols, c = tid % cols;
    uint64_t k = row0 + r;
    if (rows_bitrev) k = bitrev64(k, row_bits);
    uint64_t e = (k * (col0 + c) * tw_stride) & tw_mask;
    if (inverse) e = (0 - e) & tw_mask;
    data[tid] = f.to_mem(f.twiddled(f.from_mem(data[tid]), twiddle_pow(tw, f, e)));
}

// in-place bit-reversal permutation of every row of 2^bits values (the reference's reverse(), main_naive.cc:82-95):
// thread i swaps with bitrev(i) when i < bitrev(i)
__global__ void __launch_bounds__(256) k64_bitrev(uint64_t* data, int bits, uint64_t rows) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t len = (uint64_t)1 << bits;
    if (tid >= rows * len) return;
    const uint64_t row = tid >> bits, i = tid & (len - 1);
    const uint64_t rev = bitrev64(i, bits);
    if (i < rev) {
        uint64_t* r = data + row * len;
        const uint64_t t = r[i];
        r[i] = r[rev];
        r[rev] = t;
    }
}

// out[row][i] = canonical(data[row][i] * scale) for i < n_out
template <class F>
__global__ void __launch_bounds__(256) k64_finalize(const uint64_t* data, uint64_t len, uint64_t* out, size_t out_stride,
                                                    uint64_t n_out, uint64_t rows, typename F::W scale, typename F::Consts fc) {
    const F f(fc);
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= rows * n_out) return;
    const uint64_t row = tid / n_out, i = tid % n_out;
    out[row * out_stride + i] = f.canon(f.twiddled(f.from_mem(data[row * len + i]), scale));
}

// Probe of the double-precision field (launched once per plan by the host, which compares with integer arithmetic):
//   i < 8       out[i] = canonical(x[i] * w[i])
//   8 <= i < 12 out[i] = canonical(raw[i] mod p)            (reduction of arbitrary 64-bit words)
//   12 <= i <16 out[i] = canonical(from_mem(to_mem(x[i])))  (round trip through the memory format)
__global__ void __launch_bounds__(64) k64_fieldd_probe(const double* x, const double* w, const uint64_t* raw, uint64_t* out, FieldDConsts fc) {
    const FieldD f(fc);
    const int i = threadIdx.x;
    if (i < 8) out[i] = f.canon(f.mulw(x[i], w[i]));
    else if (i < 12) out[i] = f.canon(f.from_raw(raw[i]));
    else if (i < 16) out[i] = f.canon(f.from_mem(f.to_mem(x[i])));
}

// ---------------------------------------------------------------------------
// k64_garner: recombination of residues modulo two or three NTT primes of like size (59 or 50 bits) into the product coefficient
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
