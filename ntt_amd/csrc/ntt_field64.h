// Prime fields for transforms longer than any 32-bit NTT prime allows (2-adicity), in two arithmetic flavours
// behind one interface, so that every kernel of ntt_kernels64.h is written once:
//
//   Field64   odd p < 2^60, unsigned 64-bit registers, Montgomery reduction with R = 2^64 (a modular multiply is
//             about thirty 32-bit instructions, eleven of them multiplies);
//   FieldD    odd p <= 1.5 * 10^15 (2^50.4), double-precision registers holding exact integers: the product
//             x*w = h + l is split without error by one multiply and one fused multiply-add, the quotient is
//             q = round(h / p), and x*w - q*p = fma(-q, p, h) + l is exact -- six floating-point instructions, all
//             of them exact integer arithmetic, so the results are the same canonical residues bit for bit.
//
// Both keep data in the ordinary domain and constants (twiddles, scale factors) in the form the multiply wants
// (Montgomery form / balanced doubles), use lazy ranges between butterflies and produce canonical values only at
// the final store.  In memory every value is an unsigned 64-bit word below 2p, whichever field wrote it.
// The reference reaches the same residues for its large modulus through four 32-bit primes and a CRT
// (main_naive.cc:178-230); its modulus 1337006139375617 = 19*2^46 + 1 is itself NTT-friendly and below the FieldD
// limit, which is what the single n = 2^28 product relies on.
#pragma once
#include <stdint.h>

#include "ntt_field.h"

// The double-precision field below is exact integer arithmetic: every product, sum and fused multiply-add is written out
// and must be executed as written.  Contraction of a multiply and an add into one fused operation (hipcc's default for
// device code) changes roundings; the build passes -ffp-contract=off, and this says the same for any other build line.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace nttamd {

NTT_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

struct Field64Consts {
    uint64_t p;      // modulus
    uint64_t np;     // -p^-1 mod 2^64
    uint64_t r1;     // R mod p
    uint64_t r2;     // R^2 mod p
    uint64_t root16F[8];  // Montgomery forms of w16^e, e = 0..7
    uint64_t root16I[8];  // Montgomery forms of w16^-e
};

struct Field64 {
    typedef uint64_t V;   // register value
    typedef uint64_t W;   // twiddle (Montgomery form, < 2p)
    typedef Field64Consts Consts;
    static constexpr int SHORT_MUL_WAVES = 3;   // waves per SIMD k64_short_mul is compiled for (register budget 168)
    uint64_t p, np, B, r2;
    NTT_HD explicit Field64(const Field64Consts& c) : p(c.p), np(c.np), B(2 * c.p), r2(c.r2) {}

    // (hi:lo + m*p) / 2^64; in [0,2p) for hi:lo < 2^64 * p
    NTT_HD uint64_t redc(uint64_t hi, uint64_t lo) const {
        const uint64_t m = lo * np;
        return hi + mulhi64(m, p) + (lo != 0 ? 1u : 0u);
    }
    // x < 2^64 (any), wm < 2p Montgomery form of w, with x*wm < 2^64*p: x*w mod p in [0,2p)
    NTT_HD uint64_t mulw(uint64_t x, uint64_t wm) const {
        const unsigned __int128 t = (unsigned __int128)x * wm;   // one 128-bit product: high and low words share the partial products
        return redc((uint64_t)(t >> 64), (uint64_t)t);
    }
    NTT_HD static uint64_t csub(uint64_t x, uint64_t m) { return x >= m ? x - m : x; }
    NTT_HD uint64_t add(uint64_t u, uint64_t v) const { return csub(u + v, B); }
    NTT_HD uint64_t sub_raw(uint64_t u, uint64_t v) const { return u - v + B; }
    NTT_HD uint64_t sub(uint64_t u, uint64_t v) const { return csub(u - v + B, B); }
    // forward butterfly, u, v in [0,2p) -> [0,2p)
    NTT_HD void dif(uint64_t& u, uint64_t& v, uint64_t wm) const {
        const uint64_t s = add(u, v);
        v = mulw(sub_raw(u, v), wm);
        u = s;
    }
    NTT_HD void dif1(uint64_t& u, uint64_t& v) const {
        const uint64_t s = add(u, v);
        v = sub(u, v);
        u = s;
    }
    // inverse butterfly, x, y in [0,4p) -> [0,4p)
    NTT_HD void dit(uint64_t& x, uint64_t& y, uint64_t wm) const {
        const uint64_t xr = csub(x, B), t = mulw(y, wm);
        x = xr + t;
        y = xr - t + B;
    }
    NTT_HD void dit1(uint64_t& x, uint64_t& y) const {
        const uint64_t xr = csub(x, B), t = csub(y, B);
        x = xr + t;
        y = xr - t + B;
    }
    // Trivial butterflies next to a multiply.  dif1_lazy: inputs in [0,2p), outputs in [0,4p) -- good for mulw (any word
    // below 2^64 / 2), to_mem and mulpw, not for another butterfly (settle first).  dit1_lazy: inputs in [0,2p) (what mulw
    // and from_mem deliver), outputs in [0,4p) like dit1.
    NTT_HD void dif1_lazy(uint64_t& u, uint64_t& v) const {
        const uint64_t s = u + v;
        v = u - v + B;
        u = s;
    }
    NTT_HD void dit1_lazy(uint64_t& x, uint64_t& y) const {
        const uint64_t s = x + y;
        y = x - y + B;
        x = s;
    }
    NTT_HD uint64_t settle(uint64_t x) const { return csub(x, B); }          // [0,4p) -> [0,2p)
    NTT_HD uint64_t canon4(uint64_t x) const { return csub(csub(x, B), p); }  // [0,4p) -> [0,p)
    // any 64-bit x -> x mod p in [0,p):  redc(x * R2) = x*R, redc(that) = x
    NTT_HD uint64_t reduce(uint64_t x, uint64_t r2) const {
        const uint64_t xr = redc(mulhi64(x, r2), x * r2);       // x*R mod p, [0,2p)
        return csub(redc(0, xr), p);
    }

    // ---- the interface shared with FieldD ----
    NTT_HD V from_mem(uint64_t raw) const { return raw; }                 // stored words are below 2p
    NTT_HD uint64_t to_mem(V x) const { return csub(x, B); }              // [0,4p) -> [0,2p)
    NTT_HD W tw_from_mem(uint64_t raw) const { return raw; }
    NTT_HD W twmul(W a, W b) const { return mulw(a, b); }                 // Montgomery forms multiply to a Montgomery form
    NTT_HD V mulpw(V a, V b) const { return mulw(csub(a, B), csub(b, B)); }   // a*b*2^-64: the scale constant carries 2^64
    NTT_HD uint64_t canon(V x) const { return canon4(x); }
    NTT_HD V twiddled(V x, W w) const { return mulw(x, w); }              // x in [0,4p)
    NTT_HD V from_raw(uint64_t x) const { return reduce(x, r2); }          // any 64-bit word -> [0,p)
};

// ---------------------------------------------------------------------------------------------
// Double-precision flavour.  Register values are integers held in doubles, congruent to the true value mod p and
// bounded in magnitude; the bounds below are what keeps every step exact (all quantities are integers below 2^53
// and every quotient estimate is below 2^51).  Table twiddles are balanced (|w| <= p/2), the product of two of them
// is below 0.6 p, and chains of such products (the pass twiddles of k64_tile) stay below 0.76 p:
//   mulw(x, w)   needs |x w| / p < 2^51, i.e. |x| <= 2.5 p for |w| <= 0.6 p (p <= 1.5e15) and |x| <= 1.9 p for
//                |w| <= 0.76 p (those only ever meet values below 1.25 p); returns |r| <= 1.25 p
//   reduce(x)    needs |x| < 2^52; returns |r| <= p/2 + 1
//   dif          inputs |.| <= 1.25 p -> sum reduced (<= p/2), difference (<= 2.5 p) multiplied (<= 1.25 p)
//   dit          x reduced first, y <= 2.5 p multiplied; outputs <= 1.75 p
//   dif1_lazy    (trivial butterfly whose outputs go into a table-twiddle multiply, a store or the pointwise product, all
//                of which reduce) inputs <= 1.25 p -> outputs <= 2.5 p, unreduced; `settle` before any other use
//   dit1_lazy    (trivial butterfly fed by multiplies or loads) inputs <= 1.25 p -> outputs <= 2.5 p, which the next
//                level (reduce / multiply by a fixed root <= p/2), the twiddle of the next round and the stores accept
// Memory format: reduced value + p, an unsigned word in [p/2, 3p/2] (below 2^52, so the conversion in both
// directions is one addition of a constant and one bit operation on the high word).
// ---------------------------------------------------------------------------------------------
struct FieldDConsts {
    double p;        // modulus
    double u;        // 1/p rounded
    double c40;      // 2^40 mod p, balanced
    double root16F[8];  // w16^e mod p, balanced
    double root16I[8];  // w16^-e mod p, balanced
};

constexpr double FIELDD_MAX_P = 1.5e15;   // 1.5 * 1.25 * ... see the bounds above: 2.5 p * 0.6 p / p must stay below 2^51

struct FieldD {
    typedef double V;
    typedef double W;
    typedef FieldDConsts Consts;
    static constexpr int SHORT_MUL_WAVES = 4;   // waves per SIMD k64_short_mul is compiled for (130 registers without the cap)
    double p, u, c40;
    NTT_HD explicit FieldD(const FieldDConsts& c) : p(c.p), u(c.u), c40(c.c40) {}

    static constexpr double MAGIC_ROUND = 6755399441055744.0;   // 1.5 * 2^52: adding and subtracting it rounds to an integer
    static constexpr double TWO52 = 4503599627370496.0;

    NTT_HD static double bits_to_double(uint64_t b) { return __builtin_bit_cast(double, b); }
    NTT_HD static uint64_t double_to_bits(double d) { return __builtin_bit_cast(uint64_t, d); }

    // x*w mod p, exact.  The function body must not be re-associated or contracted: every operation is written out.
    NTT_HD double mulw(double x, double w) const {
        const double h = x * w;
        const double l = __builtin_fma(x, w, -h);            // x*w = h + l exactly
        const double q = __builtin_fma(h, u, MAGIC_ROUND) - MAGIC_ROUND;   // round(h / p), off by at most one
        return __builtin_fma(-q, p, h) + l;                  // both terms exact integers
    }
    NTT_HD double reduce(double x) const {
        const double q = __builtin_fma(x, u, MAGIC_ROUND) - MAGIC_ROUND;
        return __builtin_fma(-q, p, x);
    }
    NTT_HD void dif(double& a, double& b, double w) const {
        const double s = a + b, d = a - b;
        a = reduce(s);
        b = mulw(d, w);
    }
    NTT_HD void dif1(double& a, double& b) const {
        const double s = a + b, d = a - b;
        a = reduce(s);
        b = reduce(d);
    }
    NTT_HD void dif1_lazy(double& a, double& b) const {
        const double s = a + b, d = a - b;
        a = s;
        b = d;
    }
    NTT_HD void dit1_lazy(double& x, double& y) const {
        const double s = x + y, d = x - y;
        x = s;
        y = d;
    }
    NTT_HD double settle(double x) const { return reduce(x); }
    NTT_HD void dit(double& x, double& y, double w) const {
        const double xr = reduce(x), t = mulw(y, w);
        x = xr + t;
        y = xr - t;
    }
    NTT_HD void dit1(double& x, double& y) const {
        const double xr = reduce(x), yr = reduce(y);
        x = xr + yr;
        y = xr - yr;
    }
    // stored word (below 2^52) -> value - p: magnitude at most p for anything a field kernel or the loader stored
    NTT_HD double from_mem(uint64_t raw) const { return bits_to_double(raw | 0x4330000000000000ull) - (TWO52 + p); }
    // any register value -> reduced value + p in [p/2, 3p/2]
    NTT_HD uint64_t to_mem(double x) const { return double_to_bits(reduce(x) + (TWO52 + p)) & 0x000FFFFFFFFFFFFFull; }
    NTT_HD double tw_from_mem(uint64_t raw) const { return bits_to_double(raw); }   // tables hold balanced doubles
    NTT_HD double twmul(double a, double b) const { return mulw(a, b); }             // |a|, |b| <= p/2 -> |r| <= 0.57 p
    NTT_HD double mulpw(double a, double b) const { return mulw(reduce(a), b); }       // |b| <= 2.5 p: one reduced factor suffices
    NTT_HD double twiddled(double x, double w) const { return mulw(x, w); }          // |x| <= 2.5 p
    NTT_HD uint64_t canon(double x) const {
        double r = reduce(x);
        r = r < 0.0 ? r + p : r;
        return double_to_bits(r + TWO52) & 0x000FFFFFFFFFFFFFull;
    }
    // any 64-bit word -> value mod p (magnitude <= p/2 + 1): x = a * 2^40 + b with a < 2^24, b < 2^40
    NTT_HD double from_raw(uint64_t x) const {
        const double a = (double)(uint32_t)(x >> 40);
        const double b = bits_to_double((x & 0xFFFFFFFFFFull) | 0x4330000000000000ull) - TWO52;
        return reduce(mulw(a, c40) + b);
    }
};

// Two-level twiddle lookup: w^e = W1[e >> t0] * W0[e & (2^t0 - 1)], e < 2^logtab (entries in the field's twiddle form).
struct Twiddle64 {
    const uint64_t* w0;  // 2^t0 entries: w^i
    const uint64_t* w1;  // 2^(logtab - t0) entries: w^(i * 2^t0)
    int t0, logtab;
};

template <class F>
NTT_HD typename F::W twiddle_pow(const Twiddle64& tw, const F& f, uint64_t e) {
    const uint64_t lo = e & ((1ull << tw.t0) - 1), hi = e >> tw.t0;
    const typename F::W a = f.tw_from_mem(tw.w1[hi]);
    if (lo == 0) return a;
    return f.twmul(a, f.tw_from_mem(tw.w0[lo]));
}

}  // namespace nttamd
