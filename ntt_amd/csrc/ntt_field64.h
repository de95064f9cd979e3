// 64-bit prime field for transforms longer than any 32-bit NTT prime allows
// (2-adicity): odd p < 2^61, Montgomery reduction with R = 2^64.
//
// Same conventions as the 32-bit field (ntt_field.h): data in the ordinary
// domain, constants in Montgomery form, lazy ranges [0,2p) forward / [0,4p)
// inverse, canonical values only at the final store.  The reference's
// large-modulus path reaches the same residues through four 32-bit primes and
// a CRT (main_naive.cc:178-230); its modulus 1337006139375617 = 19*2^46 + 1 is
// itself NTT-friendly, which is what the single n = 2^28 product relies on.
#pragma once
#include <stdint.h>

#include "ntt_field.h"

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
    uint64_t p, np, B;
    NTT_HD explicit Field64(const Field64Consts& c) : p(c.p), np(c.np), B(2 * c.p) {}

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
    NTT_HD uint64_t canon4(uint64_t x) const { return csub(csub(x, B), p); }  // [0,4p) -> [0,p)
    // any 64-bit x -> x mod p in [0,p):  redc(x * R2) = x*R, redc(that) = x
    NTT_HD uint64_t reduce(uint64_t x, uint64_t r2) const {
        const uint64_t xr = redc(mulhi64(x, r2), x * r2);       // x*R mod p, [0,2p)
        return csub(redc(0, xr), p);
    }
};

// Two-level twiddle lookup: w^e = W1[e >> t0] * W0[e & (2^t0 - 1)] (both Montgomery form), e < 2^logtab.
struct Twiddle64 {
    const uint64_t* w0;  // 2^t0 entries: w^i
    const uint64_t* w1;  // 2^(logtab - t0) entries: w^(i * 2^t0)
    int t0, logtab;
};

NTT_HD uint64_t twiddle_pow(const Twiddle64& tw, const Field64& f, uint64_t e) {
    const uint64_t lo = e & ((1ull << tw.t0) - 1), hi = e >> tw.t0;
    const uint64_t a = tw.w1[hi];
    if (lo == 0) return a;
    return f.mulw(a, tw.w0[lo]);  // Montgomery forms multiply to the Montgomery form of the product, [0,2p)
}

}  // namespace nttamd
