// Host-side plan construction: number-theoretic constants and twiddle tables
// for one (modulus, transform length) pair.  Pure host code, no device calls,
// so it is unit-testable on its own.
//
// Restates the set-up arithmetic of the reference: pow() (main_naive.cc:69-80),
// the per-stage roots g^((p-1)/len) and their inverses (main_naive.cc:107-108),
// n^-1 = n^(p-2) (main_naive.cc:132), the Montgomery32 constructor constants
// (Montgomery32.h:19-40) and the CRT constants M, M_i, M_i^-1
// (main_naive.cc:189-200).  The primitive root is 3, as in the reference.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "ntt_field.h"

namespace nttamd {

typedef unsigned __int128 u128;

inline uint64_t h_mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((u128)a * b % m); }
inline uint64_t h_powmod(uint64_t base, uint64_t exp, uint64_t m) {
    uint64_t r = 1 % m;
    base %= m;
    while (exp) {
        if (exp & 1) r = h_mulmod(r, base, m);
        base = h_mulmod(base, base, m);
        exp >>= 1;
    }
    return r;
}
inline uint32_t h_neg_inv32(uint32_t p) {  // -p^-1 mod 2^32, p odd
    uint32_t inv = p;
    for (int i = 0; i < 5; ++i) inv *= 2u - p * inv;
    return 0u - inv;
}
inline uint32_t h_mont(uint64_t x, uint32_t p) { return (uint32_t)(((u128)(x % p) << 32) % p); }

// Deterministic Miller-Rabin for 64-bit integers (the first twelve prime bases decide every n < 3.3 * 10^24).
// Every inverse on the transform paths is a Fermat inverse x^(p-2), which is only an inverse when p is prime, and a
// pseudoprime can pass the generator test 3^((p-1)/2) = -1: moduli are admitted to the direct transform routes only
// when this says prime (composite moduli are still served, by the CRT routes).
inline bool is_prime_u64(uint64_t n) {
    if (n < 2) return false;
    static const uint64_t small[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (uint64_t q : small) {
        if (n == q) return true;
        if (n % q == 0) return false;
    }
    uint64_t d = n - 1;
    int r = 0;
    while ((d & 1) == 0) { d >>= 1; ++r; }
    for (uint64_t a : small) {
        uint64_t x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int i = 1; i < r; ++i) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) { composite = false; break; }
        }
        if (composite) return false;
    }
    return true;
}

inline int ceil_log2(uint64_t x) {
    int l = 0;
    while ((1ull << l) < x) ++l;
    return l;
}

// Transform length for n coefficients per operand: smallest power of two >= 2n-1, at least 2.
inline int transform_log2(int n) {
    int l = ceil_log2((uint64_t)(2 * (int64_t)n - 1));
    return l < 1 ? 1 : l;
}

constexpr int MAX_LOG_L = 18 + 6;  // deepest supported: two strided levels over a 4096-point core

// p usable for a direct (single-modulus) transform of length 2^logL?
inline bool modulus_is_ntt_friendly(uint64_t p, int logL, std::string* why = nullptr) {
    if (p >= (1ull << 31) || p < 3 || (p & 1) == 0) { if (why) *why = "modulus must be odd and below 2^31 for the direct path"; return false; }
    const uint64_t L = 1ull << logL;
    if ((p - 1) % L != 0) { if (why) *why = "transform length does not divide p-1"; return false; }
    if (!is_prime_u64(p)) { if (why) *why = "modulus is not prime"; return false; }
    const uint64_t w = h_powmod(3, (p - 1) / L, p);
    if (h_powmod(w, L / 2, p) != p - 1) { if (why) *why = "3 is not a generator of the required 2-power subgroup mod p (p not prime?)"; return false; }
    const uint64_t linv = h_powmod(L % p, p - 2, p);
    if (h_mulmod(linv, L % p, p) != 1) { if (why) *why = "transform length not invertible mod p (p not prime?)"; return false; }
    return true;
}

enum PlanKind { PLAN_SMALL = 0, PLAN_FUSED8192 = 1, PLAN_MULTIPASS = 2 };

// Host image of everything a plan uploads.
struct PlanHost {
    uint32_t p = 0;
    int logL = 0;
    bool lazy = false;  // p < 2^30
    PlanKind kind = PLAN_SMALL;
    int Q = 0;          // multipass: log2 of the strided radix (L = 4096 << Q)
    FieldConsts fc{};
    uint32_t scale_m = 0;             // Montgomery form of L^-1 * R
    uint32_t scale_t = 0;             // Montgomery form of L^-1 (stand-alone inverse transform: no pointwise product to undo)
    std::vector<uint32_t> twF, twI;   // small: wL^(+-e), e < L/2
    bool small16 = false;             // small: the radix-16 kernel applies (L >= 32 and 64 | p-1)
    std::vector<uint32_t> rtF[13], rtI[13];   // small16: round tables for blocks of 2^b, b = 5..logL
    std::vector<uint32_t> tw1F, tw2F, tw1I, tw2I;  // core tables
    std::vector<uint32_t> tw1I_t;     // tw1I with the scale of the stand-alone inverse transform (L^-1)
    std::vector<uint32_t> ow;         // fused8192: [w8192^j | w8192^-j], j < 4096
    std::vector<uint32_t> T1, T3;     // multipass strided twiddles, L entries each
};

inline void fill_core_tables(PlanHost& h, uint64_t scale /* plain value folded into tw1I */, uint64_t scale_t /* into tw1I_t */) {
    const uint32_t p = h.p;
    const uint64_t w4096 = h_powmod(3, (p - 1) / 4096, p), w4096i = h_powmod(w4096, p - 2, p);
    const uint64_t w256 = h_powmod(w4096, 16, p), w256i = h_powmod(w256, p - 2, p);
    h.tw1F.resize(256 * 16); h.tw1I.resize(256 * 16); h.tw2F.resize(16 * 16); h.tw2I.resize(16 * 16); h.tw1I_t.resize(256 * 16);
    for (int pos = 0; pos < 16; ++pos) {
        const int k = bitrev(pos, 4);
        const uint64_t stepF = h_powmod(w4096, k, p), stepI = h_powmod(w4096i, k, p);
        uint64_t vF = 1, vI = scale % p, vT = scale_t % p;
        for (int t = 0; t < 256; ++t) {
            h.tw1F[t * 16 + pos] = h_mont(vF, p);
            h.tw1I[t * 16 + pos] = h_mont(vI, p);
            h.tw1I_t[t * 16 + pos] = h_mont(vT, p);
            vF = h_mulmod(vF, stepF, p); vI = h_mulmod(vI, stepI, p); vT = h_mulmod(vT, stepI, p);
        }
        const uint64_t s2F = h_powmod(w256, k, p), s2I = h_powmod(w256i, k, p);
        uint64_t uF = 1, uI = 1;
        for (int c = 0; c < 16; ++c) {
            h.tw2F[c * 16 + pos] = h_mont(uF, p);
            h.tw2I[c * 16 + pos] = h_mont(uI, p);
            uF = h_mulmod(uF, s2F, p); uI = h_mulmod(uI, s2I, p);
        }
    }
}

// Build the host image.  Returns false (with *err) when (p, logL) is unsupported.
inline bool build_plan_host(uint32_t p, int logL, PlanHost& h, std::string* err) {
    std::string why;
    if (logL < 1 || logL > MAX_LOG_L) { if (err) *err = "transform length out of range"; return false; }
    if (!modulus_is_ntt_friendly(p, logL, &why)) { if (err) *err = why; return false; }
    h.p = p; h.logL = logL; h.lazy = p < (1u << 30);
    const uint64_t L = 1ull << logL;
    const uint64_t R1 = (1ull << 32) % p;
    h.fc.p = p; h.fc.np = h_neg_inv32(p); h.fc.r1 = (uint32_t)R1; h.fc.r2 = (uint32_t)h_mulmod(R1, R1, p);
    for (int e = 0; e < 32; ++e) { h.fc.rootF[e] = 0; h.fc.rootFn[e] = 0; h.fc.rootI[e] = 0; }
    if ((p - 1) % 64 == 0) {
        const uint64_t w64 = h_powmod(3, (p - 1) / 64, p), w64i = h_powmod(w64, p - 2, p);
        uint64_t a = 1, b = 1;
        for (int e = 0; e < 32; ++e) {
            h.fc.rootF[e] = h_mont(a, p); h.fc.rootFn[e] = p - h.fc.rootF[e]; h.fc.rootI[e] = h_mont(b, p);
            a = h_mulmod(a, w64, p); b = h_mulmod(b, w64i, p);
        }
    }
    const uint64_t wL = h_powmod(3, (p - 1) / L, p), wLi = h_powmod(wL, p - 2, p);
    const uint64_t Linv = h_powmod(L % p, p - 2, p);
    const uint64_t scale = h_mulmod(Linv, R1, p);  // L^-1 * R: undoes the R^-1 of the pointwise product
    h.scale_m = h_mont(scale, p);
    h.scale_t = h_mont(Linv, p);
    if (logL <= 12) {
        h.kind = PLAN_SMALL;
        h.twF.resize(L / 2 ? L / 2 : 1); h.twI.resize(h.twF.size());
        uint64_t a = 1, b = 1;
        for (uint64_t e = 0; e < L / 2; ++e) {
            h.twF[e] = h_mont(a, p); h.twI[e] = h_mont(b, p);
            a = h_mulmod(a, wL, p); b = h_mulmod(b, wLi, p);
        }
        h.small16 = logL >= 5 && (p - 1) % 64 == 0;
#ifdef NTT_NO_SMALL16   // diagnostic builds: the radix-2 kernel for every short product
        h.small16 = false;
#endif
        if (h.small16) {
            for (int bl = logL; bl > 4; bl -= 4) {   // the rounds of k_small16
                const uint64_t wB = h_powmod(wL, 1ull << (logL - bl), p), wBi = h_powmod(wB, p - 2, p);
                const size_t sub = (size_t)1 << (bl - 4);
                h.rtF[bl].resize(sub << 4); h.rtI[bl].resize(sub << 4);
                for (int pos = 0; pos < 16; ++pos) {
                    const uint64_t sF = h_powmod(wB, (uint64_t)bitrev(pos, 4), p), sI = h_powmod(wBi, (uint64_t)bitrev(pos, 4), p);
                    uint64_t vF = 1, vI = 1;
                    for (size_t j = 0; j < sub; ++j) {
                        h.rtF[bl][(j << 4) + pos] = h_mont(vF, p); h.rtI[bl][(j << 4) + pos] = h_mont(vI, p);
                        vF = h_mulmod(vF, sF, p); vI = h_mulmod(vI, sI, p);
                    }
                }
            }
        }
    } else if (logL == 13) {
        h.kind = PLAN_FUSED8192;
        fill_core_tables(h, scale, Linv);
        h.ow.resize(8192);
        uint64_t a = 1, b = 1;
        for (int j = 0; j < 4096; ++j) {
            h.ow[j] = h_mont(a, p); h.ow[4096 + j] = h_mont(b, p);
            a = h_mulmod(a, wL, p); b = h_mulmod(b, wLi, p);
        }
    } else {
        h.kind = PLAN_MULTIPASS;
        if (logL > 18) { if (err) *err = "transform lengths above 2^18 are not implemented yet"; return false; }
        h.Q = logL - 12;
        fill_core_tables(h, scale, Linv);
        const int R = 1 << h.Q;
        const uint64_t C = L >> h.Q;
        h.T1.resize(L); h.T3.resize(L);
        for (int pos = 0; pos < R; ++pos) {
            const int k = bitrev(pos, h.Q);
            const uint64_t sF = h_powmod(wL, k, p), sI = h_powmod(wLi, k, p);
            uint64_t a = 1, b = 1;
            for (uint64_t j = 0; j < C; ++j) {
                const size_t idx = ((size_t)(pos >> 2) * C + j) * 4 + (pos & 3);
                h.T1[idx] = h_mont(a, p); h.T3[idx] = h_mont(b, p);
                a = h_mulmod(a, sF, p); b = h_mulmod(b, sI, p);
            }
        }
    }
    return true;
}

// ---- CRT constants (main_naive.cc:179-200) ----
static const uint32_t kCrtPrimes[4] = {1004535809u, 1224736769u, 469762049u, 998244353u};

struct CrtHost {
    u128 M;
    u128 Mi[4];
    uint32_t Mi_inv[4];
};
inline CrtHost crt_host_consts() {
    CrtHost c;
    c.M = 1;
    for (int i = 0; i < 4; ++i) c.M *= kCrtPrimes[i];
    for (int i = 0; i < 4; ++i) {
        c.Mi[i] = c.M / kCrtPrimes[i];
        c.Mi_inv[i] = (uint32_t)h_powmod((uint64_t)(c.Mi[i] % kCrtPrimes[i]), kCrtPrimes[i] - 2, kCrtPrimes[i]);
    }
    return c;
}
// true when every coefficient of the integer product of two length-n polynomials with
// entries < p is below M, i.e. the CRT reconstruction is exact
inline bool crt_bound_ok(int n, uint64_t p) {
    const u128 M = crt_host_consts().M;
    const u128 pm1 = (u128)(p - 1);
    if (pm1 != 0 && pm1 > (~(u128)0) / pm1) return false;
    const u128 sq = pm1 * pm1;
    if (sq != 0 && (u128)n > (~(u128)0) / sq) return false;
    return sq * (u128)n < M;
}

// ---- wide CRT basis: 59-bit primes c*2^32 + 1 for which 3 generates the 2-power subgroup ----
// Used where neither the four 31-bit primes (product < 2^119, lengths <= 2^21) nor the modulus itself can carry
// the transform: any p < 2^64, lengths up to 2^32.
static const uint64_t kWidePrimes[3] = {576460451655712769ull, 576460065108656129ull, 576459820295520257ull};

inline int bit_length(uint64_t x) {
    int b = 0;
    while (x) { ++b; x >>= 1; }
    return b;
}
// The same for the double-precision flavour of the 64-bit field (primes below 1.5e15, 50.4 bits each): a product in it
// costs about two thirds of one in Montgomery integers, so this basis is taken wherever it needs no more primes.
static const uint64_t kWidePrimesD[3] = {1499510521987073ull, 1499458982379521ull, 1499407442771969ull};

// Basis and number of primes such that every coefficient of the integer product, at most n*(p-1)^2 < 2^bits, stays
// below their product (59-bit primes: q1*q2 > 2^117, q1*q2*q3 > 2^176; 50-bit primes: > 2^100 and > 2^151).
// k = 0: not even three 59-bit primes suffice.
struct WideBasis {
    const uint64_t* q;
    int k;
};
inline WideBasis wide_crt_basis(uint64_t n, uint64_t p, bool double_precision_ok) {
    const int bits = bit_length(n) + 2 * bit_length(p - 1);
    if (double_precision_ok && bits <= 100) return {kWidePrimesD, 2};
    if (bits <= 117) return {kWidePrimes, 2};
    if (double_precision_ok && bits <= 151) return {kWidePrimesD, 3};
    if (bits <= 176) return {kWidePrimes, 3};
    return {nullptr, 0};
}

}  // namespace nttamd
