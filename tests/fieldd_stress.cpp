// Host-side stress of the exact double-precision field (test infrastructure): the header the GPU kernels use, compiled
// for the CPU -- the arithmetic is IEEE-754 binary64 with fused multiply-add on both -- and tens of millions of random
// operands at and inside the range limits its butterflies rely on, every result checked against 128-bit integers.
// usage: fieldd_stress <samples per modulus>      prints "FIELDD STRESS OK <count>" or the first failure
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

#include "ntt_field64.h"

using namespace nttamd;
typedef __int128 i128;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() {   // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline int64_t rnd_sym(int64_t bound) {   // uniform in [-bound, bound]
    return (int64_t)(rnd() % (uint64_t)(2 * bound + 1)) - bound;
}
static inline bool congruent(double r, i128 exact, int64_t p) {
    if (r != floor(r)) return false;
    i128 d = (i128)(int64_t)r - exact;
    return d % p == 0;
}

int main(int argc, char** argv) {
    const long samples = argc > 1 ? atol(argv[1]) : 1000000;
    const int64_t moduli[] = {1337006139375617ll, 1499999999999989ll /* largest accepted is 1.5e15 */, 998244353ll, 7340033ll,
                              1125899906842679ll /* just above 2^50 */};
    long total = 0;
    for (int64_t p : moduli) {
        FieldDConsts c;
        c.p = (double)p; c.u = 1.0 / (double)p;
        const uint64_t c40 = ((uint64_t)1 << 40) % (uint64_t)p;
        c.c40 = c40 > (uint64_t)p / 2 ? -(double)((uint64_t)p - c40) : (double)c40;
        for (int e = 0; e < 8; ++e) { c.root16F[e] = 0; c.root16I[e] = 0; }
        const FieldD f(c);
        const int64_t xmax = (int64_t)(2.5 * (double)p), wmax = (int64_t)(0.6 * (double)p), b125 = (int64_t)(1.25 * (double)p);
        const int64_t lim52 = ((int64_t)1 << 52) - 1;
        for (long s = 0; s < samples; ++s) {
            // multiply at the widest operand ranges, every 16th sample pinned to an edge
            int64_t x = rnd_sym(xmax), w = rnd_sym(wmax);
            if ((s & 15) == 0) { x = (s & 16) ? xmax : -xmax; w = (s & 32) ? wmax : -wmax; }
            const double r = f.mulw((double)x, (double)w);
            if (!congruent(r, (i128)x * w, p) || fabs(r) > 1.25 * (double)p + 2) {
                printf("mulw p=%lld x=%lld w=%lld r=%.0f\n", (long long)p, (long long)x, (long long)w, r); return 1;
            }
            // chained pass twiddles (products of table twiddles, up to 0.76 p) only ever meet values up to 1.25 p
            {
                const int64_t w76 = (int64_t)(0.76 * (double)p);
                int64_t x2 = rnd_sym(b125), w2 = rnd_sym(w76);
                if ((s & 15) == 3) { x2 = (s & 16) ? b125 : -b125; w2 = (s & 32) ? w76 : -w76; }
                const double r2 = f.mulw((double)x2, (double)w2);
                if (!congruent(r2, (i128)x2 * w2, p) || fabs(r2) > 1.25 * (double)p + 2) {
                    printf("mulw (chained twiddle) p=%lld x=%lld w=%lld r=%.0f\n", (long long)p, (long long)x2, (long long)w2, r2); return 1;
                }
            }
            // reduction of anything below 2^52
            const int64_t y = (s & 15) == 1 ? ((s & 16) ? lim52 : -lim52) : rnd_sym(lim52);
            const double q = f.reduce((double)y);
            if (!congruent(q, (i128)y, p) || fabs(q) > (double)(p / 2) + 1) { printf("reduce p=%lld y=%lld r=%.0f\n", (long long)p, (long long)y, q); return 1; }
            // forward butterfly with the unreduced trivial level behind it: inputs up to 1.25 p, then a multiply of the lazy sum
            double a = (double)rnd_sym(b125), b = (double)rnd_sym(b125);
            const i128 ea = (i128)(int64_t)a + (int64_t)b, eb = (i128)(int64_t)a - (int64_t)b;
            f.dif1_lazy(a, b);
            const int64_t t = rnd_sym(p / 2);
            const double m1 = f.twiddled(a, (double)t), m2 = f.from_mem(f.to_mem(b));   // the two consumers of lazy values: multiply, store
            if (!congruent(m1, ea * t, p) || fabs(m1) > 1.25 * (double)p + 2 || !congruent(m2, eb, p) || fabs(m2) > (double)(p / 2) + 1) {
                printf("lazy dif p=%lld\n", (long long)p); return 1;
            }
            // inverse butterfly after an unreduced trivial level: inputs up to 2.5 p
            double u = (double)rnd_sym(xmax), v = (double)rnd_sym(xmax);
            const i128 eu = (i128)(int64_t)u + (i128)(int64_t)v * t, ev = (i128)(int64_t)u - (i128)(int64_t)v * t;
            f.dit(u, v, (double)t);
            if (!congruent(u, eu, p) || !congruent(v, ev, p) || fabs(u) > 1.75 * (double)p + 3 || fabs(v) > 1.75 * (double)p + 3) {
                printf("dit p=%lld\n", (long long)p); return 1;
            }
            // pointwise product: any held value times a value up to 2.5 p
            const double pw = f.mulpw((double)y, (double)x);
            if (!congruent(pw, (i128)y * x, p) || fabs(pw) > 1.25 * (double)p + 2) { printf("mulpw p=%lld\n", (long long)p); return 1; }
            // arbitrary 64-bit words in, canonical words out
            const uint64_t raw = (s & 15) == 2 ? ~(uint64_t)0 - (uint64_t)(s >> 4) % 3 : rnd();
            const double fr = f.from_raw(raw);
            if (!congruent(fr, (i128)(raw % (uint64_t)p), p) || fabs(fr) > (double)(p / 2) + 1) { printf("from_raw p=%lld raw=%llu\n", (long long)p, (unsigned long long)raw); return 1; }
            if (f.canon(f.mulw((double)x, (double)w)) != (uint64_t)((((i128)x * w) % p + p) % p)) { printf("canon p=%lld\n", (long long)p); return 1; }
            ++total;
        }
    }
    printf("FIELDD STRESS OK %ld\n", total);
    return 0;
}
