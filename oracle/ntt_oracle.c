/*
 * CPU oracle for the polynomial-multiplication path -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's serial/OpenMP algorithm, written
 * from the reference's behaviour (not copied from it).  Only tests/, the
 * smoke check and bench.py's cpu_baseline leg may load it; the shipped GPU
 * library never does.  Pinned against nttdata/0-3, filetest/0-3 and the
 * compiled reference (oracle/_ref) in tests/test_oracle.py.
 */
#include "ntt_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* main_naive.cc:69-80 -- right-to-left binary exponentiation; operands are
 * reduced first so every product fits 64 bits when mod < 2^32. */
uint64_t oracle_pow(uint64_t base, uint64_t exp, uint64_t mod) {
    uint64_t acc = 1;
    base %= mod;
    for (; exp != 0; exp >>= 1) {
        if (exp & 1u) acc = acc * base % mod;
        base = base * base % mod;
    }
    return acc;
}

/* main_naive.cc:82-95 -- rev[i] is built from rev[i/2]; each pair is swapped once. */
static void build_rev(int *rev, int n, int bit) {
    rev[0] = 0;
    for (int i = 1; i < n; ++i) rev[i] = (rev[i >> 1] >> 1) | ((i & 1) << (bit - 1));
}

void oracle_reverse(uint64_t *a, int n, int bit) {
    int *rev = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    build_rev(rev, n, bit);
    for (int i = 0; i < n; ++i) {
        int j = rev[i];
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    free(rev);
}

/* main_naive.cc:96-137 -- decimation in time: permute, then log2(n) stages of
 * blocks of length len whose twiddle is a running power of the len-th root
 * g^((p-1)/len) (inverse: g^((p-1)-(p-1)/len)); the inverse ends with a
 * multiplication by n^(p-2). */
void oracle_ntt(uint64_t *a, int n, int invert, uint64_t p, uint64_t g) {
    int bit = 0;
    while ((1 << bit) < n) ++bit;
    oracle_reverse(a, n, bit);
    for (int len = 2; len <= n; len <<= 1) {
        uint64_t e = (p - 1) / (uint64_t)len;
        uint64_t root = oracle_pow(g, invert ? (p - 1) - e : e, p);
        int half = len >> 1;
        for (int blk = 0; blk < n; blk += len) {
            uint64_t w = 1;
            for (int j = 0; j < half; ++j) {
                uint64_t u = a[blk + j];
                uint64_t v = a[blk + j + half] * w % p;
                uint64_t s = u + v;
                if (s >= p) s -= p;
                a[blk + j] = s;
                a[blk + j + half] = (u >= v) ? u - v : u + p - v;
                w = w * root % p;
            }
        }
    }
    if (invert) {
        uint64_t ninv = oracle_pow((uint64_t)n, p - 2, p);
        for (int i = 0; i < n; ++i) a[i] = a[i] * ninv % p;
    }
}

/* main.cc:540-566 and 571-630 -- radix-4 form of the same transform.  Values are kept in 64-bit words here
 * (the reference uses int arrays and 1LL products); p < 2^31 keeps every intermediate below 2^64. */
void oracle_ntt_base4(uint64_t *a, int n, int invert, uint64_t p, uint64_t g) {
    int digits = 0;
    for (int t = n; t > 1; t >>= 2) ++digits;
    for (int i = 0; i < n; ++i) {            /* base-4 digit reversal */
        int r = 0, v = i;
        for (int d = 0; d < digits; ++d) { r = (r << 2) | (v & 3); v >>= 2; }
        if (i < r) { uint64_t t = a[i]; a[i] = a[r]; a[r] = t; }
    }
    for (int len = 4; len <= n; len <<= 2) {
        uint64_t w1 = invert ? oracle_pow(g, (p - 1) - (p - 1) / (uint64_t)len, p) : oracle_pow(g, (p - 1) / (uint64_t)len, p);
        uint64_t w2 = w1 * w1 % p, w3 = w2 * w1 % p;
        int quarter = len >> 2;
        uint64_t im = oracle_pow(w1, (uint64_t)quarter, p);   /* primitive fourth root of unity */
        for (int base = 0; base < n; base += len) {
            uint64_t t1 = 1, t2 = 1, t3 = 1;
            for (int j = 0; j < quarter; ++j) {
                uint64_t x0 = a[base + j];
                uint64_t x1 = a[base + j + quarter] * t1 % p;
                uint64_t x2 = a[base + j + 2 * quarter] * t2 % p;
                uint64_t x3 = a[base + j + 3 * quarter] * t3 % p;
                uint64_t y1 = x1 * im % p, y3 = x3 * im % p;
                a[base + j] = (x0 + x1 + x2 + x3) % p;
                a[base + j + quarter] = (x0 + y1 + (p - x2) + (p - y3)) % p;
                a[base + j + 2 * quarter] = (x0 + (p - x1) + x2 + (p - x3)) % p;
                a[base + j + 3 * quarter] = (x0 + (p - y1) + (p - x2) + y3) % p;
                t1 = t1 * w1 % p; t2 = t2 * w2 % p; t3 = t3 * w3 % p;
            }
        }
    }
    if (invert) {
        uint64_t ninv = oracle_pow((uint64_t)n % p, p - 2, p);
        for (int i = 0; i < n; ++i) a[i] = a[i] * ninv % p;
    }
}

/* main_naive.cc:139-177 -- zero-pad both inputs to 2n, transform, multiply
 * pointwise, transform back, keep 2n-1 coefficients. */
void oracle_ntt_multiply(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p) {
    int len = n << 1;
    uint64_t *fa = (uint64_t *)calloc((size_t)len, sizeof(uint64_t));
    uint64_t *fb = (uint64_t *)calloc((size_t)len, sizeof(uint64_t));
    memcpy(fa, a, sizeof(uint64_t) * (size_t)n);
    memcpy(fb, b, sizeof(uint64_t) * (size_t)n);
    oracle_ntt(fa, len, 0, p, 3);
    oracle_ntt(fb, len, 0, p, 3);
    for (int i = 0; i < len; ++i) fa[i] = fa[i] * fb[i] % p;
    oracle_ntt(fa, len, 1, p, 3);
    for (int i = 0; i < 2 * n - 1; ++i) result[i] = fa[i];
    free(fa);
    free(fb);
}

/* main_naive.cc:178-230 -- residues modulo four fixed primes, one product per
 * prime, then sum_i M_i * ((r_i * M_i^-1) mod p_i) mod M, finally mod p. */
static const uint64_t CRT_PRIMES[4] = {1004535809ull, 1224736769ull, 469762049ull, 998244353ull};

void oracle_crt_ntt_multiply(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p) {
    int out_len = 2 * n - 1;
    u128 M = 1;
    u128 Mi[4];
    uint64_t Mi_inv[4];
    uint64_t *res[4];
    for (int i = 0; i < 4; ++i) M *= CRT_PRIMES[i];
    for (int i = 0; i < 4; ++i) {
        Mi[i] = M / CRT_PRIMES[i];
        Mi_inv[i] = oracle_pow((uint64_t)(Mi[i] % CRT_PRIMES[i]), CRT_PRIMES[i] - 2, CRT_PRIMES[i]);
    }
    uint64_t *ta = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    uint64_t *tb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    for (int i = 0; i < 4; ++i) {
        res[i] = (uint64_t *)calloc((size_t)out_len, sizeof(uint64_t));
        for (int j = 0; j < n; ++j) {
            ta[j] = a[j] % CRT_PRIMES[i];
            tb[j] = b[j] % CRT_PRIMES[i];
        }
        oracle_ntt_multiply(ta, tb, res[i], n, CRT_PRIMES[i]);
    }
    for (int j = 0; j < out_len; ++j) {
        u128 acc = 0;
        for (int i = 0; i < 4; ++i) {
            uint64_t t = res[i][j] * Mi_inv[i] % CRT_PRIMES[i];
            acc = (acc + Mi[i] * t) % M;
        }
        result[j] = (uint64_t)(acc % p);
    }
    for (int i = 0; i < 4; ++i) free(res[i]);
    free(ta);
    free(tb);
}

/* main_naive.cc:253-257 */
void oracle_poly_multiply_u64(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p) {
    if (p > (1ull << 32)) oracle_crt_ntt_multiply(a, b, result, n, p);
    else oracle_ntt_multiply(a, b, result, n, p);
}

/* main.cc:92-98 -- accumulates into ab (caller zeroes it). */
void oracle_poly_multiply_schoolbook(const int *a, const int *b, int *ab, int n, int p) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            ab[i + j] = (int)(((long long)a[i] * b[j] % p + ab[i + j]) % p);
}

void oracle_schoolbook_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p) {
    for (int k = 0; k < 2 * n - 1; ++k) ab[k] = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            ab[i + j] = (uint64_t)(((u128)a[i] * b[j] + ab[i + j]) % p);
}

/* main_openMP.cc:116-217 -- same arithmetic as the serial path; the loops
 * over butterfly blocks, the permutation and the element-wise steps are
 * OpenMP parallel-for regions, the bit-reversal table is built once per call. */
int oracle_openmp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void omp_permute(uint64_t *a, int n, const int *rev, int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < n; ++i) {
        int j = rev[i];
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

static void omp_stages(uint64_t *a, int n, int invert, uint64_t p, int threads) {
    (void)threads;
    for (int len = 2; len <= n; len <<= 1) {
        uint64_t e = (p - 1) / (uint64_t)len;
        uint64_t root = oracle_pow(3, invert ? (p - 1) - e : e, p);
        int half = len >> 1;
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int blk = 0; blk < n; blk += len) {
            uint64_t w = 1;
            for (int j = 0; j < half; ++j) {
                uint64_t u = a[blk + j];
                uint64_t v = a[blk + j + half] * w % p;
                uint64_t s = u + v;
                if (s >= p) s -= p;
                a[blk + j] = s;
                a[blk + j + half] = (u >= v) ? u - v : u + p - v;
                w = w * root % p;
            }
        }
    }
    if (invert) {
        uint64_t ninv = oracle_pow((uint64_t)n, p - 2, p);
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int i = 0; i < n; ++i) a[i] = a[i] * ninv % p;
    }
}

void oracle_ntt_multiply_openmp(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p, int threads) {
    if (threads <= 0) threads = oracle_openmp_max_threads();
    int len = n << 1;
    int bit = 0;
    while ((1 << bit) < len) ++bit;
    uint64_t *fa = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)len);
    uint64_t *fb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)len);
    int *rev = (int *)malloc(sizeof(int) * (size_t)len);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < len; ++i) {
        fa[i] = i < n ? a[i] : 0;
        fb[i] = i < n ? b[i] : 0;
    }
    build_rev(rev, len, bit);
    omp_permute(fa, len, rev, threads);
    omp_permute(fb, len, rev, threads);
    omp_stages(fa, len, 0, p, threads);
    omp_stages(fb, len, 0, p, threads);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < len; ++i) fa[i] = fa[i] * fb[i] % p;
    omp_permute(fa, len, rev, threads);
    omp_stages(fa, len, 1, p, threads);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < 2 * n - 1; ++i) result[i] = fa[i];
    free(fa);
    free(fb);
    free(rev);
}

/* Montgomery32.h:19-40 -- R = 2^32; N' = -N^-1 mod R; R2 = R^2 mod N;
 * R_mod_N = REDC(R2).  The inverse is found by Newton iteration here (the
 * reference uses the extended Euclid algorithm; the value is unique). */
int oracle_mont32_init(oracle_mont32 *m, uint32_t N) {
    if (N == 0 || (N & 1u) == 0) return -1;
    uint32_t inv = N; /* correct to 3 bits */
    for (int i = 0; i < 5; ++i) inv *= 2u - N * inv;
    m->N = N;
    m->N_inv_neg = 0u - inv;
    uint64_t r = (1ull << 32) % N;
    m->R2 = (uint32_t)(r * r % N);
    m->R_mod_N = oracle_mont32_redc(m, m->R2);
    return 0;
}

/* Montgomery32.h:43-52 */
uint32_t oracle_mont32_redc(const oracle_mont32 *m, uint64_t T) {
    uint32_t q = (uint32_t)T * m->N_inv_neg;
    uint32_t t = (uint32_t)((T + (uint64_t)q * m->N) >> 32);
    return t >= m->N ? t - m->N : t;
}

/* Montgomery.h:74-90 -- both operands enter the domain through REDC(x*R2),
 * one REDC multiplies, one leaves the domain. */
uint64_t oracle_mont64fields_multiply(uint64_t N, uint64_t a, uint64_t b) {
    oracle_mont32 m;
    if (oracle_mont32_init(&m, (uint32_t)N) != 0) return 0;
    uint64_t aR = oracle_mont32_redc(&m, a * m.R2);
    uint64_t bR = oracle_mont32_redc(&m, b * m.R2);
    return oracle_mont32_redc(&m, oracle_mont32_redc(&m, aR * bR));
}

/* barrett.cc:17-26 */
uint64_t oracle_barrett64_reduce(uint64_t m, uint64_t x) {
    u128 mu = ((u128)1 << 64) / m;
    uint64_t q = (uint64_t)(((u128)x * mu) >> 64);
    uint64_t r = x - q * m;
    return r >= m ? r - m : r;
}

/* main_mpi_CRT.cc:30-39 */
uint64_t oracle_barrett62_reduce(uint64_t m, uint64_t x) {
    uint64_t mu = (1ull << 62) / m;
    uint64_t q = (uint64_t)(((u128)x * mu) >> 62);
    uint64_t r = x - q * m;
    return r >= m ? r - m : r;
}

/* main_naive.cc:13-30 */
int oracle_read_in(const char *path, uint64_t *a, uint64_t *b, int *n, uint64_t *p, int cap) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    unsigned long long pv;
    if (fscanf(f, "%d %llu", n, &pv) != 2 || *n < 0 || *n > cap) { fclose(f); return -1; }
    *p = pv;
    for (int pass = 0; pass < 2; ++pass) {
        uint64_t *dst = pass ? b : a;
        for (int i = 0; i < *n; ++i) {
            unsigned long long v;
            if (fscanf(f, "%llu", &v) != 1) { fclose(f); return -1; }
            dst[i] = v;
        }
    }
    fclose(f);
    return 0;
}

/* main_naive.cc:32-52 */
int oracle_check_out(const char *path, const uint64_t *ab, int n) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    for (int i = 0; i < 2 * n - 1; ++i) {
        unsigned long long v;
        if (fscanf(f, "%llu", &v) != 1) { fclose(f); return -1; }
        if (v != ab[i]) { fclose(f); return 1; }
    }
    fclose(f);
    return 0;
}

/* main_naive.cc:54-67 */
int oracle_write_out(const char *path, const uint64_t *ab, int n) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    for (int i = 0; i < 2 * n - 1; ++i) fprintf(f, "%llu\n", (unsigned long long)ab[i]);
    fclose(f);
    return 0;
}

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_fill_random(uint64_t *dst, size_t count, uint64_t p, uint64_t seed) {
    uint64_t s = seed;
    for (size_t i = 0; i < count; ++i) dst[i] = splitmix64(&s) % p;
}
