/*
 * CPU oracle for the polynomial-multiplication path (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the reference's serial algorithm.  It exists so the
 * tests, the smoke check and the benchmark's cpu_baseline leg have something
 * to compare the GPU library with; nothing in the shipped library links,
 * loads or calls it.
 *
 * Every function cites the reference file:line whose behaviour it restates.
 * Pinned by: nttdata/0-3 (.in/.out), filetest/0-3 and the compiled reference
 * (oracle/_ref) on seeded inputs -- see tests/test_oracle.py.
 */
#ifndef NTT_ORACLE_H
#define NTT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* square-and-multiply, mod < 2^32 (main_naive.cc:69-80) */
uint64_t oracle_pow(uint64_t base, uint64_t exp, uint64_t mod);

/* bit-reversal permutation of a[0..n), n = 2^bit (main_naive.cc:82-95) */
void oracle_reverse(uint64_t *a, int n, int bit);

/* in-place radix-4 DIT transform of n = 4^k values: base-4 digit reversal (main.cc:540-566), then per level
 * len = 4, 16, ... the four-input butterfly with twiddles w^j, w^2j, w^3j and the fourth root w^(len/4)
 * (main.cc:571-630); same output as oracle_ntt for the same n, p, g */
void oracle_ntt_base4(uint64_t *a, int n, int invert, uint64_t p, uint64_t g);

/* in-place radix-2 DIT transform with primitive root g; invert != 0 runs the
 * inverse and scales by n^-1 (main_naive.cc:96-137) */
void oracle_ntt(uint64_t *a, int n, int invert, uint64_t p, uint64_t g);

/* result[0..2n-2] = a*b mod p, n a power of two, p < 2^32 prime with
 * 2n | p-1 and primitive root 3 (main_naive.cc:139-177) */
void oracle_ntt_multiply(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p);

/* same product for p > 2^32 through the four fixed NTT primes and a 128-bit
 * CRT sum (main_naive.cc:178-230) */
void oracle_crt_ntt_multiply(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p);

/* dispatch on p > 2^32 exactly as the reference main does (main_naive.cc:253-257) */
void oracle_poly_multiply_u64(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p);

/* O(n^2) accumulate-into-ab product of the course skeleton (main.cc:92-98) */
void oracle_poly_multiply_schoolbook(const int *a, const int *b, int *ab, int n, int p);

/* schoolbook product for any 64-bit modulus (128-bit intermediates); used to
 * cross-check the CRT path, which no reference golden file covers */
void oracle_schoolbook_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p);

/* OpenMP algorithm of the reference: parallel-for over butterfly blocks per
 * stage, table bit-reversal, '%' arithmetic (main_openMP.cc:116-217).
 * threads <= 0 means "OpenMP default". */
void oracle_ntt_multiply_openmp(const uint64_t *a, const uint64_t *b, uint64_t *result, int n, uint64_t p, int threads);
int oracle_openmp_max_threads(void);

/* Montgomery, R = 2^32 (Montgomery32.h:19-52) */
typedef struct {
    uint32_t N;
    uint32_t N_inv_neg; /* -N^-1 mod 2^32 */
    uint32_t R2;        /* 2^64 mod N */
    uint32_t R_mod_N;   /* 2^32 mod N */
} oracle_mont32;
int oracle_mont32_init(oracle_mont32 *m, uint32_t N); /* 0 ok, -1 if N even or 0 (the reference throws) */
uint32_t oracle_mont32_redc(const oracle_mont32 *m, uint64_t T);

/* Montgomery with 64-bit fields: multiply() of plain operands through four
 * REDCs (Montgomery.h:55-90) */
uint64_t oracle_mont64fields_multiply(uint64_t N, uint64_t a, uint64_t b);

/* Barrett, mu = floor(2^64/m), x < 2^64 (barrett.cc:12-31) */
uint64_t oracle_barrett64_reduce(uint64_t m, uint64_t x);
/* Barrett, mu = floor(2^62/m), shift 62, x < 2^62 (main_mpi_CRT.cc:25-40) */
uint64_t oracle_barrett62_reduce(uint64_t m, uint64_t x);

/* harness text format (main_naive.cc:13-67): ".in" = n p then n + n values;
 * ".out" = 2n-1 values.  Return 0 on success. */
int oracle_read_in(const char *path, uint64_t *a, uint64_t *b, int *n, uint64_t *p, int cap);
int oracle_check_out(const char *path, const uint64_t *ab, int n); /* 0 equal, 1 mismatch, -1 io */
int oracle_write_out(const char *path, const uint64_t *ab, int n);

/* deterministic generator used for synthetic fixtures and the benchmark
 * (splitmix64 stream, value % p) */
void oracle_fill_random(uint64_t *dst, size_t count, uint64_t p, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
