// Host-code sanitizer run of the library (test infrastructure): the library's sources are compiled for the CPU over the
// HIP stand-in of tests/cpu_twin together with this driver under AddressSanitizer and UndefinedBehaviorSanitizer, and a
// sweep over the entry points is checked against the oracle.  Catches out-of-bounds indexing in kernels (the stand-in
// runs them as ordinary loops over heap memory), leaks of plan/workspace bookkeeping and undefined integer behaviour in
// the host-side plan mathematics.  Built and run by tests/test_host_sanitizers.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/ntt_amd.h"
extern "C" {
#include "../../oracle/ntt_oracle.h"
}

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static const uint64_t BIGP = 1337006139375617ull;

static std::vector<uint64_t> want_product(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b, uint64_t p) {
    const size_t n = a.size();
    size_t m = 1;
    while (m < n) m <<= 1;
    std::vector<uint64_t> out(2 * n - 1);
    if (m <= 2048 || 2 * m > (1u << 18)) {   // short: the 128-bit schoolbook product; long: not used here
        oracle_schoolbook_u64(a.data(), b.data(), out.data(), (int)n, p);
        return out;
    }
    std::vector<uint64_t> az(m, 0), bz(m, 0), full(2 * m, 0);
    memcpy(az.data(), a.data(), n * 8);
    memcpy(bz.data(), b.data(), n * 8);
    oracle_poly_multiply_u64(az.data(), bz.data(), full.data(), (int)m, p);
    memcpy(out.data(), full.data(), (2 * n - 1) * 8);
    return out;
}

static void one_product(size_t n, uint64_t p, uint64_t seed) {
    std::vector<uint64_t> a(n), b(n), out(2 * n - 1 + 3, 0xABABABABABABABABull);
    oracle_fill_random(a.data(), n, p, seed);
    oracle_fill_random(b.data(), n, p, seed + 1);
    const int rc = ntt_poly_multiply_u64(a.data(), b.data(), out.data(), (int)n, p);
    CHECK(rc == NTT_OK, "n=%zu p=%llu rc=%d %s", n, (unsigned long long)p, rc, ntt_last_error());
    if (rc != NTT_OK) return;
    const std::vector<uint64_t> want = want_product(a, b, p);
    CHECK(memcmp(out.data(), want.data(), (2 * n - 1) * 8) == 0, "wrong product n=%zu p=%llu", n, (unsigned long long)p);
    for (int k = 0; k < 3; ++k) CHECK(out[2 * n - 1 + k] == 0xABABABABABABABABull, "wrote past 2n-1, n=%zu", n);
}

int main() {
    CHECK(ntt_init(1) == NTT_OK, "init: %s", ntt_last_error());
    const uint64_t primes32[] = {7340033ull, 998244353ull, 1224736769ull};
    const size_t sizes[] = {1, 2, 3, 17, 100, 1000, 2048, 4097, 8192, 20000, 65536};
    for (uint64_t p : primes32)
        for (size_t n : sizes) one_product(n, p, n * 7 + p % 11);
    for (int policy : {NTT_ROUTE_AUTO, NTT_ROUTE_REFERENCE}) {
        ntt_set_route_policy(policy);
        for (size_t n : {(size_t)1, (size_t)5, (size_t)700, (size_t)4096, (size_t)5000, (size_t)40000, (size_t)131072}) one_product(n, BIGP, n + 3);
    }
    ntt_set_route_policy(NTT_ROUTE_AUTO);
    for (uint64_t p : {(uint64_t)1000000007, ~(uint64_t)0 - 58, (uint64_t)6, (uint64_t)2, (uint64_t)4294967311})
        for (size_t n : {(size_t)1, (size_t)9, (size_t)300, (size_t)1500}) one_product(n, p, n + p % 13);
    // batched, 32-bit rows and 64-bit rows, odd batch counts
    {
        const int n = 1024; const uint32_t p = 998244353u; const size_t batch = 37;
        std::vector<uint32_t> a(batch * n), b(batch * n), out(batch * (2 * n - 1));
        std::vector<uint64_t> tmp(n);
        for (size_t r = 0; r < batch; ++r) {
            oracle_fill_random(tmp.data(), n, p, 100 + r); for (int i = 0; i < n; ++i) a[r * n + i] = (uint32_t)tmp[i];
            oracle_fill_random(tmp.data(), n, p, 900 + r); for (int i = 0; i < n; ++i) b[r * n + i] = (uint32_t)tmp[i];
        }
        CHECK(ntt_poly_multiply_batched(a.data(), b.data(), out.data(), n, p, batch, 1) == NTT_OK, "batched: %s", ntt_last_error());
        for (size_t r : {(size_t)0, (size_t)18, batch - 1}) {
            std::vector<uint64_t> x(n), y(n), w(2 * n);
            for (int i = 0; i < n; ++i) { x[i] = a[r * n + i]; y[i] = b[r * n + i]; }
            oracle_ntt_multiply(x.data(), y.data(), w.data(), n, p);
            bool same = true;
            for (int i = 0; i < 2 * n - 1; ++i) same = same && out[r * (2 * n - 1) + i] == (uint32_t)w[i];
            CHECK(same, "batched row %zu", r);
        }
    }
    {
        const int n = 2048; const size_t batch = 5;
        std::vector<uint64_t> a(batch * n), b(batch * n), out(batch * (2 * n - 1)), w(2 * n);
        oracle_fill_random(a.data(), a.size(), BIGP, 5); oracle_fill_random(b.data(), b.size(), BIGP, 6);
        CHECK(ntt_poly_multiply_batched_u64(a.data(), b.data(), out.data(), n, BIGP, batch, 1) == NTT_OK, "batched u64: %s", ntt_last_error());
        for (size_t r = 0; r < batch; ++r) {
            oracle_crt_ntt_multiply(a.data() + r * n, b.data() + r * n, w.data(), n, BIGP);
            CHECK(memcmp(out.data() + r * (2 * n - 1), w.data(), (2 * n - 1) * 8) == 0, "batched u64 row %zu", r);
        }
    }
    // stand-alone transforms
    for (int n : {1, 2, 8, 32, 64, 4096, 8192, 16384, 1 << 18}) {
        std::vector<uint64_t> x(n), y(n);
        oracle_fill_random(x.data(), n, 998244353ull, n);
        y = x;
        CHECK(ntt_transform_u64(y.data(), n, 0, 998244353ull) == NTT_OK, "transform: %s", ntt_last_error());
        std::vector<uint64_t> ref = x;
        oracle_ntt(ref.data(), n, 0, 998244353ull, 3);
        CHECK(y == ref, "forward transform n=%d", n);
        CHECK(ntt_transform_u64(y.data(), n, 1, 998244353ull) == NTT_OK && y == x, "inverse transform n=%d", n);
    }
    // the same on 32-bit residues: several rows per call, batches that do not fill the last workgroup, a modulus without a
    // 64th root of unity (radix-2 kernel)
    for (uint32_t p : {998244353u, 1224736769u, 257u}) {
        for (int n : {2, 16, 32, 256, 4096, 8192, 32768}) {
            if ((p - 1) % (uint32_t)n) continue;
            const size_t batch = 3;
            std::vector<uint32_t> x(batch * n + 2, 0xCDCDCDCDu), y;
            std::vector<uint64_t> tmp(n);
            for (size_t r = 0; r < batch; ++r) {
                oracle_fill_random(tmp.data(), n, p, 40 + r + n);
                for (int i = 0; i < n; ++i) x[r * n + i] = (uint32_t)tmp[i];
            }
            y = x;
            CHECK(ntt_transform_device_u32(y.data(), n, batch, 0, p, nullptr) == NTT_OK, "transform u32: %s", ntt_last_error());
            for (size_t r = 0; r < batch; ++r) {
                for (int i = 0; i < n; ++i) tmp[i] = x[r * n + i];
                oracle_ntt(tmp.data(), n, 0, p, 3);
                bool same = true;
                for (int i = 0; i < n; ++i) same = same && y[r * n + i] == (uint32_t)tmp[i];
                CHECK(same, "u32 forward transform n=%d p=%u row %zu", n, p, r);
            }
            CHECK(ntt_transform_device_u32(y.data(), n, batch, 1, p, nullptr) == NTT_OK && y == x, "u32 round trip n=%d p=%u", n, p);
        }
    }
    // argument errors leave memory alone
    {
        uint64_t one[4] = {1, 1, 1, 1}, out[7] = {9, 9, 9, 9, 9, 9, 9};
        CHECK(ntt_poly_multiply_u64(one, one, out, 4, 1) != NTT_OK && out[0] == 9, "modulus 1 accepted");
        CHECK(ntt_poly_multiply_u64(one, one, out, 0, 97) != NTT_OK, "n = 0 accepted");
        CHECK(ntt_poly_multiply_u64(nullptr, one, out, 4, 97) != NTT_OK, "null accepted");
    }
    ntt_shutdown();
    CHECK(ntt_init(1) == NTT_OK, "re-init");
    one_product(3000, 998244353ull, 1);
    ntt_shutdown();
    if (failures) { fprintf(stderr, "%d failure(s)\n", failures); return 1; }
    printf("HOST SANITIZE OK\n");
    return 0;
}
