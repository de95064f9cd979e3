// Link glue for running the reference's own program on libnttamd.so (test infrastructure, built only where the reference
// tree is present; see the Makefile recipe `_ref/main_naive_dropin`).
//
// /root/reference/main_naive.cc is compiled unmodified, where it lies.  Its two multiply functions are then marked weak
// in the object file, and the strong definitions below -- the binding a maintainer would write -- forward them to the
// library's drop-in symbol, so the reference's own main(), fRead, fCheck and fWrite run around the GPU product
// (call sites main_naive.cc:253-257).
//
// The reference opens its data under the absolute directory "/nttdata/" (main_naive.cc:15,34).  Where that directory
// cannot be created, NTT_DATA_DIR names the directory to use instead: fopen64/fopen below (what std::ifstream::open
// ends up calling) redirect exactly that prefix and nothing else.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "ntt_amd.h"

void NTT_multiply(uint64_t *a, uint64_t *b, uint64_t *result, int n, uint64_t p) { poly_multiply_u64(a, b, result, n, p); }
void CRT_NTT_multiply_serial(uint64_t *a, uint64_t *b, uint64_t *result, int n, uint64_t p) { poly_multiply_u64(a, b, result, n, p); }

static std::string redirect(const char *path) {
    const char *dir = getenv("NTT_DATA_DIR");
    if (dir && path && strncmp(path, "/nttdata/", 9) == 0) return std::string(dir) + "/" + (path + 9);
    return path ? std::string(path) : std::string();
}

extern "C" FILE *fopen64(const char *path, const char *mode) {
    typedef FILE *(*fn_t)(const char *, const char *);
    static fn_t next = (fn_t)dlsym(RTLD_NEXT, "fopen64");
    return next(redirect(path).c_str(), mode);
}
extern "C" FILE *fopen(const char *path, const char *mode) {
    typedef FILE *(*fn_t)(const char *, const char *);
    static fn_t next = (fn_t)dlsym(RTLD_NEXT, "fopen");
    return next(redirect(path).c_str(), mode);
}
