/*
 * ntt_amd -- polynomial multiplication modulo NTT-friendly primes on AMD MI355X (gfx950).
 *
 * C ABI of libnttamd.so.  Plain pointers and sizes only; every entry point is
 * what a binding for the reference's multiply path would call.  The library is
 * HIP-only: there is no CPU fallback, and every call fails (status < 0, message
 * in ntt_last_error()) when no GPU is usable.
 *
 * Conventions (same as the reference programs): a and b hold n coefficients in
 * [0, p); the product has 2n-1 coefficients, each fully reduced to [0, p);
 * p is prime with primitive root 3 and transform length L | p-1, where
 * L = smallest power of two >= 2n-1 (L = 2n when n is a power of two).
 * Moduli above 2^32 -- and any modulus that is not NTT-friendly -- go through
 * four fixed 31-bit NTT primes and a 128-bit CRT merge (the reference's route,
 * main_naive.cc:178-230).  Sizes and moduli beyond that basis (n*(p-1)^2 >= 2^118.9
 * or transform lengths above 2^18) are computed in the 64-bit field of p itself when
 * p is NTT-friendly, and otherwise through two or three 50- or 59-bit NTT primes with a
 * Garner recombination: every modulus 2 <= p < 2^64 is served.
 *
 * Threads: every entry point takes one library-wide lock, so concurrent callers are
 * serialised, not refused (the reference is single-threaded at this boundary).  The
 * device-pointer entry points only enqueue work: it is ordered on the stream they are
 * given, and work enqueued on different streams or devices may overlap on the hardware.
 * A multi-device host call runs one host thread per device inside the library.
 */
#ifndef NTT_AMD_H
#define NTT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define NTT_OK 0
#define NTT_ERR_ARG (-1)      /* bad size / pointer / modulus */
#define NTT_ERR_MODULUS (-2)  /* modulus unsupported for this n (see ntt_last_error) */
#define NTT_ERR_DEVICE (-3)   /* HIP runtime error or no GPU */
#define NTT_ERR_WORKSPACE (-4)/* caller workspace too small */
#define NTT_ERR_IO (-5)       /* data file missing or malformed */

/* Library lifetime.  ntt_init(n_gpus) selects devices 0..n_gpus-1 (n_gpus <= 0: all visible);
 * it replaces the reference's pool/global set-up calls (pthread pool creation main_pthread.cc:755-761,
 * init_global_crt_values main_openMP.cc:337-357).  Optional: the first multiply initialises lazily. */
int ntt_init(int n_gpus);
void ntt_shutdown(void);
int ntt_device_count(void);
/* message for the last non-zero status on this thread ("" if none) */
const char *ntt_last_error(void);
/* Device-pointer calls given workspace == NULL use a workspace the library owns: one per stream, on the current device,
 * grown to the largest call seen and kept until ntt_shutdown (one n = 2^28 product pins 8 GiB) -- the static pools of the
 * reference's threaded programs (main_pthread.cc:270-273) are the same idea.  At most 8 streams per device keep one; a
 * ninth takes the place of the least recently used (after a device-wide wait).  ntt_release_workspace frees the one of
 * `stream` after waiting for that stream, or, with all_streams != 0, every one of the current device after waiting for
 * the device.  Call it before destroying a stream whose handle may be reused. */
int ntt_release_workspace(void *stream, int all_streams);

/* How a modulus above 2^31 is served.  The reference splits every such product over four 32-bit NTT primes and merges
 * the residues with 128-bit arithmetic (CRT_NTT_multiply_serial, main_naive.cc:178-230).  Its one large modulus,
 * 1337006139375617 = 19 * 2^46 + 1, is itself NTT-friendly, and for such primes up to 1.5e15 the library has a direct
 * transform in exact double-precision modular arithmetic that gives the same residues at about half the cost.
 *   NTT_ROUTE_AUTO       (default) the fastest exact route: the direct 64-bit field where the modulus allows it, the
 *                        four-prime CRT otherwise
 *   NTT_ROUTE_REFERENCE  the reference's own choice: the four-prime CRT wherever it applies
 * ntt_set_route_policy returns the previous policy (-1 for an unknown one); it is process-wide and affects the calls
 * that choose a route themselves (poly_multiply_u64, the *_u64 batched entry points, ntt_workspace_bytes,
 * ntt_describe_plan), not the explicit per-route entry points below. */
#define NTT_ROUTE_AUTO 0
#define NTT_ROUTE_REFERENCE 1
int ntt_set_route_policy(int policy);

/* Drop-in for the course skeleton's entry point `void poly_multiply(int *a, int *b, int *ab, int n, int p)`
 * (main.cc:92-98, call site main.cc:1268-1269).  Writes ab[0..2n-2]; like every NTT path of the
 * reference it overwrites rather than accumulates.  Aborts the process with a message on failure,
 * because the reference signature has no error channel. */
void poly_multiply(const int *a, const int *b, int *ab, int n, int p);

/* Drop-in for the graded programs' 64-bit entry points: `NTT_multiply` (main_naive.cc:139-177) when
 * p <= 2^32 and `CRT_NTT_multiply_serial` (main_naive.cc:178-230) when p > 2^32 -- the dispatch of
 * main_naive.cc:253-257 (and main_openMP.cc:431-435, main_pthread.cc:778-782, main_mpi_CRT.cc:578).
 * Host pointers, synchronous.  Aborts on failure like poly_multiply. */
void poly_multiply_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p);

/* Status-returning forms of the two calls above (same arguments, same results). */
int ntt_poly_multiply_i32(const int *a, const int *b, int *ab, int n, int p);
int ntt_poly_multiply_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p);

/* Batch of independent products with 32-bit residues, host memory.  a, b: batch rows of n values;
 * ab: batch rows of 2n-1 values, all contiguous.  The batch is split into contiguous shards over
 * n_gpus devices (<= 0: all initialised devices) with no inter-device communication -- the
 * replacement for running the reference once per product. */
int ntt_poly_multiply_batched(const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p,
                              size_t batch, int n_gpus);

/* Same, 64-bit coefficients and any modulus. */
int ntt_poly_multiply_batched_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p,
                                  size_t batch, int n_gpus);

/* Device-resident batch on the current HIP device: d_a, d_b, d_ab are device pointers, rows
 * a_stride / b_stride / ab_stride elements apart (>= n, n, 2n-1).  Work is enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = default stream) and the call returns without
 * synchronising.  `workspace` is device memory of at least ntt_workspace_bytes(n, p, batch) bytes
 * (NULL: the library uses and grows its own, one per stream); the size depends on the route that serves (n, p), so
 * ask again after ntt_set_route_policy.  This is the path the benchmark times. */
size_t ntt_workspace_bytes(int n, uint64_t p, size_t batch);
int ntt_poly_multiply_batched_device(const uint32_t *d_a, const uint32_t *d_b, uint32_t *d_ab, int n, uint32_t p,
                                     size_t batch, size_t a_stride, size_t b_stride, size_t ab_stride,
                                     void *workspace, size_t workspace_bytes, void *stream);
int ntt_poly_multiply_batched_device_u64(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_ab, int n, uint64_t p,
                                         size_t batch, size_t a_stride, size_t b_stride, size_t ab_stride,
                                         void *workspace, size_t workspace_bytes, void *stream);

/* The products of ntt_poly_multiply_batched_device for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference, whose multiply
 * allocates its temporaries on every call (new[]/delete[], main_naive.cc:143-144,175-176).  The call allocates nothing,
 * on the device or through the runtime: the intermediates of the multi-pass path live in a static device array of
 * 32 MiB that every GPU the library runs on holds until the process ends, and the twiddle tables must already be cached --
 * call ntt_init() and run one product with the same n and p through ntt_poly_multiply_batched_device on this device
 * first.  Limits: batch * 2 * L * 4 bytes <= 32 MiB for transform lengths L > 8192 (16 products at n = 2^17, 128 at
 * n = 2^14); shorter products need no scratch and any batch is accepted.  Larger calls fail with NTT_ERR_WORKSPACE,
 * unprepared ones with NTT_ERR_ARG / NTT_ERR_DEVICE.  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU (two streams) race; calls on one stream are ordered and safe.  Nothing in the array needs clearing between
 * calls (every word used is written before it is read). */
int ntt_poly_multiply_batched_device_noalloc(const uint32_t *d_a, const uint32_t *d_b, uint32_t *d_ab, int n, uint32_t p,
                                             size_t batch, size_t a_stride, size_t b_stride, size_t ab_stride, void *stream);

/* The CRT recombination on its own (the merge loop of main_naive.cc:216-223 / the slice merge of
 * main_mpi_CRT.cc:512-522): d_res holds the four residue vectors (moduli 1004535809, 1224736769,
 * 469762049, 998244353 in this order), res_stride elements apart, count values each; writes
 * d_out[i] = (CRT(res[0..3][i]) mod M) mod p.  Device pointers, enqueued on `stream`. */
int ntt_crt_merge_device(const uint32_t *d_res, size_t res_stride, uint64_t *d_out, size_t count, uint64_t p,
                         void *stream);

/* Residue products for one CRT prime only (prime_index 0..3, order as above): d_res[row] =
 * (a[row] mod q) * (b[row] mod q) mod q, the per-rank work of main_mpi_CRT.cc:470-492.  Lets a
 * multi-GPU caller place primes on different devices and exchange residues itself. */
int ntt_crt_residue_product_device(const uint64_t *d_a, const uint64_t *d_b, uint32_t *d_res, int n, int prime_index,
                                   size_t batch, size_t a_stride, size_t b_stride, size_t res_stride,
                                   void *workspace, size_t workspace_bytes, void *stream);
uint32_t ntt_crt_prime(int prime_index);

/* ---- 64-bit prime field (odd p < 2^60 with 2^log_root | p-1 and 3 generating that subgroup) ----
 * Building blocks of one very long product (transform length beyond the 2-adicity of any 32-bit NTT
 * prime), on one GPU or sharded over ranks with an all-to-all between the two axes of the four-step
 * matrix.  They take the place of the reference's distributed single transform, which replicates the
 * array and all-reduces it after every stage (main_mpi_t.cc:139-291).  All pointers are device
 * pointers, all calls asynchronous on `stream`.  log_root = log2 of the full transform length of the
 * product; every call of one product passes the same value (it selects the root-of-unity tables).
 *
 * ntt64_transform_device: in-place transform of length 2^log_len along the middle axis of an
 *   [outer][2^log_len][inner] array.  Forward: natural order in, bit-reversed order out, values
 *   in [0,2p).  Inverse: bit-reversed in, natural out, unscaled, values in [0,4p).
 * ntt64_twiddle_device: data[r][c] *= w^(+-k(r) * (col0+c)), w of order 2^log_root, for a rows x cols
 *   tile whose first row/column are row0/col0 of the four-step matrix; k(r) = row0+r, bit-reversed
 *   over row_bits bits when rows_bitrev != 0 (the order a forward row-axis transform leaves).
 * ntt64_fourstep_rows_device: the row-axis half of a four-step transform of length 2^log_total on a column slab
 *   [2^log_rows][cols_local] holding columns col0 .. col0+cols_local-1 of the 2^log_rows x 2^(log_total-log_rows) matrix
 *   (element i = j + C*r at row r, column j): the first log_rows levels of the long transform, i.e. the transforms
 *   along the rows axis AND the twiddles w^(j*k) that join the two axes, in the same passes over memory.  Inverse: the
 *   mirror (twiddles first).  cols_local must be a power of two.  This plus one all-to-all plus
 * ntt64_fourstep_cols_device is the whole transform: it runs the remaining log_cols levels on the layout the
 *   all-to-all of column slabs leaves, without re-tiling: 2^log_parts parts (one per sending rank; the rank count is a
 *   power of two and at most 16, i.e. log_parts <= 4 -- more ranks are refused with NTT_ERR_ARG),
 *   part_stride words apart, each holding `rows` rows of 2^(log_cols-log_parts) consecutive columns.  `rows` may be a
 *   sub-range of a part (pass d_rows + first_row * 2^(log_cols-log_parts)), so that arriving chunks can be transformed
 *   while later ones are still in flight.
 * ntt64_fourstep_rows_load_device / ntt64_fourstep_rows_store_device: the same row-axis passes with the element-wise
 *   ends of a product folded in.  load: forward, reading any 64-bit words from d_src (a slab of the same shape; rows
 *   from rows_valid on count as zero and are never read -- the zero padding of the operands) and leaving the result in
 *   d_slab.  store: inverse on d_slab, the last pass writing the scaled canonical coefficients (what
 *   ntt64_finalize_device produces) to d_out, same shape.
 * ntt64_fourstep_cols_product_device: the middle of a four-step product between the exchanges, on the layout of
 *   ntt64_fourstep_cols_device: forward column-axis levels of d_a_rows and d_b_rows, pointwise product, inverse
 *   column-axis levels, result in d_a_rows (d_b_rows is overwritten with its spectrum).  When a part's transforms end
 *   in full 4096-word chunks the last forward passes, the product and the first inverse pass are one kernel.
 * ntt64_pointwise_device: a[i] = a[i]*b[i]*c mod p, c a constant of the arithmetic in use for p (2^-64 in Montgomery
 *   integers, 1 in double precision) that ntt64_finalize_device undoes.
 * ntt64_load_device: dst[row][i] = i < n ? src[row*src_stride + i] mod p : 0, i < len (may be in place).
 * ntt64_finalize_device: out[row*out_stride + i] = data[row*len + i] * 2^-log_total * 2^64 mod p, canonical.
 * ntt64_poly_multiply_device: the whole product on one GPU: d_out[0..2n-2] = a*b mod p.  Same results
 *   as the CRT route of ntt_poly_multiply_batched_device_u64 wherever both apply.
 * ntt64_poly_multiply_batched_device: `batch` such products in the same passes over memory (rows of a, b and the
 *   results a_stride, b_stride, out_stride words apart); workspace from ntt64_workspace_bytes_batched, or NULL for
 *   the library's own. */
int ntt64_transform_device(uint64_t *d_data, uint64_t p, int log_root, int log_len, size_t inner, size_t outer,
                           int inverse, void *stream);
int ntt64_fourstep_rows_device(uint64_t *d_slab, uint64_t p, int log_total, int log_rows, size_t cols_local, uint64_t col0,
                               int inverse, void *stream);
int ntt64_fourstep_cols_device(uint64_t *d_rows, uint64_t p, int log_total, int log_cols, int log_parts, size_t part_stride,
                               size_t rows, int inverse, void *stream);
int ntt64_fourstep_rows_load_device(const uint64_t *d_src, size_t rows_valid, uint64_t *d_slab, uint64_t p, int log_total,
                                    int log_rows, size_t cols_local, uint64_t col0, void *stream);
int ntt64_fourstep_rows_store_device(uint64_t *d_slab, uint64_t *d_out, uint64_t p, int log_total, int log_rows,
                                     size_t cols_local, uint64_t col0, void *stream);
int ntt64_fourstep_cols_product_device(uint64_t *d_a_rows, uint64_t *d_b_rows, uint64_t p, int log_total, int log_cols,
                                       int log_parts, size_t part_stride, size_t rows, void *stream);
int ntt64_twiddle_device(uint64_t *d_data, uint64_t p, int log_root, size_t rows, size_t cols, uint64_t row0,
                         uint64_t col0, int row_bits, int rows_bitrev, int inverse, void *stream);
int ntt64_pointwise_device(uint64_t *d_a, const uint64_t *d_b, size_t count, uint64_t p, int log_root, void *stream);
int ntt64_load_device(const uint64_t *d_src, size_t src_stride, size_t n, uint64_t *d_dst, size_t len, size_t rows,
                      uint64_t p, int log_root, void *stream);
int ntt64_finalize_device(const uint64_t *d_data, size_t len, uint64_t *d_out, size_t out_stride, size_t n_out,
                          size_t rows, uint64_t p, int log_root, int log_total, void *stream);
size_t ntt64_workspace_bytes(size_t n);
int ntt64_poly_multiply_device(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n, uint64_t p,
                               void *workspace, size_t workspace_bytes, void *stream);
size_t ntt64_workspace_bytes_batched(size_t n, size_t batch);
int ntt64_poly_multiply_batched_device(const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n, uint64_t p,
                                       size_t batch, size_t a_stride, size_t b_stride, size_t out_stride, void *workspace,
                                       size_t workspace_bytes, void *stream);

/* Stand-alone transform in natural order, in place: the reference's `NTT(a, n, invert, p, 3)`
 * (main_naive.cc:96-137) including its bit-reversal (main_naive.cc:82-95) and, for invert != 0, the
 * n^-1 scaling (main_naive.cc:131-136).  n a power of two with n | p-1, p < 2^60; values are reduced
 * mod p on the way in and canonical on the way out.  Host form (one row) and device form (batch rows). */
int ntt_transform_u64(uint64_t *a, int n, int invert, uint64_t p);
int ntt_transform_device_u64(uint64_t *d_a, size_t n, size_t batch, int invert, uint64_t p, void *stream);
/* The same transform on 32-bit residues (p < 2^31, n = 2 .. 2^18 with n | p-1, values in [0, p)), in place on batch rows
 * of n values: three passes of 8 bytes per value through a library-owned work array of the same size (none for
 * n <= 4096).  The two entry points above take this route too whenever the modulus allows it, with 8-byte words at the
 * two ends only. */
int ntt_transform_device_u32(uint32_t *d_a, size_t n, size_t batch, int invert, uint32_t p, void *stream);

/* Data files of the reference harness, host-only (no GPU needed): `<id>.in` = "n p", n coefficients of a,
 * n of b, white-space separated (fRead, main_naive.cc:13-30); `<id>.out` = 2n-1 values, one per line
 * (fCheck main_naive.cc:32-52, fWrite :54-67).  The files are mapped and scanned once instead of going
 * through iostream extraction.  ntt_io_read_header gives n and p so the caller can size its arrays;
 * ntt_io_read_in then fills a[0..n-1], b[0..n-1].  flags: NTT_IO_SIDECAR_READ takes the arrays from the
 * binary sidecar `<path>.bin` when it was made from the text file as it is now (same size and
 * modification time -- the text file stays the source of truth); NTT_IO_SIDECAR_WRITE (re)creates the
 * sidecar after a text parse (silently skipped when the directory is read-only).
 * ntt_io_check_out: 1 when the first `count` values of the file equal ab, 0 when they differ or the file
 * is short (fCheck's "wrong"), NTT_ERR_IO when it cannot be opened.  ntt_io_read_out reads them instead.
 * ntt_io_write_out writes `count` values, one per line, byte-identical to fWrite's output. */
#define NTT_IO_SIDECAR_READ 1
#define NTT_IO_SIDECAR_WRITE 2
int ntt_io_read_header(const char *path, int64_t *n, uint64_t *p);
int ntt_io_read_in(const char *path, uint64_t *a, uint64_t *b, int64_t n, int flags);
int ntt_io_check_out(const char *path, const uint64_t *ab, int64_t count);
int ntt_io_read_out(const char *path, uint64_t *ab, int64_t count);
int ntt_io_write_out(const char *path, const uint64_t *ab, int64_t count);

/* Per-kernel timing for benchmarks (no counterpart in the reference, whose only clock is the
 * chrono pair around the multiply, main_naive.cc:251-261).  While enabled, every kernel launch is
 * bracketed by HIP events on the stream it is launched on.  ntt_profile_read waits for the
 * recorded events, writes a comma-separated list of kernel labels into names and, per label,
 * the summed duration in ms and the number of launches; returns the number of labels. */
int ntt_profile_enable(int on);
int ntt_profile_read(char *names, size_t names_len, double *total_ms, int *launches, int max_kernels);

/* Name and per-call launch count of the kernels a (n, p) product uses, for profiling tools:
 * writes a comma-separated list into buf. */
int ntt_describe_plan(int n, uint64_t p, char *buf, size_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* NTT_AMD_H */
