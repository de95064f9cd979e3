// Text data files of the reference harness, read and written at memory speed.
//
// Formats (main_naive.cc:13-67): <id>.in = "n p" followed by n coefficients of a and n of b,
// whitespace separated; <id>.out = 2n-1 product coefficients, one per line.  The reference goes
// through iostream extraction (fRead main_naive.cc:13-30, fCheck :32-50) and `fout << x << '\n'`
// (fWrite :52-67); at n = 131072 that costs tens of milliseconds per file, two orders of
// magnitude more than the GPU multiply.  Here the file is mapped and scanned once.
//
// Optional binary sidecar "<path>.bin": the parsed arrays plus the size and modification time
// of the text file they came from.  The text file stays the source of truth -- a sidecar is used
// only while both still match, and is rewritten after any text parse when asked for.
//
// Host-only code: no device calls, usable without a GPU.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/ntt_amd.h"

namespace nttamd {
void set_error_text(const std::string& s);  // ntt_api.cpp
}

namespace {

int io_fail(const std::string& msg) {
    nttamd::set_error_text(msg);
    return NTT_ERR_IO;
}

struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    uint64_t mtime_ns = 0;
    int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); fd = -1; return false; }
        size = (size_t)st.st_size;
        mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
        if (size == 0) { data = ""; return true; }
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);  // one batched fault-in
        if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
        madvise(m, size, MADV_SEQUENTIAL);
        data = (const char*)m;
        return true;
    }
    ~MappedFile() {
        if (data && size) munmap((void*)data, size);
        if (fd >= 0) ::close(fd);
    }
};

// the six white-space characters of the "C" locale, which is what istream extraction skips
inline bool is_space(char c) { return c == ' ' || (unsigned)(c - '\t') < 5u; }

// Next unsigned decimal token at or after *cur.  1 = token read, 0 = end of data, -1 = malformed
// (a character that is neither digit nor white space, or a value beyond 64 bits).
inline int next_u64(const char*& cur, const char* end, uint64_t& out) {
    while (cur < end && is_space(*cur)) ++cur;
    if (cur >= end) return 0;
    if (*cur == '+') ++cur;  // istream extraction accepts a leading plus sign
    const char* first = cur;
    uint64_t v = 0;
    while (cur < end) {
        const unsigned d = (unsigned)(*cur - '0');
        if (d > 9) break;
        v = v * 10 + d;
        ++cur;
    }
    const size_t digits = (size_t)(cur - first);
    if (digits == 0) return -1;
    if (cur < end && !is_space(*cur)) return -1;
    if (digits >= 20) {  // only now can 64 bits have overflowed: redo with checks (leading zeros allowed)
        v = 0;
        for (const char* q = first; q < cur; ++q) {
            const unsigned d = (unsigned)(*q - '0');
            if (v > (UINT64_MAX - d) / 10) return -1;
            v = v * 10 + d;
        }
    }
    out = v;
    return 1;
}

// Worker count for a buffer of `bytes` bytes: one thread per 256 KiB, at most 8.
inline int worker_count(size_t bytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    size_t t = bytes / (256u << 10);
    if (t > 8) t = 8;
    if (hw && t > hw) t = hw;
    return t < 1 ? 1 : (int)t;
}

// Parses the first `want` tokens of [data, data+size) into dst[0..split) followed by dst2[0..want-split).
// With several workers the buffer is cut at white space into one piece each; a piece is parsed into a
// private array (its token count is not known beforehand) and copied to its place once the counts of the
// pieces before it are known.  Returns 1 on success, 0 when the data holds fewer tokens, -1 on a malformed
// token among the first `want`.
int parse_tokens(const char* data, size_t size, uint64_t* dst, size_t want, size_t split = ~(size_t)0, uint64_t* dst2 = nullptr) {
    auto put = [&](size_t i, uint64_t v) { if (i < split) dst[i] = v; else dst2[i - split] = v; };
    const int T = worker_count(size);
    if (T == 1) {
        const char* cur = data;
        const char* end = data + size;
        for (size_t i = 0; i < want; ++i) {
            uint64_t v;
            const int r = next_u64(cur, end, v);
            if (r != 1) return r;
            put(i, v);
        }
        return 1;
    }
    std::vector<const char*> cut(T + 1);
    cut[0] = data; cut[T] = data + size;
    for (int t = 1; t < T; ++t) {
        const char* c = data + size / T * t;
        while (c < data + size && !is_space(*c)) ++c;  // never split a token
        cut[t] = c;
    }
    for (int t = 1; t <= T; ++t) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    std::vector<std::vector<uint64_t>> local(T);
    std::vector<int> bad(T, 0);  // piece stopped at a malformed token (after local[t].size() good ones)
    auto parse_piece = [&](int t) {
        const char* cur = cut[t];
        const char* end = cut[t + 1];
        std::vector<uint64_t>& out = local[t];
        out.reserve((size_t)(end - cur) / 2 + 1);  // a token takes at least two bytes, separator included
        while (out.size() < want) {
            uint64_t v;
            const int r = next_u64(cur, end, v);
            if (r == 0) break;
            if (r < 0) { bad[t] = 1; break; }
            out.push_back(v);
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(parse_piece, t);
        parse_piece(0);
        for (auto& x : th) x.join();
    }
    size_t at = 0;
    for (int t = 0; t < T && at < want; ++t) {
        const size_t take = local[t].size() < want - at ? local[t].size() : want - at;
        for (size_t k = 0; k < take; ++k) put(at + k, local[t][k]);
        at += take;
        if (at < want && bad[t]) return -1;
    }
    return at >= want ? 1 : 0;
}

// ---- sidecar ----
struct SidecarHeader {
    char magic[8];       // "NTTBIN01"
    uint64_t text_size;  // size of the text file the arrays were parsed from
    uint64_t text_mtime_ns;
    uint64_t n;
    uint64_t p;
};
const char kMagic[8] = {'N', 'T', 'T', 'B', 'I', 'N', '0', '1'};

bool sidecar_load(const std::string& bin_path, const MappedFile& text, int64_t* n, uint64_t* p, uint64_t* a, uint64_t* b,
                  int64_t want_n) {
    MappedFile bin;
    if (!bin.open(bin_path.c_str()) || bin.size < sizeof(SidecarHeader)) return false;
    SidecarHeader h;
    memcpy(&h, bin.data, sizeof h);
    if (memcmp(h.magic, kMagic, 8) != 0 || h.text_size != text.size || h.text_mtime_ns != text.mtime_ns) return false;
    if (h.n == 0 || h.n > (uint64_t)INT64_MAX / 16 || bin.size != sizeof h + 2 * h.n * sizeof(uint64_t)) return false;
    if (n) *n = (int64_t)h.n;
    if (p) *p = h.p;
    if (a && b) {
        if ((int64_t)h.n != want_n) return false;
        memcpy(a, bin.data + sizeof h, h.n * sizeof(uint64_t));
        memcpy(b, bin.data + sizeof h + h.n * sizeof(uint64_t), h.n * sizeof(uint64_t));
    }
    return true;
}

void sidecar_store(const std::string& bin_path, const MappedFile& text, uint64_t n, uint64_t p, const uint64_t* a,
                   const uint64_t* b) {
    // written under a temporary name and renamed, so a reader never sees a partial file; failure to
    // write (read-only data directory) is not an error -- the text file remains authoritative
    const std::string tmp = bin_path + ".tmp" + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    SidecarHeader h;
    memcpy(h.magic, kMagic, 8);
    h.text_size = text.size; h.text_mtime_ns = text.mtime_ns; h.n = n; h.p = p;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(a, sizeof(uint64_t), n, f) == n && fwrite(b, sizeof(uint64_t), n, f) == n;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), bin_path.c_str()) != 0) unlink(tmp.c_str());
}

}  // namespace

extern "C" {

int ntt_io_read_header(const char* path, int64_t* n, uint64_t* p) {
    if (!path || !n || !p) return io_fail("ntt_io_read_header: null argument");
    MappedFile f;
    if (!f.open(path)) return io_fail(std::string("cannot open ") + path + ": " + strerror(errno));
    const char* cur = f.data;
    const char* end = f.data + f.size;
    uint64_t nn = 0, pp = 0;
    if (next_u64(cur, end, nn) != 1 || next_u64(cur, end, pp) != 1 || nn < 1 || nn > (uint64_t)INT64_MAX / 16)
        return io_fail(std::string(path) + ": malformed header (expected \"n p\")");
    *n = (int64_t)nn;
    *p = pp;
    return NTT_OK;
}

int ntt_io_read_in(const char* path, uint64_t* a, uint64_t* b, int64_t n, int flags) {
    if (!path || !a || !b || n < 1) return io_fail("ntt_io_read_in: bad argument");
    MappedFile f;
    if (!f.open(path)) return io_fail(std::string("cannot open ") + path + ": " + strerror(errno));
    const std::string bin_path = std::string(path) + ".bin";
    if ((flags & NTT_IO_SIDECAR_READ) && sidecar_load(bin_path, f, nullptr, nullptr, a, b, n)) return NTT_OK;
    const char* cur = f.data;
    const char* end = f.data + f.size;
    uint64_t nn = 0, pp = 0;
    if (next_u64(cur, end, nn) != 1 || next_u64(cur, end, pp) != 1) return io_fail(std::string(path) + ": malformed header");
    if ((int64_t)nn != n) return io_fail(std::string(path) + ": header n differs from the n passed in");
    const int r = parse_tokens(cur, (size_t)(end - cur), a, 2 * (size_t)n, (size_t)n, b);
    if (r != 1) return io_fail(std::string(path) + (r == 0 ? ": fewer than 2n coefficients" : ": malformed coefficient"));
    if (flags & NTT_IO_SIDECAR_WRITE) sidecar_store(bin_path, f, nn, pp, a, b);
    return NTT_OK;
}

int ntt_io_check_out(const char* path, const uint64_t* ab, int64_t count) {
    if (!path || !ab || count < 0) return io_fail("ntt_io_check_out: bad argument");
    MappedFile f;
    if (!f.open(path)) return io_fail(std::string("cannot open ") + path + ": " + strerror(errno));
    std::vector<uint64_t> got((size_t)count);
    if (parse_tokens(f.data, f.size, got.data(), got.size()) != 1) return 0;  // short or malformed: "wrong", as fCheck reports
    return memcmp(got.data(), ab, (size_t)count * sizeof(uint64_t)) == 0 ? 1 : 0;
}

int ntt_io_read_out(const char* path, uint64_t* ab, int64_t count) {
    if (!path || !ab || count < 0) return io_fail("ntt_io_read_out: bad argument");
    MappedFile f;
    if (!f.open(path)) return io_fail(std::string("cannot open ") + path + ": " + strerror(errno));
    if (parse_tokens(f.data, f.size, ab, (size_t)count) != 1)
        return io_fail(std::string(path) + ": fewer than " + std::to_string(count) + " well-formed values");
    return NTT_OK;
}

int ntt_io_write_out(const char* path, const uint64_t* ab, int64_t count) {
    if (!path || !ab || count < 0) return io_fail("ntt_io_write_out: bad argument");
    // every worker formats a contiguous range of values into its own buffer; the buffers are written in order
    const int T = worker_count((size_t)count * 8);
    std::vector<std::vector<char>> bufs(T);
    std::vector<size_t> used(T, 0);
    auto format_piece = [&](int t) {
        const int64_t lo = count / T * t, hi = (t == T - 1) ? count : count / T * (t + 1);
        bufs[t].resize((size_t)(hi - lo) * 21 + 1);  // 20 digits + newline per value at most
        char* w = bufs[t].data();
        for (int64_t i = lo; i < hi; ++i) {
            uint64_t v = ab[i];
            char tmp[20];
            int k = 0;
            while (v >= 100) { const unsigned r = (unsigned)(v % 100); v /= 100; tmp[k++] = (char)('0' + r % 10); tmp[k++] = (char)('0' + r / 10); }
            if (v >= 10) { tmp[k++] = (char)('0' + v % 10); v /= 10; }
            tmp[k++] = (char)('0' + v);
            while (k) *w++ = tmp[--k];
            *w++ = '\n';
        }
        used[t] = (size_t)(w - bufs[t].data());
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(format_piece, t);
        format_piece(0);
        for (auto& x : th) x.join();
    }
    FILE* f = fopen(path, "wb");
    if (!f) return io_fail(std::string("cannot create ") + path + ": " + strerror(errno));
    bool ok = true;
    for (int t = 0; t < T; ++t) ok = ok && fwrite(bufs[t].data(), 1, used[t], f) == used[t];
    if (fclose(f) != 0 || !ok) return io_fail(std::string("write failed: ") + path);
    return NTT_OK;
}

}  // extern "C"
