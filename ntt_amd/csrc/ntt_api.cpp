// C ABI of libnttamd.so: plan cache, workspace management, kernel launches.
// Compiled as HIP for gfx950 (the product) and, with NTT_CPU_TWIN, as plain C++
// over a test-only HIP stand-in so the host logic and kernel index arithmetic
// can be exercised without a GPU.  The product build contains no CPU compute path.
#ifdef NTT_CPU_TWIN
#include "hip_cpu_shim.h"
#else
#include <hip/hip_runtime.h>
#define NTT_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#endif
// launch bracketed by profiling events when ntt_profile_enable(1) is in effect
#define NTT_LAUNCH_P(label, kernel, grid, block, stream, ...)      \
    do {                                                           \
        const int prof_slot_ = prof_begin(label, stream);          \
        NTT_LAUNCH(kernel, grid, block, stream, __VA_ARGS__);      \
        prof_end(prof_slot_, stream);                              \
    } while (0)

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ntt_amd.h"
#include "ntt_kernels.h"
#include "ntt_kernels64.h"
#include "ntt_transform32.h"
#include "ntt_plan.h"

using namespace nttamd;

namespace {

thread_local std::string g_err;

// Every entry point of the C ABI takes this lock: the library's state (device table, plan caches, workspaces) is shared,
// so concurrent callers are serialised rather than left to race.  Recursive, because entry points call one another.
std::recursive_mutex g_api_mutex;
#define NTT_API_LOCK std::lock_guard<std::recursive_mutex> api_lock_(g_api_mutex)

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace
namespace nttamd {
void set_error_text(const std::string& s) { g_err = s; }  // for the library's other translation units
}
namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(NTT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return NTT_OK;
        if (ptr) { hipError_t e = hipFree(ptr); (void)e; ptr = nullptr; bytes = 0; }
        HIP_TRY(hipMalloc(&ptr, need));
        bytes = need;
        return NTT_OK;
    }
    void release() {
        if (ptr) { hipError_t e = hipFree(ptr); (void)e; }
        ptr = nullptr; bytes = 0;
    }
};

// page-locked host memory (DMA source/target of the host-pointer entry points)
struct PinBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return NTT_OK;
        release();
        HIP_TRY(hipHostMalloc(&ptr, need, 0));
        bytes = need;
        return NTT_OK;
    }
    void release() {
        if (ptr) { hipError_t e = hipHostFree(ptr); (void)e; }
        ptr = nullptr; bytes = 0;
    }
};

// Scratch of the entry point that allocates nothing (ntt_poly_multiply_batched_device_noalloc): the two intermediate
// arrays of the multi-pass path live in this file-scope device array instead of a hipMalloc'ed workspace.  Every GPU the
// library runs on holds its own copy at its own address for the life of the process; host code never uses the array's
// name as an address -- the per-device address is looked up once (Device::scratch, set in get_device).
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)32 << 20;
static __device__ __align__(256) unsigned char g_noalloc_scratch[NOALLOC_SCRATCH_BYTES];

struct Plan {
    PlanHost h;  // host image without the big vectors after upload (kept: small)
    DevBuf tables;
    const uint32_t *twF = nullptr, *twI = nullptr, *ow = nullptr, *T1 = nullptr, *T3 = nullptr;
    CoreTables core{};
    CoreTables core_t{};   // the same with the inverse table scaled for the stand-alone transform
    RoundTables32 rt{};
};

// 64-bit field: constants of both arithmetic flavours, the two-level twiddle tables for the root of order
// 2^log_root and the round tables of the LDS kernels.  fp: the double-precision flavour serves this modulus
// (p <= 1.5e15); the tables then hold balanced doubles instead of Montgomery forms.
struct Plan64 {
    bool fp = false;
    Field64Consts fc{};
    FieldDConsts fd{};
    Twiddle64 tw{};
    RoundTables64 rt{};
    int log_root = 0;
    DevBuf tables;
};

struct Device {
    int id = 0;
    hipStream_t stream = nullptr;
    void* scratch = nullptr;   // this device's address of g_noalloc_scratch
    std::map<std::pair<uint32_t, int>, std::unique_ptr<Plan>> plans;
    std::map<std::pair<uint64_t, int>, std::unique_ptr<struct Plan64>> plans64;
    // library-owned workspaces, one per stream the NULL-workspace calls were enqueued on: calls on different streams may
    // overlap on the device, so they must not share scratch memory.  A buffer only ever grows; growing frees the old
    // one with hipFree, which waits for the device first, so nothing in flight loses its memory.  At most
    // MAX_STREAM_WORKSPACES streams keep one: a further stream takes over the least recently used entry's place (after
    // a device-wide wait -- the stream it belonged to may no longer exist, so it cannot be waited for by handle), and
    // ntt_release_workspace() frees them on request.
    static constexpr size_t MAX_STREAM_WORKSPACES = 8;
    struct StreamWork { DevBuf buf; uint64_t last_use = 0; };
    std::map<hipStream_t, StreamWork> work;
    uint64_t work_clock = 0;
    int workspace_for(hipStream_t st, size_t need, void** ws) {
        auto it = work.find(st);
        if (it == work.end()) {
            if (work.size() >= MAX_STREAM_WORKSPACES) {
                auto lru = work.begin();
                for (auto k = work.begin(); k != work.end(); ++k)
                    if (k->second.last_use < lru->second.last_use) lru = k;
                HIP_TRY(hipDeviceSynchronize());
                lru->second.buf.release();
                work.erase(lru);
            }
            it = work.emplace(st, StreamWork()).first;
        }
        it->second.last_use = ++work_clock;
        int rc = it->second.buf.ensure(need);
        if (rc == NTT_OK) *ws = it->second.buf.ptr;
        return rc;
    }
    DevBuf io_a, io_b, io_out;  // device side of the host-pointer entry points
    PinBuf pin;                 // page-locked staging for the same
    // two independent staging sets for host-pointer calls too large for `pin`: chunks alternate between them so that
    // one chunk's copy back overlaps the next chunk's copy in (each set has its own stream, buffers and workspace)
    struct StageSet {
        PinBuf pin;
        DevBuf a, b, out, ws;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
    } stage[2];
};

std::map<int, std::unique_ptr<Device>> g_devs;  // created on first use of each device
int g_visible = 0;                                // devices this process may use (ntt_init)
bool g_inited = false;

struct Device;
int current_device(Device** out);

int do_init(int n_gpus) {
    if (g_inited) {
        if (n_gpus > 0 && n_gpus < g_visible) g_visible = n_gpus;
        return NTT_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(NTT_ERR_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    g_visible = (n_gpus > 0 && n_gpus < count) ? n_gpus : count;
    Device* dev = nullptr;   // per-device set-up of the caller's current device: stream, address of the static scratch
    const int rc = current_device(&dev);
    if (rc == NTT_OK) g_inited = true;   // a failed set-up leaves the library uninitialised, so the next call reports the cause again
    return rc;
}

int get_device(int id, Device** out) {
    auto it = g_devs.find(id);
    if (it != g_devs.end()) { *out = it->second.get(); return NTT_OK; }
    std::unique_ptr<Device> dev(new Device());
    dev->id = id;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(id));
    HIP_TRY(hipStreamCreate(&dev->stream));
    {   // with this device current: where its copy of the static scratch array lives
        void* sp = nullptr;
        hipError_t e = hipGetSymbolAddress(&sp, HIP_SYMBOL(g_noalloc_scratch));
        if (e != hipSuccess || !sp) {
            hipError_t e2 = hipStreamDestroy(dev->stream); (void)e2;
            e2 = hipSetDevice(prev); (void)e2;
            return fail(NTT_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        }
        dev->scratch = sp;
    }
    HIP_TRY(hipSetDevice(prev));
    *out = dev.get();
    g_devs[id] = std::move(dev);
    return NTT_OK;
}

int current_device(Device** out) {
    int d = 0;
    HIP_TRY(hipGetDevice(&d));
    return get_device(d, out);
}

// ---- optional per-kernel timing with HIP events on the launch stream ----
// Events come from pools that are created once and reused by every profiling session, so that a launch inside a
// session costs two hipEventRecord calls and nothing else.  An event belongs to the device that was current when it
// was created and can only be recorded on that device's streams, so there is one pool per device (the per-device
// host threads of a multi-device call each draw from their own).  prof_begin returns the record's slot (-1: not
// recorded) and prof_end closes exactly that slot.
struct ProfRec { const char* label; hipEvent_t e0, e1; bool closed; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;       // records of the current session
struct ProfPool { std::vector<hipEvent_t> events; size_t used = 0; };   // every event ever created on one device
std::map<int, ProfPool> g_prof_pools;
constexpr size_t PROF_MAX_RECORDS = 1 << 16;  // a session stops recording beyond this many launches

bool prof_event(ProfPool& pool, hipEvent_t* e) {
    if (pool.used < pool.events.size()) { *e = pool.events[pool.used++]; return true; }
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return false;
    pool.events.push_back(ev);
    ++pool.used;
    *e = ev;
    return true;
}

std::mutex g_prof_mutex;   // launches may come from the per-device threads of a multi-device call

int prof_begin(const char* label, hipStream_t st) {
    if (!g_prof_on) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    if (g_prof.size() >= PROF_MAX_RECORDS) return -1;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) return -1;
    ProfPool& pool = g_prof_pools[dev_id];
    const size_t used_before = pool.used;
    ProfRec r; r.label = label; r.e0 = nullptr; r.e1 = nullptr; r.closed = false;
    if (!prof_event(pool, &r.e0) || !prof_event(pool, &r.e1) || hipEventRecord(r.e0, st) != hipSuccess) {
        pool.used = used_before;  // the events go back to the pool
        return -1;
    }
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void prof_end(int slot, hipStream_t st) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    if ((size_t)slot >= g_prof.size()) return;
    g_prof[slot].closed = hipEventRecord(g_prof[slot].e1, st) == hipSuccess;
}

// upload helper: appends a vector to a staging image, returns its offset (16-byte aligned)
size_t stage(std::vector<uint32_t>& img, const std::vector<uint32_t>& v) {
    while (img.size() % 4) img.push_back(0);
    size_t off = img.size();
    img.insert(img.end(), v.begin(), v.end());
    return off;
}

int get_plan(Device* dev, uint32_t p, int logL, Plan** out) {
    auto key = std::make_pair(p, logL);
    auto it = dev->plans.find(key);
    if (it != dev->plans.end()) { *out = it->second.get(); return NTT_OK; }
    std::unique_ptr<Plan> pl(new Plan());
    std::string err;
    if (!build_plan_host(p, logL, pl->h, &err)) return fail(NTT_ERR_MODULUS, err);
    std::vector<uint32_t> img;
    PlanHost& h = pl->h;
    size_t o_twF = stage(img, h.twF), o_twI = stage(img, h.twI), o_ow = stage(img, h.ow);
    size_t o_1F = stage(img, h.tw1F), o_2F = stage(img, h.tw2F), o_1I = stage(img, h.tw1I), o_2I = stage(img, h.tw2I);
    size_t o_T1 = stage(img, h.T1), o_T3 = stage(img, h.T3), o_1It = stage(img, h.tw1I_t);
    size_t o_rtF[13] = {0}, o_rtI[13] = {0};
    for (int b = 0; b < 13; ++b) { o_rtF[b] = stage(img, h.rtF[b]); o_rtI[b] = stage(img, h.rtI[b]); }
    if (img.empty()) img.push_back(0);
    int rc = pl->tables.ensure(img.size() * sizeof(uint32_t));
    if (rc != NTT_OK) return rc;
    HIP_TRY(hipMemcpy(pl->tables.ptr, img.data(), img.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const uint32_t* base = (const uint32_t*)pl->tables.ptr;
    pl->twF = base + o_twF; pl->twI = base + o_twI; pl->ow = base + o_ow;
    pl->core.tw1F = base + o_1F; pl->core.tw2F = base + o_2F; pl->core.tw1I = base + o_1I; pl->core.tw2I = base + o_2I;
    pl->T1 = base + o_T1; pl->T3 = base + o_T3;
    pl->core_t = pl->core; pl->core_t.tw1I = base + o_1It;
    for (int b = 0; b < 13; ++b) {
        pl->rt.f[b] = h.rtF[b].empty() ? nullptr : base + o_rtF[b];
        pl->rt.i[b] = h.rtI[b].empty() ? nullptr : base + o_rtI[b];
    }
    // the big host vectors are no longer needed
    std::vector<uint32_t>().swap(h.T1); std::vector<uint32_t>().swap(h.T3);
    *out = pl.get();
    dev->plans[key] = std::move(pl);
    return NTT_OK;
}

size_t pipeline_workspace_bytes(int logL, size_t batch) {
    if (logL <= 13) return 0;
    return 2 * batch * ((size_t)1 << logL) * sizeof(uint32_t);
}

// forward column passes of both operands and the fused core: leaves the block-wise product spectrum's inverse core output
// in tmpA (what the inverse column pass reads)
template <int Q, bool LZ, bool FULL, int MINW, typename TIn>
void launch_fwd_core_v(const Plan& pl, const TIn* a, const TIn* b, int n, size_t batch, size_t sa, size_t sb, uint32_t* tmpA,
                       uint32_t* tmpB, hipStream_t st) {
    const int C = CORE_N;
    const unsigned chunks = C / 256;
    NTT_LAUNCH_P("k_fwd_strided", (k_fwd_strided<Q, LZ, true, FULL, MINW, TIn>), dim3((unsigned)(chunks * batch), 1, 2), dim3(256),
                 st, a, b, sa, sb, n, tmpA, tmpB, C, pl.T1, pl.h.fc);
    NTT_LAUNCH_P("k_core_mul", (k_core_mul<LZ>), dim3((unsigned)(batch << Q)), dim3(CORE_T), st, (const uint32_t*)tmpA,
                 (const uint32_t*)tmpB, tmpA, pl.h.fc, pl.core);
}

template <int Q, bool LZ, bool FULL, int MINW, typename TIn, typename TOut>
void launch_multipass_v(const Plan& pl, const TIn* a, const TIn* b, TOut* out, int n, size_t batch, size_t sa, size_t sb,
                        size_t so, uint32_t* tmpA, uint32_t* tmpB, hipStream_t st) {
    const int C = CORE_N;
    const unsigned chunks = C / 256;
    launch_fwd_core_v<Q, LZ, FULL, MINW, TIn>(pl, a, b, n, batch, sa, sb, tmpA, tmpB, st);
    NTT_LAUNCH_P("k_inv_strided", (k_inv_strided<Q, LZ, true, FULL, MINW, TOut>), dim3((unsigned)(chunks * batch)), dim3(256), st,
                 (const uint32_t*)tmpA, out, so, 2 * n - 1, C, pl.T3, pl.h.fc);
}

constexpr int multipass_minw(int Q) { return Q == 6 ? 2 : 1; }

template <int Q, bool LZ, typename TIn, typename TOut>
void launch_multipass(const Plan& pl, const TIn* a, const TIn* b, TOut* out, int n, size_t batch, size_t sa, size_t sb,
                      size_t so, uint32_t* tmpA, uint32_t* tmpB, hipStream_t st) {
    const bool full = (2 * (size_t)n == ((size_t)CORE_N << Q));
    constexpr int MINW = multipass_minw(Q);
    if (full) launch_multipass_v<Q, LZ, true, MINW>(pl, a, b, out, n, batch, sa, sb, so, tmpA, tmpB, st);
    else launch_multipass_v<Q, LZ, false, MINW>(pl, a, b, out, n, batch, sa, sb, so, tmpA, tmpB, st);
}

// The four-prime route at transform lengths 2^14 .. 2^18: one launch reduces the operands modulo the four primes and runs
// their forward column passes (the operands are read once), the fused core runs once per prime, and one launch runs the
// inverse column pass of all four primes and merges in LDS (the residues never reach HBM).
#ifndef NTT_CRT_INV_CW
#define NTT_CRT_INV_CW 64
#endif
#ifndef NTT_CRT_FWD_CW
#define NTT_CRT_FWD_CW 64
#endif
constexpr int CRT_FWD_CW = NTT_CRT_FWD_CW;
constexpr int CRT_INV_CW = NTT_CRT_INV_CW;   // columns per workgroup of the four-prime column passes (4 * CW threads)

template <int Q>
void launch_crt_multipass(Plan* const (&pl)[4], const uint64_t* a, const uint64_t* b, uint64_t* out, int n, size_t batch, size_t sa,
                          size_t sb, size_t so, uint32_t* planes, size_t plane_stride, const CrtConsts& c, hipStream_t st) {
    const bool full = (2 * (size_t)n == ((size_t)CORE_N << Q));
    constexpr int MINW = multipass_minw(Q);
    constexpr int CW = CRT_INV_CW;
    // planes: A of the four primes, then B of the four primes
    CrtFwdArgs fa;
    CrtInvArgs ia;
    for (int i = 0; i < 4; ++i) {
        fa.T1[i] = pl[i]->T1;
        fa.outA[i] = planes + i * plane_stride;
        fa.outB[i] = planes + (4 + i) * plane_stride;
        fa.fc[i] = pl[i]->h.fc;
        ia.src[i] = fa.outA[i];
        ia.T3[i] = pl[i]->T3;
        ia.fc[i] = pl[i]->h.fc;
    }
    const unsigned grid = (unsigned)((CORE_N / CW) * batch);
    constexpr int FW = CRT_FWD_CW;
    const unsigned fgrid = (unsigned)((CORE_N / FW) * batch);
    if (full)
        NTT_LAUNCH_P("k_fwd_strided_crt", (k_fwd_strided_crt<Q, FW, true, MINW>), dim3(fgrid, 1, 2), dim3(4 * FW), st, a, b, sa, sb, n,
                     CORE_N, fa);
    else
        NTT_LAUNCH_P("k_fwd_strided_crt", (k_fwd_strided_crt<Q, FW, false, MINW>), dim3(fgrid, 1, 2), dim3(4 * FW), st, a, b, sa, sb, n,
                     CORE_N, fa);
    CrtCoreArgs ca;
    for (int i = 0; i < 4; ++i) {
        ca.A[i] = fa.outA[i]; ca.B[i] = fa.outB[i]; ca.C[i] = fa.outA[i];
        ca.fc[i] = pl[i]->h.fc; ca.tb[i] = pl[i]->core;
    }
    NTT_LAUNCH_P("k_core_mul_crt", k_core_mul_crt, dim3((unsigned)(batch << Q), 4), dim3(CORE_T), st, ca);
    if (full)
        NTT_LAUNCH_P("k_inv_strided_crt", (k_inv_strided_crt<Q, CW, true>), dim3(grid), dim3(4 * CW), st, ia, out, so, 2 * n - 1, CORE_N, c);
    else
        NTT_LAUNCH_P("k_inv_strided_crt", (k_inv_strided_crt<Q, CW, false>), dim3(grid), dim3(4 * CW), st, ia, out, so, 2 * n - 1, CORE_N, c);
}

template <bool LZ, typename TIn, typename TOut>
int launch_product(const Plan& pl, const TIn* a, const TIn* b, TOut* out, int n, size_t batch, size_t sa, size_t sb,
                   size_t so, void* ws, hipStream_t st) {
    const int n_out = 2 * n - 1;
    switch (pl.h.kind) {
        case PLAN_SMALL:
            if (pl.h.small16) {
                const unsigned per_wg = 1u << (12 - pl.h.logL);   // products per workgroup
                NTT_LAUNCH_P("k_small16", (k_small16<LZ, TIn, TOut>), dim3((unsigned)((batch + per_wg - 1) / per_wg)), dim3(CORE_T), st,
                             a, b, out, sa, sb, so, n, n_out, pl.h.logL, batch, pl.h.scale_m, pl.h.fc, pl.rt);
                break;
            }
            NTT_LAUNCH_P("k_small", (k_small<LZ, TIn, TOut>), dim3((unsigned)batch), dim3(256), st, a, b, out, sa, sb, so, n, n_out,
                       pl.h.logL, pl.twF, pl.twI, pl.h.scale_m, pl.h.fc);
            break;
        case PLAN_FUSED8192:
            NTT_LAUNCH_P("k_fused8192", (k_fused8192<LZ, TIn, TOut>), dim3((unsigned)batch), dim3(CORE_T), st, a, b, out, sa, sb, so, n,
                       n_out, pl.ow, pl.h.fc, pl.core);
            break;
        case PLAN_MULTIPASS: {
            uint32_t* tmpA = (uint32_t*)ws;
            uint32_t* tmpB = tmpA + batch * ((size_t)1 << pl.h.logL);
            switch (pl.h.Q) {
                case 2: launch_multipass<2, LZ>(pl, a, b, out, n, batch, sa, sb, so, tmpA, tmpB, st); break;
                case 3: launch_multipass<3, LZ>(pl, a, b, out, n, batch, sa, sb, so, tmpA, tmpB, st); break;
                case 4: launch_multipass<4, LZ>(pl, a, b, out, n, batch, sa, sb, so, tmpA, tmpB, st); break;
                case 5: launch_multipass<5, LZ>(pl, a, b, out, n, batch, sa, sb, so, tmpA, tmpB, st); break;
                case 6: launch_multipass<6, LZ>(pl, a, b, out, n, batch, sa, sb, so, tmpA, tmpB, st); break;
                default: return fail(NTT_ERR_ARG, "unsupported strided radix");
            }
            break;
        }
    }
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

// One modulus < 2^31, device-resident operands of type TIn, output TOut.
template <typename TIn, typename TOut>
int product_device(Device* dev, const TIn* a, const TIn* b, TOut* out, int n, uint32_t p, size_t batch, size_t sa, size_t sb,
                   size_t so, void* ws, size_t ws_bytes, hipStream_t st) {
    const int logL = transform_log2(n);
    Plan* pl = nullptr;
    int rc = get_plan(dev, p, logL, &pl);
    if (rc != NTT_OK) return rc;
    const size_t need = pipeline_workspace_bytes(logL, batch);
    if (need) {
        if (ws) {
            if (ws_bytes < need) return fail(NTT_ERR_WORKSPACE, "workspace too small");
        } else {
            if ((rc = dev->workspace_for(st, need, &ws)) != NTT_OK) return rc;
        }
    }
    if (pl->h.lazy) return launch_product<true>(*pl, a, b, out, n, batch, sa, sb, so, ws, st);
    return launch_product<false>(*pl, a, b, out, n, batch, sa, sb, so, ws, st);
}

// grid.y of the per-coefficient kernels: the rows of a batch, capped at the launch limit of that dimension (the kernels
// stride over the rows beyond it)
unsigned grid_rows(size_t batch) { return (unsigned)(batch < 65535 ? batch : 65535); }

int make_crt_consts(uint64_t p, CrtConsts* c) {
    if (p < 2) return fail(NTT_ERR_ARG, "modulus must be at least 2");
    // the primes of the basis in ascending order (Garner's digits then never need a negative intermediate)
    int order[4] = {0, 1, 2, 3};
    std::sort(order, order + 4, [](int x, int y) { return kCrtPrimes[x] < kCrtPrimes[y]; });
    uint64_t m[4];
    for (int i = 0; i < 4; ++i) {
        m[i] = kCrtPrimes[order[i]];
        c->m[i] = (uint32_t)m[i];
        c->nm[i] = h_neg_inv32((uint32_t)m[i]);
        c->src[i] = order[i];
    }
    auto inv = [](uint64_t x, uint64_t q) { return h_powmod(x % q, q - 2, q); };
    c->k1 = h_mont((uint32_t)inv(m[0], m[1]), (uint32_t)m[1]);
    c->c2 = h_mont((uint32_t)(m[0] % m[2]), (uint32_t)m[2]);
    c->k2 = h_mont((uint32_t)inv(h_mulmod(m[0], m[1], m[2]), m[2]), (uint32_t)m[2]);
    c->c31 = h_mont((uint32_t)(m[0] % m[3]), (uint32_t)m[3]);
    c->c32 = h_mont((uint32_t)h_mulmod(m[0], m[1], m[3]), (uint32_t)m[3]);
    c->k3 = h_mont((uint32_t)inv(h_mulmod(h_mulmod(m[0], m[1], m[3]), m[2], m[3]), m[3]), (uint32_t)m[3]);
    // tail: x mod p from the digits
    c->fp = (double)p <= FIELDD_MAX_P ? 1 : 0;
    c->pd = (double)p;
    c->ud = 1.0 / (double)p;
    auto balanced = [p](uint64_t v) { v %= p; return v > p / 2 ? -(double)(p - v) : (double)v; };
    const uint64_t e1 = m[0] % p, e2 = h_mulmod(e1, m[1] % p, p), e3 = h_mulmod(e2, m[2] % p, p);
    c->d1 = balanced(e1); c->d2 = balanced(e2); c->d3 = balanced(e3);
    c->p = p;
    c->shift = __builtin_clzll(p);
    c->d = p << c->shift;
    c->v = (uint64_t)((~(u128)0) / c->d - ((u128)1 << 64));
    return NTT_OK;
}

// The admission tests cost a primality test and a few modular powers (tens of microseconds): their verdicts are kept
// per (modulus, length), so that the per-call dispatch of a small product does not repeat them.
bool direct_path_ok(uint64_t p, int logL) {
    if (p >= (1ull << 31)) return false;
    static std::mutex m;
    static std::map<std::pair<uint64_t, int>, bool> memo;
    std::lock_guard<std::mutex> lk(m);
    const auto key = std::make_pair(p, logL);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    const bool ok = modulus_is_ntt_friendly(p, logL);
    if (memo.size() < 4096) memo[key] = ok;
    return ok;
}

// NTT_TRANSFORM32=off sends every stand-alone transform through the 64-bit field (A/B measurements and tests of that path)
bool transform32_allowed() {
    static const bool v = [] { const char* e = getenv("NTT_TRANSFORM32"); return !(e && strcmp(e, "off") == 0); }();
    return v;
}

// NTT_CRT_MERGE=separate keeps the recombination a pass of its own (A/B measurements and tests of the stand-alone merge kernel)
bool crt_fused_merge_allowed() {
    static const bool v = [] { const char* e = getenv("NTT_CRT_MERGE"); return !(e && strcmp(e, "separate") == 0); }();
    return v;
}

size_t crt_workspace_bytes(int logL, size_t batch) {
    const size_t L = (size_t)1 << logL;
    // multi-pass lengths: the column-pass results of both operands for the four primes (eight planes); shorter ones: the
    // four residue planes
    if (logL >= 14) return 8 * batch * L * sizeof(uint32_t);
    return pipeline_workspace_bytes(logL, batch) + 4 * batch * L * sizeof(uint32_t);
}

// Any modulus through the four CRT primes: device-resident 64-bit operands.
int crt_product_device(Device* dev, const uint64_t* a, const uint64_t* b, uint64_t* out, int n, uint64_t p, size_t batch,
                       size_t sa, size_t sb, size_t so, void* ws, size_t ws_bytes, hipStream_t st) {
    const int logL = transform_log2(n);
    if (!crt_bound_ok(n, p))
        return fail(NTT_ERR_MODULUS, "n*(p-1)^2 exceeds the product of the four CRT primes: result would not be exact");
    if (logL > 21) return fail(NTT_ERR_MODULUS, "CRT primes support transform lengths up to 2^21");
    const size_t need = crt_workspace_bytes(logL, batch);
    if (ws) {
        if (ws_bytes < need) return fail(NTT_ERR_WORKSPACE, "workspace too small");
    } else {
        int rc = dev->workspace_for(st, need, &ws);
        if (rc != NTT_OK) return rc;
    }
    const size_t L = (size_t)1 << logL;
    const size_t pipe = pipeline_workspace_bytes(logL, batch);
    uint32_t* res = (uint32_t*)((char*)ws + pipe);
    const size_t res_stride = batch * L;
    CrtConsts c;
    int rc = make_crt_consts(p, &c);
    if (rc != NTT_OK) return rc;
    // multi-pass lengths: the residue step rides on one forward column pass for all four primes and the merge on the
    // inverse one.  The basis is fixed: its second prime is the one kept fully reduced, the kernels rely on that
    Plan* pl[4];
    bool multipass = true;
    for (int i = 0; i < 4; ++i) {
        if ((rc = get_plan(dev, kCrtPrimes[i], logL, &pl[i])) != NTT_OK) return rc;
        multipass = multipass && pl[i]->h.kind == PLAN_MULTIPASS && pl[i]->h.lazy == (i != 1);
    }
    if (multipass && crt_fused_merge_allowed()) {
        uint32_t* planes = (uint32_t*)ws;
        switch (pl[0]->h.Q) {
            case 2: launch_crt_multipass<2>(pl, a, b, out, n, batch, sa, sb, so, planes, res_stride, c, st); break;
            case 3: launch_crt_multipass<3>(pl, a, b, out, n, batch, sa, sb, so, planes, res_stride, c, st); break;
            case 4: launch_crt_multipass<4>(pl, a, b, out, n, batch, sa, sb, so, planes, res_stride, c, st); break;
            case 5: launch_crt_multipass<5>(pl, a, b, out, n, batch, sa, sb, so, planes, res_stride, c, st); break;
            case 6: launch_crt_multipass<6>(pl, a, b, out, n, batch, sa, sb, so, planes, res_stride, c, st); break;
            default: return fail(NTT_ERR_ARG, "unsupported strided radix");
        }
        HIP_TRY(hipGetLastError());
        return NTT_OK;
    }
    for (int i = 0; i < 4; ++i) {
        rc = product_device<uint64_t, uint32_t>(dev, a, b, res + i * res_stride, n, kCrtPrimes[i], batch, sa, sb, L, ws, pipe, st);
        if (rc != NTT_OK) return rc;
    }
    const size_t count = (size_t)(2 * n - 1);
    NTT_LAUNCH_P("k_crt_merge", k_crt_merge, dim3((unsigned)((count + 255) / 256), grid_rows(batch)), dim3(256), st, (const uint32_t*)res,
               res_stride, out, count, batch, so, L, c);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}


// ---------------------------------------------------------------------------------------------
// 64-bit field path
// ---------------------------------------------------------------------------------------------
bool modulus64_test(uint64_t p, int log_root, std::string* why);
bool modulus64_ok(uint64_t p, int log_root, std::string* why) {
    static std::mutex m;
    static std::map<std::pair<uint64_t, int>, std::pair<bool, std::string>> memo;
    std::lock_guard<std::mutex> lk(m);
    const auto key = std::make_pair(p, log_root);
    auto it = memo.find(key);
    if (it == memo.end()) {
        std::string w;
        const bool ok = modulus64_test(p, log_root, &w);
        if (memo.size() >= 4096) { *why = w; return ok; }
        it = memo.emplace(key, std::make_pair(ok, w)).first;
    }
    *why = it->second.second;
    return it->second.first;
}
bool modulus64_test(uint64_t p, int log_root, std::string* why) {
    if (p >= (1ull << 60) || p < 3 || (p & 1) == 0) { *why = "64-bit path needs an odd modulus below 2^60"; return false; }
    if (log_root < 1 || log_root > 40) { *why = "root order out of range"; return false; }
    const uint64_t L = 1ull << log_root;
    if ((p - 1) % L != 0) { *why = "transform length does not divide p-1"; return false; }
    if (!is_prime_u64(p)) { *why = "modulus is not prime"; return false; }
    const uint64_t w = h_powmod(3, (p - 1) / L, p);
    if (h_powmod(w, L / 2, p) != p - 1) { *why = "3 is not a generator of the required 2-power subgroup mod p"; return false; }
    // the inverses the route uses (all Fermat inverses) must really invert
    if (h_mulmod(h_powmod(L % p, p - 2, p), L % p, p) != 1 || h_mulmod(h_powmod(w, p - 2, p), w, p) != 1) {
        *why = "transform length or root not invertible mod p";
        return false;
    }
    return true;
}

uint64_t h_mont64(uint64_t x, uint64_t p) { return (uint64_t)(((u128)(x % p) << 64) % p); }

// NTT_FIELD64=int keeps every modulus on the integer flavour (A/B measurements and tests of that path)
bool fieldd_allowed() {
    static const bool v = [] { const char* e = getenv("NTT_FIELD64"); return !(e && strcmp(e, "int") == 0); }();
    return v;
}

// constant c (plain residue) in the form the plan's multiply wants, as a 64-bit word: Montgomery form, or the bits of the
// balanced double (|c| <= p/2)
uint64_t tw_word(const Plan64& pl, uint64_t c) {
    const uint64_t p = pl.fc.p;
    c %= p;
    if (!pl.fp) return h_mont64(c, p);
    const double d = c > p / 2 ? -(double)(p - c) : (double)c;
    uint64_t bits;
    memcpy(&bits, &d, sizeof bits);
    return bits;
}
template <class F> typename F::W scale_arg(uint64_t word);
template <> uint64_t scale_arg<Field64>(uint64_t word) { return word; }
template <> double scale_arg<FieldD>(uint64_t word) { double d; memcpy(&d, &word, sizeof d); return d; }
template <class F> const typename F::Consts& consts_of(const Plan64& pl);
template <> const Field64Consts& consts_of<Field64>(const Plan64& pl) { return pl.fc; }
template <> const FieldDConsts& consts_of<FieldD>(const Plan64& pl) { return pl.fd; }

// factor the pointwise product leaves on the data (undone by the final scale): 2^-64 in Montgomery arithmetic, none in doubles
uint64_t pointwise_compensation(const Plan64& pl) { return pl.fp ? 1 : pl.fc.r1; }

// Known-answer probe of the double-precision field as this binary executes it, run once when a plan that uses it is
// built: multiplies at the edges of the ranges its exactness rests on (|x| up to 2.5 p against |w| up to 0.6 p, either
// sign), the reduction of arbitrary 64-bit words and the round trip through the memory format, computed on the device
// and compared with 128-bit integer arithmetic here.  The field is exact only when every multiply, add and fused
// multiply-add is executed as written; a library built with floating-point contraction left on (or any other
// value-changing floating-point option) fails this and refuses the plan instead of returning wrong residues.
int fieldd_probe(const Plan64& pl) {
    constexpr int K = 16;
    const uint64_t p = pl.fc.p;
    const int64_t sp = (int64_t)p;
    const int64_t xe = sp * 2 + sp / 2, we = sp / 2 + sp / 10;      // 2.5 p and 0.6 p
    const int64_t xs[8] = {xe, -xe, xe - 1, -(xe - 1), xe, sp + sp / 4, -(sp + sp / 4), 1};
    const int64_t ws[8] = {we, we, -we, -(we - 1), -(we - 2), sp / 2, -(sp / 2), -1};
    const uint64_t raws[4] = {~(uint64_t)0, p, p - 1, ((uint64_t)1 << 63) + 12345};
    const int64_t mems[4] = {xe, -xe, 0, sp / 2};
    double hx[K] = {0}, hw[K] = {0};
    uint64_t hraw[K] = {0}, want[K] = {0}, got[K] = {0};
    auto canon_i = [&](__int128 v) -> uint64_t { __int128 r = v % (__int128)sp; if (r < 0) r += sp; return (uint64_t)r; };
    for (int i = 0; i < 8; ++i) { hx[i] = (double)xs[i]; hw[i] = (double)ws[i]; want[i] = canon_i((__int128)xs[i] * ws[i]); }
    for (int i = 0; i < 4; ++i) { hraw[8 + i] = raws[i]; want[8 + i] = raws[i] % p; }
    for (int i = 0; i < 4; ++i) { hx[12 + i] = (double)mems[i]; want[12 + i] = canon_i(mems[i]); }
    DevBuf buf;
    int rc = buf.ensure(4 * K * sizeof(uint64_t));
    if (rc != NTT_OK) return rc;
    double* dx = (double*)buf.ptr;
    double* dw = dx + K;
    uint64_t* draw = (uint64_t*)(dw + K);
    uint64_t* dout = draw + K;
    hipError_t e = hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(draw, hraw, sizeof hraw, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        NTT_LAUNCH(k64_fieldd_probe, dim3(1), dim3(64), (hipStream_t) nullptr, (const double*)dx, (const double*)dw, (const uint64_t*)draw, dout, pl.fd);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost);
    buf.release();
    if (e != hipSuccess) return fail(NTT_ERR_DEVICE, std::string("double-precision field probe: ") + hipGetErrorString(e));
    for (int i = 0; i < K; ++i)
        if (got[i] != want[i])
            return fail(NTT_ERR_DEVICE, "the double-precision field of this build is not exact (known-answer probe " + std::to_string(i) +
                                        " failed): the library must be compiled with -ffp-contract=off and no value-changing floating-point options");
    return NTT_OK;
}

int get_plan64(Device* dev, uint64_t p, int log_root, Plan64** out) {
    auto key = std::make_pair(p, log_root);
    auto it = dev->plans64.find(key);
    if (it != dev->plans64.end()) { *out = it->second.get(); return NTT_OK; }
    std::string why;
    if (!modulus64_ok(p, log_root, &why)) return fail(NTT_ERR_MODULUS, why);
    std::unique_ptr<Plan64> pl(new Plan64());
    pl->log_root = log_root;
    pl->fp = fieldd_allowed() && (double)p <= FIELDD_MAX_P;
    Field64Consts& fc = pl->fc;
    fc.p = p;
    uint64_t inv = p;  // Newton iteration for p^-1 mod 2^64
    for (int i = 0; i < 6; ++i) inv *= 2 - p * inv;
    fc.np = 0 - inv;
    fc.r1 = h_mont64(1, p);
    fc.r2 = h_mont64(fc.r1, p);
    FieldDConsts& fd = pl->fd;
    fd.p = (double)p;
    fd.u = 1.0 / (double)p;
    {
        const uint64_t c40 = (1ull << 40) % p;
        fd.c40 = c40 > p / 2 ? -(double)(p - c40) : (double)c40;
    }
    auto balanced = [&](uint64_t c) { return c > p / 2 ? -(double)(p - c) : (double)c; };
    const uint64_t wL = h_powmod(3, (p - 1) >> log_root, p);
    for (int e = 0; e < 8; ++e) { fc.root16F[e] = 0; fc.root16I[e] = 0; fd.root16F[e] = 0; fd.root16I[e] = 0; }
    if (log_root >= 4) {
        const uint64_t w16 = h_powmod(wL, 1ull << (log_root - 4), p), w16i = h_powmod(w16, p - 2, p);
        uint64_t a = 1, b = 1;
        for (int e = 0; e < 8; ++e) {
            fc.root16F[e] = h_mont64(a, p); fc.root16I[e] = h_mont64(b, p);
            fd.root16F[e] = balanced(a); fd.root16I[e] = balanced(b);
            a = h_mulmod(a, w16, p); b = h_mulmod(b, w16i, p);
        }
    } else {  // shorter transforms only use the low exponents that exist: w16^(e * 16/2^Q) with 2^Q <= root order
        for (int e = 0; e < 8; ++e) {
            if ((e << log_root) % 16 == 0) {
                const uint64_t a = h_powmod(wL, (uint64_t)(e << log_root) / 16, p);
                const uint64_t b = h_powmod(wL, (1ull << log_root) - (uint64_t)(e << log_root) / 16, p);
                fc.root16F[e] = h_mont64(a, p); fc.root16I[e] = h_mont64(b, p);
                fd.root16F[e] = balanced(a); fd.root16I[e] = balanced(b);
            }
        }
    }
    // image: [two-level table w0 | w1 | round tables forward b = 5..min(12, log_root) | the same, inverse]
    const int t0 = log_root > LOCAL64_LOG ? log_root - LOCAL64_LOG : 0;
    const size_t n0 = (size_t)1 << t0, n1 = (size_t)1 << (log_root - t0);
    std::vector<uint64_t> img(n0 + n1);
    uint64_t a = 1;
    for (size_t i = 0; i < n0; ++i) { img[i] = tw_word(*pl, a); a = h_mulmod(a, wL, p); }
    const uint64_t wstep = h_powmod(wL, n0, p);
    a = 1;
    for (size_t i = 0; i < n1; ++i) { img[n0 + i] = tw_word(*pl, a); a = h_mulmod(a, wstep, p); }
    size_t off_f[14] = {0}, off_i[14] = {0};
    const int bmax = log_root < 12 ? log_root : 12;
    for (int dir = 0; dir < 2; ++dir) {
        for (int b = 5; b <= bmax; ++b) {
            (dir ? off_i : off_f)[b] = img.size();
            uint64_t wB = h_powmod(wL, 1ull << (log_root - b), p);      // root of order 2^b
            if (dir) wB = h_powmod(wB, p - 2, p);
            const size_t sub = (size_t)1 << (b - 4);
            img.resize(img.size() + (sub << 4));
            uint64_t* T = img.data() + (dir ? off_i : off_f)[b];
            for (int pos = 0; pos < 16; ++pos) {
                const uint64_t stepw = h_powmod(wB, (uint64_t)bitrev(pos, 4), p);   // w_B^bitrev4(pos)
                uint64_t v = 1;
                for (size_t j = 0; j < sub; ++j) { T[(j << 4) + pos] = tw_word(*pl, v); v = h_mulmod(v, stepw, p); }
            }
        }
    }
    int rc = pl->tables.ensure(img.size() * sizeof(uint64_t));
    if (rc != NTT_OK) return rc;
    HIP_TRY(hipMemcpy(pl->tables.ptr, img.data(), img.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    const uint64_t* base = (const uint64_t*)pl->tables.ptr;
    pl->tw.w0 = base;
    pl->tw.w1 = base + n0;
    pl->tw.t0 = t0;
    pl->tw.logtab = log_root;
    for (int b = 0; b < 14; ++b) {
        pl->rt.f[b] = (b >= 5 && b <= bmax) ? base + off_f[b] : nullptr;
        pl->rt.i[b] = (b >= 5 && b <= bmax) ? base + off_i[b] : nullptr;
    }
    if (pl->fp && (rc = fieldd_probe(*pl)) != NTT_OK) { pl->tables.release(); return rc; }
    *out = pl.get();
    dev->plans64[key] = std::move(pl);
    return NTT_OK;
}

unsigned grid_for(uint64_t count) { return (unsigned)((count + 255) / 256); }

// ---- passes of one transform along the middle axis of [outer][2^log_len][inner] ----
struct Step64 {
    enum Kind { TILE, REG, LOCAL, SMALL } kind;
    int cur;   // log2 of the block length the pass starts from (forward direction)
    int q;     // levels it performs
};

bool is_pow2(uint64_t x) { return x && (x & (x - 1)) == 0; }

// joined: the axis is one of the two of a four-step matrix and its passes carry the twiddles that join them (Axis64::col_bits
// > 0).  Only the tiled and the register passes apply those, so the contiguous-chunk kernels are not used then -- a slab one
// column wide (inner == 1) would otherwise end in them and lose its twiddles w^(col0 * k).
std::vector<Step64> plan_steps(int log_len, size_t inner, size_t outer, bool joined = false) {
    std::vector<Step64> steps;
    int stop = 0;
    Step64 last{Step64::LOCAL, 0, 0};
    if (inner == 1 && !joined) {
        stop = log_len < LOCAL64_LOG ? log_len : LOCAL64_LOG;
        const bool whole_chunks = (((uint64_t)outer << log_len) & (LOCAL64_N - 1)) == 0;
        last = Step64{whole_chunks ? Step64::LOCAL : Step64::SMALL, stop, stop};
    }
    for (int cur = log_len; cur > stop;) {
        const int remaining = cur - stop;
        int q = 0;
        if (remaining >= TILE64_MIN_LOGR && is_pow2(inner)) {
            // as few passes as possible; of those, the earlier (longer-stride) ones get the fewer rows: a tile whose rows lie
            // megabytes apart is the slowest shape (3.2 ms against 2.3 ms per pass over 2^29 words for 512 rows)
            const int npass = (remaining + TILE64_MAX_LOGR - 1) / TILE64_MAX_LOGR;
            q = remaining / npass;
            if (remaining - q > TILE64_MAX_LOGR * (npass - 1)) q = remaining - TILE64_MAX_LOGR * (npass - 1);
            if (remaining - q > 0 && remaining - q < TILE64_MIN_LOGR) q = remaining - TILE64_MIN_LOGR;   // leave a tileable rest
            const uint64_t columns = ((uint64_t)1 << (cur - q)) * inner;
            if (q < TILE64_MIN_LOGR || q > TILE64_MAX_LOGR || columns % ((uint64_t)TILE64_N >> q) != 0) q = 0;
        }
        if (q) {
            steps.push_back({Step64::TILE, cur, q});
        } else {
            q = remaining < 4 ? remaining : 4;
            steps.push_back({Step64::REG, cur, q});
        }
        cur -= q;
    }
    if (stop) steps.push_back(last);
    return steps;
}

struct TileExtra {           // fusions of a product's first and last pass
    const uint64_t* src = nullptr; uint64_t src_valid = 0;               // forward: raw operands, rows zero beyond src_valid
    uint64_t* out = nullptr; uint64_t n_out = 0; uint64_t scale_word = 0; // inverse: scaled canonical output
    const uint64_t* other = nullptr;                                      // inverse local pass: pointwise product first
    uint64_t src_stride = 0, out_stride = 0;                              // rows of the operands / results in caller memory
    // forward tile pass of a product: the same pass over the second operand's work array in the same launch
    uint64_t* data2 = nullptr; const uint64_t* src2 = nullptr; uint64_t src_stride2 = 0;
};

// Where the passes of one transform live and which twiddles join them: the middle axis of [outer][2^log_len][inner].
// col_bits > 0: the array is a slab of a four-step matrix -- its `inner` columns are columns col0.. of 2^col_bits, and
// the passes are the first log_len levels of the transform of the whole matrix (length 2^(log_len + col_bits)), so that
// the twiddles between the two axes of the matrix come with them.
struct Axis64 {
    int log_len = 0;
    size_t inner = 1, outer = 1;
    int col_bits = 0;
    uint64_t col0 = 0;
};

int log2_exact(uint64_t x) {
    int l = 0;
    while (((uint64_t)1 << l) < x) ++l;
    return l;
}

template <class F>
int launch_pass(const Plan64& pl, uint64_t* d, bool tile, const Pass64& g, bool inverse, const TileExtra& ex, hipStream_t st) {
    const typename F::Consts& fc = consts_of<F>(pl);
    if (!tile) {
        const uint64_t threads = g.n_blocks * g.columns;
        if (threads / 256 >= (1ull << 31)) return fail(NTT_ERR_ARG, "transform too large for one launch");
        const unsigned grid = (unsigned)((threads + 255) / 256);
        const typename F::W scale = scale_arg<F>(ex.scale_word);
        const uint64_t* no_src = nullptr;
        uint64_t* no_out = nullptr;
        const dim3 grid2(grid, (!inverse && ex.data2) ? 2 : 1);      // both operands of a product in one launch
#define NTT_REG_CASE(Q)                                                                                                                  \
    if (inverse && ex.out) NTT_LAUNCH_P("k64_strided", (k64_strided<F, Q, true, false, true>), grid2, dim3(256), st, d, no_src, ex.out, ex.n_out, scale, g, fc, pl.tw, no_out, no_src, (uint64_t)0); \
    else if (inverse) NTT_LAUNCH_P("k64_strided", (k64_strided<F, Q, true>), grid2, dim3(256), st, d, no_src, no_out, (uint64_t)0, scale, g, fc, pl.tw, no_out, no_src, (uint64_t)0);               \
    else if (ex.src) NTT_LAUNCH_P("k64_strided", (k64_strided<F, Q, false, true, false>), grid2, dim3(256), st, d, ex.src, no_out, (uint64_t)0, scale, g, fc, pl.tw, ex.data2, ex.src2, (uint64_t)ex.src_stride2);   \
    else NTT_LAUNCH_P("k64_strided", (k64_strided<F, Q, false>), grid2, dim3(256), st, d, no_src, no_out, (uint64_t)0, scale, g, fc, pl.tw, ex.data2, no_src, (uint64_t)0)
        switch (g.q) {
            case 1: NTT_REG_CASE(1); break;
            case 2: NTT_REG_CASE(2); break;
            case 3: NTT_REG_CASE(3); break;
            default: NTT_REG_CASE(4); break;
        }
#undef NTT_REG_CASE
    } else {
        const uint64_t tiles = g.n_blocks * g.tiles_per_block;
        if (tiles >= (1ull << 31)) return fail(NTT_ERR_ARG, "transform too large for one launch");
        const unsigned grid = (unsigned)tiles;
        const typename F::W scale = scale_arg<F>(ex.scale_word);
        uint64_t* const no_d = nullptr;
        const uint64_t* const no_s = nullptr;
        if (!inverse) {
            const dim3 grid2(grid, ex.data2 ? 2 : 1);      // both operands of a product in one launch
            if (ex.src) NTT_LAUNCH_P("k64_tile", (k64_tile<F, false, true, false>), grid2, dim3(TILE64_T), st, d, ex.src, no_d, (uint64_t)0, scale, g, fc, pl.tw, pl.rt, ex.data2, ex.src2, (uint64_t)ex.src_stride2);
            else NTT_LAUNCH_P("k64_tile", (k64_tile<F, false, false, false>), grid2, dim3(TILE64_T), st, d, no_s, no_d, (uint64_t)0, scale, g, fc, pl.tw, pl.rt, ex.data2, no_s, (uint64_t)0);
        } else {
            if (ex.out) NTT_LAUNCH_P("k64_tile", (k64_tile<F, true, false, true>), dim3(grid), dim3(TILE64_T), st, d, no_s, ex.out, ex.n_out, scale, g, fc, pl.tw, pl.rt, no_d, no_s, (uint64_t)0);
            else NTT_LAUNCH_P("k64_tile", (k64_tile<F, true, false, false>), dim3(grid), dim3(TILE64_T), st, d, no_s, no_d, (uint64_t)0, scale, g, fc, pl.tw, pl.rt, no_d, no_s, (uint64_t)0);
        }
    }
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

template <class F>
int run_step(const Plan64& pl, uint64_t* d, const Step64& s, const Axis64& ax, bool inverse, const TileExtra& ex, hipStream_t st) {
    const typename F::Consts& fc = consts_of<F>(pl);
    const int log_len = ax.log_len;
    const size_t inner = ax.inner, outer = ax.outer;
    if (s.kind == Step64::TILE || s.kind == Step64::REG) {
        Pass64 g{};
        g.columns = ((uint64_t)1 << (s.cur - s.q)) * inner;
        g.step = g.columns;
        g.block_stride = g.columns << s.q;
        g.n_blocks = (uint64_t)outer << (log_len - s.cur);
        g.tw_mask = ((uint64_t)1 << pl.log_root) - 1;
        g.q = s.q;
        g.src_valid = ex.src_valid;
        g.src_stride = ex.src_stride;
        g.out_stride = ex.out_stride;
        g.hi_shift = 63;
        g.lo_mask = ~(uint64_t)0;
        if (ax.col_bits == 0) {           // plain transform: the twiddle index is the column's position in the block, x / inner
            if (is_pow2(inner)) g.lo_shift = log2_exact(inner); else g.inner_div = inner;
            g.tw_stride = (uint64_t)1 << (pl.log_root - s.cur);
        } else {                           // slab of a matrix: jq = (x / inner) * 2^col_bits + col0 + x % inner
            const int li = log2_exact(inner);
            g.lo_mask = inner - 1;
            g.hi_shift = li;
            g.hi_lshift = ax.col_bits;
            g.jq_off = ax.col0;
            g.tw_stride = (uint64_t)1 << (pl.log_root - s.cur - ax.col_bits);
        }
        if (s.kind == Step64::TILE) g.tiles_per_block = g.columns / ((uint64_t)TILE64_N >> s.q);
        return launch_pass<F>(pl, d, s.kind == Step64::TILE, g, inverse, ex, st);
    }
    if (s.kind == Step64::LOCAL) {
        const uint64_t n_chunks = ((uint64_t)outer << log_len) >> LOCAL64_LOG;
        const unsigned grid = (unsigned)(n_chunks < (1u << 20) ? n_chunks : (1u << 20));
        const typename F::W scale = scale_arg<F>(ex.scale_word);
        if (!inverse) NTT_LAUNCH_P("k64_local", (k64_local<F, false, false, false>), dim3(grid), dim3(LOCAL64_T), st, d, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, scale, s.q, n_chunks, fc, pl.rt);
        else if (ex.other) NTT_LAUNCH_P("k64_local", (k64_local<F, true, true, false>), dim3(grid), dim3(LOCAL64_T), st, d, ex.other, (uint64_t*)nullptr, (uint64_t)0, scale, s.q, n_chunks, fc, pl.rt);
        else NTT_LAUNCH_P("k64_local", (k64_local<F, true, false, false>), dim3(grid), dim3(LOCAL64_T), st, d, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, scale, s.q, n_chunks, fc, pl.rt);
    } else {
        const uint64_t n_blocks = (uint64_t)outer << (log_len - s.q);
        const unsigned grid = (unsigned)(n_blocks < (1u << 20) ? n_blocks : (1u << 20));
        if (inverse) NTT_LAUNCH_P("k64_small", (k64_small<F, true>), dim3(grid), dim3(256), st, d, s.q, n_blocks, fc, pl.tw);
        else NTT_LAUNCH_P("k64_small", (k64_small<F, false>), dim3(grid), dim3(256), st, d, s.q, n_blocks, fc, pl.tw);
    }
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

// in-place transform along the axis
template <class F>
int transform64_f(const Plan64& pl, uint64_t* d, const Axis64& ax, bool inverse, hipStream_t st) {
    const std::vector<Step64> steps = plan_steps(ax.log_len, ax.inner, ax.outer, ax.col_bits > 0);
    const TileExtra none;
    int rc;
    if (!inverse) {
        for (const Step64& s : steps) if ((rc = run_step<F>(pl, d, s, ax, false, none, st)) != NTT_OK) return rc;
    } else {
        for (size_t i = steps.size(); i-- > 0;) if ((rc = run_step<F>(pl, d, steps[i], ax, true, none, st)) != NTT_OK) return rc;
    }
    return NTT_OK;
}

// Whether the outermost pass of a transform along `ax` (first forward / last inverse) can read raw caller words or write
// scaled canonical results itself: the tiled and the register-only pass can, the contiguous-chunk passes cannot.
bool outer_pass_takes_io(const Axis64& ax) {
    if (ax.log_len <= 0) return false;
    const std::vector<Step64> steps = plan_steps(ax.log_len, ax.inner, ax.outer, ax.col_bits > 0);
    return !steps.empty() && (steps.front().kind == Step64::TILE || steps.front().kind == Step64::REG);
}

// The transform with `io` (source of raw words, or destination and scale of the results) applied in its outermost pass;
// only for axes for which outer_pass_takes_io() holds.
template <class F>
int transform64_io(const Plan64& pl, uint64_t* d, const Axis64& ax, bool inverse, const TileExtra& io, hipStream_t st) {
    const std::vector<Step64> steps = plan_steps(ax.log_len, ax.inner, ax.outer, ax.col_bits > 0);
    const TileExtra none;
    int rc;
    if (!inverse) {
        for (size_t i = 0; i < steps.size(); ++i)
            if ((rc = run_step<F>(pl, d, steps[i], ax, false, i == 0 ? io : none, st)) != NTT_OK) return rc;
    } else {
        for (size_t i = steps.size(); i-- > 0;)
            if ((rc = run_step<F>(pl, d, steps[i], ax, true, i == 0 ? io : none, st)) != NTT_OK) return rc;
    }
    return NTT_OK;
}

int transform64_axis(const Plan64& pl, uint64_t* d, const Axis64& ax, bool inverse, hipStream_t st) {
    if (ax.log_len + ax.col_bits > pl.log_root) return fail(NTT_ERR_ARG, "transform longer than the plan's root order");
    if (ax.log_len == 0 || ax.outer == 0 || ax.inner == 0) return NTT_OK;
    return pl.fp ? transform64_f<FieldD>(pl, d, ax, inverse, st) : transform64_f<Field64>(pl, d, ax, inverse, st);
}

int transform64(const Plan64& pl, uint64_t* d, int log_len, size_t inner, size_t outer, bool inverse, hipStream_t st) {
    Axis64 ax;
    ax.log_len = log_len; ax.inner = inner; ax.outer = outer;
    return transform64_axis(pl, d, ax, inverse, st);
}

// The column axis of a four-step matrix in the layout an all-to-all of column slabs leaves: the axis index
// c = part * Cl + l (part < 2^log_parts, l < Cl = 2^(log_cols - log_parts)) of row r lives at
// part * part_stride + r * Cl + l, r < rows.  Forward: one register pass over the parts (their twiddles depend on l
// only), then plain transforms of length Cl on every (part, row); inverse: the mirror.
// the register pass over the 2^log_parts parts of the split layout: `rows` rows of cl columns per part, parts
// part_stride apart, twiddle index = position inside the part
Pass64 parts_pass_geometry(const Plan64& pl, int log_cols, int log_parts, uint64_t part_stride, uint64_t rows) {
    const uint64_t cl = (uint64_t)1 << (log_cols - log_parts);
    Pass64 g{};
    g.columns = rows * cl;
    g.step = part_stride;
    g.block_stride = 0;
    g.n_blocks = 1;
    g.q = log_parts;
    g.hi_shift = 63;
    g.lo_mask = cl - 1;
    g.tw_stride = (uint64_t)1 << (pl.log_root - log_cols);
    g.tw_mask = ((uint64_t)1 << pl.log_root) - 1;
    return g;
}

int transform64_split(const Plan64& pl, uint64_t* d, int log_cols, int log_parts, uint64_t part_stride, uint64_t rows, bool inverse,
                      hipStream_t st) {
    if (log_parts < 0 || log_parts > 4 || log_parts > log_cols) return fail(NTT_ERR_ARG, "split layout supports 1 to 16 parts");
    if (log_cols > pl.log_root) return fail(NTT_ERR_ARG, "transform longer than the plan's root order");
    if (rows == 0) return NTT_OK;
    const int log_cl = log_cols - log_parts;
    auto parts_pass = [&]() -> int {
        if (log_parts == 0) return NTT_OK;
        const Pass64 g = parts_pass_geometry(pl, log_cols, log_parts, part_stride, rows);
        const TileExtra none;
        return pl.fp ? launch_pass<FieldD>(pl, d, false, g, inverse, none, st) : launch_pass<Field64>(pl, d, false, g, inverse, none, st);
    };
    auto rows_of_parts = [&]() -> int {
        for (uint64_t part = 0; part < ((uint64_t)1 << log_parts); ++part) {
            int rc = transform64(pl, d + part * part_stride, log_cl, 1, rows, inverse, st);
            if (rc != NTT_OK) return rc;
        }
        return NTT_OK;
    };
    int rc;
    if (!inverse) {
        if ((rc = parts_pass()) != NTT_OK) return rc;
        return rows_of_parts();
    }
    if ((rc = rows_of_parts()) != NTT_OK) return rc;
    return parts_pass();
}

// NTT_CORE64=off keeps the last forward passes, the pointwise product and the first inverse pass as separate kernels
// (A/B measurements)
bool fused_core_allowed() {
    static const bool v = [] { const char* e = getenv("NTT_CORE64"); return !(e && strcmp(e, "off") == 0); }();
    return v;
}

void pointwise64(const Plan64& pl, uint64_t* a, const uint64_t* b, uint64_t count, hipStream_t st);

// Forward column-axis transforms of two arrays in the split layout, their pointwise product and the inverse column-axis
// transform of it, result in `a`: the middle of a four-step product, between the exchanges.  When the transforms inside
// a part end in full chunks the last forward passes, the product and the first inverse pass are one kernel.
template <class F>
int product64_split_f(const Plan64& pl, uint64_t* a, uint64_t* b, int log_cols, int log_parts, uint64_t part_stride, uint64_t rows,
                      hipStream_t st) {
    const int log_cl = log_cols - log_parts;
    const uint64_t cl = (uint64_t)1 << log_cl;
    const TileExtra none;
    int rc;
    auto parts_pass = [&](uint64_t* d, bool inverse) -> int {
        if (log_parts == 0) return NTT_OK;
        return launch_pass<F>(pl, d, false, parts_pass_geometry(pl, log_cols, log_parts, part_stride, rows), inverse, none, st);
    };
    if ((rc = parts_pass(a, false)) != NTT_OK || (rc = parts_pass(b, false)) != NTT_OK) return rc;
    Axis64 ax;
    ax.log_len = log_cl; ax.inner = 1; ax.outer = rows;
    const std::vector<Step64> steps = log_cl > 0 ? plan_steps(log_cl, 1, rows) : std::vector<Step64>();
    const bool core = !steps.empty() && steps.back().kind == Step64::LOCAL && steps.back().q == LOCAL64_LOG && fused_core_allowed();
    const size_t n_outer = core ? steps.size() - 1 : steps.size();
    for (uint64_t part = 0; part < ((uint64_t)1 << log_parts); ++part) {
        uint64_t* pa = a + part * part_stride;
        uint64_t* pb = b + part * part_stride;
        for (size_t i = 0; i < n_outer; ++i)
            if ((rc = run_step<F>(pl, pa, steps[i], ax, false, none, st)) != NTT_OK || (rc = run_step<F>(pl, pb, steps[i], ax, false, none, st)) != NTT_OK) return rc;
        if (core) {
            const uint64_t n_chunks = (rows << log_cl) >> LOCAL64_LOG;
            const unsigned grid = (unsigned)(n_chunks < (1u << 20) ? n_chunks : (1u << 20));
            NTT_LAUNCH_P("k64_core_mul", (k64_core_mul<F>), dim3(grid), dim3(LOCAL64_T), st, pa, (const uint64_t*)pb, n_chunks, consts_of<F>(pl), pl.rt);
            HIP_TRY(hipGetLastError());
        } else {
            pointwise64(pl, pa, pb, rows * cl, st);
            HIP_TRY(hipGetLastError());
        }
        for (size_t i = n_outer; i-- > 0;)
            if ((rc = run_step<F>(pl, pa, steps[i], ax, true, none, st)) != NTT_OK) return rc;
    }
    return parts_pass(a, true);
}

// element-wise kernels, by field
template <class F>
void launch_load64(const Plan64& pl, const uint64_t* src, size_t src_stride, uint64_t n, uint64_t* dst, uint64_t len, uint64_t rows, hipStream_t st) {
    NTT_LAUNCH_P("k64_load", (k64_load<F>), dim3(grid_for(rows * len)), dim3(256), st, src, src_stride, n, dst, len, rows, consts_of<F>(pl));
}
void load64(const Plan64& pl, const uint64_t* src, size_t src_stride, uint64_t n, uint64_t* dst, uint64_t len, uint64_t rows, hipStream_t st) {
    if (pl.fp) launch_load64<FieldD>(pl, src, src_stride, n, dst, len, rows, st); else launch_load64<Field64>(pl, src, src_stride, n, dst, len, rows, st);
}
void pointwise64(const Plan64& pl, uint64_t* a, const uint64_t* b, uint64_t count, hipStream_t st) {
    if (pl.fp) NTT_LAUNCH_P("k64_pointwise", (k64_pointwise<FieldD>), dim3(grid_for(count)), dim3(256), st, a, b, count, pl.fd);
    else NTT_LAUNCH_P("k64_pointwise", (k64_pointwise<Field64>), dim3(grid_for(count)), dim3(256), st, a, b, count, pl.fc);
}
// out[row][i] = canonical(data[row][i] * scale), scale a plain residue
void finalize64(const Plan64& pl, const uint64_t* data, uint64_t len, uint64_t* out, size_t out_stride, uint64_t n_out, uint64_t rows,
                uint64_t scale, hipStream_t st) {
    const uint64_t word = tw_word(pl, scale);
    if (pl.fp) NTT_LAUNCH_P("k64_finalize", (k64_finalize<FieldD>), dim3(grid_for(rows * n_out)), dim3(256), st, data, len, out, out_stride, n_out, rows, scale_arg<FieldD>(word), pl.fd);
    else NTT_LAUNCH_P("k64_finalize", (k64_finalize<Field64>), dim3(grid_for(rows * n_out)), dim3(256), st, data, len, out, out_stride, n_out, rows, scale_arg<Field64>(word), pl.fc);
}
void twiddle64(const Plan64& pl, uint64_t* d, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0, int row_bits, int rows_bitrev,
               int inverse, hipStream_t st) {
    const uint64_t mask = ((uint64_t)1 << pl.log_root) - 1;
    if (pl.fp) NTT_LAUNCH_P("k64_twiddle", (k64_twiddle<FieldD>), dim3(grid_for(rows * cols)), dim3(256), st, d, rows, cols, row0, col0, row_bits, rows_bitrev, inverse, (uint64_t)1, mask, pl.fd, pl.tw);
    else NTT_LAUNCH_P("k64_twiddle", (k64_twiddle<Field64>), dim3(grid_for(rows * cols)), dim3(256), st, d, rows, cols, row0, col0, row_bits, rows_bitrev, inverse, (uint64_t)1, mask, pl.fc, pl.tw);
}

// `rows` products in the field at once, operands and results in caller memory, two work arrays of rows * L words.  When
// the first pass of the transform is a tile pass, the operand load (reduction of arbitrary words, zero padding) rides on
// it, the pointwise product on the first pass of the inverse, and scaling, canonical reduction and the copy to `out`
// on its last pass.
template <class F>
int product64_rows(const Plan64& pl, const uint64_t* a, size_t sa, const uint64_t* b, size_t sb, uint64_t* out, size_t so, size_t n,
                   int logL, size_t rows, uint64_t* wa, uint64_t* wb, hipStream_t st) {
    const size_t L = (size_t)1 << logL;
    const uint64_t p = pl.fc.p;
    const uint64_t Linv = h_powmod((uint64_t)L % p, p - 2, p);
    const uint64_t scale = h_mulmod(Linv, pointwise_compensation(pl), p);
    if (logL <= LOCAL64_LOG && fused_core_allowed()) {      // short products: one kernel, several products per workgroup
        const uint64_t n_chunks = ((uint64_t)rows + (LOCAL64_N >> logL) - 1) / (LOCAL64_N >> logL);
        const unsigned grid = (unsigned)(n_chunks < (1u << 20) ? n_chunks : (1u << 20));
        if (logL >= 6)   // first and last round straight from / to caller memory
            NTT_LAUNCH_P("k64_short_mul", (k64_short_mul<F, true>), dim3(grid), dim3(LOCAL64_T), st, a, (uint64_t)sa, b, (uint64_t)sb, out, (uint64_t)so,
                         (uint64_t)n, (uint64_t)(2 * n - 1), (uint64_t)rows, logL, scale_arg<F>(tw_word(pl, scale)), consts_of<F>(pl), pl.rt);
        else
            NTT_LAUNCH_P("k64_short_mul", (k64_short_mul<F, false>), dim3(grid), dim3(LOCAL64_T), st, a, (uint64_t)sa, b, (uint64_t)sb, out, (uint64_t)so,
                         (uint64_t)n, (uint64_t)(2 * n - 1), (uint64_t)rows, logL, scale_arg<F>(tw_word(pl, scale)), consts_of<F>(pl), pl.rt);
        HIP_TRY(hipGetLastError());
        return NTT_OK;
    }
    const std::vector<Step64> steps = plan_steps(logL, 1, rows);
    Axis64 ax;
    ax.log_len = logL;
    ax.outer = rows;
    int rc;
    // the first pass (tiled or register-only) can take the operands as they are and the last one write the results
    const bool outer_pass = steps.front().kind == Step64::TILE || steps.front().kind == Step64::REG;
    const bool fused = steps.size() >= 2 && outer_pass && steps.back().kind == Step64::LOCAL;
    if (!fused) {
        load64(pl, a, sa, n, wa, L, rows, st);
        load64(pl, b, sb, n, wb, L, rows, st);
        if ((rc = transform64_f<F>(pl, wa, ax, false, st)) != NTT_OK) return rc;
        if ((rc = transform64_f<F>(pl, wb, ax, false, st)) != NTT_OK) return rc;
        pointwise64(pl, wa, wb, rows * L, st);
        if ((rc = transform64_f<F>(pl, wa, ax, true, st)) != NTT_OK) return rc;
        finalize64(pl, wa, L, out, so, 2 * n - 1, rows, scale, st);
        HIP_TRY(hipGetLastError());
        return NTT_OK;
    }
    const TileExtra none;
    // transforms that end in full chunks: the last forward pass of both operands, the pointwise product and the first
    // inverse pass are one kernel
    const bool core = steps.back().q == LOCAL64_LOG && fused_core_allowed();
    const size_t n_outer = core ? steps.size() - 1 : steps.size();
    for (size_t i = 0; i < n_outer; ++i) {
        if (steps[i].kind == Step64::TILE || steps[i].kind == Step64::REG) {   // one launch takes the pass over both work arrays
            TileExtra ex;
            ex.data2 = wb;
            if (i == 0) { ex.src = a; ex.src_stride = sa; ex.src2 = b; ex.src_stride2 = sb; ex.src_valid = n; }
            if ((rc = run_step<F>(pl, wa, steps[i], ax, false, ex, st)) != NTT_OK) return rc;
        } else {                                                               // the chunk pass when the fused core is switched off
            if ((rc = run_step<F>(pl, wa, steps[i], ax, false, none, st)) != NTT_OK) return rc;
            if ((rc = run_step<F>(pl, wb, steps[i], ax, false, none, st)) != NTT_OK) return rc;
        }
    }
    if (core) {
        const uint64_t n_chunks = ((uint64_t)rows << logL) >> LOCAL64_LOG;
        const unsigned grid = (unsigned)(n_chunks < (1u << 20) ? n_chunks : (1u << 20));
        NTT_LAUNCH_P("k64_core_mul", (k64_core_mul<F>), dim3(grid), dim3(LOCAL64_T), st, wa, (const uint64_t*)wb, n_chunks, consts_of<F>(pl), pl.rt);
        HIP_TRY(hipGetLastError());
    }
    for (size_t i = n_outer; i-- > 0;) {
        TileExtra ex;
        if (i + 1 == steps.size()) ex.other = wb;
        if (i == 0) { ex.out = out; ex.n_out = 2 * n - 1; ex.scale_word = tw_word(pl, scale); ex.out_stride = so; }
        if ((rc = run_step<F>(pl, wa, steps[i], ax, true, ex, st)) != NTT_OK) return rc;
    }
    return NTT_OK;
}

// products that share one pass over the work arrays: as many as fit 256 MiB per array (128 at L = 2^18, one from 2^25)
size_t field64_group_rows(int logL, size_t batch) {
    const size_t fit = logL >= 25 ? 1 : ((size_t)1 << (25 - logL));
    return batch < 1 ? 1 : (batch < fit ? batch : fit);
}
size_t field64_workspace_bytes(int logL, size_t batch) {
    return 2 * field64_group_rows(logL, batch) * ((size_t)1 << logL) * sizeof(uint64_t);
}

// whole products in the 64-bit field, a group of rows at a time through one workspace
int field64_product_device(Device* dev, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint64_t p, size_t batch,
                           size_t sa, size_t sb, size_t so, void* ws, size_t ws_bytes, hipStream_t st) {
    int logL = ceil_log2(2 * (uint64_t)n - 1);
    if (logL < 1) logL = 1;
    Plan64* pl = nullptr;
    int rc = get_plan64(dev, p, logL, &pl);
    if (rc != NTT_OK) return rc;
    const size_t L = (size_t)1 << logL, group = field64_group_rows(logL, batch), need = field64_workspace_bytes(logL, batch);
    if (ws) {
        if (ws_bytes < need) return fail(NTT_ERR_WORKSPACE, "workspace too small");
    } else {
        if ((rc = dev->workspace_for(st, need, &ws)) != NTT_OK) return rc;
    }
    uint64_t* wa = (uint64_t*)ws;
    uint64_t* wb = wa + group * L;
    for (size_t row = 0; row < batch; row += group) {
        const size_t rows = batch - row < group ? batch - row : group;
        rc = pl->fp ? product64_rows<FieldD>(*pl, a + row * sa, sa, b + row * sb, sb, out + row * so, so, n, logL, rows, wa, wb, st)
                    : product64_rows<Field64>(*pl, a + row * sa, sa, b + row * sb, sb, out + row * so, so, n, logL, rows, wa, wb, st);
        if (rc != NTT_OK) return rc;
    }
    return NTT_OK;
}

size_t wide_crt_workspace_bytes(size_t n, uint64_t p, size_t batch) {
    const int logL = ceil_log2(2 * (uint64_t)n - 1) < 1 ? 1 : ceil_log2(2 * (uint64_t)n - 1);
    const size_t k = (size_t)wide_crt_basis(n, p, fieldd_allowed()).k;
    return field64_workspace_bytes(logL, batch) + k * batch * (2 * n - 1) * sizeof(uint64_t);
}

int make_wide_crt_consts(uint64_t p, const WideBasis& basis, WideCrtConsts* c) {
    c->k = basis.k;
    for (int i = 0; i < 3; ++i) {
        const uint64_t q = basis.q[i];
        uint64_t inv = q;
        for (int it = 0; it < 6; ++it) inv *= 2 - q * inv;
        c->q[i] = q; c->nq[i] = 0 - inv;
    }
    const uint64_t q1 = basis.q[0], q2 = basis.q[1], q3 = basis.q[2];
    c->inv1_m2 = h_mont64(h_powmod(q1 % q2, q2 - 2, q2), q2);
    c->inv1_m3 = h_mont64(h_powmod(q1 % q3, q3 - 2, q3), q3);
    c->inv2_m3 = h_mont64(h_powmod(q2 % q3, q3 - 2, q3), q3);
    c->p = p;
    c->shift = __builtin_clzll(p);
    c->d = p << c->shift;
    c->v = (uint64_t)((~(u128)0) / c->d - ((u128)1 << 64));
    c->q1_p = q1 % p;
    c->q12_p = h_mulmod(q1 % p, q2 % p, p);
    return NTT_OK;
}

// Any modulus below 2^64, transform lengths up to 2^32: one 64-bit-field product per wide prime, then Garner.
int wide_crt_product_device(Device* dev, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint64_t p, size_t batch,
                            size_t sa, size_t sb, size_t so, void* ws, size_t ws_bytes, hipStream_t st) {
    const WideBasis basis = wide_crt_basis(n, p, fieldd_allowed());
    const int k = basis.k;
    if (k == 0 || p < 2) return fail(NTT_ERR_MODULUS, "n*(p-1)^2 exceeds the product of the three wide CRT primes");
    const size_t need = wide_crt_workspace_bytes(n, p, batch);
    if (ws) {
        if (ws_bytes < need) return fail(NTT_ERR_WORKSPACE, "workspace too small");
    } else {
        int rc = dev->workspace_for(st, need, &ws);
        if (rc != NTT_OK) return rc;
    }
    const size_t field_ws = need - (size_t)k * batch * (2 * n - 1) * sizeof(uint64_t);
    uint64_t* res = (uint64_t*)((char*)ws + field_ws);
    const size_t count = 2 * n - 1, res_stride = batch * count;
    for (int i = 0; i < k; ++i) {
        int rc = field64_product_device(dev, a, b, res + (size_t)i * res_stride, n, basis.q[i], batch, sa, sb, count, ws, field_ws, st);
        if (rc != NTT_OK) return rc;
    }
    WideCrtConsts c;
    int rc = make_wide_crt_consts(p, basis, &c);
    if (rc != NTT_OK) return rc;
    NTT_LAUNCH_P("k64_garner", k64_garner, dim3((unsigned)((count + 255) / 256), grid_rows(batch)), dim3(256), st, (const uint64_t*)res,
                 res_stride, out, count, batch, so, count, c);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

// Which arithmetic serves (n, p), in the order tried:
//   32-bit direct  p < 2^31 NTT-friendly and L <= 2^18 (every case the reference's NTT_multiply covers)
//   64-bit field in double precision   prime p <= 1.5e15 with L | p-1: the reference's large modulus among them.  Same
//                  residues as the CRT route at about half its cost (three transforms of one 64-bit array instead of
//                  twelve of 32-bit ones plus the merge); skipped under NTT_ROUTE_REFERENCE
//   four-prime CRT any p with n*(p-1)^2 below the product of the primes, L <= 2^18 (the reference's route for p > 2^32)
//   64-bit field   p < 2^60 with L | p-1: lengths and moduli beyond the above, integer Montgomery arithmetic
//   wide CRT       everything else up to L = 2^32: two or three NTT primes of 50 bits (double-precision field) or 59 bits
//                  (integer field), whichever basis needs fewer transforms' worth of work; Garner recombination
enum Route { ROUTE_DIRECT32, ROUTE_CRT, ROUTE_FIELD64, ROUTE_WIDECRT, ROUTE_NONE };
constexpr int MAX_LOG_L32 = 18;
std::atomic<int> g_route_policy{NTT_ROUTE_AUTO};

Route choose_route(uint64_t p, size_t n, std::string* why) {
    int logL = ceil_log2(2 * (uint64_t)n - 1);
    if (logL < 1) logL = 1;
    if (logL <= MAX_LOG_L32 && direct_path_ok(p, logL)) return ROUTE_DIRECT32;
    std::string w64;
    const bool crt = logL <= MAX_LOG_L32 && n < (1u << 30) && crt_bound_ok((int)n, p);
    if (crt && g_route_policy.load() == NTT_ROUTE_AUTO && fieldd_allowed() && p < (1ull << 60) && (double)p <= FIELDD_MAX_P &&
        modulus64_ok(p, logL, &w64))
        return ROUTE_FIELD64;
    if (crt) return ROUTE_CRT;
    if (modulus64_ok(p, logL, &w64)) return ROUTE_FIELD64;
    if (p >= 2 && logL <= 32 && wide_crt_basis(n, p, fieldd_allowed()).k != 0) return ROUTE_WIDECRT;
    if (why) *why = "no exact route for this modulus and length: the four-prime CRT needs n*(p-1)^2 < 2^118.9 and length <= 2^18; "
                    "the 64-bit field needs " + w64;
    return ROUTE_NONE;
}

int check_args(const void* a, const void* b, const void* ab, int n, size_t batch) {
    if (!a || !b || !ab) return fail(NTT_ERR_ARG, "null pointer");
    if (n < 1) return fail(NTT_ERR_ARG, "n must be at least 1");
    if (n > (1 << 30)) return fail(NTT_ERR_ARG, "n too large");
    if (batch >= (1ull << 31) / 64) return fail(NTT_ERR_ARG, "batch too large for one launch");
    return NTT_OK;
}

// ---- persistent copy workers ------------------------------------------------------------------
// The staging copies of large host-pointer calls are memory-bound on the CPU side; a handful of threads that live
// as long as the library (the analogue of the reference's worker pool, main_pthread.cc:296-441, here only for
// memcpy) move them at several times the rate of one.
// Set in the per-device host threads of a multi-device call: their staging copies run in the calling thread (the
// devices' threads are the parallelism there, and the pool serves one caller at a time).
thread_local bool t_inline_copies = false;

class CopyPool {
  public:
    // Created on first use, stopped and joined by ntt_shutdown() (and created again by the next call that needs it).  A
    // process that exits without ntt_shutdown() leaves the workers blocked on their condition variable: the instance
    // is deliberately never destroyed by a static destructor.
    static CopyPool& get() {
        std::lock_guard<std::mutex> lk(instance_mutex());
        if (!instance()) instance() = new CopyPool();
        return *instance();
    }
    static void shutdown() {
        std::lock_guard<std::mutex> lk(instance_mutex());
        CopyPool* p = instance();
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk2(p->m_);
            p->stop_ = true;
            ++p->generation_;
        }
        p->cv_work_.notify_all();
        for (auto& t : p->workers_) t.join();
        delete p;
        instance() = nullptr;
    }
    void copy(void* dst, const void* src, size_t bytes, size_t piece = (size_t)1 << 20) {
        int parts = (int)((bytes + piece - 1) / piece);
        if (parts > (int)workers_.size() + 1) parts = (int)workers_.size() + 1;
        if (parts <= 1 || t_inline_copies) { memcpy(dst, src, bytes); return; }
        const size_t per = (((bytes + parts - 1) / parts) + 63) & ~(size_t)63;  // ceil: parts * per >= bytes, no tail is lost
        run(parts, [=](int i) {
            const size_t lo = (size_t)i * per;
            if (lo >= bytes) return;
            const size_t len = lo + per < bytes ? per : bytes - lo;
            memcpy((char*)dst + lo, (const char*)src + lo, len);
        });
    }
    // fn(part, lo, hi) over [0, count) cut into at most workers+1 ranges of at least `grain` items
    void for_range(size_t count, size_t grain, const std::function<void(int, size_t, size_t)>& fn) {
        int parts = (int)((count + grain - 1) / grain);
        if (parts > (int)workers_.size() + 1) parts = (int)workers_.size() + 1;
        if (parts <= 1 || t_inline_copies) { fn(0, 0, count); return; }
        const size_t per = (((count + parts - 1) / parts) + 15) & ~(size_t)15;  // ceil: parts * per >= count
        run(parts, [&](int i) {
            const size_t lo = (size_t)i * per;
            if (lo >= count) return;
            fn(i, lo, lo + per < count ? lo + per : count);
        });
    }
    int max_parts() const { return (int)workers_.size() + 1; }
  private:
    CopyPool() {
        unsigned hw = std::thread::hardware_concurrency();
        const char* omp = getenv("OMP_NUM_THREADS");   // the share of the machine this process was given, where set
        if (omp && atoi(omp) > 0 && (unsigned)atoi(omp) < hw) hw = (unsigned)atoi(omp);
        int n = hw > 2 ? (int)hw / 2 - 1 : 0;
        if (n > 7) n = 7;
        const char* forced = getenv("NTT_COPY_WORKERS");   // test hook: exact number of helper threads (0..15)
        if (forced && forced[0] >= '0' && forced[0] <= '9') n = atoi(forced) > 15 ? 15 : atoi(forced);
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    static CopyPool*& instance() { static CopyPool* p = nullptr; return p; }
    static std::mutex& instance_mutex() { static std::mutex m; return m; }
    void run(int parts, const std::function<void(int)>& fn) {
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; parts_ = parts; next_ = 0; remaining_ = parts; ++generation_;
        }
        cv_work_.notify_all();
        work();  // the caller takes its share
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [this] { return remaining_ == 0; });
        fn_ = nullptr;
    }
    void work() {
        for (;;) {
            int i;
            const std::function<void(int)>* fn;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!fn_ || next_ >= parts_) return;
                i = next_++; fn = fn_;
            }
            (*fn)(i);
            std::lock_guard<std::mutex> lk(m_);
            if (--remaining_ == 0) cv_done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(int)>* fn_ = nullptr;
    int parts_ = 0, next_ = 0, remaining_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

size_t stage_limit_bytes() {  // NTT_STAGE_LIMIT_MB: test hook, lets small calls exercise the chunked path
    static const size_t v = [] {
        const char* e = getenv("NTT_STAGE_LIMIT_MB");
        return (e && atoi(e) > 0) ? (size_t)atoi(e) << 20 : (size_t)64 << 20;
    }();
    return v;
}

// Host-pointer call too large for one staging buffer, on one device: the rows go through in chunks that alternate
// between two staging sets.  Per chunk: rows -> page-locked memory (copy workers) -> device, product, device ->
// page-locked memory; the copy out to the caller's array happens when the set comes up for reuse, so it overlaps the
// other set's transfers.  Synchronous: everything has landed in `ab` on return.
template <typename T, typename F>
int run_host_pipelined(Device* dev, const T* a, const T* b, T* ab, int n, uint64_t p, size_t batch, F&& run) {
    HIP_TRY(hipSetDevice(dev->id));
    const size_t row_in = (size_t)n * sizeof(T), row_out = (size_t)(2 * n - 1) * sizeof(T);
    size_t rows = (stage_limit_bytes() / 4) / (2 * row_in + row_out);
    if (rows < 1) rows = 1;
    if (rows > batch) rows = batch;
    const size_t ws_bytes = ntt_workspace_bytes(n, p, rows);
    int rc;
    for (auto& s : dev->stage) {
        if (!s.stream) HIP_TRY(hipStreamCreate(&s.stream));
        if (!s.done) HIP_TRY(hipEventCreate(&s.done));
        if ((rc = s.pin.ensure(rows * (2 * row_in + row_out))) != NTT_OK) return rc;
        if ((rc = s.a.ensure(rows * row_in)) != NTT_OK) return rc;
        if ((rc = s.b.ensure(rows * row_in)) != NTT_OK) return rc;
        if ((rc = s.out.ensure(rows * row_out)) != NTT_OK) return rc;
        if (ws_bytes && (rc = s.ws.ensure(ws_bytes)) != NTT_OK) return rc;
    }
    CopyPool& pool = CopyPool::get();
    struct Pending { T* dst = nullptr; size_t bytes = 0; } pending[2];
    auto drain = [&](int k) -> int {  // wait for set k's last chunk and hand its results to the caller
        if (!pending[k].dst) return NTT_OK;
        HIP_TRY(hipEventSynchronize(dev->stage[k].done));
        char* po = (char*)dev->stage[k].pin.ptr + 2 * rows * row_in;
        pool.copy(pending[k].dst, po, pending[k].bytes);
        pending[k].dst = nullptr;
        return NTT_OK;
    };
    rc = NTT_OK;
    size_t chunk = 0;
    for (size_t lo = 0; lo < batch && rc == NTT_OK; lo += rows, ++chunk) {
        const size_t cnt = lo + rows < batch ? rows : batch - lo;
        const int k = (int)(chunk & 1);
        Device::StageSet& s = dev->stage[k];
        if ((rc = drain(k)) != NTT_OK) break;
        char* pa = (char*)s.pin.ptr;
        char* pb = pa + rows * row_in;
        char* po = pb + rows * row_in;
        pool.copy(pa, a + lo * n, cnt * row_in);
        pool.copy(pb, b + lo * n, cnt * row_in);
        hipError_t e = hipMemcpyAsync(s.a.ptr, pa, cnt * row_in, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s.b.ptr, pb, cnt * row_in, hipMemcpyHostToDevice, s.stream);
        if (e != hipSuccess) { rc = fail(NTT_ERR_DEVICE, std::string("staging copy: ") + hipGetErrorString(e)); break; }
        rc = run(dev, (const T*)s.a.ptr, (const T*)s.b.ptr, (T*)s.out.ptr, cnt, ws_bytes ? s.ws.ptr : nullptr, ws_bytes, s.stream);
        if (rc != NTT_OK) break;
        e = hipMemcpyAsync(po, s.out.ptr, cnt * row_out, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess) e = hipEventRecord(s.done, s.stream);
        if (e != hipSuccess) { rc = fail(NTT_ERR_DEVICE, std::string("staging copy: ") + hipGetErrorString(e)); break; }
        pending[k].dst = ab + lo * (size_t)(2 * n - 1);
        pending[k].bytes = cnt * row_out;
    }
    if (rc != NTT_OK) {  // nothing may still be in flight towards memory the caller owns
        hipError_t e = hipDeviceSynchronize();
        (void)e;
        return rc;
    }
    const int first = (int)(chunk & 1);  // the older of the two outstanding chunks
    if ((rc = drain(first)) != NTT_OK) return rc;
    return drain(first ^ 1);
}

// Host-pointer driver for one device: H2D, run, D2H.  Calls up to stage_limit_bytes() bytes go through the
// device's page-locked staging buffer (a host memcpy each way, then true asynchronous DMA) -- pageable
// copies of a few MB cost several times the multiply itself; larger calls use the runtime's own
// pageable path.  `finish` collects what must run after the stream has been synchronised.
typedef std::vector<std::function<void()>> FinishList;

template <typename T, typename F>
int run_host_shard(Device* dev, const T* a, const T* b, T* ab, int n, size_t batch, F&& run, FinishList& finish) {
    if (batch == 0) return NTT_OK;
    HIP_TRY(hipSetDevice(dev->id));
    const size_t in_bytes = batch * (size_t)n * sizeof(T), out_bytes = batch * (size_t)(2 * n - 1) * sizeof(T);
    int rc;
    if ((rc = dev->io_a.ensure(in_bytes)) != NTT_OK) return rc;
    if ((rc = dev->io_b.ensure(in_bytes)) != NTT_OK) return rc;
    if ((rc = dev->io_out.ensure(out_bytes)) != NTT_OK) return rc;
    const bool staged = 2 * in_bytes + out_bytes <= stage_limit_bytes() && dev->pin.ensure(2 * in_bytes + out_bytes) == NTT_OK;
    char* pa = staged ? (char*)dev->pin.ptr : nullptr;
    char* pb = staged ? pa + in_bytes : nullptr;
    char* po = staged ? pb + in_bytes : nullptr;
    if (staged) {
        CopyPool& pool = CopyPool::get();
        pool.copy(pa, a, in_bytes, (size_t)512 << 10);
        HIP_TRY(hipMemcpyAsync(dev->io_a.ptr, pa, in_bytes, hipMemcpyHostToDevice, dev->stream));
        pool.copy(pb, b, in_bytes, (size_t)512 << 10);
        HIP_TRY(hipMemcpyAsync(dev->io_b.ptr, pb, in_bytes, hipMemcpyHostToDevice, dev->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(dev->io_a.ptr, a, in_bytes, hipMemcpyHostToDevice, dev->stream));
        HIP_TRY(hipMemcpyAsync(dev->io_b.ptr, b, in_bytes, hipMemcpyHostToDevice, dev->stream));
    }
    rc = run(dev, (const T*)dev->io_a.ptr, (const T*)dev->io_b.ptr, (T*)dev->io_out.ptr, batch, nullptr, (size_t)0, dev->stream);
    if (rc != NTT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(staged ? (void*)po : (void*)ab, dev->io_out.ptr, out_bytes, hipMemcpyDeviceToHost, dev->stream));
    if (staged) finish.push_back([=] { CopyPool::get().copy(ab, po, out_bytes, (size_t)512 << 10); });
    return NTT_OK;
}

// dst[i] = low word of src[i]; returns the largest value seen, saturated to 2^32-1 when any high word is set.
// Written so that the loop vectorises (no 64-bit compare); the AVX2 clone is picked at load time where available.
#if defined(__x86_64__) && !defined(NTT_CPU_TWIN) && !defined(__HIP_DEVICE_COMPILE__)
#define NTT_HOST_SIMD __attribute__((target_clones("avx2", "default")))
#else
#define NTT_HOST_SIMD
#endif
NTT_HOST_SIMD uint32_t narrow_u64(const uint64_t* src, uint32_t* dst, size_t count) {
    uint32_t hi_or = 0, mx = 0;
    for (size_t i = 0; i < count; ++i) {
        const uint64_t v = src[i];
        const uint32_t lo = (uint32_t)v;
        hi_or |= (uint32_t)(v >> 32);
        mx = lo > mx ? lo : mx;
        dst[i] = lo;
    }
    return hi_or ? 0xFFFFFFFFu : mx;
}
NTT_HOST_SIMD void widen_u32(const uint32_t* src, uint64_t* dst, size_t count) {
    for (size_t i = 0; i < count; ++i) dst[i] = src[i];
}

// 64-bit host arrays whose values all lie below a modulus p < 2^31: narrowed to 32 bits while being staged
// (half the bytes over PCIe, and the 32-bit kernels' loads), widened again after the copy back.
// *narrowed = false (nothing enqueued) when a value >= p turns up or the call is too large to stage: the
// caller then takes the 64-bit path, which reduces arbitrary inputs.
int run_host_shard_narrow(Device* dev, const uint64_t* a, const uint64_t* b, uint64_t* ab, int n, size_t batch, uint32_t p,
                          FinishList& finish, bool* narrowed);

template <typename T, typename F>
int run_host_sharded(const T* a, const T* b, T* ab, int n, size_t batch, int n_gpus, F&& run, uint64_t p, uint32_t narrow_p = 0) {
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    int G = g_visible;
    if (n_gpus > 0 && n_gpus < G) G = n_gpus;
    if ((size_t)G > batch) G = batch ? (int)batch : 1;
    int home = 0;
    HIP_TRY(hipGetDevice(&home));
    if (G == 1 && batch * ((size_t)n * 2 + (size_t)(2 * n - 1)) * sizeof(T) > stage_limit_bytes()) {
        Device* dev = nullptr;
        if ((rc = get_device(home, &dev)) != NTT_OK) return rc;
        return run_host_pipelined(dev, a, b, ab, n, p, batch, run);
    }
    const size_t per = (batch + G - 1) / G;
    // rows [lo, hi) on one device: staged copy in, product, copy out enqueued on the device's stream
    auto shard_one = [&](Device* dev, size_t lo, size_t hi, FinishList& finish) -> int {
        bool narrowed = false;
        if constexpr (sizeof(T) == 8) {
            if (narrow_p) {
                int r = run_host_shard_narrow(dev, (const uint64_t*)a + lo * n, (const uint64_t*)b + lo * n,
                                              (uint64_t*)ab + lo * (size_t)(2 * n - 1), n, hi - lo, narrow_p, finish, &narrowed);
                if (r != NTT_OK) return r;
            }
        }
        if (narrowed) return NTT_OK;
        return run_host_shard(dev, a + lo * n, b + lo * n, ab + lo * (size_t)(2 * n - 1), n, hi - lo, run, finish);
    };
    auto sync_dev = [](Device* dev) -> int {
        hipError_t e = hipSetDevice(dev->id);
        if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
        return e == hipSuccess ? NTT_OK : fail(NTT_ERR_DEVICE, std::string("stream sync: ") + hipGetErrorString(e));
    };
    if (G == 1) {  // on the caller's current device, staging copies spread over the copy pool
        Device* dev = nullptr;
        if ((rc = get_device(home, &dev)) != NTT_OK) return rc;
        FinishList finish;
        rc = shard_one(dev, 0, batch, finish);
        const int rc2 = sync_dev(dev);   // also after an error: nothing may still be in flight towards the caller's memory
        hipError_t e0 = hipSetDevice(home);
        (void)e0;
        if (rc == NTT_OK && rc2 == NTT_OK)
            for (auto& f : finish) f();
        return rc != NTT_OK ? rc : rc2;
    }
    // Several devices (0..G-1): one host thread per device stages, launches, waits and copies back its own shard, so
    // that the host copies and the PCIe transfers of all devices overlap -- the per-prime parallel region of the
    // reference (main_openMP.cc:394-397) with devices in the place of thread teams.
    std::vector<Device*> devs((size_t)G, nullptr);
    for (int g = 0; g < G; ++g)
        if ((rc = get_device(g, &devs[(size_t)g])) != NTT_OK) return rc;   // created here: the device table is not shared state of the threads
    struct Outcome { int rc = NTT_OK; std::string err; };
    std::vector<Outcome> outcome((size_t)G);
    std::vector<std::thread> threads;
    for (int g = 0; g < G; ++g) {
        const size_t lo = (size_t)g * per, hi = lo + per < batch ? lo + per : batch;
        if (lo >= hi) break;
        threads.emplace_back([&, g, lo, hi] {
            t_inline_copies = true;
            FinishList finish;
            int r = shard_one(devs[(size_t)g], lo, hi, finish);
            const int r2 = sync_dev(devs[(size_t)g]);
            if (r == NTT_OK) r = r2;
            if (r == NTT_OK)
                for (auto& f : finish) f();
            outcome[(size_t)g].rc = r;
            if (r != NTT_OK) outcome[(size_t)g].err = g_err;   // the message lives in this thread's slot
        });
    }
    for (auto& t : threads) t.join();
    hipError_t e0 = hipSetDevice(home);
    (void)e0;
    for (const Outcome& o : outcome)
        if (o.rc != NTT_OK) return fail(o.rc, o.err);
    return NTT_OK;
}

int run_host_shard_narrow(Device* dev, const uint64_t* a, const uint64_t* b, uint64_t* ab, int n, size_t batch, uint32_t p,
                          FinishList& finish, bool* narrowed) {
    *narrowed = false;
    if (batch == 0) return NTT_OK;
    const size_t n_in = batch * (size_t)n, n_out = batch * (size_t)(2 * n - 1);
    const size_t in_bytes = n_in * sizeof(uint32_t), out_bytes = n_out * sizeof(uint32_t);
    if (2 * in_bytes + out_bytes > stage_limit_bytes()) return NTT_OK;
    HIP_TRY(hipSetDevice(dev->id));
    int rc;
    if ((rc = dev->pin.ensure(2 * in_bytes + out_bytes)) != NTT_OK) return rc;
    uint32_t* pa = (uint32_t*)dev->pin.ptr;
    uint32_t* pb = pa + n_in;
    uint32_t* po = pb + n_in;
    CopyPool& pool = CopyPool::get();
    std::vector<uint32_t> part_max((size_t)pool.max_parts() * 2, 0u);
    pool.for_range(n_in, (size_t)64 << 10, [&](int i, size_t lo, size_t hi) {
        part_max[2 * i] = narrow_u64(a + lo, pa + lo, hi - lo);
        part_max[2 * i + 1] = narrow_u64(b + lo, pb + lo, hi - lo);
    });
    uint32_t mx = 0;
    for (uint32_t v : part_max) mx = v > mx ? v : mx;
    if (mx >= p) return NTT_OK;  // unreduced input: the 64-bit loads reduce it
    if ((rc = dev->io_a.ensure(in_bytes)) != NTT_OK) return rc;
    if ((rc = dev->io_b.ensure(in_bytes)) != NTT_OK) return rc;
    if ((rc = dev->io_out.ensure(out_bytes)) != NTT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(dev->io_a.ptr, pa, in_bytes, hipMemcpyHostToDevice, dev->stream));
    HIP_TRY(hipMemcpyAsync(dev->io_b.ptr, pb, in_bytes, hipMemcpyHostToDevice, dev->stream));
    rc = product_device<uint32_t, uint32_t>(dev, (const uint32_t*)dev->io_a.ptr, (const uint32_t*)dev->io_b.ptr,
                                            (uint32_t*)dev->io_out.ptr, n, p, batch, (size_t)n, (size_t)n, (size_t)(2 * n - 1),
                                            nullptr, 0, dev->stream);
    if (rc != NTT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(po, dev->io_out.ptr, out_bytes, hipMemcpyDeviceToHost, dev->stream));
    finish.push_back([=] {
        CopyPool::get().for_range(n_out, (size_t)64 << 10, [=](int, size_t lo, size_t hi) { widen_u32(po + lo, ab + lo, hi - lo); });
    });
    *narrowed = true;
    return NTT_OK;
}

void die(const char* fn) {
    fprintf(stderr, "%s failed: %s\n", fn, g_err.c_str());
    abort();
}

}  // namespace

extern "C" {

int ntt_init(int n_gpus) {
    NTT_API_LOCK; return do_init(n_gpus); }

void ntt_shutdown(void) {
    NTT_API_LOCK;
    int home = -1;
    if (hipGetDevice(&home) != hipSuccess) home = -1;
    for (auto& kvd : g_devs) {
        Device* dev = kvd.second.get();
        hipError_t e = hipSetDevice(dev->id);
        (void)e;
        for (auto& kv : dev->plans) kv.second->tables.release();
        dev->plans.clear();
        for (auto& kv : dev->plans64) kv.second->tables.release();
        dev->plans64.clear();
        for (auto& kv : dev->work) kv.second.buf.release();
        dev->work.clear();
        dev->io_a.release(); dev->io_b.release(); dev->io_out.release(); dev->pin.release();
        for (auto& s : dev->stage) {
            s.pin.release(); s.a.release(); s.b.release(); s.out.release(); s.ws.release();
            if (s.stream) { e = hipStreamDestroy(s.stream); (void)e; s.stream = nullptr; }
            if (s.done) { e = hipEventDestroy(s.done); (void)e; s.done = nullptr; }
        }
        if (dev->stream) { e = hipStreamDestroy(dev->stream); (void)e; }
    }
    g_devs.clear();
    g_inited = false;
    CopyPool::shutdown();
    g_prof.clear(); g_prof_on = false;
    for (auto& kv : g_prof_pools) {
        hipError_t e = hipSetDevice(kv.first); (void)e;
        for (hipEvent_t ev : kv.second.events) { e = hipEventDestroy(ev); (void)e; }
    }
    g_prof_pools.clear();
    if (home >= 0) { hipError_t e = hipSetDevice(home); (void)e; }
}

int ntt_release_workspace(void* stream, int all_streams) {
    NTT_API_LOCK;
    if (!g_inited) return NTT_OK;
    Device* dev = nullptr;
    int rc = current_device(&dev);
    if (rc != NTT_OK) return rc;
    if (all_streams) {
        HIP_TRY(hipDeviceSynchronize());
        for (auto& kv : dev->work) kv.second.buf.release();
        dev->work.clear();
        return NTT_OK;
    }
    auto it = dev->work.find((hipStream_t)stream);
    if (it == dev->work.end()) return NTT_OK;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    it->second.buf.release();
    dev->work.erase(it);
    return NTT_OK;
}

int ntt_device_count(void) {
    NTT_API_LOCK; return g_inited ? g_visible : 0; }

const char* ntt_last_error(void) { return g_err.c_str(); }

uint32_t ntt_crt_prime(int i) {
    NTT_API_LOCK; return (i >= 0 && i < 4) ? kCrtPrimes[i] : 0; }

size_t ntt_workspace_bytes(int n, uint64_t p, size_t batch) {
    NTT_API_LOCK;
    if (n < 1) return 0;
    int logL = ceil_log2(2 * (uint64_t)n - 1);
    if (logL < 1) logL = 1;
    switch (choose_route(p, (size_t)n, nullptr)) {
        case ROUTE_DIRECT32: return pipeline_workspace_bytes(logL, batch);
        case ROUTE_CRT: return crt_workspace_bytes(logL, batch);
        case ROUTE_FIELD64: return field64_workspace_bytes(logL, batch);
        case ROUTE_WIDECRT: return wide_crt_workspace_bytes((size_t)n, p, batch);
        default: return 0;
    }
}

int ntt_poly_multiply_batched_device(const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_ab, int n, uint32_t p, size_t batch,
                                     size_t a_stride, size_t b_stride, size_t ab_stride, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    NTT_API_LOCK;
    int rc = check_args(d_a, d_b, d_ab, n, batch);
    if (rc != NTT_OK) return rc;
    if (a_stride < (size_t)n || b_stride < (size_t)n || ab_stride < (size_t)(2 * n - 1)) return fail(NTT_ERR_ARG, "row stride too small");
    if ((rc = do_init(0)) != NTT_OK) return rc;
    if (batch == 0) return NTT_OK;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    if (!direct_path_ok(p, transform_log2(n))) {
        std::string why;
        modulus_is_ntt_friendly(p, transform_log2(n), &why);
        return fail(NTT_ERR_MODULUS, "32-bit batched path needs an NTT-friendly modulus: " + why);
    }
    return product_device<uint32_t, uint32_t>(dev, d_a, d_b, d_ab, n, p, batch, a_stride, b_stride, ab_stride, workspace,
                                              workspace_bytes, (hipStream_t)stream);
}

// Same products as ntt_poly_multiply_batched_device, for callers that may not allocate device memory: every lookup below
// is a lookup only (no plan is built, no stream or buffer created), and the intermediates go to the static scratch.
int ntt_poly_multiply_batched_device_noalloc(const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_ab, int n, uint32_t p,
                                             size_t batch, size_t a_stride, size_t b_stride, size_t ab_stride, void* stream) {
    NTT_API_LOCK;
    int rc = check_args(d_a, d_b, d_ab, n, batch);
    if (rc != NTT_OK) return rc;
    if (a_stride < (size_t)n || b_stride < (size_t)n || ab_stride < (size_t)(2 * n - 1)) return fail(NTT_ERR_ARG, "row stride too small");
    if (!g_inited) return fail(NTT_ERR_DEVICE, "library not initialised: call ntt_init() before the allocation-free entry point");
    if (batch == 0) return NTT_OK;
    int id = 0;
    HIP_TRY(hipGetDevice(&id));
    auto dit = g_devs.find(id);
    if (dit == g_devs.end() || !dit->second->scratch)
        return fail(NTT_ERR_DEVICE, "this device has not been set up: call ntt_init() or any product entry point with it current first");
    Device* dev = dit->second.get();
    const int logL = transform_log2(n);
    auto pit = dev->plans.find(std::make_pair(p, logL));
    if (pit == dev->plans.end())
        return fail(NTT_ERR_ARG, "no cached tables for this (n, p) on this device: run one product with the same n and p through "
                                 "ntt_poly_multiply_batched_device first (it builds them), then call the allocation-free entry point");
    const Plan& pl = *pit->second;
    const size_t need = pipeline_workspace_bytes(logL, batch);
    if (need > NOALLOC_SCRATCH_BYTES) {
        const size_t per = pipeline_workspace_bytes(logL, 1);
        return fail(NTT_ERR_WORKSPACE, "batch too large for the static scratch of " + std::to_string(NOALLOC_SCRATCH_BYTES >> 20) +
                                       " MiB: at most " + std::to_string(NOALLOC_SCRATCH_BYTES / per) + " products of this length per call");
    }
    // the column pass writes every word of the scratch it uses before the core reads it: nothing has to be cleared
    if (pl.h.lazy) return launch_product<true>(pl, d_a, d_b, d_ab, n, batch, a_stride, b_stride, ab_stride, dev->scratch, (hipStream_t)stream);
    return launch_product<false>(pl, d_a, d_b, d_ab, n, batch, a_stride, b_stride, ab_stride, dev->scratch, (hipStream_t)stream);
}

int ntt_poly_multiply_batched_device_u64(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_ab, int n, uint64_t p,
                                         size_t batch, size_t a_stride, size_t b_stride, size_t ab_stride, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    NTT_API_LOCK;
    int rc = check_args(d_a, d_b, d_ab, n, batch);
    if (rc != NTT_OK) return rc;
    if (a_stride < (size_t)n || b_stride < (size_t)n || ab_stride < (size_t)(2 * n - 1)) return fail(NTT_ERR_ARG, "row stride too small");
    if ((rc = do_init(0)) != NTT_OK) return rc;
    if (batch == 0) return NTT_OK;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    std::string why;
    switch (choose_route(p, (size_t)n, &why)) {
        case ROUTE_DIRECT32:
            return product_device<uint64_t, uint64_t>(dev, d_a, d_b, d_ab, n, (uint32_t)p, batch, a_stride, b_stride, ab_stride,
                                                      workspace, workspace_bytes, (hipStream_t)stream);
        case ROUTE_CRT:
            return crt_product_device(dev, d_a, d_b, d_ab, n, p, batch, a_stride, b_stride, ab_stride, workspace, workspace_bytes,
                                      (hipStream_t)stream);
        case ROUTE_FIELD64:
            return field64_product_device(dev, d_a, d_b, d_ab, (size_t)n, p, batch, a_stride, b_stride, ab_stride, workspace,
                                          workspace_bytes, (hipStream_t)stream);
        case ROUTE_WIDECRT:
            return wide_crt_product_device(dev, d_a, d_b, d_ab, (size_t)n, p, batch, a_stride, b_stride, ab_stride, workspace,
                                           workspace_bytes, (hipStream_t)stream);
        default:
            return fail(NTT_ERR_MODULUS, why);
    }
}

int ntt_crt_merge_device(const uint32_t* d_res, size_t res_stride, uint64_t* d_out, size_t count, uint64_t p, void* stream) {
    NTT_API_LOCK;
    if (!d_res || !d_out) return fail(NTT_ERR_ARG, "null pointer");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    if (count == 0) return NTT_OK;
    CrtConsts c;
    if ((rc = make_crt_consts(p, &c)) != NTT_OK) return rc;
    NTT_LAUNCH(k_crt_merge, dim3((unsigned)((count + 255) / 256), 1), dim3(256), (hipStream_t)stream, d_res, res_stride, d_out,
               count, (size_t)1, count, (size_t)0, c);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

int ntt_crt_residue_product_device(const uint64_t* d_a, const uint64_t* d_b, uint32_t* d_res, int n, int prime_index,
                                   size_t batch, size_t a_stride, size_t b_stride, size_t res_stride, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    NTT_API_LOCK;
    int rc = check_args(d_a, d_b, d_res, n, batch);
    if (rc != NTT_OK) return rc;
    if (prime_index < 0 || prime_index > 3) return fail(NTT_ERR_ARG, "prime_index must be 0..3");
    if (a_stride < (size_t)n || b_stride < (size_t)n || res_stride < (size_t)(2 * n - 1)) return fail(NTT_ERR_ARG, "row stride too small");
    if ((rc = do_init(0)) != NTT_OK) return rc;
    if (batch == 0) return NTT_OK;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    return product_device<uint64_t, uint32_t>(dev, d_a, d_b, d_res, n, kCrtPrimes[prime_index], batch, a_stride, b_stride,
                                              res_stride, workspace, workspace_bytes, (hipStream_t)stream);
}

int ntt_poly_multiply_batched(const uint32_t* a, const uint32_t* b, uint32_t* ab, int n, uint32_t p, size_t batch, int n_gpus) {
    NTT_API_LOCK;
    int rc = check_args(a, b, ab, n, batch);
    if (rc != NTT_OK) return rc;
    if (!direct_path_ok(p, transform_log2(n))) {
        std::string why;
        modulus_is_ntt_friendly(p, transform_log2(n), &why);
        return fail(NTT_ERR_MODULUS, "32-bit batched path needs an NTT-friendly modulus: " + why);
    }
    return run_host_sharded<uint32_t>(a, b, ab, n, batch, n_gpus,
        [&](Device* dev, const uint32_t* da, const uint32_t* db, uint32_t* dout, size_t shard, void* ws, size_t ws_bytes,
            hipStream_t st) {
            return product_device<uint32_t, uint32_t>(dev, da, db, dout, n, p, shard, (size_t)n, (size_t)n,
                                                      (size_t)(2 * n - 1), ws, ws_bytes, st);
        }, (uint64_t)p);
}

int ntt_poly_multiply_batched_u64(const uint64_t* a, const uint64_t* b, uint64_t* ab, int n, uint64_t p, size_t batch, int n_gpus) {
    NTT_API_LOCK;
    int rc = check_args(a, b, ab, n, batch);
    if (rc != NTT_OK) return rc;
    std::string why;
    const Route route = choose_route(p, (size_t)n, &why);
    if (route == ROUTE_NONE) return fail(NTT_ERR_MODULUS, why);
    return run_host_sharded<uint64_t>(a, b, ab, n, batch, n_gpus,
        [&](Device* dev, const uint64_t* da, const uint64_t* db, uint64_t* dout, size_t shard, void* ws, size_t ws_bytes,
            hipStream_t st) {
            const size_t sn = (size_t)n, so = (size_t)(2 * n - 1);
            if (route == ROUTE_DIRECT32)
                return product_device<uint64_t, uint64_t>(dev, da, db, dout, n, (uint32_t)p, shard, sn, sn, so, ws, ws_bytes, st);
            if (route == ROUTE_CRT) return crt_product_device(dev, da, db, dout, n, p, shard, sn, sn, so, ws, ws_bytes, st);
            if (route == ROUTE_WIDECRT) return wide_crt_product_device(dev, da, db, dout, sn, p, shard, sn, sn, so, ws, ws_bytes, st);
            return field64_product_device(dev, da, db, dout, sn, p, shard, sn, sn, so, ws, ws_bytes, st);
        }, p, route == ROUTE_DIRECT32 ? (uint32_t)p : 0u);
}

int ntt_poly_multiply_u64(const uint64_t* a, const uint64_t* b, uint64_t* ab, int n, uint64_t p) {
    NTT_API_LOCK;
    return ntt_poly_multiply_batched_u64(a, b, ab, n, p, 1, 1);
}

int ntt_poly_multiply_i32(const int* a, const int* b, int* ab, int n, int p) {
    NTT_API_LOCK;
    if (p < 2) return fail(NTT_ERR_ARG, "modulus must be positive");
    int rc = check_args(a, b, ab, n, 1);
    if (rc != NTT_OK) return rc;
    if (direct_path_ok((uint64_t)p, transform_log2(n)))
        return ntt_poly_multiply_batched((const uint32_t*)a, (const uint32_t*)b, (uint32_t*)ab, n, (uint32_t)p, 1, 1);
    // not NTT-friendly: widen and take the CRT route
    std::vector<uint64_t> wa(n), wb(n), wo(2 * (size_t)n - 1);
    for (int i = 0; i < n; ++i) { wa[i] = (uint32_t)a[i]; wb[i] = (uint32_t)b[i]; }
    rc = ntt_poly_multiply_batched_u64(wa.data(), wb.data(), wo.data(), n, (uint64_t)p, 1, 1);
    if (rc != NTT_OK) return rc;
    for (size_t i = 0; i < wo.size(); ++i) ab[i] = (int)wo[i];
    return NTT_OK;
}

void poly_multiply(const int* a, const int* b, int* ab, int n, int p) {
    NTT_API_LOCK;
    if (ntt_poly_multiply_i32(a, b, ab, n, p) != NTT_OK) die("poly_multiply");
}

void poly_multiply_u64(const uint64_t* a, const uint64_t* b, uint64_t* ab, int n, uint64_t p) {
    NTT_API_LOCK;
    if (ntt_poly_multiply_u64(a, b, ab, n, p) != NTT_OK) die("poly_multiply_u64");
}

// ---- 64-bit field entry points -------------------------------------------------------------
int ntt64_transform_device(uint64_t* d_data, uint64_t p, int log_root, int log_len, size_t inner, size_t outer, int inverse,
                           void* stream) {
    NTT_API_LOCK;
    if (!d_data) return fail(NTT_ERR_ARG, "null pointer");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_root, &pl)) != NTT_OK) return rc;
    return transform64(*pl, d_data, log_len, inner, outer, inverse != 0, (hipStream_t)stream);
}

int ntt64_fourstep_rows_device(uint64_t* d_slab, uint64_t p, int log_total, int log_rows, size_t cols_local, uint64_t col0,
                               int inverse, void* stream) {
    NTT_API_LOCK;
    if (!d_slab) return fail(NTT_ERR_ARG, "null pointer");
    if (log_rows < 0 || log_rows > log_total || cols_local == 0 || !is_pow2(cols_local)) return fail(NTT_ERR_ARG, "slab shape: 0 <= log_rows <= log_total, cols_local a power of two");
    const int log_cols = log_total - log_rows;
    if (col0 + cols_local > ((uint64_t)1 << log_cols)) return fail(NTT_ERR_ARG, "slab columns exceed the matrix");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_total, &pl)) != NTT_OK) return rc;
    Axis64 ax;
    ax.log_len = log_rows; ax.inner = cols_local; ax.outer = 1; ax.col_bits = log_cols; ax.col0 = col0;
    if (log_cols == 0) { ax.col_bits = 0; ax.col0 = 0; }   // a matrix of one column is a plain transform
    return transform64_axis(*pl, d_slab, ax, inverse != 0, (hipStream_t)stream);
}

int ntt64_fourstep_cols_device(uint64_t* d_rows, uint64_t p, int log_total, int log_cols, int log_parts, size_t part_stride,
                               size_t rows, int inverse, void* stream) {
    NTT_API_LOCK;
    if (!d_rows) return fail(NTT_ERR_ARG, "null pointer");
    if (log_cols < 0 || log_cols > log_total || log_parts < 0 || log_parts > log_cols) return fail(NTT_ERR_ARG, "bad shape");
    if (log_parts > 0 && part_stride < (rows << (log_cols - log_parts))) return fail(NTT_ERR_ARG, "part stride smaller than a part");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_total, &pl)) != NTT_OK) return rc;
    return transform64_split(*pl, d_rows, log_cols, log_parts, (uint64_t)part_stride, (uint64_t)rows, inverse != 0, (hipStream_t)stream);
}

namespace {
// shared argument checks and plan lookup of the four-step row-axis entry points
int fourstep_rows_setup(const void* d_slab, uint64_t p, int log_total, int log_rows, size_t cols_local, uint64_t col0, Plan64** pl,
                        Axis64* ax) {
    if (!d_slab) return fail(NTT_ERR_ARG, "null pointer");
    if (log_rows < 0 || log_rows > log_total || cols_local == 0 || !is_pow2(cols_local)) return fail(NTT_ERR_ARG, "slab shape: 0 <= log_rows <= log_total, cols_local a power of two");
    const int log_cols = log_total - log_rows;
    if (col0 + cols_local > ((uint64_t)1 << log_cols)) return fail(NTT_ERR_ARG, "slab columns exceed the matrix");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    if ((rc = get_plan64(dev, p, log_total, pl)) != NTT_OK) return rc;
    ax->log_len = log_rows; ax->inner = cols_local; ax->outer = 1; ax->col_bits = log_cols; ax->col0 = col0;
    if (log_cols == 0) { ax->col_bits = 0; ax->col0 = 0; }   // a matrix of one column is a plain transform
    return NTT_OK;
}
}  // namespace

int ntt64_fourstep_rows_load_device(const uint64_t* d_src, size_t rows_valid, uint64_t* d_slab, uint64_t p, int log_total, int log_rows,
                                    size_t cols_local, uint64_t col0, void* stream) {
    NTT_API_LOCK;
    if (!d_src) return fail(NTT_ERR_ARG, "null pointer");
    Plan64* pl = nullptr;
    Axis64 ax;
    int rc = fourstep_rows_setup(d_slab, p, log_total, log_rows, cols_local, col0, &pl, &ax);
    if (rc != NTT_OK) return rc;
    const uint64_t rows = (uint64_t)1 << log_rows;
    if (rows_valid > rows) return fail(NTT_ERR_ARG, "more valid rows than the slab has");
    hipStream_t st = (hipStream_t)stream;
    TileExtra io;
    io.src = d_src;
    io.src_valid = (uint64_t)rows_valid * cols_local;   // row-major slab: the rows below rows_valid are its first elements
    if (!outer_pass_takes_io(ax)) {
        load64(*pl, d_src, 0, io.src_valid, d_slab, rows * cols_local, 1, st);
        HIP_TRY(hipGetLastError());
        if (log_rows == 0) return NTT_OK;
        return transform64_axis(*pl, d_slab, ax, false, st);
    }
    return pl->fp ? transform64_io<FieldD>(*pl, d_slab, ax, false, io, st) : transform64_io<Field64>(*pl, d_slab, ax, false, io, st);
}

int ntt64_fourstep_rows_store_device(uint64_t* d_slab, uint64_t* d_out, uint64_t p, int log_total, int log_rows, size_t cols_local,
                                     uint64_t col0, void* stream) {
    NTT_API_LOCK;
    if (!d_out) return fail(NTT_ERR_ARG, "null pointer");
    Plan64* pl = nullptr;
    Axis64 ax;
    int rc = fourstep_rows_setup(d_slab, p, log_total, log_rows, cols_local, col0, &pl, &ax);
    if (rc != NTT_OK) return rc;
    const uint64_t count = ((uint64_t)1 << log_rows) * cols_local;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t Linv = h_powmod((1ull << log_total) % p, p - 2, p);
    const uint64_t scale = h_mulmod(Linv, pointwise_compensation(*pl), p);
    if (!outer_pass_takes_io(ax)) {
        if (log_rows > 0 && (rc = transform64_axis(*pl, d_slab, ax, true, st)) != NTT_OK) return rc;
        finalize64(*pl, d_slab, count, d_out, 0, count, 1, scale, st);
        HIP_TRY(hipGetLastError());
        return NTT_OK;
    }
    TileExtra io;
    io.out = d_out;
    io.n_out = count;
    io.scale_word = tw_word(*pl, scale);
    return pl->fp ? transform64_io<FieldD>(*pl, d_slab, ax, true, io, st) : transform64_io<Field64>(*pl, d_slab, ax, true, io, st);
}

int ntt64_fourstep_cols_product_device(uint64_t* d_a_rows, uint64_t* d_b_rows, uint64_t p, int log_total, int log_cols, int log_parts,
                                       size_t part_stride, size_t rows, void* stream) {
    NTT_API_LOCK;
    if (!d_a_rows || !d_b_rows) return fail(NTT_ERR_ARG, "null pointer");
    if (log_cols < 0 || log_cols > log_total || log_parts < 0 || log_parts > log_cols || log_parts > 4) return fail(NTT_ERR_ARG, "bad shape");
    if (log_parts > 0 && part_stride < (rows << (log_cols - log_parts))) return fail(NTT_ERR_ARG, "part stride smaller than a part");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_total, &pl)) != NTT_OK) return rc;
    if (rows == 0) return NTT_OK;
    return pl->fp ? product64_split_f<FieldD>(*pl, d_a_rows, d_b_rows, log_cols, log_parts, (uint64_t)part_stride, (uint64_t)rows, (hipStream_t)stream)
                  : product64_split_f<Field64>(*pl, d_a_rows, d_b_rows, log_cols, log_parts, (uint64_t)part_stride, (uint64_t)rows, (hipStream_t)stream);
}

int ntt64_twiddle_device(uint64_t* d_data, uint64_t p, int log_root, size_t rows, size_t cols, uint64_t row0, uint64_t col0,
                         int row_bits, int rows_bitrev, int inverse, void* stream) {
    NTT_API_LOCK;
    if (!d_data) return fail(NTT_ERR_ARG, "null pointer");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_root, &pl)) != NTT_OK) return rc;
    if (rows == 0 || cols == 0) return NTT_OK;
    twiddle64(*pl, d_data, (uint64_t)rows, (uint64_t)cols, row0, col0, row_bits, rows_bitrev, inverse, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

int ntt64_pointwise_device(uint64_t* d_a, const uint64_t* d_b, size_t count, uint64_t p, int log_root, void* stream) {
    NTT_API_LOCK;
    if (!d_a || !d_b) return fail(NTT_ERR_ARG, "null pointer");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_root, &pl)) != NTT_OK) return rc;
    if (count == 0) return NTT_OK;
    pointwise64(*pl, d_a, d_b, (uint64_t)count, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

int ntt64_load_device(const uint64_t* d_src, size_t src_stride, size_t n, uint64_t* d_dst, size_t len, size_t rows, uint64_t p,
                      int log_root, void* stream) {
    NTT_API_LOCK;
    if (!d_src || !d_dst) return fail(NTT_ERR_ARG, "null pointer");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_root, &pl)) != NTT_OK) return rc;
    if (rows == 0 || len == 0) return NTT_OK;
    load64(*pl, d_src, src_stride, (uint64_t)n, d_dst, (uint64_t)len, (uint64_t)rows, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

int ntt64_finalize_device(const uint64_t* d_data, size_t len, uint64_t* d_out, size_t out_stride, size_t n_out, size_t rows,
                          uint64_t p, int log_root, int log_total, void* stream) {
    NTT_API_LOCK;
    if (!d_data || !d_out) return fail(NTT_ERR_ARG, "null pointer");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, log_root, &pl)) != NTT_OK) return rc;
    if (rows == 0 || n_out == 0) return NTT_OK;
    // scale = L^-1 * R (the R undoes the R^-1 left by the pointwise product), in Montgomery form
    const uint64_t Linv = h_powmod((1ull << log_total) % p, p - 2, p);
    finalize64(*pl, d_data, (uint64_t)len, d_out, out_stride, (uint64_t)n_out, (uint64_t)rows,
               h_mulmod(Linv, pointwise_compensation(*pl), p), (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

size_t ntt64_workspace_bytes(size_t n) {
    NTT_API_LOCK;
    if (n < 1) return 0;
    const int logL = ceil_log2(2 * (uint64_t)n - 1) < 1 ? 1 : ceil_log2(2 * (uint64_t)n - 1);
    return 2 * ((size_t)1 << logL) * sizeof(uint64_t);
}

int ntt64_poly_multiply_device(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n, uint64_t p, void* workspace,
                               size_t workspace_bytes, void* stream) {
    NTT_API_LOCK;
    if (!d_a || !d_b || !d_out) return fail(NTT_ERR_ARG, "null pointer");
    if (n < 1 || n > ((size_t)1 << 34)) return fail(NTT_ERR_ARG, "n out of range");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    return field64_product_device(dev, d_a, d_b, d_out, n, p, 1, n, n, 2 * n - 1, workspace, workspace_bytes, (hipStream_t)stream);
}

int ntt_set_route_policy(int policy) {
    if (policy != NTT_ROUTE_AUTO && policy != NTT_ROUTE_REFERENCE) { fail(NTT_ERR_ARG, "unknown route policy"); return -1; }
    return g_route_policy.exchange(policy);
}

size_t ntt64_workspace_bytes_batched(size_t n, size_t batch) {
    NTT_API_LOCK;
    if (n < 1) return 0;
    const int logL = ceil_log2(2 * (uint64_t)n - 1) < 1 ? 1 : ceil_log2(2 * (uint64_t)n - 1);
    return field64_workspace_bytes(logL, batch);
}

int ntt64_poly_multiply_batched_device(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n, uint64_t p, size_t batch,
                                       size_t a_stride, size_t b_stride, size_t out_stride, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    NTT_API_LOCK;
    if (batch == 0) return NTT_OK;
    if (!d_a || !d_b || !d_out) return fail(NTT_ERR_ARG, "null pointer");
    if (n < 1 || n > ((size_t)1 << 34)) return fail(NTT_ERR_ARG, "n out of range");
    if (a_stride < n || b_stride < n || out_stride < 2 * n - 1) return fail(NTT_ERR_ARG, "row strides shorter than the rows");
    if (batch > ((size_t)1 << 24)) return fail(NTT_ERR_ARG, "batch too large");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    return field64_product_device(dev, d_a, d_b, d_out, n, p, batch, a_stride, b_stride, out_stride, workspace, workspace_bytes,
                                  (hipStream_t)stream);
}

}  // extern "C"

namespace {

// Natural-order transform on the 32-bit kernels (ntt_transform32.h), in place on rows of n = 2^lg values of type T.
// *served = false (and nothing launched) when this modulus / length belongs to the 64-bit field instead.
template <int Q, bool LZ, typename T>
void launch_transform32_columns(const Plan& pl, T* d_a, uint32_t* work, size_t n, size_t batch, bool invert, hipStream_t st) {
    constexpr int MINW = multipass_minw(Q);
    const unsigned chunks = CORE_N / 256;
    if (!invert)
        NTT_LAUNCH_P("k_fwd_strided", (k_fwd_strided<Q, LZ, false, false, MINW, T>), dim3((unsigned)(chunks * batch), 1, 1), dim3(256), st,
                     (const T*)d_a, (const T*)d_a, n, n, (int)n, work, work, CORE_N, pl.T1, pl.h.fc);
    else
        NTT_LAUNCH_P("k_inv_strided", (k_inv_strided<Q, LZ, true, false, MINW, T>), dim3((unsigned)(chunks * batch)), dim3(256), st,
                     (const uint32_t*)work, d_a, n, (int)n, CORE_N, pl.T3, pl.h.fc);
}

template <bool LZ, typename T>
int transform32_launch(Device* dev, const Plan& pl, T* d_a, int lg, size_t batch, bool invert, hipStream_t st) {
    const size_t n = (size_t)1 << lg;
    if (lg <= 12) {
        if (pl.h.small16) {
            const size_t per_wg = (size_t)1 << (12 - lg);
            const unsigned grid = (unsigned)((batch + per_wg - 1) / per_wg);
            if (!invert) NTT_LAUNCH_P("k_ntt_small16", (k_ntt_small16<LZ, false, T>), dim3(grid), dim3(CORE_T), st, d_a, lg, batch, pl.h.scale_t, pl.h.fc, pl.rt);
            else NTT_LAUNCH_P("k_ntt_small16", (k_ntt_small16<LZ, true, T>), dim3(grid), dim3(CORE_T), st, d_a, lg, batch, pl.h.scale_t, pl.h.fc, pl.rt);
        } else {
            const unsigned threads = n / 2 < 64 ? 64u : (n / 2 > 256 ? 256u : (unsigned)(n / 2));
            if (!invert) NTT_LAUNCH_P("k_ntt_small2", (k_ntt_small2<LZ, false, T>), dim3((unsigned)batch), dim3(threads), st, d_a, lg, pl.twF, pl.h.scale_t, pl.h.fc);
            else NTT_LAUNCH_P("k_ntt_small2", (k_ntt_small2<LZ, true, T>), dim3((unsigned)batch), dim3(threads), st, d_a, lg, pl.twI, pl.h.scale_t, pl.h.fc);
        }
        HIP_TRY(hipGetLastError());
        return NTT_OK;
    }
    // longer transforms: three passes through a 32-bit work array of the same shape
    void* ws = nullptr;
    int rc = dev->workspace_for(st, batch * n * sizeof(uint32_t), &ws);
    if (rc != NTT_OK) return rc;
    uint32_t* work = (uint32_t*)ws;
    const int Q = lg - 12;
    const unsigned rows = (unsigned)(batch << Q);
    auto columns = [&]() {
        switch (Q) {
            case 1:
                if (!invert) NTT_LAUNCH_P("k_ntt_outer2", (k_ntt_outer2<LZ, false, T>), dim3((unsigned)(16 * batch)), dim3(256), st, work, d_a, pl.ow, pl.h.fc);
                else NTT_LAUNCH_P("k_ntt_outer2", (k_ntt_outer2<LZ, true, T>), dim3((unsigned)(16 * batch)), dim3(256), st, work, d_a, pl.ow, pl.h.fc);
                break;
            case 2: launch_transform32_columns<2, LZ, T>(pl, d_a, work, n, batch, invert, st); break;
            case 3: launch_transform32_columns<3, LZ, T>(pl, d_a, work, n, batch, invert, st); break;
            case 4: launch_transform32_columns<4, LZ, T>(pl, d_a, work, n, batch, invert, st); break;
            case 5: launch_transform32_columns<5, LZ, T>(pl, d_a, work, n, batch, invert, st); break;
            default: launch_transform32_columns<6, LZ, T>(pl, d_a, work, n, batch, invert, st); break;
        }
    };
    if (!invert) {
        columns();
        NTT_LAUNCH_P("k_ntt_core_fwd", (k_ntt_core_fwd<LZ>), dim3(rows), dim3(CORE_T), st, work, pl.h.fc, pl.core);
        NTT_LAUNCH_P("k_ntt_transpose", (k_ntt_transpose<LZ, false, T>), dim3(rows), dim3(256), st, work, d_a, Q, pl.h.fc);
    } else {
        NTT_LAUNCH_P("k_ntt_transpose", (k_ntt_transpose<LZ, true, T>), dim3(rows), dim3(256), st, work, d_a, Q, pl.h.fc);
        NTT_LAUNCH_P("k_ntt_core_inv", (k_ntt_core_inv<LZ>), dim3(rows), dim3(CORE_T), st, work, pl.h.fc, pl.core_t);
        columns();
    }
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

template <typename T>
int transform32_device(Device* dev, T* d_a, size_t n, size_t batch, bool invert, uint64_t p, hipStream_t st, bool* served) {
    *served = false;
    const int lg = ceil_log2(n);
    if (lg < 1 || lg > 18 || p >= (1ull << 31) || !direct_path_ok(p, lg)) return NTT_OK;
    if (batch > ((size_t)1 << 24)) return NTT_OK;
    Plan* pl = nullptr;
    int rc = get_plan(dev, (uint32_t)p, lg, &pl);
    if (rc != NTT_OK) return rc;
    *served = true;
    return pl->h.lazy ? transform32_launch<true, T>(dev, *pl, d_a, lg, batch, invert, st)
                      : transform32_launch<false, T>(dev, *pl, d_a, lg, batch, invert, st);
}

}  // namespace

extern "C" {

int ntt_transform_device_u32(uint32_t* d_a, size_t n, size_t batch, int invert, uint32_t p, void* stream) {
    NTT_API_LOCK;
    if (!d_a) return fail(NTT_ERR_ARG, "null pointer");
    if (n < 1 || (n & (n - 1)) != 0) return fail(NTT_ERR_ARG, "transform length must be a power of two");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    if (batch == 0 || n == 1) return NTT_OK;       // a length-1 transform is the identity (values are taken as reduced)
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    bool served = false;
    if ((rc = transform32_device<uint32_t>(dev, d_a, n, batch, invert != 0, p, (hipStream_t)stream, &served)) != NTT_OK) return rc;
    if (!served) {
        std::string why;
        if (n > ((size_t)1 << 18)) why = "32-bit transforms go up to 2^18 points (ntt_transform_device_u64 serves longer ones)";
        else modulus_is_ntt_friendly(p, ceil_log2(n), &why);
        return fail(NTT_ERR_MODULUS, "32-bit transform: " + why);
    }
    return NTT_OK;
}

// ---- stand-alone transforms in natural order (the reference's NTT(), main_naive.cc:96-137) ----
int ntt_transform_device_u64(uint64_t* d_a, size_t n, size_t batch, int invert, uint64_t p, void* stream) {
    NTT_API_LOCK;
    if (!d_a) return fail(NTT_ERR_ARG, "null pointer");
    if (n < 1 || (n & (n - 1)) != 0) return fail(NTT_ERR_ARG, "transform length must be a power of two");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    if (batch == 0 || n == 1) {
        if (n == 1 && batch) {  // length-1 transform is the identity: only canonicalise
            Device* dev1 = nullptr;
            if ((rc = current_device(&dev1)) != NTT_OK) return rc;
            Plan64* pl1 = nullptr;
            if ((rc = get_plan64(dev1, p, 1, &pl1)) != NTT_OK) return rc;
            load64(*pl1, (const uint64_t*)d_a, (size_t)1, (uint64_t)1, d_a, (uint64_t)1, (uint64_t)batch, (hipStream_t)stream);
            HIP_TRY(hipGetLastError());
        }
        return NTT_OK;
    }
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    const int lg = ceil_log2(n);
    hipStream_t st = (hipStream_t)stream;
    if (transform32_allowed()) {      // moduli below 2^31: the 32-bit kernels, 8-byte words only at the two ends
        bool served = false;
        if ((rc = transform32_device<uint64_t>(dev, d_a, n, batch, invert != 0, p, st, &served)) != NTT_OK || served) return rc;
    }
    Plan64* pl = nullptr;
    if ((rc = get_plan64(dev, p, lg, &pl)) != NTT_OK) return rc;
    const uint64_t total = (uint64_t)batch * n;
    load64(*pl, (const uint64_t*)d_a, (size_t)n, (uint64_t)n, d_a, (uint64_t)n, (uint64_t)batch, st);
    uint64_t scale = 1;
    if (!invert) {
        if ((rc = transform64(*pl, d_a, lg, 1, batch, false, st)) != NTT_OK) return rc;
        NTT_LAUNCH_P("k64_bitrev", k64_bitrev, dim3(grid_for(total)), dim3(256), st, d_a, lg, (uint64_t)batch);
    } else {
        NTT_LAUNCH_P("k64_bitrev", k64_bitrev, dim3(grid_for(total)), dim3(256), st, d_a, lg, (uint64_t)batch);
        if ((rc = transform64(*pl, d_a, lg, 1, batch, true, st)) != NTT_OK) return rc;
        scale = h_powmod((uint64_t)n % p, p - 2, p);   // n^-1, as main_naive.cc:132
    }
    finalize64(*pl, (const uint64_t*)d_a, (uint64_t)n, d_a, (size_t)n, (uint64_t)n, (uint64_t)batch, scale, st);
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}

int ntt_transform_u64(uint64_t* a, int n, int invert, uint64_t p) {
    NTT_API_LOCK;
    if (!a || n < 1) return fail(NTT_ERR_ARG, "bad argument");
    int rc = do_init(0);
    if (rc != NTT_OK) return rc;
    Device* dev = nullptr;
    if ((rc = current_device(&dev)) != NTT_OK) return rc;
    const size_t bytes = (size_t)n * sizeof(uint64_t);
    if ((rc = dev->io_a.ensure(bytes)) != NTT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(dev->io_a.ptr, a, bytes, hipMemcpyHostToDevice, dev->stream));
    if ((rc = ntt_transform_device_u64((uint64_t*)dev->io_a.ptr, (size_t)n, 1, invert, p, dev->stream)) != NTT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(a, dev->io_a.ptr, bytes, hipMemcpyDeviceToHost, dev->stream));
    HIP_TRY(hipStreamSynchronize(dev->stream));
    return NTT_OK;
}

int ntt_profile_enable(int on) {
    NTT_API_LOCK;
    g_prof.clear();
    for (auto& kv : g_prof_pools) kv.second.used = 0;   // the pools' events are reused by the next session
    g_prof_on = on != 0;
    return NTT_OK;
}

int ntt_profile_read(char* names, size_t names_len, double* total_ms, int* launches, int max_kernels) {
    NTT_API_LOCK;
    if (!names || !total_ms || !launches || max_kernels <= 0) return fail(NTT_ERR_ARG, "bad argument");
    std::vector<std::string> labels;
    std::string joined;
    for (auto& r : g_prof) {
        if (!r.closed) continue;
        HIP_TRY(hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.e0, r.e1));
        size_t k = 0;
        while (k < labels.size() && labels[k] != r.label) ++k;
        if (k == labels.size()) {
            if ((int)k >= max_kernels) continue;
            labels.push_back(r.label); total_ms[k] = 0; launches[k] = 0;
        }
        total_ms[k] += ms; launches[k] += 1;
    }
    for (size_t k = 0; k < labels.size(); ++k) joined += (k ? "," : "") + labels[k];
    snprintf(names, names_len, "%s", joined.c_str());
    return (int)labels.size();
}

int ntt_describe_plan(int n, uint64_t p, char* buf, size_t buf_len) {
    NTT_API_LOCK;
    if (!buf || buf_len == 0 || n < 1) return fail(NTT_ERR_ARG, "bad argument");
    const int logL = transform_log2(n);
    std::string s, why;
    const Route route = choose_route(p, (size_t)n, &why);
    const char* pipe = logL <= 12 ? ((logL >= 5 && p < (1ull << 31) && (p - 1) % 64 == 0) ? "k_small16" : "k_small") : logL == 13 ? "k_fused8192" : "k_fwd_strided,k_core_mul,k_inv_strided";
    if (route == ROUTE_DIRECT32) s = pipe;
    else if (route == ROUTE_CRT) {
        // the products run modulo the four primes of the basis (64 divides every p_i - 1), not modulo p
        const char* per_prime = logL <= 12 ? (logL >= 5 ? "k_small16" : "k_small") : logL == 13 ? "k_fused8192" : "k_fwd_strided,k_core_mul,k_inv_strided";
        s = (logL >= 14 && crt_fused_merge_allowed()) ? std::string("k_fwd_strided_crt,k_core_mul_crt,k_inv_strided_crt")
                                                      : std::string("4x(") + per_prime + "),k_crt_merge";
    }
    else if (route == ROUTE_FIELD64 || route == ROUTE_WIDECRT) {
        // the launches of one product in the 64-bit field, in order of first use
        const std::vector<Step64> steps = plan_steps(logL, 1, 1);
        const bool fused = steps.size() >= 2 && steps.front().kind != Step64::SMALL && steps.back().kind == Step64::LOCAL;
        const bool core = fused && steps.back().q == LOCAL64_LOG && fused_core_allowed();
        std::vector<std::string> names;
        auto add = [&](const char* k) { if (std::find(names.begin(), names.end(), k) == names.end()) names.push_back(k); };
        if (logL <= LOCAL64_LOG && fused_core_allowed()) add("k64_short_mul");
        else {
        if (!fused) add("k64_load");
        for (const Step64& st : steps)
            add(st.kind == Step64::TILE ? "k64_tile" : st.kind == Step64::REG ? "k64_strided" : st.kind == Step64::SMALL ? "k64_small"
                : core ? "k64_core_mul" : "k64_local");
        if (!fused) { add("k64_pointwise"); add("k64_finalize"); }
        }
        if (route == ROUTE_WIDECRT) add("k64_garner");
        for (size_t i = 0; i < names.size(); ++i) s += (i ? "," : "") + names[i];
    }
    else return fail(NTT_ERR_MODULUS, why);
    snprintf(buf, buf_len, "%s", s.c_str());
    return NTT_OK;
}

}  // extern "C"
