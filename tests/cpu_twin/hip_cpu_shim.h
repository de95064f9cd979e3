// Test-only stand-in for the HIP runtime: lets the kernel sources and the host
// launch code be compiled by g++ and executed on the CPU so index arithmetic,
// bounds and host logic can be checked without a GPU.  One workgroup runs at a
// time; its threads are cooperative fibers and __syncthreads() is a yield.
// Never part of the shipped library (the package loads only libnttamd.so).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local   /* one workgroup at a time per host thread */
#define __launch_bounds__(...)
#define __restrict__
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) (&(x))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };

namespace hipshim {
struct State {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    ucontext_t sched;
    std::vector<ucontext_t> fibers;
    std::vector<char> stacks;
    std::vector<int> done;
    const std::function<void()>* body = nullptr;
    int current = 0;
};
inline State& st() { static thread_local State s; return s; }   // launches from different host threads do not meet

inline void fiber_entry() {
    State& s = st();
    (*s.body)();
    s.done[s.current] = 1;
    swapcontext(&s.fibers[s.current], &s.sched);
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    State& s = st();
    const int T = (int)(block.x * block.y * block.z);
    const size_t STACK = 256 * 1024;
    s.blockDim = block; s.gridDim = grid; s.body = &body;
    s.fibers.resize(T); s.stacks.resize((size_t)T * STACK); s.done.assign(T, 0);
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.blockIdx = dim3(bx, by, bz);
        for (int t = 0; t < T; ++t) {
            getcontext(&s.fibers[t]);
            s.fibers[t].uc_stack.ss_sp = s.stacks.data() + (size_t)t * STACK;
            s.fibers[t].uc_stack.ss_size = STACK;
            s.fibers[t].uc_link = nullptr;
            makecontext(&s.fibers[t], (void (*)())fiber_entry, 0);
            s.done[t] = 0;
        }
        int remaining = T;
        while (remaining > 0) {  // one sweep = one barrier interval
            for (int t = 0; t < T; ++t) {
                if (s.done[t]) continue;
                s.current = t;
                s.threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                swapcontext(&s.sched, &s.fibers[t]);
                if (s.done[t]) --remaining;
            }
        }
    }
}
inline void yield_barrier() {
    State& s = st();
    swapcontext(&s.fibers[s.current], &s.sched);
}
// A DPP quad permutation (every lane reads `x` of the lane of its own quad that `ctrl` names: two selector bits per
// lane) for fibres: all threads of the workgroup must reach it together, as they do on the GPU, where it is one
// instruction of a wavefront in uniform control flow.
inline uint32_t quad_perm(uint32_t x, int ctrl) {
    static thread_local std::vector<uint32_t> lanes;
    State& s = st();
    const int T = (int)(s.blockDim.x * s.blockDim.y * s.blockDim.z);
    if ((int)lanes.size() < T) lanes.resize(T);
    const int me = s.current;
    lanes[me] = x;
    yield_barrier();
    const uint32_t v = lanes[(me & ~3) | ((ctrl >> (2 * (me & 3))) & 3)];
    yield_barrier();
    return v;
}
}  // namespace hipshim

#define threadIdx (hipshim::st().threadIdx)
#define blockIdx (hipshim::st().blockIdx)
#define blockDim (hipshim::st().blockDim)
#define gridDim (hipshim::st().gridDim)
inline void __syncthreads() { hipshim::yield_barrier(); }

inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }

// ---- runtime API subset ----
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "shim error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// NTT_TWIN_DEVICES=k makes the stand-in report k devices (all of them host memory), so that the multi-device host logic runs
inline int shim_device_count() { const char* e = getenv("NTT_TWIN_DEVICES"); const int k = e ? atoi(e) : 1; return k >= 1 && k <= 64 ? k : 1; }
inline int& shim_current_device() { static thread_local int d = 0; return d; }
inline hipError_t hipGetDeviceCount(int* n) { *n = shim_device_count(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= shim_device_count()) return hipErrorInvalidValue; shim_current_device() = d; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = shim_current_device(); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
template <class T> inline hipError_t hipGetSymbolAddress(void** p, T* sym) { *p = (void*)sym; return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)1; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

#define NTT_LAUNCH(kernel, grid, block, stream, ...) \
    do { (void)(stream); hipshim::launch((grid), (block), [&]() { kernel(__VA_ARGS__); }); } while (0)
