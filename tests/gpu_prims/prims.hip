// Test-only kernels that apply the library's arithmetic primitives (ntt_field.h, ntt_field64.h, the 2-by-1 remainder of
// the CRT merge) to caller-supplied vectors, one element per thread, so that each primitive can be compared with Python
// integers on edge values in isolation.  Built by __graft_entry__.build() into tests/gpu_prims/libnttprims.so; never
// part of the shipped library.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_kernels.h"
#include "ntt_kernels64.h"

using namespace nttamd;

enum Op {
    OP_REDC_LAZY = 0,     // in0 = T (u64)                       -> redc_lazy(T, p, np)
    OP_MULW_LZ = 1,       // in0 = x (any u32), in1 = wm (< p)    -> Field<true>::mulw
    OP_MULW_NLZ = 2,      //                                      -> Field<false>::mulw
    OP_REDUCE64_LZ = 3,   // in0 = x (any u64)                    -> Field<true>::reduce64
    OP_REDUCE64_NLZ = 4,  //                                      -> Field<false>::reduce64
    OP_DIF_LZ = 5,        // in0 = u | v << 32, in1 = wm          -> u' | v' << 32 of Field<true>::dif
    OP_DIF_NLZ = 6,
    OP_DIT_LZ = 7,        // in0 = x | y << 32, in1 = wm          -> x' | y' << 32 of Field<true>::dit
    OP_DIT_NLZ = 8,
    OP_F64_MULW = 9,      // in0 = x, in1 = wm                    -> Field64::mulw
    OP_F64_REDUCE = 10,   // in0 = x (any u64)                    -> Field64::reduce(x, r2)
    OP_F64_DIF = 11,      // in0 = u, in1 = v, in2 = wm           -> out = u', out2 = v'
    OP_F64_DIT = 12,
    OP_REM_2BY1 = 13,     // in0 = u1, in1 = u0 (u1 < p)          -> (u1:u0) mod p with d, v, shift from the consts
    OP_REM128 = 14,       // same through the 64-bit-field file's copy (rem128)
    OP_CANON_INV_LZ = 15, // in0 = x in [0,4p)                    -> Field<true>::canon_inv
    OP_SUB_LZ = 16,       // in0 = u | v << 32 (both < 2p)        -> Field<true>::sub
    // double-precision field: values travel as signed 64-bit integers
    OP_FD_MULW = 17,      // in0 = x, in1 = w                     -> FieldD::mulw(x, w)
    OP_FD_REDUCE = 18,    // in0 = x                              -> FieldD::reduce(x)
    OP_FD_FROM_RAW = 19,  // in0 = any 64-bit word                -> FieldD::from_raw
    OP_FD_MEM = 20,       // in0 = x                              -> out = to_mem(x), out2 = canon(x)
    OP_FD_FROM_MEM = 21,  // in0 = stored word (< 2^52)           -> from_mem
    OP_FD_DIF = 22,       // in0 = u, in1 = v, in2 = w            -> out = u', out2 = v'
    OP_FD_DIT = 23,
};

struct PrimConsts {
    FieldConsts fc;
    Field64Consts fc64;
    FieldDConsts fd;
    uint64_t d, v;
    int shift;
};

__global__ void k_prims(int op, const uint64_t* in0, const uint64_t* in1, const uint64_t* in2, uint64_t* out, uint64_t* out2,
                        size_t count, PrimConsts c) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Field<true> fl(c.fc);
    const Field<false> fn(c.fc);
    const Field64 f64(c.fc64);
    const FieldD fdd(c.fd);
    const double dx = (double)(int64_t)in0[i], dy = in1 ? (double)(int64_t)in1[i] : 0.0, dz = in2 ? (double)(int64_t)in2[i] : 0.0;
    const uint64_t x = in0[i], y = in1 ? in1[i] : 0, z = in2 ? in2[i] : 0;
    uint64_t r = 0, r2 = 0;
    uint32_t u = (uint32_t)x, w = (uint32_t)(x >> 32);
    switch (op) {
        case OP_REDC_LAZY: r = redc_lazy(x, c.fc.p, c.fc.np); break;
        case OP_MULW_LZ: r = fl.mulw((uint32_t)x, (uint32_t)y); break;
        case OP_MULW_NLZ: r = fn.mulw((uint32_t)x, (uint32_t)y); break;
        case OP_REDUCE64_LZ: r = fl.reduce64(x, c.fc.r1, c.fc.r2); break;
        case OP_REDUCE64_NLZ: r = fn.reduce64(x, c.fc.r1, c.fc.r2); break;
        case OP_DIF_LZ: fl.dif(u, w, (uint32_t)y, c.fc.p - (uint32_t)y); r = u | ((uint64_t)w << 32); break;
        case OP_DIF_NLZ: fn.dif(u, w, (uint32_t)y, c.fc.p - (uint32_t)y); r = u | ((uint64_t)w << 32); break;
        case OP_DIT_LZ: fl.dit(u, w, (uint32_t)y); r = u | ((uint64_t)w << 32); break;
        case OP_DIT_NLZ: fn.dit(u, w, (uint32_t)y); r = u | ((uint64_t)w << 32); break;
        case OP_F64_MULW: r = f64.mulw(x, y); break;
        case OP_F64_REDUCE: r = f64.reduce(x, c.fc64.r2); break;
        case OP_F64_DIF: { uint64_t a = x, b = y; f64.dif(a, b, z); r = a; r2 = b; break; }
        case OP_F64_DIT: { uint64_t a = x, b = y; f64.dit(a, b, z); r = a; r2 = b; break; }
        case OP_REM_2BY1: r = rem_2by1(x, y, c.d, c.v, c.shift); break;
        case OP_REM128: r = rem128(x, y, c.d, c.v, c.shift); break;
        case OP_CANON_INV_LZ: r = fl.canon_inv((uint32_t)x); break;
        case OP_SUB_LZ: r = fl.sub(u, w); break;
        case OP_FD_MULW: r = (uint64_t)(int64_t)fdd.mulw(dx, dy); break;
        case OP_FD_REDUCE: r = (uint64_t)(int64_t)fdd.reduce(dx); break;
        case OP_FD_FROM_RAW: r = (uint64_t)(int64_t)fdd.from_raw(x); break;
        case OP_FD_MEM: r = fdd.to_mem(dx); r2 = fdd.canon(dx); break;
        case OP_FD_FROM_MEM: r = (uint64_t)(int64_t)fdd.from_mem(x); break;
        case OP_FD_DIF: { double a = dx, b = dy; fdd.dif(a, b, dz); r = (uint64_t)(int64_t)a; r2 = (uint64_t)(int64_t)b; break; }
        case OP_FD_DIT: { double a = dx, b = dy; fdd.dit(a, b, dz); r = (uint64_t)(int64_t)a; r2 = (uint64_t)(int64_t)b; break; }
    }
    out[i] = r;
    if (out2) out2[i] = r2;
}

extern "C" int prims_run(int op, const uint64_t* in0, const uint64_t* in1, const uint64_t* in2, uint64_t* out, uint64_t* out2,
                         size_t count, const PrimConsts* c, void* stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_prims, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, in0, in1, in2, out,
                       out2, count, *c);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" size_t prims_consts_size(void) { return sizeof(PrimConsts); }
// byte offsets the Python side fills (keeps the test independent of struct padding rules)
extern "C" void prims_consts_fill(PrimConsts* c, uint32_t p32, uint32_t np32, uint32_t r1_32, uint32_t r2_32, uint64_t p64,
                                  uint64_t np64, uint64_t r1_64, uint64_t r2_64, uint64_t d, uint64_t v, int shift) {
    *c = PrimConsts{};
    c->fc.p = p32; c->fc.np = np32; c->fc.r1 = r1_32; c->fc.r2 = r2_32;
    c->fc64.p = p64; c->fc64.np = np64; c->fc64.r1 = r1_64; c->fc64.r2 = r2_64;
    c->d = d; c->v = v; c->shift = shift;
    c->fd.p = (double)p64; c->fd.u = 1.0 / (double)p64;
    const uint64_t c40 = (1ull << 40) % p64;
    c->fd.c40 = c40 > p64 / 2 ? -(double)(p64 - c40) : (double)c40;
}
