// Modular arithmetic for odd moduli p < 2^31 with R = 2^32 Montgomery
// reduction, shared by the device kernels and the host-side plan builder.
//
// Data stays in the ordinary domain; every constant it is multiplied by
// (twiddles, scale factors) is stored in Montgomery form c*R mod p, so
// redc(x * cR) = x*c mod p and no domain conversion of the data is needed.
// The only operand pair without a Montgomery-form member is the pointwise
// product, which yields a*b*R^-1; the inverse-transform scale constant
// carries the compensating factor R.
//
// Semantics restated from the reference: Montgomery32::REDC
// (Montgomery32.h:43-52) -- here without the final conditional subtraction
// ("lazy": result in [0,2p)) -- and its constructor constants
// (Montgomery32.h:19-40).  The reference's Barrett residue step
// (main_mpi_CRT.cc:35-39) is replaced by reduce64(), which produces the same
// canonical residue.
#pragma once
#include <stdint.h>

#include <type_traits>

#if defined(__HIPCC__)
#define NTT_HD __host__ __device__ __forceinline__
#else
#define NTT_HD inline
#endif

namespace nttamd {

// Per-modulus constants handed to kernels by value (lands in SGPRs).
struct FieldConsts {
    uint32_t p;       // modulus, odd, < 2^31
    uint32_t np;      // -p^-1 mod 2^32
    uint32_t r1;      // R mod p      (Montgomery form of 1)
    uint32_t r2;      // R^2 mod p    (Montgomery form of R)
    uint32_t rootF[32];   // Montgomery form of  w64^e, e = 0..31 (w64 = primitive 64th root; zeros when 64 does not divide p-1)
    uint32_t rootFn[32];  // Montgomery form of -w64^e  (= p - rootF[e])
    uint32_t rootI[32];   // Montgomery form of  w64^-e
};

// makes a per-thread value opaque to the optimiser, so that the indices, row numbers and masks derived from it are
// recomputed where they are used instead of being kept in registers across the whole kernel
#if defined(__HIP_DEVICE_COMPILE__)
#define NTT_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define NTT_OPAQUE(x) asm volatile("" : "+r"(x))
#endif
// a value that is the same in every lane of a wavefront, moved to a scalar register (so that what is indexed with it --
// kernel arguments, constants -- is fetched by scalar loads)
#if defined(__HIP_DEVICE_COMPILE__)
#define NTT_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define NTT_UNIFORM(x) (x)
#endif

// (T + m*p) / 2^32 with m = T*np mod 2^32.  For T < 2^32 * p the result is in [0, 2p).
NTT_HD uint32_t redc_lazy(uint64_t T, uint32_t p, uint32_t np) {
    uint32_t m = (uint32_t)T * np;
    return (uint32_t)((T + (uint64_t)m * p) >> 32);
}

// x - m if x >= m else x, for x < 2m.
NTT_HD uint32_t csub(uint32_t x, uint32_t m) { return x >= m ? x - m : x; }

// x of the lane of the own quad (four consecutive lanes) that the DPP quad permutation `ctrl` names (two selector bits per
// lane: 0xB1 = lanes 1,0,3,2, i.e. lane ^ 1; 0x4E = lanes 2,3,0,1, i.e. lane ^ 2): one full-rate vector instruction, no LDS.
// Uniform control flow only.  (The CPU build of the test suite supplies a stand-in; the host pass of the HIP compiler
// never executes it.)
#if defined(__HIP_DEVICE_COMPILE__)
#define NTT_QUAD_PERM(x, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, false))
#elif defined(NTT_CPU_TWIN)
#define NTT_QUAD_PERM(x, ctrl) hipshim::quad_perm((x), (ctrl))
#else
#define NTT_QUAD_PERM(x, ctrl) (x)
#endif

// Compile-time loop: f(integral_constant<int, I>) for I in [I0, N).  The register transforms below
// index their arrays only with these constants, so the arrays stay in registers whatever the
// optimiser's unrolling limits are.
template <int I, int N, class F>
NTT_HD void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Field policy.  LZ (p < 2^30): values between butterflies live in [0, 2p) on
// the forward (decimation-in-frequency) side and in [0, 4p) on the inverse
// (decimation-in-time) side; one conditional subtraction per butterfly.
// !LZ (2^30 <= p < 2^31): values are kept fully reduced in [0, p).
template <bool LZ>
struct Field {
    uint32_t p, np, B;  // B = bound modulus: 2p (LZ) or p

    NTT_HD explicit Field(const FieldConsts& c) : p(c.p), np(c.np), B(LZ ? 2u * c.p : c.p) {}

    // x any 32-bit value, wm = Montgomery form (< p) of w: returns x*w mod p in [0, B).
    NTT_HD uint32_t mulw(uint32_t x, uint32_t wm) const {
        uint32_t r = redc_lazy((uint64_t)x * wm, p, np);
        return LZ ? r : csub(r, p);
    }
    // a, b in [0, B): a*b*R^-1 mod p in [0, B)
    NTT_HD uint32_t mulpw(uint32_t a, uint32_t b) const { return mulw(a, b); }

    NTT_HD uint32_t add(uint32_t u, uint32_t v) const { return csub(u + v, B); }   // u, v in [0,B)
    NTT_HD uint32_t sub_raw(uint32_t u, uint32_t v) const { return u - v + B; }    // (0, 2B), unreduced
    // (u - v) mod B for u, v in [0,B): a negative difference wraps to a huge value, so the smaller of d and d + B
    // is the reduced one -- three plain instructions, no carry flag (the carry form costs a wait state on gfx950)
    NTT_HD uint32_t sub(uint32_t u, uint32_t v) const {
        const uint32_t d = u - v, e = d + B;
        return d < e ? d : e;
    }

    // Gentleman-Sande butterfly, u, v in [0,B) -> [0,B):  u' = u+v, v' = (u-v)*w.  The difference is
    // never formed: (u-v)*w = u*w + v*(-w) is accumulated in 64 bits (two multiply-adds) and reduced
    // once; u*w + v*(p-w) < 2*B*p <= 2^32*p keeps the reduction in range.  wn = Montgomery form of -w.
    NTT_HD void dif(uint32_t& u, uint32_t& v, uint32_t wm, uint32_t wn) const {
        uint32_t s = add(u, v);
        uint32_t r = redc_lazy((uint64_t)u * wm + (uint64_t)v * wn, p, np);
        v = LZ ? r : csub(r, p);
        u = s;
    }
    NTT_HD void dif1(uint32_t& u, uint32_t& v) const {  // w = 1
        uint32_t s = add(u, v);
        v = sub(u, v);
        u = s;
    }

    // Cooley-Tukey butterfly for the inverse.  LZ: x, y in [0,4p) -> [0,4p)
    // (Harvey's lazy form); !LZ: [0,p) -> [0,p).   x' = x + y*w, y' = x - y*w
    NTT_HD void dit(uint32_t& x, uint32_t& y, uint32_t wm) const {
        if (LZ) {
            uint32_t xr = csub(x, B);
            uint32_t t = redc_lazy((uint64_t)y * wm, p, np);  // [0,2p) for any 32-bit y
            x = xr + t;
            y = xr - t + B;
        } else {
            uint32_t t = mulw(y, wm);
            uint32_t s = add(x, t);
            y = sub(x, t);
            x = s;
        }
    }
    NTT_HD void dit1(uint32_t& x, uint32_t& y) const {  // w = 1
        if (LZ) {
            uint32_t xr = csub(x, B), t = csub(y, B);
            x = xr + t;
            y = xr - t + B;
        } else {
            uint32_t s = add(x, y);
            y = sub(x, y);
            x = s;
        }
    }
    // as dit1 but for inputs already in [0,B) (directly after a twiddle multiply)
    NTT_HD void dit1_reduced(uint32_t& x, uint32_t& y) const {
        if (LZ) {
            uint32_t s = x + y;
            y = x - y + B;
            x = s;
        } else {
            dit1(x, y);
        }
    }

    // inverse-side value (LZ: [0,4p); !LZ: [0,p)) -> canonical [0,p)
    NTT_HD uint32_t canon_inv(uint32_t x) const { return LZ ? csub(csub(x, B), p) : x; }
    // forward-side value ([0,B)) -> canonical
    NTT_HD uint32_t canon_fwd(uint32_t x) const { return LZ ? csub(x, p) : x; }

    // any 64-bit x -> x mod p in [0,B).  lo + hi*2^32 = redc(lo*R + hi*R^2).  LZ (p < 2^30): both products are below
    // 2^32*p, their sum and the reduction's own addend still fit in 64 bits, and one reduction lands in [0,3p).
    // Otherwise the two words are reduced separately.  No branch at run time, neither on the data nor on p: a branch
    // per element ends the basic block and serialises the loads of the column passes behind it.
    NTT_HD uint32_t reduce64(uint64_t x, uint32_t r1, uint32_t r2) const {
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        if (LZ) {
            const uint32_t r = redc_lazy((uint64_t)lo * r1 + (uint64_t)hi * r2, p, np);
            return csub(r, B);
        }
        const uint32_t a = csub(redc_lazy((uint64_t)lo * r1, p, np), p);  // [0,p)
        const uint32_t b = csub(redc_lazy((uint64_t)hi * r2, p, np), p);  // [0,p)
        return csub(a + b, p);
    }
};

// bit reversal of the low `bits` bits
NTT_HD constexpr int bitrev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// In-register radix-2^Q decimation-in-frequency transform: natural order in,
// x[pos] = X[bitrev_Q(pos)] out.  rootF holds Montgomery forms of w64^e; the
// 2^Q-th root is w64^(64 >> Q).  ZERO_UPPER: x[2^(Q-1)..] are known zeros.
// LAZY_OUT: every output except x[0] is about to be multiplied by a twiddle
// (mulw accepts any 32-bit value), so the last stage skips their reduction and
// leaves them in [0, 2B).
template <int Q, bool LZ, bool ZERO_UPPER = false, bool LAZY_OUT = false>
NTT_HD void dif_reg(uint32_t (&x)[1 << Q], const Field<LZ>& f, const uint32_t (&rootF)[32], const uint32_t (&rootFn)[32]) {
    constexpr int N = 1 << Q;
    static_for<0, Q>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int h = N >> (s + 1);
        static_for<0, N>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr ((i & h) == 0) {
                constexpr int e = ((i & (h - 1)) << s) << (6 - Q);  // exponent in units of w64
                if constexpr (ZERO_UPPER && s == 0) {
                    x[i + h] = (e == 0) ? x[i] : f.mulw(x[i], rootF[e]);
                } else if constexpr (e == 0) {
                    if constexpr (LAZY_OUT && s == Q - 1) {
                        const uint32_t u = x[i], v = x[i + h];
                        x[i] = (i == 0) ? f.add(u, v) : u + v;
                        x[i + h] = f.sub_raw(u, v);
                    } else {
                        f.dif1(x[i], x[i + h]);
                    }
                } else {
                    f.dif(x[i], x[i + h], rootF[e], rootFn[e]);
                }
            }
        });
    });
}

// Mirror of dif_reg: x[pos] = X[bitrev_Q(pos)] in, natural order out, times 2^Q.
// FIRST_REDUCED: inputs are in [0,B) (fresh from a twiddle multiply).
template <int Q, bool LZ, bool FIRST_REDUCED = false>
NTT_HD void dit_reg(uint32_t (&x)[1 << Q], const Field<LZ>& f, const uint32_t (&rootI)[32]) {
    constexpr int N = 1 << Q;
    static_for<0, Q>([&](auto S) {
        constexpr int s = Q - 1 - decltype(S)::value;
        constexpr int h = N >> (s + 1);
        static_for<0, N>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr ((i & h) == 0) {
                constexpr int e = ((i & (h - 1)) << s) << (6 - Q);
                if constexpr (e == 0) {
                    if constexpr (FIRST_REDUCED && s == Q - 1) f.dit1_reduced(x[i], x[i + h]);
                    else f.dit1(x[i], x[i + h]);
                } else {
                    f.dit(x[i], x[i + h], rootI[e]);
                }
            }
        });
    });
}

}  // namespace nttamd
