// Exchange microbenchmark: the 16 x 16 transposition between two radix-16 rounds of a transform (16 threads trade their 16
// registers so that thread i ends up with register i of everybody), (a) through LDS the way the 4096-point core does it
// (16 writes, barrier, 16 reads, conflict-free swizzle), (b) inside the wavefront with ds_bpermute shuffles and selects
// (four butterfly steps of 16 shuffles each: a register-indexed exchange has to go through log2(16) rounds because a lane
// cannot pick a register by a run-time index), (c) radix-2 levels ACROSS lanes with no transposition at all: at lane distance
// d = 1 .. 32 lanes l and l^d trade one register of each pair (v[2i], v[2i+1]), after which every lane holds both inputs
// of a butterfly -- DPP moves (quad_perm for d = 1, 2; row_shl / row_shr with bank masks for d = 4; row_ror:8 for d = 8;
// gfx950 has no row_xmask) and v_permlane16_swap / v_permlane32_swap for d = 16, 32; (d) the one-value-per-lane form of the
// same levels, where each lane fetches its partner's value through a DPP operand and computes only its own output.
// Prints time per exchange per wave.   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_xchg.hip -o tools/ubench_xchg
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;

__device__ __forceinline__ void work(uint32_t (&v)[16], uint32_t c) {   // a little arithmetic between exchanges, so that nothing folds away
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = v[i] * 2654435761u + c + i;
}

// (a) LDS: slot (row, col) -> row * 16 + (col ^ row): writes by rows, reads by columns, both conflict-free
__global__ void __launch_bounds__(256) k_lds(uint32_t* out, uint32_t seed) {
    __shared__ uint32_t lds[4096];
    const int t = threadIdx.x, hi = t >> 4, lo = t & 15;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) {
        work(v, (uint32_t)it);
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[hi * 256 + lo * 16 + (i ^ lo)] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = lds[hi * 256 + i * 16 + (lo ^ i)];
        __syncthreads();
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

// (b) shuffles: transpose of a 16 x 16 block held by 16 consecutive lanes, butterfly steps d = 8, 4, 2, 1:
// lanes l and l^d swap the registers whose index has bit d different from the lane's bit d
__global__ void __launch_bounds__(256) k_shfl(uint32_t* out, uint32_t seed) {
    const int t = threadIdx.x, lo = t & 15;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) {
        work(v, (uint32_t)it);
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            const bool up = (lo & d) != 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((i & d) == 0) {
                    // pair (i, i+d): a lane with bit d clear keeps v[i] and trades v[i+d]; the partner the other way round
                    const uint32_t give = up ? v[i] : v[i + d];
                    const uint32_t got = (uint32_t)__shfl_xor((int)give, d, 64);
                    if (up) v[i] = got; else v[i + d] = got;
                }
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

// (c) one radix-2 level across lanes at distance D for the eight register pairs of a lane: the lane whose bit D is clear
// gives v[2i+1] and takes the partner's v[2i]; the other lane the other way round.  Afterwards (v[2i], v[2i+1]) are the two
// inputs of one butterfly in one lane.
template <int D>
__device__ __forceinline__ void pair_swap(uint32_t& x0, uint32_t& x1, int lane) {
    if constexpr (D == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);   // x0[32..63] <-> x1[0..31]
        x0 = r[0]; x1 = r[1];
    } else if constexpr (D == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);   // odd rows of x0 <-> even rows of x1
        x0 = r[0]; x1 = r[1];
    } else if constexpr (D == 8) {
        const uint32_t t = x0;
        x0 = __builtin_amdgcn_update_dpp(x0, x1, 0x128, 0xF, 0xC, false);  // row_ror:8, lanes 8..15 of a row take x1 of lane - 8
        x1 = __builtin_amdgcn_update_dpp(x1, t, 0x128, 0xF, 0x3, false);   // lanes 0..7 take x0 of lane + 8
    } else if constexpr (D == 4) {
        const uint32_t t = x0;
        x0 = __builtin_amdgcn_update_dpp(x0, x1, 0x114, 0xF, 0xA, false);  // row_shr:4 into banks 1, 3
        x1 = __builtin_amdgcn_update_dpp(x1, t, 0x104, 0xF, 0x5, false);   // row_shl:4 into banks 0, 2
    } else {
        constexpr int ctrl = D == 2 ? 0x4E : 0xB1;                          // quad_perm [2,3,0,1] / [1,0,3,2]
        const bool up = (lane & D) != 0;
        const uint32_t give = up ? x0 : x1;
        const uint32_t got = __builtin_amdgcn_update_dpp(0u, give, ctrl, 0xF, 0xF, false);
        x0 = up ? got : x0;
        x1 = up ? x1 : got;
    }
}

template <int D>
__device__ __forceinline__ void level_swap(uint32_t (&v)[16], int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pair_swap<D>(v[2 * i], v[2 * i + 1], lane);
        const uint32_t u = v[2 * i], w = v[2 * i + 1];     // stand-in for the butterfly: sum and difference
        v[2 * i] = u + w; v[2 * i + 1] = u - w;
    }
}

template <int LEVELS>   // 6: all lane distances; 4: within a row of sixteen lanes only (d = 1 .. 8)
__global__ void __launch_bounds__(256) k_dpp_swap(uint32_t* out, uint32_t seed) {
    const int t = threadIdx.x, lane = t & 63;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) {
        work(v, (uint32_t)it);
        if (LEVELS == 6) { level_swap<32>(v, lane); level_swap<16>(v, lane); }
        level_swap<8>(v, lane); level_swap<4>(v, lane); level_swap<2>(v, lane); level_swap<1>(v, lane);
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

// (d) one value per lane: every lane reads its partner's value through a DPP operand and keeps sum or difference
template <int CTRL, int D>
__device__ __forceinline__ void level_single(uint32_t (&v)[16], int lane) {
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t other = __builtin_amdgcn_update_dpp(0u, v[i], CTRL, 0xF, 0xF, false);
        v[i] = up ? other - v[i] : v[i] + other;
    }
}
__global__ void __launch_bounds__(256) k_dpp_single(uint32_t* out, uint32_t seed) {
    const int t = threadIdx.x, lane = t & 63;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) {
        work(v, (uint32_t)it);
        level_single<0x128, 8>(v, lane); level_single<0x4E, 2>(v, lane); level_single<0xB1, 1>(v, lane);
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

template <class K>
int run(const char* name, K kern, uint32_t* d_out) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_launch = ms / reps;
    // waves per SIMD = blocks * 4 waves / (256 CUs * 4 SIMDs) = 8; each does ITERS exchanges
    const double ns_per_exchange_per_wave_per_simd = per_launch * 1e6 / ITERS / 8.0;
    printf("%-28s %.3f ms per launch  %.1f ns per exchange+work per wave per SIMD\n", name, per_launch, ns_per_exchange_per_wave_per_simd);
    return 0;
}

__global__ void __launch_bounds__(256) k_none(uint32_t* out, uint32_t seed) {
    const int t = threadIdx.x;
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + t * 16 + i;
    for (int it = 0; it < ITERS; ++it) work(v, (uint32_t)it);
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * 256 + t] = r;
}

int main() {
    uint32_t* d_out = nullptr;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    if (run("work only", k_none, d_out)) return 1;
    if (run("LDS exchange (core's way)", k_lds, d_out)) return 1;
    if (run("wave shuffles (bpermute)", k_shfl, d_out)) return 1;
    if (run("6 cross-lane levels, pair swap", k_dpp_swap<6>, d_out)) return 1;
    if (run("4 cross-lane levels, pair swap", k_dpp_swap<4>, d_out)) return 1;
    if (run("3 levels (d=8,2,1), one value", k_dpp_single, d_out)) return 1;
    CHECK(hipFree(d_out));
    return 0;
}
