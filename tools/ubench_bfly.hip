// Butterfly-level micro-benchmark: VALU cost of candidate butterfly formulations on gfx950.
// Each thread runs radix-16 register rounds (32 butterflies) repeatedly; prints wave-cycles
// per butterfly at the measured clock-independent rate (G butterflies/s chip-wide).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "../ntt_amd/csrc/kernels.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace nttx;

template <class A, bool FWD>
__global__ __launch_bounds__(256, 4) void k(uint64_t* out, const typename A::TwElem* tw, typename A::Ctx c, int iters)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x[16];
    typename A::Tw w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = out[tid * 16 + i];
#pragma unroll
    for (int i = 1; i < 16; i++) w[i] = A::load(tw, (tid & 3) * 16 + i, c);
    for (int it = 0; it < iters; it++) {
        radix<A, 4, FWD>(x, RegTw<A, 4>{w}, c, false);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[tid * 16 + i] = x[i];
}

template <class A, bool FWD>
void run(const char* name, typename A::Ctx c, uint64_t q)
{
    const int blocks = 256 * 8, threads = 256, iters = 256;
    uint64_t *out; typename A::TwElem* tw;
    size_t n = (size_t)blocks * threads * 16;
    CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMalloc(&tw, 64 * 16));
    uint64_t* h = (uint64_t*)malloc(n * 8);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s % q; }
    CHECK(hipMemcpy(out, h, n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(tw, h + 1000, 64 * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<A, FWD><<<blocks, threads>>>(out, tw, c, 8);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<A, FWD><<<blocks, threads>>>(out, tw, c, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double bf = (double)blocks * threads * iters * 32;
    double rate = bf / (ms * 1e-3);
    printf("%-28s %8.1f G butterflies/s  %.2f ms  => %.1f cyc/wave-bfly @2.4GHz, %.1f @1.93GHz\n", name, rate / 1e9, ms,
           1024.0 * 64 * 2.4e9 / rate, 1024.0 * 64 * 1.93e9 / rate);
    CHECK(hipFree(out)); CHECK(hipFree(tw)); free(h);
}

int main()
{
    const uint64_t q = 0x1fffffffffe00001ull;
    MontCtx mc{q, 0x6000040000200001ull, (uint64_t)((((unsigned __int128)1) << 64) % q), 0x1000000000000ull};
    GoldilocksCtx gc{0xfffeffff00010001ull};
    run<Mont<false>, true>("mont strict fwd", mc, q);
    run<Mont<false>, false>("mont strict bwd", mc, q);
    Shoup61Ctx sc{q, 2 * q, 4 * q, 0 - q, {1, (uint64_t)((((unsigned __int128)1) << 64) / q)}, {5, 40}, {7, 56}, 0};
    run<Shoup61, true>("shoup61 lazy fwd", sc, q);
    run<Shoup61, false>("shoup61 lazy bwd", sc, q);
    run<Mont<true>, true>("mont wide fwd", mc, q);
    run<Goldilocks, true>("goldilocks fwd", gc, Goldilocks::Q);
    run<Goldilocks, false>("goldilocks bwd", gc, Goldilocks::Q);
    return 0;
}
