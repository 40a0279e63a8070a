// Butterfly-level micro-benchmark: VALU cost of candidate butterfly formulations on gfx950.
// Each thread runs radix-16 register rounds (32 butterflies) repeatedly; prints wave-cycles
// per butterfly at the measured clock-independent rate (G butterflies/s chip-wide).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../ntt_amd/csrc/kernels.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace nttx;

template <class A, bool FWD>
__global__ __launch_bounds__(256, 4) void k(uint64_t* out, const typename A::TwElem* tw, typename A::Ctx c, int iters,
                                            uint64_t* stamps)
{
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        stamps[2 * (tid >> 6)] = c1 - c0;
        stamps[2 * (tid >> 6) + 1] = r1 - r0;
    }
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
    uint64_t* stamps;
    const int nw = blocks * threads / 64;
    CHECK(hipMalloc(&stamps, nw * 16));
    k<A, FWD><<<blocks, threads>>>(out, tw, c, 8, stamps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<A, FWD><<<blocks, threads>>>(out, tw, c, iters, stamps);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double bf = (double)blocks * threads * iters * 32;
    double rate = bf / (ms * 1e-3);
    std::vector<uint64_t> hs(nw * 2);
    CHECK(hipMemcpy(hs.data(), stamps, nw * 16, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (int i = 0; i < nw; i++) mhz.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double clock = mhz[nw / 2] * 1e6;
    printf("%-28s %8.1f G butterflies/s  %.2f ms  clock %4.0f MHz => %.1f cycles per wave-butterfly per SIMD\n", name, rate / 1e9, ms,
           clock / 1e6, 1024.0 * 64 * clock / rate);
    CHECK(hipFree(stamps));
    CHECK(hipFree(out)); CHECK(hipFree(tw)); free(h);
}

int main()
{
    const uint64_t q = 0x1fffffffffe00001ull;
    MontCtx mc{q, 0x6000040000200001ull, (uint64_t)((((unsigned __int128)1) << 64) % q), 0x1000000000000ull};
    GoldilocksCtx gc{0xfffeffff00010001ull, 0, 0};
    run<Mont<false>, true>("mont strict fwd", mc, q);
    run<Mont<false>, false>("mont strict bwd", mc, q);
    Shoup61Ctx sc = make_shoup61_ctx(q, 5, 7, 0);
    run<Shoup61, true>("shoup61 lazy fwd", sc, q);
    run<Shoup61, false>("shoup61 lazy bwd", sc, q);
    run<Shoup61Near, true>("shoup61 near fwd", sc, q);
    run<Shoup61Near, false>("shoup61 near bwd", sc, q);
    run<Mont<true>, true>("mont wide fwd", mc, q);
    run<Goldilocks, true>("goldilocks fwd", gc, Goldilocks::Q);
    run<Goldilocks, false>("goldilocks bwd", gc, Goldilocks::Q);
    return 0;
}
