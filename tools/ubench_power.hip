// Board power per kind of VALU / LDS instruction on gfx950.  Every SIMD of the chip runs four waves of
// 16 independent copies of ONE instruction in a loop (inline asm, nothing folded), launch after launch
// for about 0.7 s, while a host thread reads the board power (hwmon power1_input) every 20 ms; the
// second half of the readings is the plateau.  The shader clock comes from s_memtime / s_memrealtime.
// Output per instruction: wave-instructions per second over the chip, clock, board power, and the
// energy per wave-instruction above the power of the emptiest loop measured (s_nop).
// The step of a transform runs at the board's power limit, so what an instruction costs there is
// energy, not only an issue slot.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 8192;
struct Stamp {
    uint64_t cycles, ticks;
};
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

enum Op { NOP, ADD32, ADD64_PAIR, LSHL_ADD64, MAD64, MAD64_ZERO_HI, MUL_LO, MUL_HI, MAD24, FMA64, CNDMASK, XOR32, LDS_RW64, NOPS };

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, Stamp* stamps, uint32_t seed)
{
    __shared__ uint64_t lds[256 * 17];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[16];
    uint32_t x = tid * 2654435761u + seed, y = seed ^ 0x9e3779b9u;
    uint64_t z = ((uint64_t)x << 32) | y;
#pragma unroll
    for (int i = 0; i < 16; i++)
        a[i] = (uint64_t)(tid + i) * 0x9e3779b97f4a7c15ull + seed;
    if (OP == LDS_RW64) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            lds[threadIdx.x * 17 + i] = a[i];
    }
    uint64_t* mine = lds + threadIdx.x * 17;
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
        if (OP == NOP) {
#define S(i) asm volatile("s_nop 3");
            REP16(S)
#undef S
        } else if (OP == ADD32) {
#define S(i) { uint32_t d; asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == XOR32) {
#define S(i) { uint32_t d; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == ADD64_PAIR) {
#define S(i) { uint32_t l, h; asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc" : "=&v"(l), "=&v"(h) : "v"((uint32_t)a[i]), "v"((uint32_t)(a[i] >> 32)), "v"(x), "v"(y) : "vcc"); a[i] = ((uint64_t)h << 32) | l; }
            REP16(S)
#undef S
        } else if (OP == LSHL_ADD64) {
#define S(i) { uint64_t d; asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a[i]), "v"(z)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == MAD64) {
#define S(i) { uint64_t d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x), "v"(a[i]) : "vcc"); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == MAD64_ZERO_HI) { // factors below 2^16: what a multiplier costs when most of its array idles
#define S(i) { uint64_t d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"((uint32_t)a[i] & 0xffffu), "v"(x & 0xffffu), "v"(a[i] & 0xffffffffull) : "vcc"); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == MUL_LO) {
#define S(i) { uint32_t d; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == MUL_HI) {
#define S(i) { uint32_t d; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i] | 0x80000000u), "v"(x | 0x80000000u)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == MAD24) {
#define S(i) { uint32_t d; asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == FMA64) {
#define S(i) { uint64_t d; asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(d) : "v"(a[i]), "v"(z)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == CNDMASK) {
#define S(i) { uint32_t d; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == LDS_RW64) { // one 64-bit LDS write and one read per copy (conflict-free, padded rows)
#define S(i) { mine[i] = a[i] + z; asm volatile("" ::: "memory"); a[i] = mine[(i + 1) & 15]; }
            REP16(S)
#undef S
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++)
        s ^= a[i];
    out[tid] = s;
    if ((threadIdx.x & 63) == 0)
        stamps[tid >> 6] = Stamp{c1 - c0, r1 - r0};
}

static double board_watts()
{
    FILE* f = fopen("/sys/class/drm/card0/device/hwmon/hwmon0/power1_input", "r");
    if (!f)
        return 0;
    long long uw = 0;
    if (fscanf(f, "%lld", &uw) != 1)
        uw = 0;
    fclose(f);
    return uw / 1e6;
}

static double base_watts = 0;

template <int OP> void run(const char* name, int instrs_per_copy, double seconds)
{
    const int W = 4, blocks = 256 * W, threads = 256, nwaves = blocks * threads / 64;
    uint64_t* out;
    Stamp* stamps;
    CHECK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads));
    CHECK(hipMalloc(&stamps, sizeof(Stamp) * nwaves));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, stamps, 12345u);
    CHECK(hipDeviceSynchronize());
    std::atomic<bool> stop{false};
    std::vector<double> watts;
    std::thread sampler([&] {
        while (!stop.load()) {
            watts.push_back(board_watts());
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    CHECK(hipEventRecord(e0));
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int r = 0; r < 8; r++, launches++)
            k<OP><<<blocks, threads>>>(out, stamps, 777u + launches);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    stop.store(true);
    sampler.join();
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h(nwaves);
    CHECK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * nwaves, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (auto& s : h)
        mhz.push_back((double)s.cycles / (double)s.ticks * 100.0);
    std::sort(mhz.begin(), mhz.end());
    std::vector<double> late(watts.begin() + watts.size() / 2, watts.end());
    std::sort(late.begin(), late.end());
    const double w = late.empty() ? 0 : late[late.size() / 2];
    if (OP == NOP)
        base_watts = w;
    const double n = (double)ITERS * 16 * instrs_per_copy * nwaves * launches; // wave-instructions issued
    const double rate = n / (ms * 1e-3);
    printf("%-36s %7.1f G wave-instr/s  clock %5.0f MHz  %6.0f W  %+7.2f nJ per wave-instr over the s_nop loop  (%d launches, %.0f ms)\n",
           name, rate / 1e9, mhz[mhz.size() / 2], w, (w - base_watts) / rate * 1e9, launches, ms);
    CHECK(hipFree(out));
    CHECK(hipFree(stamps));
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 0.7;
    run<NOP>("s_nop 3 (baseline: waves resident)", 1, seconds);
    run<ADD32>("v_add_u32", 1, seconds);
    run<XOR32>("v_xor_b32", 1, seconds);
    run<CNDMASK>("v_cndmask_b32", 1, seconds);
    run<ADD64_PAIR>("v_add_co + s_nop + v_addc", 2, seconds);
    run<LSHL_ADD64>("v_lshl_add_u64", 1, seconds);
    run<MAD24>("v_mad_u32_u24", 1, seconds);
    run<MUL_LO>("v_mul_lo_u32", 1, seconds);
    run<MUL_HI>("v_mul_hi_u32", 1, seconds);
    run<MAD64>("v_mad_u64_u32 (full-width factors)", 1, seconds);
    run<MAD64_ZERO_HI>("v_mad_u64_u32 (16-bit factors, +3 and)", 1, seconds);
    run<FMA64>("v_fma_f64", 1, seconds);
    run<LDS_RW64>("ds_write_b64 + ds_read_b64 (+add64)", 2, seconds);
    run<NOP>("s_nop 3 again", 1, seconds);
    return 0;
}
