// Issue cost of single VALU instructions on gfx950, in SHADER cycles (clock-independent).
//
// Every wave runs ITERS x 16 independent copies of one instruction (inline asm, so nothing is
// folded) between two s_memtime stamps; s_memrealtime (100 MHz) beside them gives the clock the
// chip held.  With W waves per SIMD the cost per wave-instruction per SIMD is
// cycles / (ITERS * 16 * W).  Usage: ubench_issue [waves_per_simd = 1 2 4 8]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 8192;

struct Stamp {
    uint64_t cycles, ticks;
};

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, Stamp* stamps, uint32_t seed)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[16];
    uint32_t x = tid * 2654435761u + seed, y = seed ^ 0x9e3779b9u;
    uint64_t z = ((uint64_t)x << 32) | y;
    uint64_t smask = __builtin_amdgcn_read_exec() ^ seed;
    asm volatile("" : "+s"(smask));
#pragma unroll
    for (int i = 0; i < 16; i++)
        a[i] = (uint64_t)(tid + i) * 0x9e3779b97f4a7c15ull + seed;
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
#define LO(i) "v"((uint32_t)a[i])
        if (OP == 0) {
#define S(i) { uint32_t d; asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 1) {
#define S(i) { uint64_t d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x), "v"(a[i]) : "vcc"); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 2) {
#define S(i) { uint32_t d; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 3) {
#define S(i) { uint32_t d; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 4) {
#define S(i) { uint64_t d; asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a[i]), "v"(z)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 5) {
#define S(i) { uint32_t d; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 6) {
#define S(i) { float d; asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(d) : "v"(__uint_as_float((uint32_t)a[i])), "v"(__uint_as_float(x))); a[i] = __float_as_uint(d); }
            REP16(S)
#undef S
        } else if (OP == 7) {
#define S(i) { uint64_t d; asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(d) : "v"(a[i]), "v"(z)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 8) {
#define S(i) { uint64_t d; asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(d) : "v"(a[i]), "v"(z)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 9) {
#define S(i) { uint32_t l, h; asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc" : "=&v"(l), "=&v"(h) : "v"((uint32_t)a[i]), "v"((uint32_t)(a[i] >> 32)), "v"(x), "v"(y) : "vcc"); a[i] = ((uint64_t)h << 32) | l; }
            REP16(S)
#undef S
        } else if (OP == 10) {
#define S(i) { uint32_t l, h; asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc" : "=&v"(l), "=&v"(h) : "v"((uint32_t)a[i]), "v"((uint32_t)(a[i] >> 32)), "v"(x), "v"(y) : "vcc"); a[i] = ((uint64_t)h << 32) | l; }
            REP16(S)
#undef S
        } else if (OP == 11) {
#define S(i) { uint32_t d; asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x), "v"(y)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 12) {
#define S(i) { uint32_t d; asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 13) {
#define S(i) { uint32_t d; asm volatile("v_lshl_add_u32 %0, %1, 21, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 14) {
#define S(i) { uint64_t d; asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(d) : "v"(a[i])); a[i] = d ^ z; }
            REP16(S)
#undef S
        } else if (OP == 15) {
#define S(i) { uint32_t d; asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 16) {
#define S(i) { uint32_t d; asm volatile("v_alignbit_b32 %0, %1, %2, 29" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 17) {
#define S(i) { uint32_t d; asm volatile("v_sub_co_u32 %0, vcc, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x) : "vcc"); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 18) {
#define S(i) { uint64_t d; asm volatile("v_pk_mov_b32 %0, %1, 0 op_sel:[1,0]" : "=v"(d) : "v"(a[i])); a[i] = d + z; }
            REP16(S)
#undef S
        } else if (OP == 19) {
#define S(i) { uint64_t d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"((uint32_t)a[i]), "s"(seed) : "vcc"); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 20) {
#define S(i) { uint32_t d; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 21) {
#define S(i) { uint32_t d; asm volatile("v_bfe_u32 %0, %1, 3, 29" : "=v"(d) : "v"((uint32_t)a[i])); a[i] = d + x; }
            REP16(S)
#undef S
        } else if (OP == 22) {
#define S(i) { uint32_t d; asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 23) {
#define S(i) { uint32_t d; asm volatile("v_min_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 24) {
#define S(i) { uint32_t d; asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 25) {
#define S(i) { uint32_t d; asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 26) {
#define S(i) { uint32_t d; asm volatile("v_sub_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 27) {
#define S(i) { uint32_t d; asm volatile("v_or_b32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 28) {
#define S(i) { uint32_t d; asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 30) {
#define S(i) { uint32_t d; asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 31) {
#define S(i) { uint32_t d; asm volatile("v_not_b32 %0, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 34) {
#define S(i) { uint32_t d; asm volatile("v_bfe_u32 %0, %1, 3, 29" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 35) {
#define S(i) { uint32_t d; asm volatile("v_and_or_b32 %0, %1, %2, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 36) {
#define S(i) { uint32_t d; asm volatile("v_max_u32 %0, %1, %2" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 37) {
#define S(i) { uint32_t d; asm volatile("v_lshrrev_b32 %0, %2, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 38) {
#define S(i) { uint32_t d; asm volatile("v_add_u32 %0, 0x12345, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 39) {
#define S(i) { uint32_t d; asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 29) {
#define S(i) { uint32_t d; asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"((uint32_t)a[i]), "v"(x), "s"(smask)); a[i] = d; }
            REP16(S)
#undef S
        } else if (OP == 32) {
#define S(i) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"((uint32_t)a[i]), "v"(x) : "vcc"); }
            REP16(S)
#undef S
        } else if (OP == 33) {
#define S(i) { uint32_t l, h; asm volatile("v_sub_co_u32 %0, vcc, %2, %4\n\ts_nop 1\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "=&v"(l), "=&v"(h) : "v"((uint32_t)a[i]), "v"((uint32_t)(a[i] >> 32)), "v"(x), "v"(y) : "vcc"); a[i] = ((uint64_t)h << 32) | l; }
            REP16(S)
#undef S
        } else if (OP == 40) {
#define S(i) { uint32_t t, h; uint64_t d; asm volatile("v_lshrrev_b32 %1, 31, %4\n\tv_and_b32 %2, 0x7fffffff, %4\n\tv_mad_u64_u32 %0, vcc, %1, %5, %3" : "=v"(d), "=&v"(t), "=&v"(h) : "v"(a[i]), "v"((uint32_t)(a[i] >> 32)), "s"(seed) : "vcc"); a[i] = d ^ ((uint64_t)h << 32); }
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

template <int OP>
void run(const char* name, int waves_per_simd, int extra_per_asm, int instrs_per_rep)
{
    // 256-thread blocks: 4 waves, one per SIMD; waves_per_simd blocks per CU
    const int blocks = 256 * waves_per_simd, threads = 256;
    uint64_t* out;
    Stamp* stamps;
    const int nwaves = blocks * threads / 64;
    CHECK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads));
    CHECK(hipMalloc(&stamps, sizeof(Stamp) * nwaves));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, stamps, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, threads>>>(out, stamps, 777u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h(nwaves);
    CHECK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * nwaves, hipMemcpyDeviceToHost));
    std::vector<double> cyc, mhz;
    for (auto& s : h) {
        cyc.push_back((double)s.cycles);
        mhz.push_back((double)s.cycles / (double)s.ticks * 100.0);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(mhz.begin(), mhz.end());
    const double clock = mhz[mhz.size() / 2];
    const double n = (double)ITERS * 16 * instrs_per_rep * waves_per_simd;
    // whole launch (event time x measured clock) and the median wave's own span
    printf("%-34s W=%d  launch %6.2f  wave %6.2f cycles/wave-instr/SIMD  (%d instr per rep%s)  clock %.0f MHz  %.3f ms\n", name,
           waves_per_simd, ms * 1e-3 * clock * 1e6 / n, cyc[cyc.size() / 2] / n, instrs_per_rep,
           extra_per_asm ? ", incl. glue" : "", clock, ms);
    CHECK(hipFree(out));
    CHECK(hipFree(stamps));
}

int main(int argc, char** argv)
{
    std::vector<int> ws;
    for (int i = 1; i < argc; i++)
        ws.push_back(atoi(argv[i]));
    if (ws.empty())
        ws = {1, 2, 4, 8};
    for (int w : ws) {
        run<0>("v_add_u32", w, 0, 1);
        run<12>("v_and_b32", w, 0, 1);
        run<20>("v_xor_b32", w, 0, 1);
        run<23>("v_min_u32", w, 0, 1);
        run<11>("v_add3_u32", w, 0, 1);
        run<13>("v_lshl_add_u32", w, 0, 1);
        run<16>("v_alignbit_b32", w, 0, 1);
        run<5>("v_cndmask_b32", w, 0, 1);
        run<17>("v_sub_co_u32", w, 0, 1);
        run<9>("v_add_co + v_addc (no nop)", w, 0, 2);
        run<10>("v_add_co + s_nop 1 + v_addc", w, 0, 2);
        run<4>("v_lshl_add_u64", w, 0, 1);
        run<14>("v_lshlrev_b64 (+xor64)", w, 1, 3);
        run<18>("v_pk_mov_b32 (+add64)", w, 1, 2);
        run<21>("v_bfe_u32 (+add)", w, 1, 2);
        run<24>("v_lshrrev_b32 (const shift)", w, 0, 1);
        run<37>("v_lshrrev_b32 (vgpr shift)", w, 0, 1);
        run<25>("v_lshlrev_b32", w, 0, 1);
        run<26>("v_sub_u32", w, 0, 1);
        run<27>("v_or_b32", w, 0, 1);
        run<28>("v_ashrrev_i32", w, 0, 1);
        run<30>("v_mov_b32", w, 0, 1);
        run<31>("v_not_b32", w, 0, 1);
        run<34>("v_bfe_u32", w, 0, 1);
        run<35>("v_and_or_b32", w, 0, 1);
        run<36>("v_max_u32", w, 0, 1);
        run<38>("v_add_u32 literal", w, 0, 1);
        run<39>("v_and_b32 literal", w, 0, 1);
        run<29>("v_cndmask_b32 sgpr mask", w, 0, 1);
        run<32>("v_cmp_lt_u32 vcc", w, 0, 1);
        run<33>("csub (4 instr + 2 s_nop 1)", w, 0, 4);
        run<40>("topbit reduce (3 instr, +2 glue)", w, 1, 5);
        run<1>("v_mad_u64_u32 (vgpr)", w, 0, 1);
        run<19>("v_mad_u64_u32 (sgpr factor, 0)", w, 0, 1);
        run<2>("v_mul_lo_u32", w, 0, 1);
        run<3>("v_mul_hi_u32", w, 0, 1);
        run<15>("v_mad_u32_u24", w, 0, 1);
        run<22>("v_mul_u32_u24", w, 0, 1);
        run<6>("v_fma_f32", w, 0, 1);
        run<7>("v_pk_fma_f32", w, 0, 1);
        run<8>("v_fma_f64", w, 0, 1);
    }
    return 0;
}
