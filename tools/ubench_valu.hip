// VALU integer throughput micro-benchmark for gfx950: issue rate of the 32-bit multiply
// family that a 64-bit modular multiply is built from.  Prints giga lane-ops per second.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc[ILP];
    uint32_t x = tid * 2654435761u + seed, y = seed ^ 0x9e3779b9u;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = (uint64_t)(tid + i) * 0x9e3779b97f4a7c15ull + seed;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) {  // v_mad_u64_u32
                acc[i] = (uint64_t)(uint32_t)acc[i] * (uint64_t)x + acc[i];
            } else if (OP == 1) {  // v_mul_lo_u32
                uint32_t a = (uint32_t)acc[i];
                a = a * x + 0; acc[i] = a ^ y;
            } else if (OP == 2) {  // v_mul_hi_u32
                uint32_t a = (uint32_t)acc[i];
                acc[i] = __umulhi(a, x) + y;
            } else if (OP == 3) {  // 64-bit add (2 VALU)
                acc[i] = acc[i] + (((uint64_t)y << 32) | x);
            } else if (OP == 4) {  // v_mul_u32_u24
                uint32_t a = (uint32_t)acc[i];
                acc[i] = __umul24(a, x) + y;
            } else if (OP == 5) {  // 64x64 -> hi via __umul64hi
                acc[i] = __umul64hi(acc[i], ((uint64_t)x << 32) | y) + acc[i];
            } else if (OP == 6) {  // 64-bit low multiply
                acc[i] = acc[i] * ((((uint64_t)x << 32) | y) | 1);
            } else if (OP == 7) {  // Montgomery product (generic q, k): the modular multiply itself
                const uint64_t q = 0x1fffffffffe00001ull, kq = 0x6000040000200001ull;
                uint64_t a = acc[i], b = (((uint64_t)x << 32) | y) >> 3;
                uint64_t lo = a * b, hi = __umul64hi(a, b);
                uint64_t m = lo * kq;
                uint64_t mq = __umul64hi(m, q);
                uint64_t r = hi - mq;
                if (hi < mq) r += q;
                acc[i] = r;
            } else if (OP == 8) {  // DFMA
                double d = __longlong_as_double((acc[i] & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
                d = fma(d, 1.0000001, 1e-9);
                acc[i] = __double_as_longlong(d);
            } else if (OP == 10) {  // 64-bit add via v_lshl_add_u64 (non-constant addend)
                acc[i] = acc[i] + acc[(i + 1) % ILP];
            } else if (OP == 11) {  // 64-bit compare + select
                acc[i] = (acc[i] < acc[(i + 1) % ILP]) ? acc[i] + 1 : acc[(i + 3) % ILP];
            } else if (OP == 12) {  // 32-bit add3
                uint32_t a = (uint32_t)acc[i];
                acc[i] = a + x + (uint32_t)acc[(i + 1) % ILP];
            } else if (OP == 13) {  // 64-bit sub
                acc[i] = acc[i] - acc[(i + 1) % ILP];
            } else if (OP == 20) {  // v_pk_mov_b32: (hi half, 0)
                uint64_t d;
                asm volatile("v_pk_mov_b32 %0, %1, 0 op_sel:[1,0]" : "=v"(d) : "v"(acc[i]));
                acc[i] = d;
            } else if (OP == 21) {  // v_lshrrev_b64
                uint64_t d;
                asm volatile("v_lshrrev_b64 %0, 1, %1" : "=v"(d) : "v"(acc[i]));
                acc[i] = d;
            } else if (OP == 22) {  // v_mov_b32
                uint32_t d;
                asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"((uint32_t)acc[i]));
                acc[i] = d;
            } else if (OP == 23) {  // v_max3_u32
                uint32_t d;
                asm volatile("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"((uint32_t)acc[i]), "v"(x), "v"(y));
                acc[i] = d;
            } else if (OP == 24) {  // v_lshl_add_u64
                uint64_t d;
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(acc[i]), "v"(acc[(i + 1) % ILP]));
                acc[i] = d;
            } else if (OP == 25) {  // v_cmp_eq_u32 to an SGPR pair + v_cndmask from it
                uint32_t d;
                uint64_t m;
                asm volatile("v_cmp_eq_u32 %1, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %2, %3, %1"
                             : "=&v"(d), "=&s"(m) : "v"((uint32_t)acc[i]), "v"(x));
                acc[i] = d;
            } else if (OP == 26) {  // v_cmp_lt_u64 + cndmask
                uint32_t d;
                uint64_t m;
                asm volatile("v_cmp_lt_u64 %1, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, 0, -1, %1"
                             : "=&v"(d), "=&s"(m) : "v"(acc[i]), "v"(acc[(i + 1) % ILP]));
                acc[i] = acc[i] ^ d;
            } else if (OP == 27) {  // v_mov_b64
                uint64_t d;
                asm volatile("v_mov_b64 %0, %1" : "=v"(d) : "v"(acc[i]));
                acc[i] = d;
            } else if (OP == 28) {  // add_co + addc with s_nop between
                uint32_t l, h;
                asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
                             : "=&v"(l), "=&v"(h) : "v"((uint32_t)acc[i]), "v"((uint32_t)(acc[i] >> 32)), "v"(x), "v"(y) : "vcc");
                acc[i] = ((uint64_t)h << 32) | l;
            } else if (OP == 9) {  // v_mad_u32_u24
                uint32_t a = (uint32_t)acc[i];
                acc[i] = __umul24(a, x) + a;
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[tid] = s;
}

template <int OP>
void run(const char* name, double ops_per_iter)
{
    const int blocks = 256 * 8, threads = 256;
    uint64_t* out;
    CHECK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) k<OP><<<blocks, threads>>>(out, 777u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double n = (double)blocks * threads * ITERS * ILP * reps;
    double gops = n / (ms * 1e-3) / 1e9;
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs, 64 lanes
    double cyc = 1024.0 * 64.0 * 2.4 / (gops * ops_per_iter);
    printf("%-28s %10.1f G lane-iter/s  (%.1f ms)  ~%.2f cycles/wave-instr/SIMD at 2.4GHz assuming %g instr/iter\n",
           name, gops, ms / reps, cyc, ops_per_iter);
    CHECK(hipFree(out));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs %d  clock %d kHz  memclk %d kHz  bus %d bit\n", p.name, p.multiProcessorCount,
           p.clockRate, p.memoryClockRate, p.memoryBusWidth);
    run<0>("v_mad_u64_u32", 1);
    run<1>("v_mul_lo_u32 (+xor)", 2);
    run<2>("v_mul_hi_u32 (+add)", 2);
    run<3>("add64 (2 valu)", 2);
    run<4>("v_mul_u32_u24 (+add)", 2);
    run<9>("v_mad_u32_u24", 1);
    run<5>("umul64hi + add64", 1);
    run<6>("mul64 lo", 1);
    run<7>("montmul generic", 1);
    run<8>("dfma (+and/or)", 1);
    run<10>("add64 reg+reg", 1);
    run<11>("cmp64+select+add", 1);
    run<12>("add3 u32", 1);
    run<13>("sub64 reg-reg", 1);
    run<22>("v_mov_b32", 1);
    run<20>("v_pk_mov_b32", 1);
    run<27>("v_mov_b64", 1);
    run<21>("v_lshrrev_b64", 1);
    run<23>("v_max3_u32", 1);
    run<24>("v_lshl_add_u64", 1);
    run<25>("cmp_eq_u32 + cndmask", 2);
    run<26>("cmp_lt_u64 + cndmask + xor", 4);
    run<28>("add_co/addc asm", 2);
    return 0;
}
