// Memory-only models of the 8-stage pass kernels' tile shapes on gfx950: every kernel reads each
// element of a 2 GiB batch (4096 x 2^16 u64) once and writes it once, in place, with the access shapes
// and cache policy (nt) of the real passes and no arithmetic -- what the memory system gives each
// shape.  One workgroup of 256 threads per tile of 4096 elements, 16 elements per thread.
//
//   strided   256 rows x 16 columns of one transform, rows 256 elements apart (first forward pass)
//   xb        the same 256-element block of 16 consecutive transforms: 16 runs of 2 KiB, 512 KiB apart
//             (cross-batch tile of the pass with the last eight stages)
// Access forms of a side (load or store):
//   A8    8 bytes per lane, a wave instruction = four 128-byte segments in four rows / runs
//   C16   16 bytes per lane, a wave instruction = 1 KiB contiguous
//   C8    8 bytes per lane, a wave instruction = 512 bytes contiguous
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int NT = 2;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(uint64_t* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0xffffffff, 0x00020000);
}

enum Form { A8, C16, C8 };

// element offsets (from the tile's first element) of access i of a thread, for a tile made of 16
// runs of 256 elements `run_stride` apart (xb), in the given form
template <int FORM>
__device__ __forceinline__ uint32_t xb_offset(int i, uint32_t t, uint32_t run_stride)
{
    if (FORM == A8) { // thread (sigma = t >> 4, rlow = t & 15), element rlow + 16 i of run sigma
        return (t >> 4) * run_stride + (t & 15) + 16u * i;
    } else if (FORM == C16) { // wave w owns runs 4w..4w+3; access i (0..7): half i & 1 of run 4w + (i >> 1)
        const uint32_t w = t >> 6, lane = t & 63;
        return (4 * w + (i >> 1)) * run_stride + (i & 1) * 128u + 2 * lane;
    } else { // C8: access i (0..15): quarter i & 3 of run 4w + (i >> 2)
        const uint32_t w = t >> 6, lane = t & 63;
        return (4 * w + (i >> 2)) * run_stride + (i & 3) * 64u + lane;
    }
}

template <int LOAD, int STORE, int LAUX = NT, int SAUX = NT>
__global__ __launch_bounds__(256) void xb_kernel(uint64_t* d, uint32_t s0 /* log2 blocks per transform */)
{
    const uint64_t tile = blockIdx.x;
    const uint32_t run_stride = 1u << (s0 + 8);
    uint64_t* base = d + ((((tile >> s0) << 4) << s0) + (tile & ((1ull << s0) - 1))) * 256;
    const __amdgpu_buffer_rsrc_t r = rsrc(base);
    const uint32_t t = threadIdx.x;
    uint64_t v[16];
    if (LOAD == C16) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(r, xb_offset<C16>(i, t, run_stride) * 8, 0, LAUX);
            v[2 * i] = ((uint64_t)x.y << 32) | x.x;
            v[2 * i + 1] = ((uint64_t)x.w << 32) | x.z;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32x2_t x = __builtin_amdgcn_raw_buffer_load_b64(r, xb_offset<LOAD>(i, t, run_stride) * 8, 0, LAUX);
            v[i] = ((uint64_t)x.y << 32) | x.x;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
        v[i] += 1;
    __syncthreads(); // the real kernels exchange through LDS here: every load has landed before any store
    if (STORE == C16) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32x4_t x = {(uint32_t)v[2 * i], (uint32_t)(v[2 * i] >> 32), (uint32_t)v[2 * i + 1], (uint32_t)(v[2 * i + 1] >> 32)};
            __builtin_amdgcn_raw_buffer_store_b128(x, r, xb_offset<C16>(i, t, run_stride) * 8, 0, SAUX);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32x2_t x = {(uint32_t)v[i], (uint32_t)(v[i] >> 32)};
            __builtin_amdgcn_raw_buffer_store_b64(x, r, xb_offset<STORE>(i, t, run_stride) * 8, 0, SAUX);
        }
    }
}

// strided tile, both sides A8 (as the first forward / last backward pass); G > 0: XCD-grouped order
template <int LAUX = NT, int SAUX = NT>
__global__ __launch_bounds__(256) void strided_kernel(uint64_t* d, uint32_t G)
{
    uint64_t tile = blockIdx.x;
    if (G)
        tile = ((uint64_t)(blockIdx.x / (8 * G)) * 8 + blockIdx.x % 8) * G + (blockIdx.x / 8) % G;
    uint64_t* base = d + (tile >> 4) * 65536 + (tile & 15) * 16;
    const __amdgpu_buffer_rsrc_t r = rsrc(base);
    const uint32_t t = threadIdx.x;
    const uint32_t lane = ((t >> 4) * 256 + (t & 15)) * 8; // row t >> 4, column t & 15
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u32x2_t x = __builtin_amdgcn_raw_buffer_load_b64(r, lane, i * 16 * 256 * 8, LAUX);
        v[i] = ((uint64_t)x.y << 32) | x.x;
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
        v[i] += 1;
    __syncthreads();
    // second round's layout: rows 16 (t >> 4) + i
    const uint32_t lane2 = ((t >> 4) * 16 * 256 + (t & 15)) * 8;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u32x2_t x = {(uint32_t)v[i], (uint32_t)(v[i] >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(x, r, lane2, i * 256 * 8, SAUX);
    }
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

// `ubench_tiles power`: every shape runs back to back for 0.7 s while the board power (hwmon) is sampled
// every 20 ms; the plateau (median of the second half) is printed beside the rate.
static bool with_power = false;
static double run_seconds = 0.7;

template <class F> void timeit(const char* name, F launch, double bytes)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; r++) launch();
    CHECK(hipDeviceSynchronize());
    if (with_power) {
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
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < run_seconds) {
            for (int r = 0; r < 8; r++, launches++)
                launch();
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        stop.store(true);
        sampler.join();
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> late(watts.begin() + watts.size() / 2, watts.end());
        std::sort(late.begin(), late.end());
        const double w = late.empty() ? 0 : late[late.size() / 2];
        const double tbs = bytes * launches / (ms * 1e-3) / 1e12;
        printf("%-44s %4d launches  %.3f ms each  -> %.2f TB/s (read+write)  %5.0f W  %5.1f W per TB/s above 360 W\n", name, launches,
               ms / launches, tbs, w, (w - 360) / tbs);
        return;
    }
    float best = 1e9, sum = 0;
    const int reps = 10;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
    }
    printf("%-44s best %.3f ms  avg %.3f ms  -> %.2f TB/s (read+write)\n", name, best, sum / reps, bytes / (sum / reps * 1e-3) / 1e12);
}

// `ubench_tiles aux`: the strided tile (XCD groups of 16) and the cross-batch tile (C8 loads, A8 stores) under every
// pair of cache policies of the buffer instructions (aux bits: 1 = sc0, 2 = nt, 16 = sc1), with board power.
template <int L, int S> void aux_case(uint64_t* d, unsigned grid, double bytes)
{
    char name[64];
    snprintf(name, sizeof name, "strided  load aux %2d  store aux %2d", L, S);
    timeit(name, [&] { strided_kernel<L, S><<<grid, 256>>>(d, 16); }, bytes);
    snprintf(name, sizeof name, "xb C8->A8  load aux %2d  store aux %2d", L, S);
    timeit(name, [&] { xb_kernel<C8, A8, L, S><<<grid, 256>>>(d, 8); }, bytes);
}
template <int L> void aux_row(uint64_t* d, unsigned grid, double bytes)
{
    aux_case<L, 0>(d, grid, bytes);
    aux_case<L, 1>(d, grid, bytes);
    aux_case<L, 2>(d, grid, bytes);
    aux_case<L, 3>(d, grid, bytes);
    aux_case<L, 16>(d, grid, bytes);
    aux_case<L, 18>(d, grid, bytes);
}

int main(int argc, char** argv)
{
    with_power = argc > 1 && argv[1][0] == 'p';
    const size_t n = (size_t)4096 * 65536;
    uint64_t* d;
    CHECK(hipMalloc(&d, n * 8));
    CHECK(hipMemset(d, 1, n * 8));
    const double bytes = 2.0 * n * 8;
    const unsigned grid = (unsigned)(n / 4096);
    if (argc > 1 && argv[1][0] == 'a') {
        with_power = true;
        run_seconds = 0.45;
        aux_row<0>(d, grid, bytes);
        aux_row<1>(d, grid, bytes);
        aux_row<2>(d, grid, bytes);
        aux_row<3>(d, grid, bytes);
        aux_row<16>(d, grid, bytes);
        aux_row<18>(d, grid, bytes);
        return 0;
    }
    for (int round = 0; round < (with_power ? 1 : 2); round++) {
        timeit("strided A8 -> A8, plain order", [&] { strided_kernel<><<<grid, 256>>>(d, 0); }, bytes);
        timeit("strided A8 -> A8, XCD groups of 16", [&] { strided_kernel<><<<grid, 256>>>(d, 16); }, bytes);
        timeit("xb A8 -> C16 (fwd contig as shipped)", [&] { xb_kernel<A8, C16><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb A8 -> C8", [&] { xb_kernel<A8, C8><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb A8 -> A8", [&] { xb_kernel<A8, A8><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb C16 -> A8 (bwd contig as shipped)", [&] { xb_kernel<C16, A8><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb C16 (default policy) -> A8 (nt)", [&] { xb_kernel<C16, A8, 0, NT><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb C16 (default) -> A8 (default)", [&] { xb_kernel<C16, A8, 0, 0><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb C8 (default) -> A8 (default)", [&] { xb_kernel<C8, A8, 0, 0><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb A8 (default) -> C16 (default)", [&] { xb_kernel<A8, C16, 0, 0><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb C8 -> A8", [&] { xb_kernel<C8, A8><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb C16 -> C16", [&] { xb_kernel<C16, C16><<<grid, 256>>>(d, 8); }, bytes);
        timeit("xb C8 -> C8", [&] { xb_kernel<C8, C8><<<grid, 256>>>(d, 8); }, bytes);
        // plain contiguous tiles (s0 = 0 makes the 16 runs adjacent: one 32 KiB run)
        timeit("contiguous 32 KiB tile C16 -> C16", [&] { xb_kernel<C16, C16><<<grid, 256>>>(d, 0); }, bytes);
        timeit("contiguous 32 KiB tile C8 -> C8", [&] { xb_kernel<C8, C8><<<grid, 256>>>(d, 0); }, bytes);
    }
    return 0;
}
