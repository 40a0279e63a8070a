// Memory-pattern micro-benchmark for the transform passes on gfx950: what HBM rate the access
// shapes of the pass kernels can reach without any arithmetic.  All kernels work in place on a
// 2 GiB buffer (batch 4096 x 2^16 u64) and move every byte once in and once out.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// contiguous 32 KiB tile per 256-thread block, 16 B per lane, 8 accesses per thread
__global__ __launch_bounds__(256) void inplace16(uint64_t* d)
{
    ulonglong2* p = reinterpret_cast<ulonglong2*>(d) + (size_t)blockIdx.x * 2048 + threadIdx.x;
    ulonglong2 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = p[i * 256];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i].x += 1; p[i * 256] = v[i]; }
}
// same tile, 8 B per lane, 16 accesses per thread
__global__ __launch_bounds__(256) void inplace8(uint64_t* d)
{
    uint64_t* p = d + (size_t)blockIdx.x * 4096 + threadIdx.x;
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = p[i * 256];
#pragma unroll
    for (int i = 0; i < 16; i++) p[i * 256] = v[i] + 1;
}
// strided tile of the column pass: 256 rows x 16 columns, rows 256 elements apart, 8 B per lane
__global__ __launch_bounds__(256) void strided8(uint64_t* d)
{
    const size_t tile = blockIdx.x;
    uint64_t* p = d + (tile >> 4) * 65536 + (tile & 15) * 16 + (threadIdx.x & 15) + (size_t)(threadIdx.x >> 4) * 256;
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = p[(size_t)i * 16 * 256];
#pragma unroll
    for (int i = 0; i < 16; i++) p[(size_t)i * 16 * 256] = v[i] + 1;
}
// strided tile 256 rows x 16 columns with 16 B per lane (each thread owns two adjacent columns, 32 rows)
__global__ __launch_bounds__(128) void strided16(uint64_t* d)
{
    const size_t tile = blockIdx.x;
    ulonglong2* p = reinterpret_cast<ulonglong2*>(d + (tile >> 4) * 65536 + (tile & 15) * 16 + (size_t)(threadIdx.x >> 3) * 256) + (threadIdx.x & 7);
    ulonglong2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = p[(size_t)i * 16 * 128];
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i].x += 1; p[(size_t)i * 16 * 128] = v[i]; }
}
// strided tile, 256 rows x 32 columns (256-byte row segments), 16 B per lane, 256 threads
__global__ __launch_bounds__(256) void strided16w(uint64_t* d)
{
    const size_t tile = blockIdx.x;
    ulonglong2* p = reinterpret_cast<ulonglong2*>(d + (tile >> 3) * 65536 + (tile & 7) * 32 + (size_t)(threadIdx.x >> 4) * 256) + (threadIdx.x & 15);
    ulonglong2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = p[(size_t)i * 16 * 128];
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i].x += 1; p[(size_t)i * 16 * 128] = v[i]; }
}
// grid-stride streaming copy in place, 16 B per lane
__global__ __launch_bounds__(256) void stream16(uint64_t* d, size_t n16)
{
    ulonglong2* p = reinterpret_cast<ulonglong2*>(d);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        ulonglong2 v = p[i];
        v.x += 1;
        p[i] = v;
    }
}

// one launch, every block sweeps its own set of 32 KiB tiles `reps` times (in place): the
// region is re-read from wherever it stays resident (L2 if small per XCD, else Infinity Cache)
__global__ __launch_bounds__(256) void resident16(uint64_t* d, int tiles_per_block, int reps)
{
    for (int r = 0; r < reps; r++)
        for (int t = 0; t < tiles_per_block; t++) {
            ulonglong2* p = reinterpret_cast<ulonglong2*>(d) + ((size_t)t * gridDim.x + blockIdx.x) * 2048 + threadIdx.x;
            ulonglong2 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = p[i * 256];
#pragma unroll
            for (int i = 0; i < 8; i++) { v[i].x += 1; p[i * 256] = v[i]; }
        }
}

template <class F> void timeit(const char* name, F launch, double bytes)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e9, sum = 0;
    for (int r = 0; r < 5; r++) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
    }
    printf("%-34s best %.3f ms  avg %.3f ms  -> %.2f TB/s (read+write)\n", name, best, sum / 5, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t n = (size_t)4096 * 65536;
    uint64_t* d;
    CHECK(hipMalloc(&d, n * 8));
    CHECK(hipMemset(d, 1, n * 8));
    const double bytes = 2.0 * n * 8;
    timeit("stream16 grid=2048", [&] { stream16<<<2048, 256>>>(d, n / 2); }, bytes);
    timeit("stream16 grid=8192", [&] { stream16<<<8192, 256>>>(d, n / 2); }, bytes);
    timeit("inplace16 tile 32K", [&] { inplace16<<<n / 4096, 256>>>(d); }, bytes);
    timeit("inplace8 tile 32K", [&] { inplace8<<<n / 4096, 256>>>(d); }, bytes);
    timeit("strided8 256x16 (8B/lane)", [&] { strided8<<<n / 4096, 256>>>(d); }, bytes);
    timeit("strided16 256x16 (16B/lane,128thr)", [&] { strided16<<<n / 4096, 128>>>(d); }, bytes);
    timeit("strided16w 256x32 (16B/lane)", [&] { strided16w<<<n / 8192, 256>>>(d); }, bytes);
    // cache residency: the same in-place pass repeated over a small region (launches back to back)
    for (size_t mib : {8, 16, 32, 64, 128, 192, 256, 512, 2048}) {
        const size_t elems = mib << 17; // MiB -> u64 count
        const int reps = (int)(2048 / mib);
        char name[64];
        snprintf(name, sizeof name, "inplace16 region %zu MiB x%d", mib, reps);
        timeit(name, [&] { for (int r = 0; r < reps; r++) inplace16<<<elems / 4096, 256>>>(d); }, bytes);
    }
    for (size_t mib : {16, 32, 64, 128, 192, 224, 512, 2048}) {
        const int grid = 1024, tiles = (int)((mib << 20) / 32768 / grid);
        const int reps = (int)(8192 / mib);
        char name[64];
        snprintf(name, sizeof name, "resident16 %zu MiB x%d one launch", mib, reps);
        timeit(name, [&] { resident16<<<grid, 256>>>(d, tiles, reps); }, 2.0 * (double)(mib << 20) * reps);
    }
    return 0;
}
