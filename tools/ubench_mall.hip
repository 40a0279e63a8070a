// Working-set sweep for the Infinity Cache on gfx950: the same in-place read+write sweep (one
// workgroup per 32 KiB tile, 8 bytes per lane, every byte once in and once out) repeated over a
// working set of S MiB, S from well below to far above the 256 MiB of the cache, with the nt and
// the default cache policy.  Reports the rate and the board power read from hwmon while the
// queue is busy.  If a working set that fits is served faster or at lower power than the 2 GiB one,
// a batch worked on in chunks of that size gains what the second pass re-reads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(uint64_t* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0xffffffff, 0x00020000);
}

template <int AUX>
__global__ __launch_bounds__(256) void sweep(uint64_t* d)
{
    const __amdgpu_buffer_rsrc_t r = rsrc(d + (size_t)blockIdx.x * 4096);
    const uint32_t lane = threadIdx.x * 8;
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u32x2_t x = __builtin_amdgcn_raw_buffer_load_b64(r, lane, i * 2048, AUX);
        v[i] = ((uint64_t)x.y << 32) | x.x;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        v[i] += 1;
        const u32x2_t x = {(uint32_t)v[i], (uint32_t)(v[i] >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(x, r, lane, i * 2048, AUX);
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

template <int AUX> void run(uint64_t* d, size_t mib, const char* policy)
{
    const size_t bytes = mib << 20;
    const unsigned grid = (unsigned)(bytes / 32768);
    const int reps = (int)((192ull << 30) / bytes); // 192 GiB swept per measurement
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; r++)
        sweep<AUX><<<grid, 256>>>(d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++)
        sweep<AUX><<<grid, 256>>>(d);
    CHECK(hipEventRecord(e1));
    std::this_thread::sleep_for(std::chrono::milliseconds(40));
    const double w = board_watts(); // the queue is still busy: 192 GiB take more than 60 ms
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("working set %5zu MiB  %-7s  %6d sweeps  %8.3f ms  %6.2f TB/s (read+write)  %6.0f W\n", mib, policy, reps, ms,
           2.0 * bytes * reps / (ms * 1e-3) / 1e12, w);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main()
{
    uint64_t* d;
    CHECK(hipMalloc(&d, (size_t)2048 << 20));
    CHECK(hipMemset(d, 1, (size_t)2048 << 20));
    const size_t sizes[] = {8, 16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 2048};
    for (int round = 0; round < 2; round++)
        for (size_t s : sizes) {
            run<2>(d, s, "nt");
            run<0>(d, s, "default");
        }
    return 0;
}
