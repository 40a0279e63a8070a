// How fast can a pageable host slice reach the device and come back?  Compares, for one buffer:
//   (a) hipMemcpy from/to pageable memory (what the runtime does by itself),
//   (b) hipHostRegister in place, async copies, hipHostUnregister,
//   (c) staging through two pinned buffers filled by host threads, copies overlapped.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_hostcopy.hip -o tools/ubench_hostcopy -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                                    \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_copy(char* dst, const char* src, size_t bytes, int threads)
{
    if (threads <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = (bytes + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        const size_t off = (size_t)t * per;
        if (off >= bytes)
            break;
        const size_t len = std::min(per, bytes - off);
        pool.emplace_back([=] { std::memcpy(dst + off, src + off, len); });
    }
    for (auto& th : pool)
        th.join();
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? (size_t)std::atol(argv[1]) : 2048;
    const size_t bytes = mib << 20;
    char* host = static_cast<char*>(std::aligned_alloc(1 << 21, bytes < (1u << 21) ? (1u << 21) : bytes));
    std::memset(host, 1, bytes);
    void* dev = nullptr;
    CK(hipMalloc(&dev, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipMemcpy(dev, host, 1 << 20, hipMemcpyHostToDevice)); // warm up

    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        CK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        double t2 = now();
        std::printf("(a) pageable  %zu MiB: H2D %.1f GB/s  D2H %.1f GB/s  total %.1f ms\n", mib, bytes / (t1 - t0) / 1e9,
                    bytes / (t2 - t1) / 1e9, (t2 - t0) * 1e3);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        hipError_t re = hipHostRegister(host, bytes, hipHostRegisterDefault);
        if (re != hipSuccess) {
            std::printf("(b) hipHostRegister: %s -- skipped\n", hipGetErrorString(re));
            (void)hipGetLastError();
            break;
        }
        double t1 = now();
        CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t3 = now();
        CK(hipHostUnregister(host));
        double t4 = now();
        std::printf("(b) register  %zu MiB: register %.1f ms  H2D %.1f GB/s  D2H %.1f GB/s  unregister %.1f ms  total %.1f ms\n",
                    mib, (t1 - t0) * 1e3, bytes / (t2 - t1) / 1e9, bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3, (t4 - t0) * 1e3);
    }
    for (int threads : {1, 4, 8, 16}) {
        for (size_t chunk_mib : {16, 64}) {
            const size_t chunk = chunk_mib << 20;
            char* pin[2];
            hipEvent_t done[2];
            for (int i = 0; i < 2; i++) {
                CK(hipHostMalloc(reinterpret_cast<void**>(&pin[i]), chunk, hipHostMallocDefault));
                CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
            }
            double t0 = now();
            size_t k = 0;
            for (size_t off = 0; off < bytes; off += chunk, k++) {
                const size_t len = std::min(chunk, bytes - off);
                const int b = (int)(k & 1);
                if (k >= 2)
                    CK(hipEventSynchronize(done[b]));
                par_copy(pin[b], host + off, len, threads);
                CK(hipMemcpyAsync(static_cast<char*>(dev) + off, pin[b], len, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(done[b], s));
            }
            CK(hipStreamSynchronize(s));
            double t1 = now();
            // way back: copy chunk k to pinned, then host threads move it out while chunk k+1 is in flight
            k = 0;
            size_t prev_off = 0, prev_len = 0;
            int prev_b = -1;
            for (size_t off = 0; off < bytes; off += chunk, k++) {
                const size_t len = std::min(chunk, bytes - off);
                const int b = (int)(k & 1);
                CK(hipMemcpyAsync(pin[b], static_cast<char*>(dev) + off, len, hipMemcpyDeviceToHost, s));
                CK(hipEventRecord(done[b], s));
                if (prev_b >= 0) {
                    CK(hipEventSynchronize(done[prev_b]));
                    par_copy(host + prev_off, pin[prev_b], prev_len, threads);
                }
                prev_b = b;
                prev_off = off;
                prev_len = len;
            }
            CK(hipEventSynchronize(done[prev_b]));
            par_copy(host + prev_off, pin[prev_b], prev_len, threads);
            double t2 = now();
            std::printf("(c) staged    %zu MiB, %2d threads, %2zu MiB chunks: H2D %.1f GB/s  D2H %.1f GB/s  total %.1f ms\n", mib,
                        threads, chunk_mib, bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9, (t2 - t0) * 1e3);
            for (int i = 0; i < 2; i++) {
                CK(hipHostFree(pin[i]));
                CK(hipEventDestroy(done[i]));
            }
        }
    }
    CK(hipFree(dev));
    std::free(host);
    return 0;
}
