// Host-side memory checker run (AddressSanitizer on host code only): drives the host entry points of
// libnttx that allocate, stage and thread -- plan creation from a root (host-built and device-built
// tables), table read-back, small and streamed host batches, the product, the entry point that
// allocates nothing, the split transform on virtual shards, error paths -- and checks round trips.
// Build and run: see tools/asan_host_paths.sh.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../include/nttx.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, nttx_last_error()); failures++; } } while (0)

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q)
{
    uint64_t r = 1;
    for (; e; e >>= 1, a = mulmod(a, a, q))
        if (e & 1)
            r = mulmod(r, a, q);
    return r;
}
static std::vector<uint64_t> residues(uint64_t q, size_t count, uint64_t seed)
{
    std::vector<uint64_t> v(count);
    uint64_t s = seed * 0x9e3779b97f4a7c15ull + 1;
    for (auto& x : v) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x = s % q;
    }
    return v;
}

static void exercise(int kind, uint64_t q, uint64_t psi, uint64_t n, const char* what)
{
    nttx_plan* plan = nullptr;
    CHECK(nttx_plan_create_from_root(kind, q, psi, n, nullptr, 0, &plan) == NTTX_OK);
    if (!plan)
        return;
    std::printf("%s: policy %s\n", what, nttx_plan_policy(plan));
    std::vector<uint64_t> fwd(n), inv(n);
    CHECK(nttx_plan_tables(plan, fwd.data(), inv.data()) == NTTX_OK);
    // batches: one vector, a small batch (direct copy), a streamed batch (>= 16 MiB: pinned chunks + copy threads)
    for (size_t batch : {(size_t)1, (size_t)5, (size_t)((48u << 20) / (n * 8) + 3)}) {
        std::vector<uint64_t> x = residues(q, n * batch, 7 + batch), y = x;
        CHECK(nttx_forward_host(plan, y.data(), batch) == NTTX_OK);
        CHECK(y != x);
        CHECK(nttx_backward_host(plan, y.data(), batch) == NTTX_OK);
        CHECK(y == x);
        CHECK(nttx_forward_host_lazy(plan, y.data(), batch) == NTTX_OK);
        CHECK(nttx_backward_host_lazy(plan, y.data(), batch) == NTTX_OK);
        for (size_t i = 0; i < y.size(); i++)
            if (y[i] % q != x[i]) { CHECK(!"lazy round trip"); break; }
    }
    { // product of (1 + X) and (1 - X) = 1 - X^2, two pairs; then a streamed product
        std::vector<uint64_t> a(2 * n, 0), b(2 * n, 0);
        for (int k = 0; k < 2; k++) { a[k * n] = 1; a[k * n + 1] = 1; b[k * n] = 1; b[k * n + 1] = q - 1; }
        CHECK(nttx_negacyclic_mul_host(plan, a.data(), b.data(), 2) == NTTX_OK);
        CHECK(a[0] == 1 && a[1] == 0 && a[2] == q - 1 && a[n] == 1 && a[n + 2] == q - 1);
        const size_t big = (40u << 20) / (n * 8) + 1;
        std::vector<uint64_t> c = residues(q, n * big, 3), d(n * big, 0), c0 = c;
        for (size_t k = 0; k < big; k++) d[k * n] = 1; // times the constant 1
        CHECK(nttx_negacyclic_mul_host(plan, c.data(), d.data(), big) == NTTX_OK);
        CHECK(c == c0);
    }
    { // the entry point that allocates nothing, from two threads at once
        auto work = [&](uint64_t seed) {
            const size_t batch = (8u << 20) / (n * 8) ? (8u << 20) / (n * 8) : 1;
            std::vector<uint64_t> x = residues(q, n * batch, seed), y = x;
            CHECK(nttx_transform_host_noalloc(plan, y.data(), batch, 0) == NTTX_OK);
            CHECK(nttx_transform_host_noalloc(plan, y.data(), batch, 1) == NTTX_OK);
            CHECK(y == x);
        };
        std::thread t1(work, 11), t2(work, 12);
        t1.join();
        t2.join();
        std::vector<uint64_t> too_big((40u << 20) / 8);
        CHECK(nttx_transform_host_noalloc(plan, too_big.data(), too_big.size() / n, 0) == NTTX_EINVAL);
    }
    { // one plan, four host threads at once: small calls and a streamed one
        auto work = [&](uint64_t seed, size_t batch) {
            std::vector<uint64_t> x = residues(q, n * batch, seed), y = x;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(nttx_forward_host(plan, y.data(), batch) == NTTX_OK);
                CHECK(nttx_backward_host(plan, y.data(), batch) == NTTX_OK);
            }
            CHECK(y == x);
        };
        const size_t streamed = (20u << 20) / (n * 8) + 1;
        std::thread t1(work, 31, 1), t2(work, 32, 2), t3(work, 33, streamed), t4(work, 34, 3);
        t1.join();
        t2.join();
        t3.join();
        t4.join();
    }
    if (n >= (1u << 16)) { // split transform on virtual shards of the one device
        for (int shards : {2, 4}) {
            std::vector<uint64_t> x = residues(q, n, 21 + shards), y = x, z = x;
            CHECK(nttx_forward_distributed(plan, y.data(), shards) == NTTX_OK);
            CHECK(nttx_forward_host(plan, z.data(), 1) == NTTX_OK);
            CHECK(y == z);
            CHECK(nttx_backward_distributed(plan, y.data(), shards) == NTTX_OK);
            CHECK(y == x);
        }
    }
    // error paths: nothing may be touched
    CHECK(nttx_forward_host(plan, nullptr, 1) == NTTX_EINVAL);
    CHECK(nttx_forward_host(nullptr, fwd.data(), 1) == NTTX_EINVAL);
    CHECK(nttx_forward_distributed(plan, fwd.data(), 3) == NTTX_EINVAL);
    nttx_plan_destroy(plan);
}

int main()
{
    std::printf("%s\n", nttx_version());
    const uint64_t q61 = 0x1fffffffffe00001ull, qgl = 0xffffffff00000001ull;
    const uint64_t psi61_2_21 = powmod(0x0b4f9cb5c38b40b1ull, 1, q61); // a primitive 2^21-th root of unity (the reference's psi of n = 2^16 is its 16th power)
    auto psi61 = [&](uint64_t n) { return powmod(psi61_2_21, (1ull << 20) / n, q61); };
    auto psigl = [&](uint64_t n) { return powmod(7, (qgl - 1) / (2 * n), qgl); };
    exercise(NTTX_MONT64, q61, psi61(1 << 16), 1 << 16, "61-bit prime, 2^16");
    exercise(NTTX_MONT64, q61, psi61(1 << 10), 1 << 10, "61-bit prime, 2^10");
    exercise(NTTX_GOLDILOCKS, qgl, psigl(1 << 16), 1 << 16, "Goldilocks, 2^16");
    exercise(NTTX_GOLDILOCKS, qgl, psigl(1 << 20), 1 << 20, "Goldilocks, 2^20 (tables generated on the device)");
    exercise(NTTX_MONT64, qgl, psigl(1 << 12), 1 << 12, "Goldilocks prime as a Montgomery table, 2^12");
    { // creation errors
        nttx_plan* p = nullptr;
        CHECK(nttx_plan_create_from_root(NTTX_MONT64, q61 - 1, 3, 1 << 10, nullptr, 0, &p) == NTTX_EINVAL && p == nullptr);
        CHECK(nttx_plan_create_from_root(NTTX_MONT64, q61, 3, 1 << 10, nullptr, 0, &p) == NTTX_EINVAL && p == nullptr);
        CHECK(nttx_plan_create_from_root(NTTX_GOLDILOCKS, qgl, psigl(1 << 9), 1 << 9, nullptr, 0, &p) == NTTX_EINVAL && p == nullptr);
    }
    if (failures == 0)
        std::printf("host paths clean\n");
    return failures ? 1 : 0;
}
