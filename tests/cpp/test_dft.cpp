// The reference's own dft.rs / ntt.rs tests, restated against the C++ host mirror
// (ntt_amd/host/dft.hpp) with the transforms running on the GPU through libnttx.so.
// Needs a GPU; tests/test_gpu_host_mirror.py builds and runs it.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../ntt_amd/host/dft.hpp"

using namespace ntt::dft;
static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            failures++;                                                      \
        }                                                                    \
    } while (0)

static std::vector<uint64_t> random_below(uint64_t q, size_t n, uint64_t seed)
{
    std::mt19937_64 rng(seed);
    std::uniform_int_distribution<uint64_t> d(0, q - 1);
    std::vector<uint64_t> a(n);
    for (auto& x : a)
        x = d(rng);
    return a;
}

// dft.rs:23-42
static void base_version()
{
    auto table = Table<uint64_t>::new_goldilock();
    auto a = random_below(table.q, 1 << 16, 1);
    auto a_cloned = a;
    table.forward_inplace(a);
    CHECK(a != a_cloned);
    table.backward_inplace(a);
    CHECK(a == a_cloned);
}

// dft.rs:45-78
static void avx_correctness()
{
    auto table = Table<uint64_t>::new_();
    auto table_avx = TableAVX2::new_();
    auto a = random_below((uint64_t)table_avx.q, 1 << 16, 2);
    std::vector<int64_t> a_avx2(a.begin(), a.end());
    auto a_cloned = a;
    table.forward_inplace(a);
    table_avx.forward_inplace(a_avx2);
    bool same = true;
    for (size_t i = 0; i < a.size(); i++)
        same &= a[i] == (uint64_t)a_avx2[i];
    CHECK(same);
    table.backward_inplace(a);
    table_avx.backward_inplace(a_avx2);
    same = true;
    for (size_t i = 0; i < a.size(); i++)
        same &= a[i] == (uint64_t)a_avx2[i];
    CHECK(same);
    CHECK(a == a_cloned);
}

// dft.rs:127-152
static void goldilock_correctness()
{
    auto table = Table<uint64_t>::new_goldilock();
    auto table_goldilock = TableGoldilock<uint64_t>::new_();
    auto a_goldilock = random_below(table_goldilock.q, 1 << 16, 3);
    auto a_cloned = a_goldilock;
    auto a = a_goldilock;
    table.forward_inplace(a);
    table_goldilock.forward_inplace(a_goldilock);
    CHECK(a == a_goldilock);
    table.backward_inplace(a);
    table_goldilock.backward_inplace(a_goldilock);
    CHECK(a == a_goldilock);
    CHECK(a == a_cloned);
}

// ntt.rs:279-295 (`2 ^ 61` is XOR there: values in 0..63)
static void test_normal_case()
{
    auto table = Table<uint64_t>::new_();
    auto a = random_below(63, 1 << 16, 4);
    auto a_clone = a;
    table.forward_inplace(a);
    table.backward_inplace(a);
    CHECK(a_clone == a);
}

// utils.rs:200-222 and the doctests at utils.rs:9-13, 42-50
static void utils_tests()
{
    using namespace ntt::utils;
    auto params = MontgomeryTransformer<uint64_t>::from_modulus(0x1fffffffffe00001ull);
    CHECK((params == MontgomeryTransformer<uint64_t>{6917533425689690113ull, 281474708275264ull, 2305843009211596801ull}));
    for (uint64_t val : {1ull, 16777208ull})
        CHECK(val == params.reduce((u128)params.transform(val)));
    CHECK(bit_reverse(2, 2) == 1);
    auto [gcd, x, y] = egcd_i128((i128)0x1fffffffffe00001ull, (i128)1 << 64);
    CHECK(gcd == (i128)0x1fffffffffe00001ull * x + ((i128)1 << 64) * y);
}

static void panics()
{
    bool threw = false;
    try {
        Table<uint64_t>::new_simple(); // odd log2 n: the reference panics in with_precomputes
    } catch (const ntt::Panic&) {
        threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
        auto t = Table<uint64_t>::new_();
        std::vector<uint64_t> too_long(1 << 17);
        t.forward_inplace(too_long);
    } catch (const ntt::Panic&) {
        threw = true;
    }
    CHECK(threw);
}

// A modulus for which the reference derives a wrong k (utils.rs:148-152): rejected as the reference's
// result would be wrong, accepted with the exact k, and then correct against plain arithmetic.
static void exact_k_modulus()
{
    const uint64_t q = 12289, n = 1024;
    const uint64_t psi = ntt::utils::mod_exp_u64(11, (q - 1) / (2 * n), q);
    bool threw = false;
    try {
        Table<uint64_t>::from_prime_and_root(q, psi, n);
    } catch (const ntt::Panic&) {
        threw = true;
    }
    CHECK(threw);
    auto t = Table<uint64_t>::from_prime_and_root(q, psi, n, {}, true);
    std::vector<uint64_t> a(n, 0);
    a[1] = 1; // X: its transform is psi^(2 brev(i) + 1) at output i
    t.forward_inplace(a);
    bool ok = true;
    for (uint64_t i = 0; i < n; i++)
        ok = ok && a[i] == ntt::utils::mod_exp_u64(psi, 2 * ntt::utils::bit_reverse(i, 10) + 1, q);
    CHECK(ok);
    auto b = random_below(q, n, 9), b0 = b;
    t.forward_inplace(b);
    CHECK(b != b0);
    t.backward_inplace(b);
    CHECK(b == b0);
}

// One transform computed by several shards must equal the unsplit transform (all three tables).
static void split_equals_unsplit()
{
    auto table = Table<uint64_t>::new_();
    auto a = random_below(table.q, 1 << 16, 7), b = a;
    table.forward_inplace(a);
    table.forward_inplace_distributed(b.data(), b.size(), 8);
    CHECK(a == b);
    table.backward_inplace_distributed(b.data(), b.size(), 4);
    table.backward_inplace(a);
    CHECK(a == b);

    auto gl = TableGoldilock<uint64_t>::new_();
    auto c = random_below(gl.q, 1 << 16, 8), d = c;
    gl.forward_inplace(c);
    gl.forward_inplace_distributed(d.data(), d.size(), 2);
    CHECK(c == d);

    auto avx = TableAVX2::new_();
    std::vector<int64_t> e(1 << 16), f;
    auto raw = random_below((uint64_t)avx.q, 1 << 16, 9);
    for (size_t i = 0; i < e.size(); i++)
        e[i] = (int64_t)raw[i];
    f = e;
    avx.forward_inplace(e);
    avx.forward_inplace_distributed(f.data(), f.size(), 8);
    CHECK(e == f);

    bool threw = false;
    try {
        table.forward_inplace_distributed(b.data(), b.size(), 3);
    } catch (const ntt::Panic&) {
        threw = true;
    }
    CHECK(threw);
}

// X^(n-1) * X = -1 and (1 + X)^2 = 1 + 2X + X^2 in Z_q[X]/(X^n + 1), all three tables.
static void negacyclic_product()
{
    auto table = Table<uint64_t>::new_();
    const size_t n = 1 << 16;
    std::vector<uint64_t> a(n, 0), b(n, 0);
    a[n - 1] = 1;
    b[1] = 1;
    table.negacyclic_mul_inplace(a, b);
    CHECK(a[0] == table.q - 1);
    for (size_t i = 1; i < n; i++)
        if (a[i] != 0) { CHECK(a[i] == 0); break; }
    CHECK(b[1] == 1 && b[0] == 0); // the host entry point leaves b alone

    auto gl = TableGoldilock<uint64_t>::new_();
    std::vector<uint64_t> c(n, 0);
    c[0] = c[1] = 1;
    auto d = c;
    gl.negacyclic_mul_inplace(c, d);
    CHECK(c[0] == 1 && c[1] == 2 && c[2] == 1 && c[3] == 0);

    auto avx = TableAVX2::new_();
    std::vector<int64_t> e(n, 0), f(n, 0);
    e[n - 1] = 3;
    f[2] = 5;
    avx.negacyclic_mul_inplace(e, f);
    CHECK(e[1] == avx.q - 15 && e[0] == 0);
}

int main()
{
    utils_tests();
    exact_k_modulus();
    negacyclic_product();
    split_equals_unsplit();
    base_version();
    avx_correctness();
    goldilock_correctness();
    test_normal_case();
    panics();
    if (failures == 0)
        std::printf("all host-mirror tests passed\n");
    return failures ? 1 : 0;
}
