// Host-side unit test of ntt_amd/csrc/arith.hpp: every arithmetic policy is run on the CPU
// (the header is __host__ __device__) against plain unsigned __int128 arithmetic, including a
// whole radix-2 transform in the reference's loop order (ntt.rs:129-171, :185-229).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <utility>
#include <vector>

#include "../../ntt_amd/csrc/arith.hpp"
#include "../../ntt_amd/host/tables.hpp"

using namespace nttx;
using u128 = unsigned __int128;
static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            if (failures < 20)                                               \
                std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
            failures++;                                                      \
        }                                                                    \
    } while (0)

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }

static Shoup61Ctx make_shoup_ctx(uint64_t q, uint64_t n_inv_plain, uint64_t s1_plain)
{
    return make_shoup61_ctx(q, n_inv_plain, s1_plain, 0);
}

// Single butterflies of the lazy policies on values at the edges of their documented ranges:
// outputs stay in range (as exact integers, no wrap) and are congruent to the plain butterfly.
template <bool NEAR>
static void shoup_butterfly_bounds(uint64_t q)
{
    using A = Shoup61T<NEAR>;
    std::mt19937_64 rng(q + NEAR);
    Shoup61Ctx c = make_shoup_ctx(q, 1, 1);
    const u128 two64 = (u128)1 << 64;
    // CT takes and leaves [0, 8q) (general) or any u64 (NEAR); GS takes and leaves [0, 4q)
    const uint64_t ct_hi = NEAR ? ~0ull : 8 * q - 1, gs_hi = 4 * q - 1;
    auto pick = [&](uint64_t hi, int it) -> uint64_t {
        const uint64_t e[] = {0, 1, q - 1, q, q + 1, 2 * q - 1, 2 * q, 4 * q - 1, 4 * q, 4 * q + 1, (1ull << 62) - 1,
                              1ull << 62, (1ull << 63) - 1, 1ull << 63, (1ull << 63) + 1, hi - 1, hi};
        uint64_t x = (it & 1) ? e[rng() % (sizeof(e) / sizeof(e[0]))] : rng();
        if (x > hi)
            x = hi == ~0ull ? x : x % (hi + 1);
        return x;
    };
    for (int it = 0; it < 400000; it++) {
        const uint64_t w = it < 4 ? (uint64_t[]){0, 1, q - 1, q / 2}[it] : rng() % q;
        const typename A::Tw t = A::make(TwPair{w, (uint64_t)(((u128)w << 64) / q)});
        {
            const uint64_t a0 = pick(ct_hi, it), b0 = pick(ct_hi, it >> 1);
            uint64_t a = a0, b = b0;
            A::ct(a, b, t, c);
            const uint64_t v = mulmod(b0 % q, w, q);
            CHECK(a % q == (uint64_t)(((u128)(a0 % q) + v) % q));
            CHECK(b % q == (uint64_t)(((u128)(a0 % q) + q - v) % q));
            CHECK(a <= ct_hi && b <= ct_hi);
            CHECK(A::finish(a, c) == a % q && A::finish(b, c) == b % q);
        }
        {
            const uint64_t a0 = pick(gs_hi, it), b0 = pick(gs_hi, it >> 1);
            uint64_t a = a0, b = b0;
            A::gs(a, b, t, c);
            CHECK(a % q == (uint64_t)(((u128)a0 + b0) % q));
            CHECK(b % q == mulmod((uint64_t)(((u128)a0 + 4 * (u128)q - b0) % q), w, q));
            CHECK(a <= gs_hi && b <= gs_hi);
            CHECK(A::canon4(a, c) == a % q && A::canon4(b, c) == b % q);
            a = a0;
            b = b0;
            A::gs_last(a, b, t, c);
            CHECK(a == mulmod((uint64_t)(((u128)a0 + b0) % q), c.scale.w, q));
            CHECK(b == mulmod((uint64_t)(((u128)a0 + 4 * (u128)q - b0) % q), c.scale1.w, q));
        }
        const uint64_t x = rng();
        CHECK(A::condition_input(x, c) == x % q);
        (void)two64;
    }
}

static void shoup_mul_bounds(uint64_t q)
{
    std::mt19937_64 rng(q);
    Shoup61Ctx c = make_shoup_ctx(q, 1, 1);
    const uint64_t ys[] = {0, 1, q - 1, q, 2 * q, 4 * q - 1, 8 * q - 1, ~0ull, 0xffffffffull, 0x100000000ull};
    for (int it = 0; it < 200000; it++) {
        uint64_t w = it < 4 ? (uint64_t[]){0, 1, q - 1, q / 2}[it] : rng() % q;
        uint64_t y = it % 16 < 10 ? ys[it % 16] : rng();
        Shoup61::Tw t = Shoup61::make(TwPair{w, (uint64_t)(((u128)w << 64) / q)});
        uint64_t r = Shoup61::mul(y, t, c);
        CHECK(r < 4 * q);
        CHECK(r % q == mulmod(y % q, w, q));
        uint64_t acc = rng() % (4 * q);
        CHECK(Shoup61::mul_add(acc, y, t, c) == acc + r);
    }
}


// Scaling by 2^-k without a quotient (mul_pow2_inv) and the group forms that hide the last
// conditional subtraction behind one comparison (gs_last_stage, finish_group): for every input,
// including those built to land at q or above before the screen, the results are the canonical
// residues the one-at-a-time routines give.
template <bool NEAR>
static void shoup_group_forms(uint64_t q, uint32_t k)
{
    using A = Shoup61T<NEAR>;
    std::mt19937_64 rng(q + k);
    const uint64_t n_inv = ntt::utils::mod_exp_u64(ntt::utils::mod_exp_u64(2, k, q), q - 2, q); // 2^-k (q prime here)
    const uint64_t s1 = rng() % q;
    const Shoup61Ctx c = make_shoup_ctx(q, n_inv, s1);
    CHECK(c.scale_shift == k);
    Shoup61Ctx general = c; // the same constants through the product by a quotient
    general.scale_shift = 0;
    const uint64_t two_k = 1ull << k;
    for (int it = 0; it < 300000; it++) {
        // x * 2^-k for x below 8q (a sum of two values below 4q): random, edges, and the largest
        // inputs, whose results reach past q; the result stays below q + x / 2^k < 2q
        const uint64_t top = 8 * q - 1; // q < 2^61
        uint64_t x = rng() % top;
        if (it % 4 == 1)
            x = (uint64_t[]){0, 1, two_k - 1, two_k, q - 1, q, q + 1, 4 * q, top - 1, top}[rng() % 10];
        if (it % 4 == 2)
            x = top - (rng() % (two_k << 3));
        const uint64_t r = A::mul_pow2_inv(x, c);
        CHECK(r < 2 * q);
        CHECK(r % q == mulmod(x % q, n_inv, q));
    }
    // whole groups: 16 registers, pairs (i, i + 8)
    int slow = 0;
    for (int it = 0; it < 100000; it++) {
        uint64_t x[16], want[16], y[16];
        for (int i = 0; i < 16; i++)
            x[i] = rng() % (4 * q);
        if (it % 8 == 0) { // sums at the top of the range: mul_pow2_inv overshoots q for some of these
            x[0] = 4 * q - 1 - rng() % 1000;
            x[8] = 4 * q - 1 - rng() % 1000;
        }
        if (it % 8 == 1) { // a difference of zero and a sum of zero
            x[1] = x[9] = rng() % (4 * q);
            x[2] = x[10] = 0;
        }
        const typename A::Tw unused = A::make(TwPair{1, 0});
        for (int i = 0; i < 8; i++) {
            want[i] = mulmod((uint64_t)(((u128)x[i] + x[i + 8]) % q), n_inv, q);
            want[i + 8] = mulmod((uint64_t)(((u128)x[i] + 4 * (u128)q - x[i + 8]) % q), s1, q);
        }
        for (int i = 0; i < 16; i++) y[i] = x[i];
        A::template gs_last_stage<16>(y, unused, c);
        for (int i = 0; i < 16; i++) CHECK(y[i] == want[i]);
        for (int i = 0; i < 16; i++) y[i] = x[i];
        A::template gs_last_stage<16>(y, unused, general);
        for (int i = 0; i < 16; i++) CHECK(y[i] == want[i]);
        for (int i = 0; i < 8; i++) { // one at a time, both contexts
            uint64_t a = x[i], b = x[i + 8];
            A::gs_last(a, b, unused, c);
            CHECK(a == want[i] && b == want[i + 8]);
            a = x[i]; b = x[i + 8];
            A::gs_last(a, b, unused, general);
            CHECK(a == want[i] && b == want[i + 8]);
        }
        // end of the forward transform: any u64 (NEAR) / [0, 8q) (general); every fourth group holds a
        // value that fold61 leaves at q or above (a multiple of 2^61 minus less than 7 c8, top bits set)
        for (int i = 0; i < 16; i++)
            x[i] = NEAR ? rng() : rng() % (8 * q);
        if (NEAR && it % 4 == 0) {
            const uint64_t kk = 1 + rng() % 7;
            x[rng() % 16] = (kk << 61) - 1 - rng() % (kk * c.c8);
            slow++;
        }
        for (int i = 0; i < 16; i++) y[i] = x[i];
        A::template finish_group<16>(y, c);
        for (int i = 0; i < 16; i++) CHECK(y[i] == x[i] % q);
    }
    (void)slow;
}

// Whole transform with a policy, radix-2, reference loop order; tables in the policy's format.
template <class A>
static void forward(std::vector<uint64_t>& a, const std::vector<typename A::TwElem>& tw, const typename A::Ctx& c)
{
    const size_t n = a.size();
    for (auto& x : a)
        x = A::condition_input(x, c);
    size_t t = n, m = 1;
    while (m < n) {
        t /= 2;
        for (size_t i = 0; i < m; i++) {
            typename A::Tw s = A::load(tw.data(), m + i, c);
            for (size_t j = 2 * i * t; j < 2 * i * t + t; j++)
                A::ct(a[j], a[j + t], s, c);
        }
        m *= 2;
    }
    for (auto& x : a)
        x = A::finish(x, c);
}

template <class A>
static void backward(std::vector<uint64_t>& a, const std::vector<typename A::TwElem>& tw, const typename A::Ctx& c)
{
    const size_t n = a.size();
    for (auto& x : a)
        x = A::condition_input(x, c);
    size_t t = 1, m = n;
    while (m > 1) {
        size_t j1 = 0, h = m / 2;
        for (size_t i = 0; i < h; i++) {
            typename A::Tw s = A::load(tw.data(), h + i, c);
            for (size_t j = j1; j < j1 + t; j++) {
                if (m == 2)
                    A::gs_last(a[j], a[j + t], s, c);
                else
                    A::gs(a[j], a[j + t], s, c);
            }
            j1 += 2 * t;
        }
        t *= 2;
        m /= 2;
    }
}

// Shoup61::pointwise is REDC: for operands below 2q (the lazy end of a forward transform) the result
// is still the canonical a * b * 2^-64 mod q.
static void pointwise_lazy_operands(uint64_t q)
{
    Shoup61Ctx sc = make_shoup61_ctx(q, 1, 1, ntt::utils::MontgomeryTransformer<uint64_t>::from_modulus_exact(q).k);
    const uint64_t r_inv = ntt::utils::mod_exp_u64((uint64_t)((((u128)1) << 64) % q), q - 2, q); // 2^-64 mod q (q prime here)
    std::mt19937_64 rng(q);
    const uint64_t edge[] = {0, 1, q - 1, q, q + 1, 2 * q - 1};
    for (int it = 0; it < 200000; it++) {
        const uint64_t a = it % 3 ? rng() % (2 * q) : edge[rng() % 6], b = it % 5 ? rng() % (2 * q) : edge[rng() % 6];
        const uint64_t got = Shoup61::pointwise(a, b, sc);
        CHECK(got == mulmod(mulmod(a % q, b % q, q), r_inv, q));
    }
}

// plain-arithmetic reference of the same transform (canonical residues)
static void ref_forward(std::vector<uint64_t>& a, const std::vector<uint64_t>& w, uint64_t q)
{
    const size_t n = a.size();
    for (auto& x : a) x %= q;
    size_t t = n, m = 1;
    while (m < n) {
        t /= 2;
        for (size_t i = 0; i < m; i++)
            for (size_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                uint64_t u = a[j], v = mulmod(a[j + t], w[m + i], q);
                a[j] = (uint64_t)(((u128)u + v) % q);
                a[j + t] = (uint64_t)(((u128)u + q - v) % q);
            }
        m *= 2;
    }
}

// corrected: the constructor that also takes odd log2 n (tables at the true bit reversal)
static void transform_tests(uint64_t q, uint64_t psi, size_t n, bool corrected = false)
{
    auto mt = ntt::host::build_montgomery_tables(q, psi, n, corrected);
    const auto& M = mt.montgomery;
    std::vector<uint64_t> plain_f(n), plain_b(n);
    for (size_t i = 0; i < n; i++) {
        plain_f[i] = M.reduce((u128)mt.powers_psi_bo[i]);
        plain_b[i] = M.reduce((u128)mt.powers_psi_inv_bo[i]);
    }
    {
        // entry bit_reverse(i) of the forward table is psi^i, entry for entry its inverse in the other
        const uint32_t bits = ntt::host::ilog2(n);
        uint64_t pw = 1;
        for (size_t i = 0; i < n; i++) {
            size_t r = 0;
            for (uint32_t b = 0; b < bits; b++)
                r |= ((i >> b) & 1) << (bits - 1 - b);
            CHECK(plain_f[r] == pw);
            CHECK(mulmod(plain_f[r], plain_b[r], q) == 1);
            pw = mulmod(pw, psi, q);
        }
    }
    std::mt19937_64 rng(n ^ q);
    std::vector<uint64_t> in(n);
    for (auto& x : in) x = rng();
    in[0] = ~0ull; in[1] = q; in[2] = q - 1; in[3] = 0;
    std::vector<uint64_t> want = in;
    ref_forward(want, plain_f, q);

    // strict Montgomery (both widths) on Montgomery-form tables
    MontCtx mc{q, M.k, (uint64_t)((((u128)1) << 64) % q), mt.n_inv_montgomery};
    {
        std::vector<uint64_t> a = in;
        forward<Mont<true>>(a, mt.powers_psi_bo, mc);
        CHECK(a == want);
        backward<Mont<true>>(a, mt.powers_psi_inv_bo, mc);
        for (size_t i = 0; i < n; i++) CHECK(a[i] == in[i] % q);
    }
    if (q < (1ull << 63)) {
        std::vector<uint64_t> a = in;
        forward<Mont<false>>(a, mt.powers_psi_bo, mc);
        CHECK(a == want);
        backward<Mont<false>>(a, mt.powers_psi_inv_bo, mc);
        for (size_t i = 0; i < n; i++) CHECK(a[i] == in[i] % q);
    }
    if (q < (1ull << 61)) {
        auto pair_of = [&](uint64_t w) { return TwPair{w, (uint64_t)(((u128)w << 64) / q)}; };
        std::vector<TwPair> f2(n), b2(n);
        for (size_t i = 0; i < n; i++) { f2[i] = pair_of(plain_f[i]); b2[i] = pair_of(plain_b[i]); }
        const uint64_t n_inv = M.reduce((u128)mt.n_inv_montgomery);
        Shoup61Ctx sc = make_shoup_ctx(q, n_inv, mulmod(plain_b[1], n_inv, q));
        std::vector<uint64_t> a = in;
        forward<Shoup61>(a, f2, sc);
        CHECK(a == want);
        backward<Shoup61>(a, b2, sc);
        for (size_t i = 0; i < n; i++) CHECK(a[i] == in[i] % q);
        if (shoup61_is_near(q)) {
            a = in;
            forward<Shoup61Near>(a, f2, sc);
            CHECK(a == want);
            backward<Shoup61Near>(a, b2, sc);
            for (size_t i = 0; i < n; i++) CHECK(a[i] == in[i] % q);
        }
    }
}

static void goldilocks_tests()
{
    const uint64_t q = Goldilocks::Q;
    GoldilocksCtx c{0x1234567812345678ull % q, 0, 0};
    std::mt19937_64 rng(5);
    // values where a carry, a borrow or a compare against q is on the edge
    std::vector<uint64_t> edge = {0, 1, 2, q - 2, q - 1, q, q + 1, q + 2, ~0ull, ~0ull - 1, 0xffffffffull, 0xfffffffeull,
                                  0x100000000ull, 0x100000001ull, 0xffffffff00000000ull, 0xfffffffeffffffffull,
                                  0xfffffffe00000001ull, 0x00000001ffffffffull, 0x8000000000000000ull, 0x7fffffff80000001ull};
    auto pick = [&](int it, int salt) -> uint64_t {
        switch ((it + salt) % 4) {
        case 0: return edge[rng() % edge.size()];
        case 1: return edge[rng() % edge.size()] + (int64_t)(rng() % 5) - 2;
        case 2: return (rng() % 3 == 0) ? (rng() & 0xffffffffull) : (rng() | 0xffffffff00000000ull);
        default: return rng();
        }
    };
    for (int it = 0; it < 400000; it++) {
        const uint64_t a = pick(it, 0), b = pick(it / 4, 1), w = pick(it / 16, 2);
        CHECK(Goldilocks::mul_strict(a, b) == mulmod(a % q, b % q, q));
        uint32_t top;
        const uint64_t m = Goldilocks::mul_fast(a, b, top);
        CHECK(top == (uint32_t)(m >> 32));
        CHECK(m == Goldilocks::mul_strict(a, b) || top == 0xffffffffu);
        const uint64_t s = Goldilocks::add_fast(a, b, top);
        CHECK(s == Goldilocks::add_strict(a, b) || top == 0xffffffffu);
        const uint64_t d = Goldilocks::sub_fast(a, b, top);
        CHECK(d == Goldilocks::sub_gs_strict(a, b));
        CHECK(d == Goldilocks::sub_ct_strict(a, b) || top == 0xffffffffu);
        // whole butterflies against the reference's statements
        Goldilocks::Tw tw{w};
        uint64_t x = a, y = b;
        Goldilocks::ct(x, y, tw, c);
        uint64_t v = Goldilocks::mul_strict(b, w);
        CHECK(x == Goldilocks::add_strict(a, v) && y == Goldilocks::sub_ct_strict(a, v));
        x = a; y = b;
        Goldilocks::gs(x, y, tw, c);
        CHECK(x == Goldilocks::add_strict(a, b) && y == Goldilocks::mul_strict(Goldilocks::sub_gs_strict(a, b), w));
        x = a; y = b;
        Goldilocks::gs_last(x, y, tw, c);
        CHECK(x == Goldilocks::mul_strict(Goldilocks::add_strict(a, b), c.n_inv));
        CHECK(y == Goldilocks::mul_strict(Goldilocks::sub_gs_strict(a, b), Goldilocks::mul_strict(w, c.n_inv)));
    }
    CHECK(Goldilocks::add_strict(q - 5, 5) == q);  // ntt_goldilock.rs:107: a sum of exactly q stays q
    CHECK(Goldilocks::add_strict(q - 5, 6) == 1);
    CHECK(Goldilocks::sub_ct_strict(q, 0) == 0);    // :120
    CHECK(Goldilocks::sub_gs_strict(q, 0) == q);    // :162-165
}


// Twiddles that are powers of two: the shift forms against the reference's statements with the
// twiddle's canonical value, for each of the 30 exponents the kernels compile in.
template <int E>
static void goldilocks_pow2_case(std::mt19937_64& rng, const std::vector<uint64_t>& edge)
{
    const uint64_t q = Goldilocks::Q;
    const uint64_t w = ntt::utils::mod_exp_u64(2, E, q); // canonical 2^E: q - 2^(E-96) for E >= 96
    GoldilocksCtx c{1, 0, 0};
    auto pick = [&](int it, int salt) -> uint64_t {
        switch ((it + salt) % 5) {
        case 0: return edge[rng() % edge.size()];
        case 1: return edge[rng() % edge.size()] + (int64_t)(rng() % 5) - 2;
        case 2: return (rng() % 3 == 0) ? (rng() & 0xffffffffull) : (rng() | 0xffffffff00000000ull);
        case 3: return rng() % q;
        default: return rng();
        }
    };
    for (int it = 0; it < 60000; it++) {
        uint64_t a = pick(it, 0), b = pick(it / 5, 1);
        if (it % 7 == 0) // a equal to the product: the zero difference of a negative twiddle
            a = Goldilocks::mul_strict(b, E >= 96 ? ntt::utils::mod_exp_u64(2, E - 96, q) : w);
        if (it % 11 == 0)
            b = 0;
        uint32_t top;
        const uint64_t m = Goldilocks::mul_pow2_fast<E % 96>(b, top);
        CHECK(top == (uint32_t)(m >> 32) || (E % 96) / 32 == 2);
        CHECK(m == Goldilocks::mul_strict(b, ntt::utils::mod_exp_u64(2, E % 96, q)) || top == 0xffffffffu);
        uint32_t tops[4];
        uint64_t x = a, y = b, xs = a, ys = b;
        Goldilocks::ct_pow2_fast<E>(x, y, tops);
        Goldilocks::ct_strict(xs, ys, Goldilocks::Tw{w}, c);
        const bool flagged = tops[0] == 0xffffffffu || tops[1] == 0xffffffffu || tops[2] == 0xffffffffu || tops[3] == 0xffffffffu;
        CHECK((x == xs && y == ys) || flagged);
        // backward form: operands in [0, q], which is what every stage after the first of an inverse
        // transform sees (add_strict and mul_reduce leave nothing larger); the forward form above takes any u64
        if (a > q) a = (it & 1) ? q : a % q;
        if (b > q) b = (it & 2) ? q : b % q;
        x = a; y = b; xs = a; ys = b;
        Goldilocks::gs_pow2_fast<E>(x, y, tops);
        Goldilocks::gs_strict(xs, ys, Goldilocks::Tw{w}, c);
        CHECK((x == xs && y == ys) || tops[0] == 0xffffffffu || tops[1] == 0xffffffffu);
    }
}
// the last inverse stage as shifts, for the transform sizes that have a compiled-in form
template <int L>
static void goldilocks_last_stage_case(std::mt19937_64& rng)
{
    const uint64_t q = Goldilocks::Q;
    const uint64_t n_inv = ntt::utils::mod_exp_u64(2, 192 - L, q);
    CHECK(mulmod(n_inv, 1ull << L, q) == 1);
    GoldilocksCtx c{n_inv, L, 0};
    const Goldilocks::Tw s1{ntt::utils::mod_exp_u64(2, 144, q)}; // tw_inv[1]
    const Goldilocks::Tw sf = Goldilocks::scaled(s1, c);
    const uint64_t edge[] = {0, 1, 2, q - 2, q - 1, q, 0xffffffffull, 0x100000000ull, 0xffffffff00000000ull, 0xfffffffeffffffffull};
    for (int it = 0; it < 200000; it++) {
        uint64_t a = it % 3 ? rng() % (q + 1) : edge[rng() % 10], b = it % 5 ? rng() % (q + 1) : edge[rng() % 10];
        if (it % 13 == 0)
            b = a; // zero difference
        if (it % 17 == 0)
            b = q - a; // sum of exactly q
        uint32_t top;
        const uint64_t m = Goldilocks::mul_ninv_fast<L>(a, top);
        CHECK(m == Goldilocks::mul_strict(a, n_inv));
        uint32_t tops[3];
        uint64_t x = a, y = b, xs = a, ys = b;
        Goldilocks::gs_last_pow2_fast<L>(x, y, tops);
        Goldilocks::gs_last_strict(xs, ys, sf, c);
        CHECK((x == xs && y == ys) || tops[0] == 0xffffffffu || tops[1] == 0xffffffffu || tops[2] == 0xffffffffu);
    }
}

template <int... X>
static void goldilocks_pow2_all(std::integer_sequence<int, X...>)
{
    std::mt19937_64 rng(96);
    const uint64_t q = Goldilocks::Q;
    const std::vector<uint64_t> edge = {0, 1, 2, q - 2, q - 1, q, q + 1, q + 2, ~0ull, ~0ull - 1, 0xffffffffull, 0xfffffffeull,
                                        0x100000000ull, 0x100000001ull, 0xffffffff00000000ull, 0xfffffffeffffffffull,
                                        0xfffffffe00000001ull, 0x00000001ffffffffull, 0x8000000000000000ull, 0x7fffffff80000001ull};
    (goldilocks_pow2_case<Goldilocks::pow2_exponent(X + 1, false)>(rng, edge), ...);
    (goldilocks_pow2_case<Goldilocks::pow2_exponent(X + 1, true)>(rng, edge), ...);
    // the exponents are those of the reference's own table (ntt_goldilock.rs:51: psi = 0xabd0a6e8aa3d8a0e, n = 2^16)
    auto t = ntt::host::build_goldilocks_tables(0xabd0a6e8aa3d8a0eull, 1 << 16);
    for (int x = 1; x < 16; x++) {
        CHECK(t.powers_psi_bo[x] == ntt::utils::mod_exp_u64(2, Goldilocks::pow2_exponent(x, false), q));
        CHECK(t.powers_psi_inv_bo[x] == ntt::utils::mod_exp_u64(2, Goldilocks::pow2_exponent(x, true), q));
    }
    CHECK(t.powers_psi_bo[16] == Goldilocks::POW2_ROOT64);
    goldilocks_last_stage_case<16>(rng);
    goldilocks_last_stage_case<20>(rng);
    goldilocks_last_stage_case<24>(rng);
    goldilocks_last_stage_case<28>(rng);
    goldilocks_last_stage_case<8>(rng);
}

int main()
{
    const uint64_t q61 = 0x1fffffffffe00001ull, psi61 = 0x15eb043c7aa2b01full;
    shoup_mul_bounds(q61);
    shoup_mul_bounds(4293918721ull);
    shoup_mul_bounds((1ull << 61) - 1); // not prime; bounds and congruence hold for any odd modulus
    shoup_mul_bounds(7681);
    shoup_butterfly_bounds<false>(q61);
    shoup_butterfly_bounds<true>(q61);
    shoup_butterfly_bounds<false>(4293918721ull);
    shoup_butterfly_bounds<false>((1ull << 61) - 1);
    shoup_butterfly_bounds<true>((1ull << 61) - 1);
    shoup_butterfly_bounds<true>((1ull << 61) - (1ull << 29) + 1); // the smallest modulus the NEAR forms take
    shoup_butterfly_bounds<false>(7681);
    shoup_group_forms<true>(q61, 16);
    shoup_group_forms<false>(q61, 16);
    shoup_group_forms<true>(q61, 12);
    shoup_group_forms<true>(q61, 20);
    shoup_group_forms<false>(4293918721ull, 16);
    shoup_group_forms<false>(998244353ull, 10);
    for (size_t n : {4, 16, 256, 4096}) {
        transform_tests(q61, ntt::utils::mod_exp_u64(psi61, (1 << 16) / n, q61), n);
        transform_tests(Goldilocks::Q, ntt::utils::mod_exp_u64(0xabd0a6e8aa3d8a0eull, (1 << 16) / n, Goldilocks::Q), n);
        transform_tests(4293918721ull, ntt::utils::mod_exp_u64(2004365341ull, (1 << 16) / n, 4293918721ull), n);
    }
    // odd log2 n: the reference's constructors panic, the corrected one builds the true table
    for (size_t n : {8, 32, 512, 2048}) {
        bool threw = false;
        try {
            (void)ntt::host::build_montgomery_tables(q61, ntt::utils::mod_exp_u64(psi61, (1 << 16) / n, q61), n);
        } catch (const ntt::Panic&) {
            threw = true;
        }
        CHECK(threw);
        transform_tests(q61, ntt::utils::mod_exp_u64(psi61, (1 << 16) / n, q61), n, true);
        transform_tests(Goldilocks::Q, ntt::utils::mod_exp_u64(0xabd0a6e8aa3d8a0eull, (1 << 16) / n, Goldilocks::Q), n, true);
    }
    pointwise_lazy_operands(q61);
    pointwise_lazy_operands(4293918721ull);
    goldilocks_tests();
    goldilocks_pow2_all(std::make_integer_sequence<int, 15>{});
    if (failures == 0)
        std::printf("all arithmetic tests passed\n");
    else
        std::printf("%d failures\n", failures);
    return failures ? 1 : 0;
}
