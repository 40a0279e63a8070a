// Host-side precompute helpers: the C++ mirror of the reference's `ntt::utils`
// module (src/utils.rs).  Same names, argument meaning and error behaviour; a Rust
// panic becomes a thrown ntt::Panic.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>

namespace ntt {

// What `panic!` is in the reference.
struct Panic : std::runtime_error {
    explicit Panic(const std::string& what) : std::runtime_error(what) {}
};

namespace utils {

using u128 = unsigned __int128;
using i128 = __int128;

// utils.rs:14-17 -- bit reversal of `n` over `bits` bits.  The shift is an integer
// division, so (like the reference) the result is only a reversal for even `bits`.
inline uint64_t bit_reverse(uint64_t n, uint32_t bits)
{
    const uint32_t shift = (64u - bits) / 2u;
    uint64_t x = n << shift;
    // reverse_bits
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    x = __builtin_bswap64(x);
    return x >> shift;
}

// utils.rs:51-64 -- extended Euclid, returns (gcd, x, y) with gcd = a*x + m*y.
inline std::tuple<i128, i128, i128> egcd_i128(i128 a, i128 m)
{
    i128 r[2] = {a, m}, s[2] = {1, 0}, t[2] = {0, 1};
    while (r[1] != 0) {
        const i128 quot = r[0] / r[1];
        const i128 rn = r[0] - quot * r[1], sn = s[0] - quot * s[1], tn = t[0] - quot * t[1];
        r[0] = r[1]; r[1] = rn;
        s[0] = s[1]; s[1] = sn;
        t[0] = t[1]; t[1] = tn;
    }
    return {r[0], s[0], t[0]};
}

// utils.rs:76-92 (u64/u128 instance)
inline uint64_t mod_exp_u64(uint64_t base, uint64_t exp, uint64_t modulus)
{
    uint64_t out = 1, acc = base;
    for (; exp > 0; exp >>= 1) {
        if (exp & 1)
            out = (uint64_t)(((u128)out * acc) % modulus);
        acc = (uint64_t)(((u128)acc * acc) % modulus);
    }
    return out;
}

// utils.rs:76-92 (i64/i128 instance)
inline int64_t mod_exp_i64(int64_t base, int64_t exp, int64_t modulus)
{
    int64_t out = 1, acc = base;
    for (; exp > 0; exp >>= 1) {
        if (exp % 2 == 1)
            out = (int64_t)(((i128)out * acc) % modulus);
        acc = (int64_t)(((i128)acc * acc) % modulus);
    }
    return out;
}

// utils.rs:105-119 draws random candidates; the mirror walks 2, 3, 4, ... instead so that
// the result is reproducible.  Returns g = c^((q-1)/n) with g^(n/2) != 1.
inline uint64_t find_nth_unity_root_u64(uint64_t n, uint64_t modulus)
{
    for (uint64_t c = 2; c < modulus; c++) {
        const uint64_t g = mod_exp_u64(c, (modulus - 1) / n, modulus);
        if (mod_exp_u64(g, n / 2, modulus) != 1)
            return g;
    }
    throw Panic("no n-th root of unity found");
}
inline int64_t find_nth_unity_root_i64(int64_t n, int64_t modulus)
{
    return (int64_t)find_nth_unity_root_u64((uint64_t)n, (uint64_t)modulus);
}

// utils.rs:126-193.  O is uint64_t or int64_t; arithmetic is on the bit pattern.
template <class O>
struct MontgomeryTransformer {
    O k;        // utils.rs:130
    O r_square; // utils.rs:133
    O modulus;  // utils.rs:134 (private there)

    // utils.rs:141-162
    static MontgomeryTransformer from_modulus(O modulus)
    {
        auto [gcd, x, y] = egcd_i128((i128)modulus, (i128)1 << 64);
        (void)y;
        if (gcd != 1)
            throw Panic("modulus is not coprime to helper modulus R=2^64");
        // utils.rs:148-152: a negative x is lifted by `modulus` (not 2^64); kept as is.
        const O k = (x < 0) ? (O)(x + (i128)modulus) : (O)x;
        const u128 r_mod_n = ((u128)1 << 64) % (u128)(uint64_t)modulus;
        const O r_square = (O)(uint64_t)((r_mod_n * r_mod_n) % (u128)(uint64_t)modulus);
        return {k, r_square, modulus};
    }

    // Same with k = modulus^-1 mod 2^64 for EVERY odd modulus: utils.rs:148-152 lifts a negative
    // Bezout coefficient by `modulus` instead of 2^64, which leaves k * modulus != 1 (and with it
    // wrong transforms) for about half of all moduli, 12289 and 65537 among them.  Not the
    // reference's behaviour: an opt-in for callers who need those moduli.
    static MontgomeryTransformer from_modulus_exact(O modulus)
    {
        MontgomeryTransformer t = from_modulus(modulus);
        auto [gcd, x, y] = egcd_i128((i128)modulus, (i128)1 << 64);
        (void)gcd;
        (void)y;
        t.k = (O)(uint64_t)(u128)x; // x mod 2^64
        return t;
    }

    // utils.rs:165-178
    O reduce(u128 a) const
    {
        const u128 mask = ~(uint64_t)0;
        const u128 m = ((a & mask) * (u128)(uint64_t)k) & mask;
        const u128 m_n = m * (u128)(uint64_t)modulus;
        const uint64_t y = (uint64_t)((a - m_n) >> 64);
        return (O)((a < m_n) ? y + (uint64_t)modulus : y);
    }
    O transform(O a) const { return reduce((u128)(uint64_t)a * (u128)(uint64_t)r_square); } // utils.rs:181-183
    O mul(O a, O b) const { return reduce((u128)(uint64_t)a * (u128)(uint64_t)b); }         // utils.rs:185-187

    bool operator==(const MontgomeryTransformer& o) const
    {
        return k == o.k && r_square == o.r_square && modulus == o.modulus;
    }
};

} // namespace utils
} // namespace ntt
