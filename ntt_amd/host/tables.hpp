// Host-side table construction: the C++ mirror of what the reference's three
// constructors compute before any transform runs (ntt.rs:90-120 + 239-261,
// ntt_avx2.rs:74-105 + 592-613, ntt_goldilock.rs:48-58 + 192-214).
#pragma once
#include <cstdint>
#include <vector>

#include "utils.hpp"

namespace ntt {
namespace host {

struct MontgomeryTables {
    utils::MontgomeryTransformer<uint64_t> montgomery;
    uint64_t n_inv_montgomery;
    uint64_t psi_inv_montgomery;
    std::vector<uint64_t> powers_psi_bo;     // psi^i * 2^64 mod q at index bit_reverse(i)
    std::vector<uint64_t> powers_psi_inv_bo; // psi^-i * 2^64 mod q at index bit_reverse(i)
};

inline uint32_t ilog2(uint64_t n)
{
    if (n == 0)
        throw Panic("argument of integer logarithm must be positive");
    return 63u - (uint32_t)__builtin_clzll(n);
}

// ntt.rs:90-120 followed by with_precomputes (ntt.rs:239-261).  The i64 instance used by
// TableAVX2 (ntt_avx2.rs:74-105, 592-613) produces the same bit patterns.
// out[bit_reverse(i)] = one * g^i for i < 2^log_n, the table the reference fills with a serial
// chain tmp = tmp * g stored at bit_reverse(i, log_n) (ntt.rs:251-258, ntt_goldilock.rs:205-211).
// The products are canonical residues, so the order of multiplication does not change them: the
// table is built by doubling instead -- out[2^k + j] = out[j] * g^(n / 2^(k+1)) for j < 2^k, because
// bit_reverse(2^k + j) = bit_reverse(j) + n / 2^(k+1) -- which walks memory sequentially (the chain
// scatters its 2^28 writes of the largest table over 2 GiB).
// Odd log_n: the reference's bit_reverse (utils.rs:14-17) is not a reversal and indexes out of bounds,
// so its constructors panic; `any_log2` (the corrected constructors) builds the table of the true
// bit reversal instead, which is what the doubling rule yields for every log_n.
template <class Mul>
inline void fill_bit_reversed_powers(std::vector<uint64_t>& out, uint32_t log_n, uint64_t one, uint64_t g, Mul mul,
                                     bool any_log2 = false)
{
    if (log_n % 2 != 0 && !any_log2)
        throw Panic("index out of bounds: bit_reverse is only a reversal for even log2(n)");
    std::vector<uint64_t> step(log_n ? log_n : 1); // step[k] = g^(n / 2^(k+1))
    uint64_t sq = g;
    for (uint32_t k = log_n; k-- > 0;) {
        step[k] = sq;
        sq = mul(sq, sq);
    }
    out[0] = one;
    for (uint32_t k = 0; k < log_n; k++) {
        const uint64_t half = 1ull << k, s = step[k];
        for (uint64_t j = 0; j < half; j++)
            out[half + j] = mul(out[j], s);
    }
}

// exact_k: the corrected setup (not the reference's behaviour): MontgomeryTransformer::from_modulus_exact,
// and odd log2(n) with the true bit reversal.
inline MontgomeryTables build_montgomery_tables(uint64_t modulus, uint64_t root, uint64_t n, bool exact_k = false)
{
    using utils::i128;
    MontgomeryTables t{exact_k ? utils::MontgomeryTransformer<uint64_t>::from_modulus_exact(modulus)
                               : utils::MontgomeryTransformer<uint64_t>::from_modulus(modulus),
                       0, 0, {}, {}};

    const i128 n_inv_raw = std::get<1>(utils::egcd_i128((i128)n, (i128)modulus));
    const uint64_t n_inv = n_inv_raw < 0 ? (uint64_t)(n_inv_raw + (i128)modulus) : (uint64_t)n_inv_raw;
    const i128 psi_inv_raw = std::get<1>(utils::egcd_i128((i128)root, (i128)modulus));
    const uint64_t psi_inv = psi_inv_raw < 0 ? (uint64_t)(psi_inv_raw + (i128)modulus) : (uint64_t)psi_inv_raw;

    t.n_inv_montgomery = t.montgomery.transform(n_inv);
    t.psi_inv_montgomery = t.montgomery.transform(psi_inv);

    const uint64_t psi_montgomery = t.montgomery.transform(root);
    uint64_t tmp_psi = t.montgomery.transform(1);
    uint64_t tmp_psi_inv = tmp_psi;

    const uint32_t log_n = ilog2(n);
    t.powers_psi_bo.assign(n, 0);
    t.powers_psi_inv_bo.assign(n, 0);
    const auto& M = t.montgomery;
    fill_bit_reversed_powers(t.powers_psi_bo, log_n, tmp_psi, psi_montgomery,
                             [&](uint64_t a, uint64_t b) { return M.mul(a, b); }, exact_k);
    fill_bit_reversed_powers(t.powers_psi_inv_bo, log_n, tmp_psi_inv, t.psi_inv_montgomery,
                             [&](uint64_t a, uint64_t b) { return M.mul(a, b); }, exact_k);
    return t;
}

constexpr uint64_t GOLDILOCKS_Q = 0xffffffff00000001ull;

// reduce_159 (ntt_goldilock.rs:63-79)
inline uint64_t goldilocks_reduce_159(uint64_t low, uint64_t middle, uint64_t high)
{
    uint64_t low_2 = low - high;
    if (low < high)
        low_2 += GOLDILOCKS_Q;
    uint64_t product = middle << 32;
    product -= product >> 32;
    uint64_t res = low_2 + product;
    if (res < product || res >= GOLDILOCKS_Q)
        res -= GOLDILOCKS_Q;
    return res;
}

// mul_reduce (ntt_goldilock.rs:183-190)
inline uint64_t goldilocks_mul_reduce(uint64_t a, uint64_t b)
{
    const utils::u128 x = (utils::u128)a * b;
    const uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
    return goldilocks_reduce_159(lo, hi & 0xffffffffull, hi >> 32);
}

struct GoldilocksTables {
    uint64_t n_inv;
    std::vector<uint64_t> powers_psi_bo;     // plain residues
    std::vector<uint64_t> powers_psi_inv_bo;
};

// TableGoldilock::new generalised over (psi, n): the reference fixes psi = 0xabd0a6e8aa3d8a0e,
// n = 2^16, n_inv = 0xfffeffff00010001 (ntt_goldilock.rs:50-53); n_inv = q - (q-1)/n reproduces
// that constant.  Tables as with_precomputes (ntt_goldilock.rs:192-214).
inline GoldilocksTables build_goldilocks_tables(uint64_t psi, uint64_t n, bool any_log2 = false)
{
    if (n == 0 || (n & (n - 1)) != 0 || (GOLDILOCKS_Q - 1) % n != 0)
        throw Panic("n must be a power of two dividing q - 1");
    GoldilocksTables t{GOLDILOCKS_Q - (GOLDILOCKS_Q - 1) / n, {}, {}};
    const uint64_t psi_inv = utils::mod_exp_u64(psi, GOLDILOCKS_Q - 2, GOLDILOCKS_Q);
    const uint32_t log_n = ilog2(n);
    t.powers_psi_bo.assign(n, 0);
    t.powers_psi_inv_bo.assign(n, 0);
    fill_bit_reversed_powers(t.powers_psi_bo, log_n, 1, psi, goldilocks_mul_reduce, any_log2);
    fill_bit_reversed_powers(t.powers_psi_inv_bo, log_n, 1, psi_inv, goldilocks_mul_reduce, any_log2);
    return t;
}

} // namespace host
} // namespace ntt
