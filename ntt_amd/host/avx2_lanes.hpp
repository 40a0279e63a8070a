// The free functions of the reference's AVX2 module (ntt_avx2.rs:704-813) on four 64-bit lanes held
// in a plain array: what they compute, lane by lane, without the x86 intrinsics.  The device code
// has no use for them (its counterparts are the functions of csrc/arith.hpp); they exist so that
// host code written against the reference's module finds the same names and results.
#pragma once
#include <array>
#include <cstdint>
#include <utility>

namespace ntt {
namespace dft {
namespace avx2 {

using Lanes = std::array<int64_t, 4>;   // the four 64-bit lanes of an __m256i, lane 0 first
using LanesF64 = std::array<double, 4>; // the four lanes of an __m256d

// ntt_avx2.rs:706-716
inline std::array<int64_t, 4> extract_m256i_to_i64(const Lanes& a) { return a; }
// ntt_avx2.rs:720-725
inline std::array<double, 4> extract_m256d_to_f64(const LanesF64& a) { return a; }

// ntt_avx2.rs:728-744: a < b ? all ones : 0 with the lanes read as UNSIGNED integers.  The reference
// assembles it from signed compares by cases on the signs; the four cases add up to the unsigned order.
inline Lanes a_lt_b(const Lanes& a, const Lanes& b)
{
    Lanes out{};
    for (int i = 0; i < 4; i++)
        out[i] = (uint64_t)a[i] < (uint64_t)b[i] ? -1 : 0;
    return out;
}

// ntt_avx2.rs:747-779: (high, low) words of the unsigned 128-bit product, from four 32 x 32
// products.  The partial sums are kept in the reference's order: low_high never overflows 64 bits.
inline std::pair<Lanes, Lanes> a_mul_b_karatsuba(const Lanes& a, const Lanes& b)
{
    Lanes high{}, low{};
    for (int i = 0; i < 4; i++) {
        const uint64_t a_low = (uint64_t)a[i] & 0xffffffffull, a_high = (uint64_t)a[i] >> 32;
        const uint64_t b_low = (uint64_t)b[i] & 0xffffffffull, b_high = (uint64_t)b[i] >> 32;
        uint64_t h = a_high * b_high;
        const uint64_t l = a_low * b_low;
        uint64_t low_high = l >> 32;
        const uint64_t middle_0 = a_high * b_low, middle_1 = a_low * b_high;
        h += middle_0 >> 32;
        h += middle_1 >> 32;
        low_high += middle_0 & 0xffffffffull;
        low_high += middle_1 & 0xffffffffull;
        h += low_high >> 32;
        high[i] = (int64_t)h;
        low[i] = (int64_t)((l & 0xffffffffull) | (low_high << 32));
    }
    return {high, low};
}

// ntt_avx2.rs:782-796: (high, low) of a - b on 128-bit values, wrapping.
inline std::pair<Lanes, Lanes> a_sub_b(const Lanes& a_high, const Lanes& a_low, const Lanes& b_high, const Lanes& b_low)
{
    Lanes high{}, low{};
    const Lanes borrow = a_lt_b(a_low, b_low);
    for (int i = 0; i < 4; i++) {
        low[i] = (int64_t)((uint64_t)a_low[i] - (uint64_t)b_low[i]);
        high[i] = (int64_t)((uint64_t)a_high[i] - (uint64_t)(borrow[i] & 1) - (uint64_t)b_high[i]);
    }
    return {high, low};
}

// ntt_avx2.rs:799-813: a < b ? all ones : 0 on unsigned 128-bit values.
inline Lanes a_lt_b_wide(const Lanes& a_high, const Lanes& a_low, const Lanes& b_high, const Lanes& b_low)
{
    const Lanes lo = a_lt_b(a_low, b_low), hi = a_lt_b(a_high, b_high);
    Lanes out{};
    for (int i = 0; i < 4; i++)
        out[i] = hi[i] | ((a_high[i] == b_high[i] ? -1 : 0) & lo[i]);
    return out;
}

} // namespace avx2
} // namespace dft
} // namespace ntt
