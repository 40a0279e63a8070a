// Modular arithmetic for the transform kernels: 64-bit Montgomery (R = 2^64) for any odd
// modulus below 2^64, and the Goldilocks prime 2^64 - 2^32 + 1 with shift-and-add reduction.
//
// Every function is __host__ __device__ so the same code is unit-tested on the CPU
// (tests/test_arith_host.py builds it with g++) before it runs on a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NTTX_HD __host__ __device__ __forceinline__
#else
#define NTTX_HD inline
#endif

namespace nttx {

// 64 x 64 -> 128 as (hi, lo).
NTTX_HD void mul_wide(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo)
{
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (uint64_t)p;
    hi = (uint64_t)(p >> 64);
#endif
}

NTTX_HD uint64_t mul_hi(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// Constants of one Montgomery plan.  `k` is q^-1 mod 2^64 as the reference's
// MontgomeryTransformer holds it (utils.rs:130); `r_mod_q` = 2^64 mod q.
struct MontCtx {
    uint64_t q;
    uint64_t k;
    uint64_t r_mod_q;
    uint64_t n_inv_m; // n^-1 in Montgomery form (ntt.rs:13)
};

// Montgomery arithmetic on canonical residues in [0, q).  WIDE selects carry-aware
// addition for q >= 2^63 (the Goldilocks prime run through Table<u64>, ntt.rs:60-62).
template <bool WIDE>
struct Mont {
    using Ctx = MontCtx;

    // REDC(a * b): a * b * 2^-64 mod q in [0, q) for a * b < q * 2^64 (utils.rs:165-178).
    // Because k * q = 1 mod 2^64, the low words of a*b and m*q are equal, so only the
    // high words need subtracting and the borrow decides the correction.
    static NTTX_HD uint64_t mul(uint64_t a, uint64_t b, const Ctx& c)
    {
        uint64_t hi, lo;
        mul_wide(a, b, hi, lo);
        uint64_t m = lo * c.k;
        uint64_t mq_hi = mul_hi(m, c.q);
        uint64_t r = hi - mq_hi;
        return (hi < mq_hi) ? r + c.q : r;
    }

    static NTTX_HD uint64_t add(uint64_t u, uint64_t v, const Ctx& c)
    {
        uint64_t s = u + v;
        if (WIDE) {
            bool carry = s < u;
            return (carry || s >= c.q) ? s - c.q : s;
        }
        return (s >= c.q) ? s - c.q : s;
    }

    static NTTX_HD uint64_t sub(uint64_t u, uint64_t v, const Ctx& c)
    {
        uint64_t d = u - v;
        return (u < v) ? d + c.q : d;
    }
    static NTTX_HD uint64_t sub_ct(uint64_t u, uint64_t v, const Ctx& c) { return sub(u, v, c); }
    static NTTX_HD uint64_t sub_gs(uint64_t u, uint64_t v, const Ctx& c) { return sub(u, v, c); }

    // The reference's first sweep (ntt.rs:125-127, 181-183) maps any u64 to its residue;
    // REDC(x * (2^64 mod q)) = x mod q does the same for the rare x >= q.
    static NTTX_HD uint64_t condition_input(uint64_t x, const Ctx& c)
    {
        return (x >= c.q) ? mul(x, c.r_mod_q, c) : x;
    }

    // factor applied to every output of the inverse transform (ntt.rs:231-235)
    static NTTX_HD uint64_t scale_factor(const Ctx& c) { return c.n_inv_m; }
};

struct GoldilocksCtx {
    uint64_t n_inv; // plain residue (ntt_goldilock.rs:12)
};

// Goldilocks arithmetic with the reference's exact per-butterfly rules, so that the
// forward transform reproduces TableGoldilock bit for bit, including the non-canonical
// value q that `res > q` (ntt_goldilock.rs:107) lets through.
struct Goldilocks {
    using Ctx = GoldilocksCtx;
    static constexpr uint64_t Q = 0xffffffff00000001ull;

    // mul_reduce (ntt_goldilock.rs:183-190) with reduce_159 (:63-79): canonical a*b mod q
    // for any u64 a, b.
    static NTTX_HD uint64_t mul(uint64_t a, uint64_t b, const Ctx&)
    {
        uint64_t hi, lo;
        mul_wide(a, b, hi, lo);
        uint64_t high = hi >> 32;
        uint64_t middle = hi & 0xffffffffull;
        uint64_t low_2 = lo - high;
        if (lo < high)
            low_2 += Q;
        uint64_t product = (middle << 32) - middle;
        uint64_t res = low_2 + product;
        if (res < product || res >= Q)
            res -= Q;
        return res;
    }

    // ntt_goldilock.rs:104-112: carry -> -q, else strictly-greater -> -q (a sum of exactly q stays)
    static NTTX_HD uint64_t add(uint64_t u, uint64_t v, const Ctx&)
    {
        uint64_t s = u + v;
        bool carry = s < u;
        return (carry || s > Q) ? s - Q : s;
    }

    // ntt_goldilock.rs:115-120: borrow -> +q, then % q
    static NTTX_HD uint64_t sub_ct(uint64_t u, uint64_t v, const Ctx&)
    {
        uint64_t d = u - v;
        if (u < v)
            d += Q;
        return (d >= Q) ? d - Q : d;
    }

    // ntt_goldilock.rs:162-165: borrow -> +q, no further reduction
    static NTTX_HD uint64_t sub_gs(uint64_t u, uint64_t v, const Ctx&)
    {
        uint64_t d = u - v;
        return (u < v) ? d + Q : d;
    }

    static NTTX_HD uint64_t condition_input(uint64_t x, const Ctx&) { return x; }
    static NTTX_HD uint64_t scale_factor(const Ctx& c) { return c.n_inv; }
};

} // namespace nttx
