// Modular arithmetic policies for the transform kernels.
//
//   Shoup61     moduli below 2^61 through the Montgomery-kind plan (Table<u64>, TableAVX2):
//               lazy butterflies on residues kept in [0, 8q) (forward) / [0, 4q) (backward),
//               products by a precomputed quotient (w, floor(w * 2^64 / q)); 9 32x32 multiplies.
//   Mont<WIDE>  any odd modulus below 2^64 (WIDE: q >= 2^63): strict 64-bit Montgomery, R = 2^64.
//   Goldilocks  q = 2^64 - 2^32 + 1 with the reference's exact per-butterfly rules.
//
// A policy provides: Ctx (plan constants), TwElem (device table entry), Tw (twiddle in
// registers), load(), ct() / gs() butterflies, gs_last() (stage 1 of the inverse with the n^-1
// factor folded in, canonical outputs), condition_input() and finish() (canonical residue at the
// end of the forward transform).
//
// Every function is __host__ __device__ so the same code is unit-tested on the CPU before it
// runs on a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NTTX_HD __host__ __device__ __forceinline__
#define NTTX_UNROLL _Pragma("unroll")
#else
#define NTTX_HD inline
#define NTTX_UNROLL
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

// 32 x 32 + 64 -> 64; callers guarantee no overflow or use only low bits.  On the device this is
// pinned to v_mad_u64_u32: left to itself hipcc lowers the low-half and high-half uses to
// v_mul_lo_u32 / v_mul_hi_u32, which issue slower on gfx950 and need extra adds.
NTTX_HD uint64_t mad32(uint32_t a, uint32_t b, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
#else
    return (uint64_t)a * (uint64_t)b + c;
#endif
}
NTTX_HD uint64_t mul32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
#else
    return (uint64_t)a * (uint64_t)b;
#endif
}
// Same with the second factor wave-uniform (an SGPR on the device: plan constants).
NTTX_HD uint64_t mad32u(uint32_t a, uint32_t b_uniform, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c) : "vcc");
    return d;
#else
    return (uint64_t)a * (uint64_t)b_uniform + c;
#endif
}
NTTX_HD uint64_t mul32u(uint32_t a, uint32_t b_uniform)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "s"(b_uniform) : "vcc");
    return d;
#else
    return (uint64_t)a * (uint64_t)b_uniform;
#endif
}
// 32-bit add kept opaque on the device so that it stays a single v_add_u32 on one half of a
// register pair (hipcc otherwise widens it into a 64-bit add plus two moves).
NTTX_HD uint32_t add32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint32_t d;
    asm("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return a + b;
#endif
}
NTTX_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
// a - b on 64-bit values as exactly two instructions (hipcc sometimes regroups a subtraction whose
// operands were assembled from halves into five).
NTTX_HD uint64_t sub64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint32_t rl, rh;
    asm("v_sub_co_u32 %0, vcc, %2, %4\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %1, vcc, %3, %5, vcc"
        : "=&v"(rl), "=&v"(rh)
        : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32))
        : "vcc");
    return ((uint64_t)rh << 32) | rl;
#else
    return a - b;
#endif
}
NTTX_HD uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
NTTX_HD uint32_t lo32(uint64_t x) { return (uint32_t)x; }
NTTX_HD uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }

// One 16-byte device table entry for policies that keep two words per twiddle.
struct alignas(16) TwPair {
    uint64_t w;
    uint64_t v;
};

// 64-bit add as ONE instruction (v_lshl_add_u64).  A carry chain (v_add_co_u32 + v_addc_co_u32) costs
// about three times as much on gfx950: two instructions that cannot issue back to back.
NTTX_HD uint64_t add64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint64_t d;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return a + b;
#endif
}
// Same with a wave-uniform second term (plan constants live in scalar registers).
NTTX_HD uint64_t add64u(uint64_t a, uint64_t b_uniform)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint64_t d;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "s"(b_uniform));
    return d;
#else
    return a + b_uniform;
#endif
}
// 2a + b, b wave-uniform
NTTX_HD uint64_t dbl_add64u(uint64_t a, uint64_t b_uniform)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    uint64_t d;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(a), "s"(b_uniform));
    return d;
#else
    return (a << 1) + b_uniform;
#endif
}
// ---------------------------------------------------------------------------------------------
// Shoup61: q < 2^61 (so 8q <= 2^64).
//
// mul(y, (w, v)) with v = floor(w * 2^64 / q) returns y*w - qhat*q mod 2^64, where qhat is
// floor(y * v / 2^64) computed from the three upper partial products only (at most 2 short).
// For any y < 2^64 the exact remainder is below q * (1 + y / 2^64) <= 2q, hence the result lies
// in [0, 4q) and is congruent to y*w.  Outputs of both reference paths are canonical residues
// (ntt.rs:173-175, ntt_avx2.rs:276-278), so any exact arithmetic mod q that ends in a canonical
// value reproduces them.
//
// Instruction costs on gfx950 (tools/ubench_issue.hip, cycles per wave instruction at four waves
// per SIMD): v_add/sub/and/or/xor/not/mov/lshrrev/ashrrev on 32 bits about 2.5; everything else
// (multiplies, v_lshl_add_u64, carry-out adds, v_cndmask, three-operand forms) 4.3 to 4.7.  There is
// no cheap multiply, so what can be trimmed is the glue around the nine or ten of a product.
//
// NEAR = true: moduli just below 2^61 (c8 = 2^61 - q < 2^29, e.g. 0x1fffffffffe00001).  Multiples of q
// are then a power of two minus a small constant, so "subtract k*q, k = the top bits of x" is
// (x with those bits cleared) + k * const: three instructions, two of them full rate.
struct Shoup61Ctx {
    uint64_t q, q2, q4;
    uint64_t nq;   // 2^64 - q
    uint32_t c8;   // NEAR: 2^61 - q
    uint32_t scale_shift; // k when `scale` is 2^-k mod q with q = 1 mod 2^k (n^-1 of a transform of 2^k points), else 0
    TwPair one;    // (1, floor(2^64 / q)): reduces an arbitrary u64 input
    TwPair scale;  // n^-1 (plain residue) and its quotient
    TwPair scale1; // tw[1] * n^-1 mod q and its quotient (twiddle of inverse stage 1, pre-scaled)
    uint64_t k;    // q^-1 mod 2^64 (utils.rs:130), for the product of two data words
    uint64_t qk;   // scale_shift != 0: (q - 1) >> scale_shift
    uint64_t qk1;  // scale_shift != 0: qk + 1
};

// True when the NEAR forms apply to q.
NTTX_HD bool shoup61_is_near(uint64_t q) { return q < (1ull << 61) && (1ull << 61) - q < (1ull << 29); }

// Plan constants from the modulus, n^-1, the (plain) twiddle of inverse stage 1 times n^-1, and k.
inline Shoup61Ctx make_shoup61_ctx(uint64_t q, uint64_t n_inv_plain, uint64_t s1_plain, uint64_t k)
{
    auto pair_of = [&](uint64_t w) { return TwPair{w, (uint64_t)((((unsigned __int128)w) << 64) / q)}; };
    Shoup61Ctx c{};
    c.q = q;
    c.q2 = 2 * q;
    c.q4 = 4 * q;
    c.nq = 0 - q;
    c.c8 = shoup61_is_near(q) ? (uint32_t)((1ull << 61) - q) : 0u;
    c.one = pair_of(1 % q);
    c.scale = pair_of(n_inv_plain);
    c.scale1 = pair_of(s1_plain);
    c.k = k;
    // n^-1 = 2^-j for a transform of 2^j points, and 2n | q - 1 for any modulus that has such a
    // transform: then scaling is a division by 2^j made exact by a multiple of q (mul_pow2_inv).
    // Limits of that routine: t < 2^24 and hi32(qk) < 2^24 for its 24-bit multiply.
    for (uint32_t j = 8; j <= 24; j++) {
        if ((uint64_t)(((unsigned __int128)n_inv_plain << j) % q) == 1 % q && ((q - 1) & ((1ull << j) - 1)) == 0 &&
            ((q - 1) >> j >> 32) < (1u << 24)) {
            c.scale_shift = j;
            c.qk = (q - 1) >> j;
            c.qk1 = c.qk + 1;
            break;
        }
    }
    return c;
}

template <bool NEAR>
struct Shoup61T {
    using Ctx = Shoup61Ctx;
    using TwElem = TwPair;
    static constexpr bool CONDITIONS_INPUT = true;
    static constexpr bool DEFERRED_CHECK = false;
    static constexpr bool HAS_POW2_TWIDDLES = false;
    static NTTX_HD uint64_t modulus(const Ctx& c) { return c.q; }
    // Inputs below this bound need no conditioning: both butterflies take values in [0, 4q), so
    // the lazily reduced inputs the reference documents for its *_lazy methods ([0, 2q),
    // ntt.rs:31-35, 45-49) run at full speed; the outputs are canonical either way.
    static NTTX_HD uint64_t input_bound(const Ctx& c) { return c.q4; }
    struct Tw {
        uint32_t w0, w1, v0, v1;
    };

    static NTTX_HD Tw make(const TwPair& e) { return Tw{lo32(e.w), hi32(e.w), lo32(e.v), hi32(e.v)}; }
    static NTTX_HD Tw load(const TwElem* __restrict__ t, uint64_t i, const Ctx&) { return make(t[i]); }

    // Quotient estimate floor(y * v / 2^64) from y1*v1 + hi32(y1*v0) + hi32(y0*v1): the dropped
    // low parts sum to less than 3 * 2^32, so the estimate is at most 2 short.
    static NTTX_HD uint64_t quot(uint32_t y0, uint32_t y1, const Tw& t)
    {
        const uint32_t ah = mulhi32(y1, t.v0);
        const uint32_t bh = mulhi32(y0, t.v1);
        return mad32(y1, t.v1, (uint64_t)ah) + bh;
    }

    // acc + y*w - qh*q mod 2^64, all by multiply-adds: -q is the 64-bit constant nq, the low
    // words chain through 64-bit addends (overflow past bit 63 is irrelevant), and of the
    // 2^32-weight cross terms only the low 32 bits matter.
    static NTTX_HD uint64_t fold(uint64_t acc, uint32_t y0, uint32_t y1, uint64_t qh, const Tw& t, const Ctx& c)
    {
        const uint32_t n0 = lo32(c.nq), n1 = hi32(c.nq);
        const uint64_t s = mad32(y0, t.w0, mad32u(lo32(qh), n0, acc));
        const uint64_t x = mad32(y0, t.w1, mad32(y1, t.w0, mad32u(lo32(qh), n1, mul32u(hi32(qh), n0))));
        return pack64(lo32(s), add32(hi32(s), lo32(x)));
    }

    // acc + (y * w mod q, lazily in [0, 4q)) for any y; exact mod 2^64.
    static NTTX_HD uint64_t mul_add(uint64_t acc, uint64_t y, const Tw& t, const Ctx& c)
    {
        const uint32_t y0 = lo32(y), y1 = hi32(y);
        return fold(acc, y0, y1, quot(y0, y1, t), t, c);
    }
    static NTTX_HD uint64_t mul(uint64_t y, const Tw& t, const Ctx& c) { return mul_add(0, y, t, c); }

    // x in [0, 2m) -> [0, m); m is a plan constant (wave-uniform).  On the device the borrow of
    // the subtraction drives the select directly: 4 instructions instead of the 5 hipcc emits
    // (it adds a separate 64-bit compare).  The s_nop's are the wait states gfx950 needs between
    // a VALU write of VCC and a VALU read of it as carry-in / select mask.  (A carry-free form --
    // add 2^64 - m, spread the sign, mask m, add -- is five instructions and measured 6 % slower.)
    static NTTX_HD uint64_t csub(uint64_t x, uint64_t m)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        uint32_t rl, rh;
        asm("v_sub_co_u32 %0, vcc, %2, %4\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32 %1, vcc, %3, %5, vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %0, %0, %2, vcc\n\t" // vcc = borrow = (x < m): keep x, else x - m
            "v_cndmask_b32 %1, %1, %3, vcc"
            : "=&v"(rl), "=&v"(rh)
            : "v"(lo32(x)), "v"(hi32(x)), "v"(lo32(m)), "v"(hi32(m))
            : "vcc");
        return pack64(rl, rh);
#else
        const uint64_t t = x - m;
        return (x < m) ? x : t;
#endif
    }

    // NEAR forms.  x - k * (2^B - e) = (x with the bits from B up cleared) + k * e for k = x >> B.
    // fold63: k * 4q, k <= 1.  [0, 2^64) -> [0, 2^63 + 4 c8)
    static NTTX_HD uint64_t fold63(uint64_t x, const Ctx& c)
    {
        return mad32u(hi32(x) >> 31, c.c8 << 2, x & 0x7fffffffffffffffull);
    }
    // fold62: k * 2q, k <= 3.  [0, 2^64) -> [0, 2^62 + 6 c8)
    static NTTX_HD uint64_t fold62(uint64_t x, const Ctx& c)
    {
        return mad32u(hi32(x) >> 30, c.c8 << 1, x & 0x3fffffffffffffffull);
    }
    // fold61: k * q, k <= 7.  [0, 2^64) -> [0, 2^61 + 7 c8), which lies below 2q
    static NTTX_HD uint64_t fold61(uint64_t x, const Ctx& c)
    {
        return mad32u(hi32(x) >> 29, c.c8, x & 0x1fffffffffffffffull);
    }

    // Cooley-Tukey (ntt.rs:146-166 on lazy residues): a' = u + v, b' = u - v + 4q with u = a less a
    // multiple of 4q and v = b*s lazily in [0, 4q).
    //   general: values in [0, 8q); u = a - 4q when a >= 4q.
    //   NEAR:    values in [0, 2^64); u = a - 4q when a >= 2^63, so u < 2^63 + 4 c8 = 2^64 - 4q and
    //            u + v < 2^64.  2u + 4q may wrap; b' itself lies in (0, 2^64), so the wrap cancels.
    static NTTX_HD void ct(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = NEAR ? fold63(a, c) : csub(a, c.q4);
        const uint64_t sum = mul_add(u, b, s, c);
        const uint64_t t = dbl_add64u(u, c.q4);
        a = sum;
        b = sub64(t, sum);                          // 2u + 4q - (u + v)
    }

    // Gentleman-Sande (ntt.rs:200-221 on lazy residues), values in [0, 4q) in and out: a' = u + v
    // less a multiple of q, b' = (u - v + 4q) * s lazily.
    //   general: a' = u + v - 4q when that is not negative.
    //   NEAR:    a' = u + v - k * 2q, k = (u + v) >> 62: below 2^62 + 6 c8 < 4q.
    static NTTX_HD void gs(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = a, v = b;
        const uint64_t sum = add64(u, v);
        a = NEAR ? fold62(sum, c) : csub(sum, c.q4);
        b = mul(sub64(add64u(u, c.q4), v), s, c);          // input (0, 8q) -> [0, 4q)
    }

    // [0, 4q) -> [0, q)
    static NTTX_HD uint64_t canon4(uint64_t x, const Ctx& c)
    {
        if (NEAR)
            return csub(fold61(x, c), c.q);
        return csub(csub(x, c.q2), c.q);
    }

    // x * 2^-k mod q for k = c.scale_shift (q = 1 mod 2^k), any x < 2^64, without a product by a
    // quotient: with t = ~x mod 2^k the sum x + (t + 1) q is a multiple of 2^k, and
    //   (x + (t + 1) q) / 2^k = (x >> k) + 1 + (t + 1) qk = (x >> k) + (qk + 1) + t * qk,  qk = (q - 1) >> k.
    // Six instructions against eleven; the result lies in [0, q + 2^(64-k)): almost always canonical,
    // always below 2q.
    static NTTX_HD uint64_t mul_pow2_inv(uint64_t x, const Ctx& c)
    {
        const uint32_t t = ~lo32(x) & ((1u << c.scale_shift) - 1);
        const uint64_t r = mad32u(t, lo32(c.qk), add64u(x >> c.scale_shift, c.qk1));
#if defined(__HIP_DEVICE_COMPILE__)
        return pack64(lo32(r), hi32(r) + __umul24(t, hi32(c.qk))); // one v_mad_u32_u24: both factors below 2^24
#else
        return pack64(lo32(r), hi32(r) + t * hi32(c.qk));
#endif
    }

    // Stage 1 of the inverse with the n^-1 factor of ntt.rs:231-235 folded in; canonical outputs.
    static NTTX_HD void gs_last(uint64_t& a, uint64_t& b, const Tw&, const Ctx& c)
    {
        const uint64_t u = a, v = b;
        if (c.scale_shift)
            a = csub(mul_pow2_inv(add64(u, v), c), c.q);
        else
            a = canon4(mul(add64(u, v), make(c.scale), c), c);
        b = canon4(mul(sub64(add64u(u, c.q4), v), make(c.scale1), c), c);
    }

    // The same stage on a whole register group (x[i] with x[i + M/2]).  NEAR: every output is first
    // brought below 2q by a route that almost never leaves it at q or above (fold61 of a product:
    // only for values within 7 c8 of a multiple of 2^61; mul_pow2_inv: with probability about
    // 2^(3-k)); one comparison of the largest high word with q's decides whether the group needs the
    // conditional subtractions at all.  7 -> 3.5 instructions per output, results identical.
    template <int M>
    static NTTX_HD void gs_last_stage(uint64_t (&x)[M], const Tw& s, const Ctx& c)
    {
        constexpr int half = M / 2;
        if (!NEAR) {
            NTTX_UNROLL
            for (int i = 0; i < half; i++)
                gs_last(x[i], x[i + half], s, c);
            return;
        }
        uint32_t top = 0;
        if (c.scale_shift) {
            NTTX_UNROLL
            for (int i = 0; i < half; i++) {
                const uint64_t u = x[i], v = x[i + half];
                x[i] = mul_pow2_inv(add64(u, v), c);
                x[i + half] = fold61(mul(sub64(add64u(u, c.q4), v), make(c.scale1), c), c);
                top = max3(top, hi32(x[i]), hi32(x[i + half]));
            }
        } else {
            NTTX_UNROLL
            for (int i = 0; i < half; i++) {
                const uint64_t u = x[i], v = x[i + half];
                x[i] = fold61(mul(add64(u, v), make(c.scale), c), c);
                x[i + half] = fold61(mul(sub64(add64u(u, c.q4), v), make(c.scale1), c), c);
                top = max3(top, hi32(x[i]), hi32(x[i + half]));
            }
        }
        if (__builtin_expect(top >= hi32(c.q), 0)) {
            NTTX_UNROLL
            for (int i = 0; i < M; i++)
                x[i] = csub(x[i], c.q);
        }
    }
    static NTTX_HD uint32_t max3(uint32_t x, uint32_t y, uint32_t z)
    {
        const uint32_t m = x > y ? x : y;
        return m > z ? m : z;
    }

    // Lazy outputs: the range [0, 2q) the reference documents for forward_inplace_lazy /
    // backward_inplace_lazy (ntt.rs:31-35, 45-49) instead of the canonical residue.
    static constexpr bool HAS_LAZY = true;
    // [0, 4q) -> [0, 2q)
    static NTTX_HD uint64_t lazy4(uint64_t x, const Ctx& c) { return NEAR ? fold61(x, c) : csub(x, c.q2); }
    static NTTX_HD void gs_last_lazy(uint64_t& a, uint64_t& b, const Tw&, const Ctx& c)
    {
        const uint64_t u = a, v = b;
        a = lazy4(mul(add64(u, v), make(c.scale), c), c);
        b = lazy4(mul(sub64(add64u(u, c.q4), v), make(c.scale1), c), c);
    }
    // end of the forward transform, lazily: general [0, 8q) -> [0, 2q); NEAR [0, 2^64) -> [0, 2^61 + 7 c8)
    static NTTX_HD uint64_t finish_lazy(uint64_t x, const Ctx& c)
    {
        return NEAR ? fold61(x, c) : csub(csub(x, c.q4), c.q2);
    }

    // ntt.rs:125-127, 181-183: the reference's first sweep maps any u64 to its residue.
    static NTTX_HD uint64_t condition_input(uint64_t x, const Ctx& c)
    {
        return (x >= c.q) ? canon4(mul(x, make(c.one), c), c) : x;
    }

    // a * b * 2^-64 mod q in [0, q) for a, b in [0, 2q) (neither has a precomputed quotient, so this
    // is REDC as in utils.rs:165-178: hi(a * b) < 4 q^2 / 2^64 < q keeps the result canonical); the
    // plan folds the factor 2^64 back in with n^-1.
    static NTTX_HD uint64_t pointwise(uint64_t a, uint64_t b, const Ctx& c)
    {
        uint64_t hi, lo;
        mul_wide(a, b, hi, lo);
        const uint64_t mq_hi = mul_hi(lo * c.k, c.q);
        const uint64_t r = hi - mq_hi;
        return (hi < mq_hi) ? r + c.q : r;
    }

    // End of the forward transform: the canonical value ntt.rs:173-175 ends with.
    //   general: [0, 8q) -> [0, q);  NEAR: [0, 2^64) -> [0, q)
    static NTTX_HD uint64_t finish(uint64_t x, const Ctx& c)
    {
        if (NEAR)
            return csub(fold61(x, c), c.q);
        return canon4(csub(x, c.q4), c);
    }
    // The same for a register group.  NEAR: fold61 leaves a value at q or above only when the input
    // lies within 7 c8 below a multiple of 2^61 (probability about 2^-37 c8 per element), so the
    // conditional subtractions hide behind one comparison of the group's largest high word.
    template <int M>
    static NTTX_HD void finish_group(uint64_t (&x)[M], const Ctx& c)
    {
        if (!NEAR) {
            NTTX_UNROLL
            for (int i = 0; i < M; i++)
                x[i] = finish(x[i], c);
            return;
        }
        uint32_t top = 0;
        NTTX_UNROLL
        for (int i = 0; i < M; i += 2) {
            x[i] = fold61(x[i], c);
            x[i + 1] = fold61(x[i + 1], c);
            top = max3(top, hi32(x[i]), hi32(x[i + 1]));
        }
        if (__builtin_expect(top >= hi32(c.q), 0)) {
            NTTX_UNROLL
            for (int i = 0; i < M; i++)
                x[i] = csub(x[i], c.q);
        }
    }
};
using Shoup61 = Shoup61T<false>;
using Shoup61Near = Shoup61T<true>;

// ---------------------------------------------------------------------------------------------
// Strict 64-bit Montgomery on canonical residues.  `k` is q^-1 mod 2^64 as the reference's
// MontgomeryTransformer holds it (utils.rs:130); twiddles are in Montgomery form as the
// reference stores them, data stays in plain form: REDC(x * wR) = x * w.
struct MontCtx {
    uint64_t q;
    uint64_t k;
    uint64_t r_mod_q;  // 2^64 mod q
    uint64_t n_inv_m;  // n^-1 in Montgomery form (ntt.rs:13)
};

template <bool WIDE> // WIDE: q >= 2^63 needs carry-aware addition (Table::new_goldilock, ntt.rs:60-62)
struct Mont {
    using Ctx = MontCtx;
    using TwElem = uint64_t;
    static constexpr bool CONDITIONS_INPUT = true;
    static constexpr bool DEFERRED_CHECK = false;
    static constexpr bool HAS_POW2_TWIDDLES = false;
    static NTTX_HD uint64_t modulus(const Ctx& c) { return c.q; }
    static NTTX_HD uint64_t input_bound(const Ctx& c) { return c.q; } // strict residues only
    struct Tw {
        uint64_t w;
    };

    static NTTX_HD Tw load(const TwElem* __restrict__ t, uint64_t i, const Ctx&) { return Tw{t[i]}; }

    // REDC(a * b): a * b * 2^-64 mod q in [0, q) for a * b < q * 2^64 (utils.rs:165-178).
    // Because k * q = 1 mod 2^64 the low words of a*b and m*q are equal, so only the high
    // words need subtracting and the borrow decides the correction.
    static NTTX_HD uint64_t mul(uint64_t a, uint64_t b, const Ctx& c)
    {
        uint64_t hi, lo;
        mul_wide(a, b, hi, lo);
        const uint64_t m = lo * c.k;
        const uint64_t mq_hi = mul_hi(m, c.q);
        const uint64_t r = hi - mq_hi;
        return (hi < mq_hi) ? r + c.q : r;
    }

    static NTTX_HD uint64_t add(uint64_t u, uint64_t v, const Ctx& c)
    {
        const uint64_t s = u + v;
        if (WIDE) {
            const bool carry = s < u;
            return (carry || s >= c.q) ? s - c.q : s;
        }
        return (s >= c.q) ? s - c.q : s;
    }

    static NTTX_HD uint64_t sub(uint64_t u, uint64_t v, const Ctx& c)
    {
        const uint64_t d = u - v;
        return (u < v) ? d + c.q : d;
    }

    static NTTX_HD void ct(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = a, v = mul(b, s.w, c);
        a = add(u, v, c);
        b = sub(u, v, c);
    }

    static NTTX_HD void gs(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = a, v = b;
        a = add(u, v, c);
        b = mul(sub(u, v, c), s.w, c);
    }

    static NTTX_HD void gs_last(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = a, v = b;
        const uint64_t sf = mul(s.w, c.n_inv_m, c); // (S * n^-1) in Montgomery form
        a = mul(add(u, v, c), c.n_inv_m, c);
        b = mul(sub(u, v, c), sf, c);
    }

    // REDC(x * (2^64 mod q)) = x mod q for the rare x >= q (ntt.rs:125-127, 181-183)
    static NTTX_HD uint64_t condition_input(uint64_t x, const Ctx& c)
    {
        return (x >= c.q) ? mul(x, c.r_mod_q, c) : x;
    }

    // a * b * 2^-64 mod q for residues a, b; the plan folds the factor 2^64 back in with n^-1.
    static NTTX_HD uint64_t pointwise(uint64_t a, uint64_t b, const Ctx& c) { return mul(a, b, c); }

    static NTTX_HD uint64_t finish(uint64_t x, const Ctx&) { return x; }
    template <int M>
    static NTTX_HD void finish_group(uint64_t (&)[M], const Ctx&) {}
    template <int M>
    static NTTX_HD void gs_last_stage(uint64_t (&x)[M], const Tw& s, const Ctx& c)
    {
        NTTX_UNROLL
        for (int i = 0; i < M / 2; i++)
            gs_last(x[i], x[i + M / 2], s, c);
    }

    // strict residues throughout: the canonical outputs already lie in the lazy range [0, 2q)
    static constexpr bool HAS_LAZY = false;
    static NTTX_HD uint64_t finish_lazy(uint64_t x, const Ctx& c) { return finish(x, c); }
    static NTTX_HD void gs_last_lazy(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c) { gs_last(a, b, s, c); }
};

// ---------------------------------------------------------------------------------------------
// Goldilocks arithmetic with the reference's exact per-butterfly rules, so that the forward
// transform reproduces TableGoldilock bit for bit, including the non-canonical value q that
// `res > q` (ntt_goldilock.rs:107) lets through.
struct GoldilocksCtx {
    uint64_t n_inv; // plain residue (ntt_goldilock.rs:12)
    uint32_t logn;  // log2 n when n_inv = 2^(192 - log2 n) is to be applied as a shift (plans with shift twiddles), else 0
    uint32_t pad_;
};

struct Goldilocks {
    using Ctx = GoldilocksCtx;
    using TwElem = uint64_t;
    static constexpr bool CONDITIONS_INPUT = false; // TableGoldilock has no input sweep
    static constexpr bool HAS_POW2_TWIDDLES = true;
    static NTTX_HD uint64_t modulus(const Ctx&) { return 0xffffffff00000001ull; }
    static NTTX_HD uint64_t input_bound(const Ctx&) { return 0xffffffff00000001ull; }
    struct Tw {
        uint64_t w;
    };
    static constexpr uint64_t Q = 0xffffffff00000001ull;
    static constexpr uint64_t EPS = 0xffffffffull; // 2^64 - q

    static NTTX_HD Tw load(const TwElem* __restrict__ t, uint64_t i, const Ctx&) { return Tw{t[i]}; }

    // ---- the reference's rules, statement by statement ---------------------------------------
    // mul_reduce (ntt_goldilock.rs:183-190) with reduce_159 (:63-79).  Whatever a and b are, the
    // result is the canonical residue a*b mod q: low_2 + product stays below 2q, so the single
    // conditional subtraction always lands in [0, q).
    static NTTX_HD uint64_t mul_strict(uint64_t a, uint64_t b)
    {
        uint64_t hi, lo;
        mul_wide(a, b, hi, lo);
        // reduce_159(low = lo, middle = hi & 0xffffffff, high = hi >> 32)
        const uint64_t high = hi32(hi);
        uint64_t low_2 = lo - high;
        if (lo < high)
            low_2 += Q;
        const uint64_t product = (uint64_t)lo32(hi) * 0xffffffffull; // (middle << 32) - middle
        uint64_t res = low_2 + product;
        if (res < product || res >= Q)
            res -= Q;
        return res;
    }
    // ntt_goldilock.rs:104-112: carry -> -q, else strictly-greater -> -q (a sum of exactly q stays)
    static NTTX_HD uint64_t add_strict(uint64_t u, uint64_t v)
    {
        const uint64_t s = u + v;
        const bool carry = s < u;
        return (carry || s > Q) ? s - Q : s;
    }
    // ntt_goldilock.rs:115-120: borrow -> +q, then % q
    static NTTX_HD uint64_t sub_ct_strict(uint64_t u, uint64_t v)
    {
        uint64_t d = u - v;
        if (u < v)
            d += Q;
        return (d >= Q) ? d - Q : d;
    }
    // ntt_goldilock.rs:162-165: borrow -> +q, no further reduction
    static NTTX_HD uint64_t sub_gs_strict(uint64_t u, uint64_t v)
    {
        const uint64_t d = u - v;
        return (u < v) ? d + Q : d;
    }

    // ---- the same results in fewer instructions -----------------------------------------------
    // Each fast routine applies only the correction its carry / borrow asks for (adding
    // 2^32 - 1 = 2^64 - q, i.e. subtracting q mod 2^64) and hands back the high word of its
    // result.  It can differ from the strict rule only when the strict rule's remaining test
    // (a compare against q without carry) fires, and then that high word is 0xffffffff.  The
    // butterflies OR those words' all-ones test into one rarely-taken branch that redoes the
    // butterfly with the strict routines, so outputs are bit-identical for every input while the
    // common path has no 64-bit compares or selects.
    //
    // The device forms are written out because hipcc cannot keep a carry in VCC across C++
    // statements; each has its C++ meaning beside it, which is what the host build compiles.
    // s_nop 1 = the two wait states gfx950 needs between a VALU write of VCC / an SGPR and a
    // VALU read of it as carry-in or select mask.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
    using Carry = uint64_t; // wave lane mask in an SGPR pair
#else
    using Carry = uint32_t; // 0 or 1
#endif

    // (hi32(p), 0), (lo32(p), 0), (lo32(p), lo32(r)) as aligned register pairs, one instruction each
    static NTTX_HD uint64_t zhi(uint64_t p)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        uint64_t d;
        asm("v_pk_mov_b32 %0, %1, 0 op_sel:[1,0]" : "=v"(d) : "v"(p));
        return d;
#else
        return p >> 32;
#endif
    }
    static NTTX_HD uint64_t zlo(uint64_t p)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        uint64_t d;
        asm("v_pk_mov_b32 %0, %1, 0 op_sel:[0,0]" : "=v"(d) : "v"(p));
        return d;
#else
        return p & 0xffffffffull;
#endif
    }
    static NTTX_HD uint64_t packlo(uint64_t p, uint64_t r)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        uint64_t d;
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(p), "v"(r));
        return d;
#else
        return (r << 32) | (p & 0xffffffffull);
#endif
    }

    // x + m * (2^32 - 1) mod 2^64 and its carry
    static NTTX_HD uint64_t mad_eps(uint32_t m, uint64_t x, Carry& carry)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        uint64_t d;
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(d), "=s"(carry) : "v"(m), "v"(x));
        return d;
#else
        const uint64_t p = (uint64_t)m * EPS, d = x + p;
        carry = d < p;
        return d;
#endif
    }

    // m * (2^32 - 1)
    static NTTX_HD uint64_t mul_eps(uint32_t m)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        uint64_t d, unused;
        asm("v_mad_u64_u32 %0, %1, %2, -1, 0" : "=v"(d), "=s"(unused) : "v"(m));
        return d;
#else
        return (uint64_t)m * EPS;
#endif
    }

    // Canonical a*b mod q unless the returned word `top` (the result's high half) is 0xffffffff.
    //
    // With the 128-bit product as 32-bit limbs x0..x3: a*b = lo + 2^64*x2 + 2^96*x3 with
    // lo = x0 + 2^32*x1, and 2^64 = eps, 2^96 = -1 (mod q), so a*b = V := lo + x2*eps - x3.
    // X = lo + x2*eps is one multiply-add with carry c.  If c is set V lies in [q, 2q), and
    // V - q = X - x3 - c + c*2^32 (mod 2^64) is exact and canonical.  If not, V < 2^64: the
    // wrapped X - x3 is V when 0 <= V, canonical unless V >= q (high word all ones), and
    // V + 2^64 > 2^64 - 2^32 (high word all ones again) when V < 0.
    static NTTX_HD uint64_t mul_fast(uint64_t a, uint64_t b, uint32_t& top)
    {
        const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
        const uint64_t p0 = mul32(a0, b0);
        const uint64_t p1 = mad32(a1, b0, zhi(p0));
        const uint64_t p2 = mad32(a0, b1, zlo(p1));  // lo32 = x1
        const uint64_t h = mad32(a1, b1, zhi(p1));   // + hi32(p2) = (x2, x3)
        const uint64_t lo = packlo(p0, p2);
        uint32_t x2, x3;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        asm("v_add_co_u32 %0, vcc, %2, %4\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %1, vcc, 0, %3, vcc"
            : "=&v"(x2), "=&v"(x3)
            : "v"(lo32(h)), "v"(hi32(h)), "v"(hi32(p2))
            : "vcc");
#else
        {
            const uint64_t hh = h + hi32(p2);
            x2 = lo32(hh);
            x3 = hi32(hh);
        }
#endif
        return reduce_limbs_fast(lo, x2, x3, top);
    }

    // lo + 2^64 x2 + 2^96 x3 (32-bit x2, x3) modulo q, by the argument above: canonical unless the
    // returned word `top` (the result's high half) is 0xffffffff.
    static NTTX_HD uint64_t reduce_limbs_fast(uint64_t lo, uint32_t x2, uint32_t x3, uint32_t& top)
    {
        Carry c;
        const uint64_t x = mad_eps(x2, lo, c);
        uint32_t tl, th;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        asm("s_nop 1\n\t"
            "v_addc_co_u32 %1, vcc, 0, %3, %5\n\t" // th = xh + c
            "v_subb_co_u32 %0, vcc, %2, %4, %5\n\t" // tl = xl - x3 - c, vcc = borrow
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %1, vcc, 0, %1, vcc" // th -= borrow
            : "=&v"(tl), "=&v"(th)
            : "v"(lo32(x)), "v"(hi32(x)), "v"(x3), "s"(c)
            : "vcc");
#else
        {
            const uint64_t t = x - x3 - c + ((uint64_t)c << 32);
            tl = lo32(t);
            th = hi32(t);
        }
#endif
        top = th;
        return pack64(tl, th);
    }
    // the same with x3 = 0
    static NTTX_HD uint64_t reduce_limbs_fast(uint64_t lo, uint32_t x2, uint32_t& top)
    {
        Carry c;
        const uint64_t x = mad_eps(x2, lo, c);
        uint32_t tl, th;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        asm("s_nop 1\n\t"
            "v_addc_co_u32 %1, vcc, 0, %3, %4\n\t"    // th = xh + c
            "v_subbrev_co_u32 %0, vcc, 0, %2, %4\n\t" // tl = xl - c, vcc = borrow
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"      // th -= borrow
            : "=&v"(tl), "=&v"(th)
            : "v"(lo32(x)), "v"(hi32(x)), "s"(c)
            : "vcc");
#else
        {
            const uint64_t t = x - c + ((uint64_t)c << 32);
            tl = lo32(t);
            th = hi32(t);
        }
#endif
        top = th;
        return pack64(tl, th);
    }

    // ---- twiddles that are powers of two -------------------------------------------------------
    // 2 has order 192 modulo q (2^96 = -1), so every root of unity of order up to 64 is +-2^k: the
    // twiddles of the first five forward stages (and the last five inverse ones) of ANY transform.
    // For a table whose psi^(n/32) is 2^39 (= 7^((q-1)/64): the reference's own psi,
    // ntt_goldilock.rs:51, and every psi = 7^((q-1)/2n)) entry x < 32 of the forward table is
    // 2^pow2_exponent(x) and that of the inverse table 2^(192 - pow2_exponent(x)).
    static constexpr int pow2_exponent(int x, bool inverse)
    {
        int u = 0;
        for (int t = x; t; t >>= 1)
            u++; // x has u bits: stage u
        int rev = 0;
        for (int i = 0; i < u; i++)
            rev |= ((x >> i) & 1) << (u - 1 - i);
        const int e = (39 * rev * (1 << (5 - u))) % 192;
        return inverse ? (192 - e) % 192 : e;
    }
    static constexpr uint64_t POW2_ROOT64 = 0x8000000000ull; // 2^39, the psi^(n/32) of such tables

    // a * 2^K mod q for 0 < K < 96, K not a multiple of 32: the 96-bit a << (K mod 32) as limbs
    // y0, y1, y2, placed K / 32 limbs up.  Canonical unless `top` (the result's high half) is 0xffffffff;
    // three shifts and four or five more instructions against the thirteen of mul_fast.
    template <int K>
    static NTTX_HD uint64_t mul_pow2_fast(uint64_t a, uint32_t& top)
    {
        static_assert(K > 0 && K < 96, "0 < K < 96");
        constexpr int S = K % 32, J = K / 32;
        const uint32_t a0 = lo32(a), a1 = hi32(a);
        // S = 0: whole limbs, nothing to shift
        const uint32_t y0 = S ? a0 << S : a0, y1 = S ? (a1 << S) | (a0 >> ((32 - S) & 31)) : a1, y2 = S ? a1 >> ((32 - S) & 31) : 0;
        if (J == 0)
            return reduce_limbs_fast(pack64(y0, y1), y2, top);        // y0 + 2^32 y1 + 2^64 y2
        if (J == 1) {
            if (S == 0)
                return reduce_limbs_fast(pack64(0, y0), y1, top);     // 2^32 y0 + 2^64 y1
            return reduce_limbs_fast(pack64(0, y0), y1, y2, top);     // 2^32 y0 + 2^64 y1 + 2^96 y2
        }
        // 2^64 y0 + 2^96 y1 + 2^128 y2 = y0 (2^32 - 1) - (y1 + 2^32 y2): both canonical (y2 < 2^31), so
        // the borrow-corrected difference is the canonical product exactly
        return sub_fast(mul_eps(y0), pack64(y1, y2), top);
    }

    // a * 2^-L mod q = a * 2^(192 - L) for 0 < L < 32 (the n^-1 of a transform of 2^L points): the limbs
    // y0, y1, y2 of a << (32 - L) five limbs up, where 2^160 = 1 - 2^32, 2^192 = 1 and 2^224 = 2^32:
    // (y0 + y1 + 2^32 y2) - 2^32 y0, both canonical, so the result is canonical exactly.
    template <int L>
    static NTTX_HD uint64_t mul_ninv_fast(uint64_t a, uint32_t& top)
    {
        static_assert(L > 0 && L < 32, "n = 2^L with 0 < L < 32");
        constexpr int S = 32 - L;
        const uint32_t a0 = lo32(a), a1 = hi32(a);
        const uint32_t y0 = a0 << S, y1 = (a1 << S) | (a0 >> (32 - S)), y2 = a1 >> (32 - S);
        return sub_fast(pack64(y1, y2) + y0, pack64(0, y0), top);
    }

    // Butterflies whose twiddle is 2^E, E in [0, 192) known at compile time; 2^E = -2^(E - 96) for
    // E >= 96.  Same contract as ct_fast / gs_fast: outputs equal the strict butterfly with the
    // twiddle's canonical value unless one of the watched words is 0xffffffff.
    //
    // Negative twiddle, forward (v = q - v' for v' = b 2^(E-96) != 0): add_strict(u, q - v') equals
    // u - v' (+ q on borrow) except when u = v' (the reference leaves q where this form gives 0), and
    // for v' = 0 except when u > q; sub_ct_strict(u, q - v') equals u + v' (- q on carry) except when that
    // lies in [q, 2^64).  So the roles of add_fast and sub_fast swap, and the one new case -- a zero
    // difference -- is watched through its high word minus one.
    static constexpr int TOPS_CT_POW2 = 4;
    template <int E>
    static NTTX_HD void ct_pow2_fast(uint64_t& a, uint64_t& b, uint32_t* tops)
    {
        const uint64_t u = a;
        const uint64_t v = mul_pow2_fast<E % 96>(b, tops[0]);
        if (E < 96) {
            a = add_fast(u, v, tops[1]);
            b = sub_fast(u, v, tops[2]);
            tops[3] = 0;
        } else {
            a = sub_fast(u, v, tops[1]);
            b = add_fast(u, v, tops[2]);
            tops[3] = tops[1] - 1; // 0xffffffff when the difference may be zero
        }
    }
    // Backward form.  Operands must lie in [0, q] -- true of every stage but the first of an inverse
    // transform, since add_strict and mul_reduce leave nothing larger: only then is the borrow-corrected
    // v - u the negative of the reference's u - v modulo q (for larger operands the reference's own
    // difference wraps modulo 2^64 and would have to be reproduced as it is).
    template <int E>
    static NTTX_HD void gs_pow2_fast(uint64_t& a, uint64_t& b, uint32_t* tops)
    {
        const uint64_t u = a, v = b;
        uint32_t unused;
        a = add_fast(u, v, tops[0]);
        // mul_reduce is canonical for any operand, and (u - v) * -2^k = (v - u) * 2^k
        b = mul_pow2_fast<E % 96>(E < 96 ? sub_fast(u, v, unused) : sub_fast(v, u, unused), tops[1]);
    }

    // u + v mod 2^64, minus q if that carried.  Differs from add_strict only when the sum did not
    // carry yet exceeds q, and then `top`, its high word, is 0xffffffff.
    static NTTX_HD uint64_t add_fast(uint64_t u, uint64_t v, uint32_t& top)
    {
        uint32_t sl, sh;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        // (The correction as one multiply-add, as in sub_fast, saves an instruction here too, but the
        // sum is a long-lived output: as an aligned register pair it pushed the backward kernels past
        // 128 VGPRs into spills.)
        uint32_t e;
        asm("v_add_co_u32 %0, vcc, %3, %5\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %1, vcc, %4, %6, vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %2, 0, -1, vcc\n\t"
            "v_add_co_u32 %0, vcc, %0, %2\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "=&v"(sl), "=&v"(sh), "=&v"(e)
            : "v"(lo32(u)), "v"(hi32(u)), "v"(lo32(v)), "v"(hi32(v))
            : "vcc");
#else
        {
            uint64_t s = u + v;
            if (s < u)
                s += EPS;
            sl = lo32(s);
            sh = hi32(s);
        }
#endif
        top = sh;
        return pack64(sl, sh);
    }

    // u - v mod 2^64, plus q if that borrowed: sub_gs_strict exactly; sub_ct_strict unless the
    // result is >= q, and then `top`, its high word, is 0xffffffff.
    static NTTX_HD uint64_t sub_fast(uint64_t u, uint64_t v, uint32_t& top)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTTX_NO_ASM_MAD)
        // The correction -(2^32 - 1) = (-65535) * 65537 is one signed 32 x 32 multiply-add on the
        // difference: 4 instructions instead of the 5 of a second borrow chain (+1.2 % pairs/s on the
        // Goldilocks workloads).
        uint32_t dl, dh, e;
        asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %2, 0, %7, vcc"
            : "=&v"(dl), "=&v"(dh), "=&v"(e)
            : "v"(lo32(u)), "v"(hi32(u)), "v"(lo32(v)), "v"(hi32(v)), "v"(0xffff0001u)
            : "vcc");
        uint64_t d, unused;
        asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(unused) : "v"(e), "s"(65537u), "v"(pack64(dl, dh)));
        top = hi32(d);
        return d;
#else
        uint64_t d = u - v;
        if (u < v)
            d -= EPS;
        top = hi32(d);
        return d;
#endif
    }

    static NTTX_HD uint32_t max3(uint32_t x, uint32_t y, uint32_t z)
    {
        const uint32_t m = x > y ? x : y;
        return m > z ? m : z;
    }

    static NTTX_HD uint64_t mul(uint64_t a, uint64_t b, const Ctx&) { return mul_strict(a, b); }

    // Butterflies in two forms.  *_fast returns a word that is 0xffffffff whenever its outputs may
    // differ from *_strict (the reference's statements); callers take the maximum of those words
    // over a group of independent butterflies and redo the group strictly if it is all ones, so a
    // group costs one branch and its butterflies can be interleaved.
    static constexpr bool DEFERRED_CHECK = true;
    static constexpr uint32_t SUSPECT = 0xffffffffu;

    // The fast butterflies write the high words to watch into `tops` (TOPS_CT / TOPS_GS of them);
    // the caller folds them into one maximum with three-input max operations.
    static constexpr int TOPS_CT = 3, TOPS_GS = 2, TOPS_GS_LAST = 3;

    // ntt_goldilock.rs:98-121
    static NTTX_HD void ct_fast(uint64_t& a, uint64_t& b, const Tw& s, const Ctx&, uint32_t* tops)
    {
        const uint64_t u = a;
        const uint64_t v = mul_fast(b, s.w, tops[0]);
        a = add_fast(u, v, tops[1]);
        b = sub_fast(u, v, tops[2]);
    }
    static NTTX_HD void ct_strict(uint64_t& a, uint64_t& b, const Tw& s, const Ctx&)
    {
        const uint64_t u = a, v = mul_strict(b, s.w);
        a = add_strict(u, v);
        b = sub_ct_strict(u, v);
    }

    // ntt_goldilock.rs:152-166
    static NTTX_HD void gs_fast(uint64_t& a, uint64_t& b, const Tw& s, const Ctx&, uint32_t* tops)
    {
        const uint64_t u = a, v = b;
        uint32_t unused;
        a = add_fast(u, v, tops[0]);
        b = mul_fast(sub_fast(u, v, unused), s.w, tops[1]);
    }
    static NTTX_HD void gs_strict(uint64_t& a, uint64_t& b, const Tw& s, const Ctx&)
    {
        const uint64_t u = a, v = b;
        a = add_strict(u, v);
        b = mul_strict(sub_gs_strict(u, v), s.w);
    }

    // The last inverse stage with both products as shifts, for a table with the standard powers of two:
    // the sum times n^-1 = 2^(192 - L), the difference times tw_inv[1] * n^-1 = 2^144 * 2^(192 - L) =
    // -2^(48 - L).  Operands in [0, q], as for gs_pow2_fast.
    template <int L>
    static NTTX_HD void gs_last_pow2_fast(uint64_t& a, uint64_t& b, uint32_t* tops)
    {
        static_assert(pow2_exponent(1, true) == 144 && L < 32, "tw_inv[1] = 2^144");
        const uint64_t u = a, v = b;
        uint32_t unused;
        a = mul_ninv_fast<L>(add_fast(u, v, tops[0]), tops[1]);
        b = mul_pow2_fast<48 - L>(sub_fast(v, u, unused), tops[2]);
    }

    // last stage + mul_reduce(., n_inv) (ntt_goldilock.rs:177-179): both products are canonical.
    // `sf` = s * n_inv, computed once per twiddle by the caller (scaled()).
    static NTTX_HD Tw scaled(const Tw& s, const Ctx& c) { return Tw{mul_strict(s.w, c.n_inv)}; }
    static NTTX_HD void gs_last_fast(uint64_t& a, uint64_t& b, const Tw& sf, const Ctx& c, uint32_t* tops)
    {
        const uint64_t u = a, v = b;
        uint32_t unused;
        a = mul_fast(add_fast(u, v, tops[0]), c.n_inv, tops[1]);
        b = mul_fast(sub_fast(u, v, unused), sf.w, tops[2]);
    }
    static NTTX_HD void gs_last_strict(uint64_t& a, uint64_t& b, const Tw& sf, const Ctx& c)
    {
        const uint64_t u = a, v = b;
        a = mul_strict(add_strict(u, v), c.n_inv);
        b = mul_strict(sub_gs_strict(u, v), sf.w);
    }

    // self-contained butterflies (one check each)
    static NTTX_HD void ct(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = a, y = b;
        uint32_t t[TOPS_CT];
        ct_fast(a, b, s, c, t);
        if (__builtin_expect(max3(t[0], t[1], t[2]) == SUSPECT, 0)) {
            a = u;
            b = y;
            ct_strict(a, b, s, c);
        }
    }
    static NTTX_HD void gs(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = a, y = b;
        uint32_t t[TOPS_GS];
        gs_fast(a, b, s, c, t);
        if (__builtin_expect(max3(t[0], t[1], 0) == SUSPECT, 0)) {
            a = u;
            b = y;
            gs_strict(a, b, s, c);
        }
    }
    static NTTX_HD void gs_last(uint64_t& a, uint64_t& b, const Tw& s, const Ctx& c)
    {
        const uint64_t u = a, y = b;
        const Tw sf = scaled(s, c);
        uint32_t t[TOPS_GS_LAST];
        gs_last_fast(a, b, sf, c, t);
        if (__builtin_expect(max3(t[0], t[1], t[2]) == SUSPECT, 0)) {
            a = u;
            b = y;
            gs_last_strict(a, b, sf, c);
        }
    }

    static NTTX_HD uint64_t condition_input(uint64_t x, const Ctx&) { return x; }
    static NTTX_HD uint64_t finish(uint64_t x, const Ctx&) { return x; }
    template <int M>
    static NTTX_HD void finish_group(uint64_t (&)[M], const Ctx&) {}
    // TableGoldilock's lazy methods run the same statements as its strict ones (ntt_goldilock.rs:25-29, 39-43)
    static constexpr bool HAS_LAZY = false;
    static NTTX_HD uint64_t finish_lazy(uint64_t x, const Ctx&) { return x; }
    // mul_reduce (ntt_goldilock.rs:183-190): canonical for any operands, q itself included
    static NTTX_HD uint64_t pointwise(uint64_t a, uint64_t b, const Ctx&) { return mul_strict(a, b); }
};

// The Goldilocks prime behind the Montgomery-kind interface (Table::new_goldilock, ntt.rs:60-62):
// Table<u64> reduces its input first (ntt.rs:125-127) and ends with canonical residues
// (ntt.rs:173-175), whatever happens in between.  The Goldilocks butterflies are correct modulo q
// on any u64, so they serve with a reduction of the input (x >= q -> x - q: 2q > 2^64) and of
// the forward output (the one non-canonical value they can produce is q itself).  Twiddles and
// n^-1 are taken out of Montgomery form when the plan is created.
struct GoldilocksCanonical : Goldilocks {
    static constexpr bool CONDITIONS_INPUT = true;
    static NTTX_HD uint64_t condition_input(uint64_t x, const Ctx&) { return x >= Q ? x - Q : x; }
    static NTTX_HD uint64_t finish(uint64_t x, const Ctx&) { return x == Q ? 0 : x; }
    template <int M>
    static NTTX_HD void finish_group(uint64_t (&x)[M], const Ctx& c)
    {
        NTTX_UNROLL
        for (int i = 0; i < M; i++)
            x[i] = finish(x[i], c);
    }
};

} // namespace nttx
