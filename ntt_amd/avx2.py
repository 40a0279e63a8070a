"""The free functions of the reference's AVX2 module (src/dft/ntt_avx2.rs:704-813) on lists of four
signed 64-bit lanes: the same names and results without the x86 intrinsics, for host code written
against that module.  The GPU path does not use them."""

_M = (1 << 64) - 1


def _u(x):
    return x & _M


def _s(x):
    x &= _M
    return x - (1 << 64) if x >> 63 else x


def extract_m256i_to_i64(a):  # ntt_avx2.rs:706-716
    return [_s(x) for x in a]


def extract_m256d_to_f64(a):  # ntt_avx2.rs:720-725
    return [float(x) for x in a]


def a_lt_b(a, b):
    """ntt_avx2.rs:728-744: -1 where a < b with the lanes read as unsigned integers, else 0."""
    return [-1 if _u(x) < _u(y) else 0 for x, y in zip(a, b)]


def a_mul_b_karatsuba(a, b):
    """ntt_avx2.rs:747-779: (high, low) words of the unsigned 128-bit products."""
    prods = [_u(x) * _u(y) for x, y in zip(a, b)]
    return [_s(p >> 64) for p in prods], [_s(p) for p in prods]


def a_sub_b(a_high, a_low, b_high, b_low):
    """ntt_avx2.rs:782-796: (high, low) of a - b on 128-bit values, wrapping."""
    diffs = [(((_u(ah) << 64) | _u(al)) - ((_u(bh) << 64) | _u(bl))) & ((1 << 128) - 1)
             for ah, al, bh, bl in zip(a_high, a_low, b_high, b_low)]
    return [_s(d >> 64) for d in diffs], [_s(d) for d in diffs]


def a_lt_b_wide(a_high, a_low, b_high, b_low):
    """ntt_avx2.rs:799-813: -1 where a < b on unsigned 128-bit values, else 0."""
    return [-1 if ((_u(ah) << 64) | _u(al)) < ((_u(bh) << 64) | _u(bl)) else 0
            for ah, al, bh, bl in zip(a_high, a_low, b_high, b_low)]
