"""Mirror of the reference's `ntt::utils` (src/utils.rs) for host-side precompute in Python."""

MASK64 = (1 << 64) - 1


def bit_reverse(n, bits):
    """utils.rs:14-17.  `(64 - bits) / 2` truncates, so only even `bits` give a true reversal."""
    shift = (64 - bits) // 2
    x = (n << shift) & MASK64
    return int(format(x, "064b")[::-1], 2) >> shift


def egcd_i128(a, m):
    """utils.rs:51-64: returns (gcd, x, y) with gcd = a*x + m*y (Rust `/` truncates toward zero)."""
    def tdiv(p, q):
        s = -1 if (p < 0) != (q < 0) else 1
        return s * (abs(p) // abs(q))

    r, s, t = (a, m), (1, 0), (0, 1)
    while r[1] != 0:
        q = tdiv(r[0], r[1])
        r = (r[1], r[0] - q * r[1])
        s = (s[1], s[0] - q * s[1])
        t = (t[1], t[0] - q * t[1])
    return r[0], s[0], t[0]


def mod_exp_u64(base, exp, modulus):
    """utils.rs:76-92"""
    out, acc = 1, base
    while exp > 0:
        if exp % 2 == 1:
            out = (out * acc) % modulus
        acc = (acc * acc) % modulus
        exp >>= 1
    return out


mod_exp_i64 = mod_exp_u64


def find_nth_unity_root_u64(n, modulus):
    """utils.rs:105-119 with the random candidate replaced by 2, 3, 4, ... (reproducible)."""
    c = 2
    while c < modulus:
        g = mod_exp_u64(c, (modulus - 1) // n, modulus)
        if mod_exp_u64(g, n // 2, modulus) != 1:
            return g
        c += 1
    raise ValueError("no n-th root of unity found")


find_nth_unity_root_i64 = find_nth_unity_root_u64


class MontgomeryTransformer:
    """utils.rs:126-193"""

    def __init__(self, k, r_square, modulus):
        self.k, self.r_square, self.modulus = k, r_square, modulus

    @classmethod
    def from_modulus(cls, modulus):
        gcd, x, _ = egcd_i128(modulus, 1 << 64)
        if gcd != 1:
            from .dft import Panic
            raise Panic("modulus is not coprime to helper modulus R=2^64")
        k = (x + modulus) & MASK64 if x < 0 else x & MASK64  # utils.rs:148-152
        r_mod_n = (1 << 64) % modulus
        return cls(k, (r_mod_n * r_mod_n) % modulus, modulus)

    def reduce(self, a):
        m = ((a & MASK64) * self.k) & MASK64
        m_n = m * self.modulus
        y = (((a - m_n) % (1 << 128)) >> 64) & MASK64
        return (y + self.modulus) & MASK64 if a < m_n else y

    def transform(self, a):
        return self.reduce(a * self.r_square)

    def mul(self, a, b):
        return self.reduce(a * b)

    def __eq__(self, o):
        return (self.k, self.r_square, self.modulus) == (o.k, o.r_square, o.modulus)
