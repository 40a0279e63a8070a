"""NTT-friendly moduli for the strict-Montgomery kernel policies (test helper).

Deterministic search results for q = c * 2^21 + 1 prime, kept only where the reference's
MontgomeryTransformer::from_modulus (utils.rs:141-162, whose k derivation is wrong whenever the
Bezout coefficient comes out negative, utils.rs:148-152) yields k * q = 1 mod 2^64.  `psi` is a
primitive 2^21-th root of unity (smallest generator candidate g = 2, 3, ...: g^((q-1)/2^21)).
tests/test_moduli_cpu.py re-derives every constant."""

# (q, psi of order 2^21, policy the plan must select)
WIDE = [  # q >= 2^63, not Goldilocks -> Mont<true> ("mont_wide")
    (0xFFFFFFFFFF000001, 0x4DDB0AED1485BDF3, "mont_wide"),
    (0xFFFFFFFFF4C00001, 0xAA3994472E9E11FD, "mont_wide"),
]
MID = [  # 2^61 <= q < 2^63 -> Mont<false> ("mont")
    (0x7FFFFFFFF9E00001, 0x486B976416398E05, "mont"),
    (0x2000000000A00001, 0x027EC9C4370CD1D2, "mont"),
    (0x2000000008200001, 0x1EC5D618C37700E6, "mont"),
]
ALL = WIDE + MID


def is_prime(n):
    """Deterministic Miller-Rabin for n < 2^64 (first twelve primes as bases)."""
    if n < 2:
        return False
    bases = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in bases:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in bases:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def root_of_unity(q, order):
    """Primitive `order`-th root of unity mod prime q (order a power of two dividing q - 1)."""
    g = 2
    while True:
        r = pow(g, (q - 1) // order, q)
        if pow(r, order // 2, q) == q - 1:
            return r
        g += 1


def psi_for(q, psi21, n):
    """Primitive 2n-th root nested in the 2^21-th root `psi21`."""
    return pow(psi21, (1 << 21) // (2 * n), q)
