"""The moduli tests/_primes.py lists for the strict-Montgomery kernels: prime, NTT-friendly, in the
range of the policy they are meant to select, and inside the reference's parity domain (its k
derivation, utils.rs:148-152, is right for them)."""
import _oracle as orc
import _primes
from ntt_amd import utils


def test_listed_moduli():
    for q, psi, policy in _primes.ALL:
        assert _primes.is_prime(q) and (q - 1) % (1 << 21) == 0
        assert pow(psi, 1 << 20, q) == q - 1 and psi == _primes.root_of_unity(q, 1 << 21)
        k = utils.MontgomeryTransformer.from_modulus(q).k
        assert k * q % (1 << 64) == 1 and k == orc.MontgomeryTransformer.from_modulus(q).k
        assert q != 0xFFFFFFFF00000001
        assert (q >= 1 << 63) == (policy == "mont_wide") and q >= 1 << 61


def test_search_is_deterministic():
    """First two primes c * 2^21 + 1 below 2^64 with a correct reference k, searching downwards."""
    found, c = [], ((1 << 64) - 1) >> 21
    while len(found) < 2:
        q = c * (1 << 21) + 1
        if q < 1 << 64 and _primes.is_prime(q) and utils.MontgomeryTransformer.from_modulus(q).k * q % (1 << 64) == 1:
            found.append(q)
        c -= 1
    assert found == [q for q, _, _ in _primes.WIDE]
