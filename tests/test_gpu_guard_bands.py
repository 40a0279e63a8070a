"""Out-of-bounds stores: device-resident batches sit between two guard bands of a known pattern, and no
transform, product or split phase may change a word of them.  Sizes and batches are chosen to end on ragged
tiles, on a partial group of sixteen transforms (the cross-batch tiles take the largest multiple of 16, plain
tiles the rest) and on single small transforms; every table kind's arithmetic takes part."""
import numpy as np
import pytest

import _oracle as orc
import _primes

pytestmark = pytest.mark.gpu

Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
QGL = 0xFFFFFFFF00000001
GUARD = 1 << 17          # words on each side (1 MiB)
PATTERN = 0x5AA55AA5DEADBEEF


@pytest.fixture(scope="module")
def amd():
    import ntt_amd
    return ntt_amd


def _tables(amd, kind, n):
    if kind == "mont61":
        return amd.Table.from_prime_and_root(Q61, pow(R61, (1 << 20) // n, Q61), n), Q61
    if kind == "goldilocks":
        return amd.TableGoldilock.from_root(pow(7, (QGL - 1) // (2 * n), QGL), n), QGL
    if kind == "montwide":
        return amd.Table.from_prime_and_root(QGL, pow(7, (QGL - 1) // (2 * n), QGL), n), QGL
    q, psi21, _ = _primes.MID[0]
    return amd.Table.from_prime_and_root(q, _primes.psi_for(q, psi21, n), n), q


@pytest.mark.parametrize("kind", ["mont61", "goldilocks", "montwide", "mont63"])
def test_guard_bands_survive(amd, kind):
    import torch

    cases = [(2, 1), (2, 1025), (4, 257), (6, 65), (8, 17), (10, 5), (12, 1), (12, 3), (14, 5), (16, 1), (16, 15),
             (16, 16), (16, 17), (16, 33), (18, 3), (20, 1)]
    for logn, batch in cases:
        n = 1 << logn
        table, q = _tables(amd, kind, n)
        words = n * batch
        buf = torch.full((GUARD + words + GUARD,), PATTERN, dtype=torch.int64, device="cuda")
        other = torch.full((GUARD + words + GUARD,), PATTERN, dtype=torch.int64, device="cuda")
        host = orc.splitmix_residues(q, words, 9000 + logn * 100 + batch)
        data, data2 = buf[GUARD:GUARD + words], other[GUARD:GUARD + words]
        data.copy_(torch.from_numpy(host.view(np.int64)))
        data2.copy_(data)
        table.forward_inplace(data)
        table.backward_inplace(data)
        torch.cuda.synchronize()
        assert torch.equal(data, data2), (kind, logn, batch, "round trip")
        table.negacyclic_mul_inplace(data, data2)   # the device entry point may use its second operand as scratch
        torch.cuda.synchronize()
        for name, t in (("data", buf), ("second operand", other)):
            assert bool((t[:GUARD] == PATTERN).all()) and bool((t[GUARD + words:] == PATTERN).all()), (kind, logn, batch, name)
