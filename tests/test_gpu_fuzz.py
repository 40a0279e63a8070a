"""A bounded, seeded slice of the randomised parity sweep (tests/_fuzz.py): every table kind -- the
lazy policies, the Goldilocks ones and the strict-Montgomery ones for 62-, 63- and 64-bit primes --
with random sizes, ragged batches, input styles and entry points, against the CPU oracle."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(280)
def test_bounded_fuzz():
    import _fuzz

    cases, seen = _fuzz.run(45.0, 20251111)
    assert cases >= 50
    assert {"mont61", "mont62", "mont63", "mont64", "goldilocks"} <= seen
