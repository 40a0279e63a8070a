"""nttx_transform_host_noalloc (the host-slice transform that stages through a static device array and
allocates nothing) against nttx_forward_host / nttx_backward_host, through the C ABI: two calls in
a row, the second smaller, so the array's leftovers from the first call must not matter."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as orc

Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
QGL = 0xFFFFFFFF00000001


def test_noalloc_matches_host_entry_points():
    import ntt_amd
    from ntt_amd._lib import lib

    n = 1 << 12
    for table, q in ((ntt_amd.Table.from_prime_and_root(Q61, pow(R61, (1 << 20) // n, Q61), n), Q61),
                     (ntt_amd.TableGoldilock.from_root(pow(7, (QGL - 1) // (2 * n), QGL), n), QGL)):
        plan = table._plan
        for batch, seed in ((37, 1), (5, 2)):            # the second call is smaller than the first
            a = orc.splitmix_residues(q, n * batch, seed)
            want, got = a.copy(), a.copy()
            assert lib.nttx_forward_host(plan, want.ctypes.data, batch) == 0
            assert lib.nttx_transform_host_noalloc(plan, got.ctypes.data, batch, 0) == 0, ntt_amd._lib.last_error()
            assert np.array_equal(got, want), "forward"
            assert lib.nttx_backward_host(plan, want.ctypes.data, batch) == 0
            assert lib.nttx_transform_host_noalloc(plan, got.ctypes.data, batch, 1) == 0
            assert np.array_equal(got, want) and np.array_equal(got, a), "backward"
    # limits and argument errors
    big = np.zeros(n * 1025, dtype=np.uint64)             # 1025 * 32 KiB > 32 MiB
    assert lib.nttx_transform_host_noalloc(plan, big.ctypes.data, 1025, 0) == ntt_amd._lib.NTTX_EINVAL
    assert b"32 MiB" in lib.nttx_last_error()
    assert lib.nttx_transform_host_noalloc(plan, big.ctypes.data, 1024, 0) == 0   # exactly 32 MiB fits
    assert lib.nttx_transform_host_noalloc(plan, big.ctypes.data, 1, 2) == ntt_amd._lib.NTTX_EINVAL
    assert lib.nttx_transform_host_noalloc(plan, None, 1, 0) == ntt_amd._lib.NTTX_EINVAL
    assert lib.nttx_transform_host_noalloc(plan, None, 0, 0) == 0
