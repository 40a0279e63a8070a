"""nttx_transform_host_noalloc (the host-slice transform that stages through a static device array and
allocates nothing) against nttx_forward_host / nttx_backward_host AND against the oracle, through the
C ABI: two calls in a row, the second smaller, so the array's leftovers from the first call must not
matter; then calls from several threads at once, which the library serialises on a per-device lock."""
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
    psi61, psigl = pow(R61, (1 << 20) // n, Q61), pow(7, (QGL - 1) // (2 * n), QGL)
    for table, ref, q in ((ntt_amd.Table.from_prime_and_root(Q61, psi61, n), orc.Table.from_prime_and_root(Q61, psi61, n), Q61),
                          (ntt_amd.TableGoldilock.from_root(psigl, n), orc.TableGoldilock.from_root(psigl, n), QGL)):
        plan = table._plan
        for batch, seed in ((37, 1), (5, 2)):            # the second call is smaller than the first
            a = orc.splitmix_residues(q, n * batch, seed)
            want, got = a.copy(), a.copy()
            assert lib.nttx_forward_host(plan, want.ctypes.data, batch) == 0
            assert lib.nttx_transform_host_noalloc(plan, got.ctypes.data, batch, 0) == 0, ntt_amd._lib.last_error()
            assert np.array_equal(got, want), "forward"
            oracle = a.copy()
            for b in range(batch):
                ref.forward_inplace(oracle[b * n:(b + 1) * n])
            assert np.array_equal(got, oracle), "forward against the oracle"
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


def test_noalloc_calls_from_several_threads_take_turns():
    """All calls on a GPU share that GPU's one array; a per-device lock held for the whole call keeps
    concurrent callers (two plans, four threads) from overwriting each other's data."""
    import threading

    import ntt_amd
    from ntt_amd._lib import lib

    n = 1 << 12
    psi61, psigl = pow(R61, (1 << 20) // n, Q61), pow(7, (QGL - 1) // (2 * n), QGL)
    cases = [(ntt_amd.Table.from_prime_and_root(Q61, psi61, n), orc.Table.from_prime_and_root(Q61, psi61, n), Q61),
             (ntt_amd.TableGoldilock.from_root(psigl, n), orc.TableGoldilock.from_root(psigl, n), QGL)]
    errors = []

    def worker(k):
        table, ref, q = cases[k % 2]
        batch = 3 + k
        for rep in range(6):
            a = orc.splitmix_residues(q, n * batch, 100 * k + rep)
            got, want = a.copy(), a.copy()
            if lib.nttx_transform_host_noalloc(table._plan, got.ctypes.data, batch, 0) != 0:
                errors.append("call failed")
                return
            for b in range(batch):
                ref.forward_inplace(want[b * n:(b + 1) * n])
            if not np.array_equal(got, want):
                errors.append("thread %d rep %d: wrong result" % (k, rep))
                return

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
