"""Lazy-output transforms (nttx_*_lazy: DFT::forward_inplace_lazy / backward_inplace_lazy, dft.rs:8,10):
every output lies in [0, 2q) (the contract at ntt.rs:31-35, 45-49) and is congruent to the oracle's
canonical output; lazy outputs fed to the other direction give the oracle's round trip; plans whose
arithmetic is strict return exactly the strict outputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as orc
import _primes

Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
Q32, R32 = 4293918721, 2004365341
QGL = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def amd():
    import ntt_amd
    return ntt_amd


def make(amd, kind, n):
    if kind == "near61":
        psi = pow(R61, (1 << 20) // n, Q61)
        return amd.Table.from_prime_and_root(Q61, psi, n), orc.Table.from_prime_and_root(Q61, psi, n), Q61
    psi = pow(R32, (1 << 16) // n, Q32)
    return amd.Table.from_prime_and_root(Q32, psi, n), orc.Table.from_prime_and_root(Q32, psi, n), Q32


@pytest.mark.parametrize("kind,policy", [("near61", "shoup61_near"), ("lazy32", "shoup61")])
@pytest.mark.parametrize("logn,batch", [(4, 3), (8, 5), (10, 3), (12, 19), (14, 2), (16, 1), (16, 33)])
def test_lazy_outputs_device(amd, kind, policy, logn, batch):
    import torch
    n = 1 << logn
    gpu, ref, q = make(amd, kind, n)
    assert gpu.policy == policy
    a = orc.splitmix_residues(q, n * batch, 900 + logn)
    fwd = a.copy()
    for b in range(batch):
        ref.forward_inplace(fwd[b * n:(b + 1) * n])
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    gpu.forward_inplace_lazy(d)
    torch.cuda.synchronize()
    lazy = d.cpu().numpy().view(np.uint64)
    assert int(lazy.max()) < 2 * q
    assert np.array_equal(lazy % np.uint64(q), fwd)
    # lazy forward output -> strict backward = the oracle's round trip (the input itself)
    gpu.backward_inplace(d)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), a)
    # strict forward -> lazy backward
    d = torch.from_numpy(fwd.view(np.int64).copy()).cuda()
    gpu.backward_inplace_lazy(d)
    torch.cuda.synchronize()
    back = d.cpu().numpy().view(np.uint64)
    assert int(back.max()) < 2 * q and np.array_equal(back % np.uint64(q), a)
    # lazy forward -> lazy backward -> lazy forward: every hand-over stays in range
    gpu.forward_inplace_lazy(d)
    torch.cuda.synchronize()
    again = d.cpu().numpy().view(np.uint64)
    assert int(again.max()) < 2 * q and np.array_equal(again % np.uint64(q), fwd)


@pytest.mark.parametrize("logn", [8, 12, 16])
def test_lazy_outputs_host(amd, logn):
    n = 1 << logn
    gpu, ref, q = make(amd, "near61", n)
    a = orc.splitmix_residues(q, n, 950 + logn)
    want = a.copy()
    ref.forward_inplace(want)
    x = a.copy()
    gpu.forward_inplace_lazy(x)
    assert int(x.max()) < 2 * q and np.array_equal(x % np.uint64(q), want)
    gpu.backward_inplace_lazy(x)
    assert int(x.max()) < 2 * q and np.array_equal(x % np.uint64(q), a)
    gpu.forward_inplace_core(x, lazy=True)
    assert np.array_equal(x % np.uint64(q), want)


def test_strict_policies_return_canonical_outputs(amd):
    """q >= 2^61 and TableGoldilock: the lazy entry points return what the strict ones return."""
    n = 1 << 12
    q, psi21, _ = _primes.MID[0]
    cases = [(amd.Table.from_prime_and_root(q, _primes.psi_for(q, psi21, n), n), q),
             (amd.TableGoldilock.from_root(pow(7, (QGL - 1) // (2 * n), QGL), n), QGL),
             (amd.Table.from_prime_and_root(QGL, pow(7, (QGL - 1) // (2 * n), QGL), n), QGL)]
    for gpu, modulus in cases:
        a = orc.splitmix_residues(modulus, n, 5)
        s, lz = a.copy(), a.copy()
        gpu.forward_inplace(s)
        gpu.forward_inplace_lazy(lz)
        assert np.array_equal(s, lz), gpu.policy
        gpu.backward_inplace(s)
        gpu.backward_inplace_lazy(lz)
        assert np.array_equal(s, lz) and np.array_equal(s, a), gpu.policy
