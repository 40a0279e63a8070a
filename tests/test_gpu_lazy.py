"""The trait's lazy methods (DFT::forward_inplace_lazy / backward_inplace_lazy, dft.rs:8,10) return what the
reference's return: its LAZY parameter is never read (ntt.rs:122, 178), so lazy == strict, bit for bit
against the oracle.

The explicit opt-in (forward_inplace_range_2q / backward_inplace_range_2q over nttx_*_lazy) keeps the
contract the reference documents for those methods (ntt.rs:31-35, 45-49): every output lies in [0, 2q)
and is congruent to the oracle's canonical output; such outputs fed to the other direction give the
oracle's round trip; plans whose arithmetic is strict return exactly the strict outputs."""
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
    gpu.forward_inplace_range_2q(d)
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
    gpu.backward_inplace_range_2q(d)
    torch.cuda.synchronize()
    back = d.cpu().numpy().view(np.uint64)
    assert int(back.max()) < 2 * q and np.array_equal(back % np.uint64(q), a)
    # lazy forward -> lazy backward -> lazy forward: every hand-over stays in range
    gpu.forward_inplace_range_2q(d)
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
    gpu.forward_inplace_range_2q(x)
    assert int(x.max()) < 2 * q and np.array_equal(x % np.uint64(q), want)
    gpu.backward_inplace_range_2q(x)
    assert int(x.max()) < 2 * q and np.array_equal(x % np.uint64(q), a)
    x %= np.uint64(q)
    gpu.forward_inplace_core(x, lazy=True)  # the core ignores LAZY, as the reference's does
    assert np.array_equal(x, want)


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
        gpu.forward_inplace_range_2q(lz)
        assert np.array_equal(s, lz), gpu.policy
        gpu.backward_inplace(s)
        gpu.backward_inplace_range_2q(lz)
        assert np.array_equal(s, lz) and np.array_equal(s, a), gpu.policy


@pytest.mark.parametrize("which", ["table", "avx2", "goldilock"])
def test_trait_lazy_methods_equal_the_oracles(amd, which):
    """Default binding: table.forward_inplace_lazy == orc.forward_inplace_lazy, bit for bit, on host slices
    and on a device-resident batch (a caller ported from the reference may treat the outputs as canonical)."""
    import torch
    n = 1 << 16
    if which == "table":
        gpu, ref, q, dt = amd.Table.new(), orc.Table.new(), Q61, np.uint64
    elif which == "avx2":
        gpu, ref, q, dt = amd.TableAVX2.new(), orc.TableAVX2.new(), Q61, np.int64
    else:
        gpu, ref, q, dt = amd.TableGoldilock.new(), orc.TableGoldilock.new(), QGL, np.uint64
    a = orc.splitmix_residues(q, n, 77).view(dt)
    want = a.copy()
    ref.forward_inplace_lazy(want)
    x = a.copy()
    gpu.forward_inplace_lazy(x)
    assert np.array_equal(x, want)
    y = a.copy()
    gpu.forward_inplace_core(y, lazy=True)
    assert np.array_equal(y, want)
    d = torch.from_numpy(np.concatenate([a, a]).view(np.int64).copy()).cuda()
    gpu.forward_inplace_lazy(d)
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(dt)
    assert np.array_equal(got[:n], want) and np.array_equal(got[n:], want)
    ref.backward_inplace_lazy(want)
    gpu.backward_inplace_lazy(x)
    assert np.array_equal(x, want) and np.array_equal(x, a)
    gpu.backward_inplace_lazy(d)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(dt)[:n], a)
