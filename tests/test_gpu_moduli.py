"""Parity of the strict-Montgomery kernel policies -- Mont<false> for 2^61 <= q < 2^63 and Mont<true>
for q >= 2^63 other than Goldilocks: REDC with R = 2^64 as utils.rs:165-187 at ntt.rs:147, 221 --
against the oracle's Table, for every pass shape (log2 n = 4..20 incl. the fused tails), ragged
batches, arbitrary u64 inputs, the negacyclic product and the split transform.  Also the small
moduli of the reference's parity domain through the lazy policies."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as orc
import _primes


@pytest.fixture(scope="module")
def amd():
    import ntt_amd
    return ntt_amd


def tables(amd, q, psi21, n):
    psi = _primes.psi_for(q, psi21, n)
    return amd.Table.from_prime_and_root(q, psi, n), orc.Table.from_prime_and_root(q, psi, n)


@pytest.mark.parametrize("q,psi21,policy", _primes.ALL)
@pytest.mark.parametrize("logn,batch", [(4, 33), (6, 5), (8, 1), (8, 37), (10, 5), (12, 33), (14, 3), (16, 1), (16, 17),
                                        (18, 3), (20, 2)])
def test_forward_backward_match_oracle(amd, q, psi21, policy, logn, batch):
    n = 1 << logn
    gpu, ref = tables(amd, q, psi21, n)
    assert gpu.policy == policy
    assert np.array_equal(gpu.powers_psi_bo, ref.powers_psi_bo)
    rng = np.random.default_rng(logn * 100 + batch)
    a = rng.integers(0, 1 << 64, n * batch, dtype=np.uint64)      # any u64: ntt.rs:125-127 reduces it
    a[:n] = orc.splitmix_residues(q, n, 7)
    a[n - 8:n] = np.array([0, 1, q - 1, q, (q + 1) & (2**64 - 1), 2**64 - 1, 2**63, q - 2], dtype=np.uint64)
    want = a.copy()
    for b in range(batch):
        ref.forward_inplace(want[b * n:(b + 1) * n])
    gpu.forward_inplace_batch(a, batch)
    assert np.array_equal(a, want), "forward"
    for b in range(batch):
        ref.backward_inplace(want[b * n:(b + 1) * n])
    gpu.backward_inplace_batch(a, batch)
    assert np.array_equal(a, want), "backward"


@pytest.mark.parametrize("q,psi21,policy", [_primes.WIDE[0], _primes.MID[0], _primes.MID[1]])
@pytest.mark.parametrize("logn,batch", [(8, 3), (10, 5), (12, 17), (16, 2), (18, 3), (20, 2)])
def test_negacyclic_product(amd, q, psi21, policy, logn, batch):
    import torch
    n = 1 << logn
    gpu, ref = tables(amd, q, psi21, n)
    a = orc.splitmix_residues(q, n * batch, 400 + logn)
    b = orc.splitmix_residues(q, n * batch, 500 + logn)
    want = a.copy()
    for i in range(batch):
        fa, fb = a[i * n:(i + 1) * n].copy(), b[i * n:(i + 1) * n].copy()
        ref.forward_inplace(fa)
        ref.forward_inplace(fb)
        prod = (fa.astype(object) * fb.astype(object) % q).astype(np.uint64)
        ref.backward_inplace(prod)
        want[i * n:(i + 1) * n] = prod
    got = a.copy()
    gpu.negacyclic_mul_inplace(got, b)
    assert np.array_equal(got, want), "host path"
    da = torch.from_numpy(a.view(np.int64).copy()).cuda()
    db = torch.from_numpy(b.view(np.int64).copy()).cuda()
    gpu.negacyclic_mul_inplace(da, db)
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy().view(np.uint64), want), "device path"


@pytest.mark.parametrize("q,psi21,policy", [_primes.WIDE[1], _primes.MID[2]])
def test_split_transform(amd, q, psi21, policy):
    n = 1 << 16
    gpu, ref = tables(amd, q, psi21, n)
    a = orc.splitmix_residues(q, n, 77)
    want = a.copy()
    ref.forward_inplace(want)
    gpu.forward_inplace_distributed(a, 4)
    assert np.array_equal(a, want)
    ref.backward_inplace(want)
    gpu.backward_inplace_distributed(a, 8)
    assert np.array_equal(a, want)


def test_avx2_table_on_a_62_bit_modulus(amd):
    """TableAVX2 takes moduli below 2^62 (ntt_avx2.rs:563-589 compares as signed): [2^61, 2^62) runs on
    the strict-Montgomery kernels and must equal both oracle restatements (dft.rs:45-78)."""
    q, psi21, _ = _primes.MID[1]
    n = 1 << 12
    psi = _primes.psi_for(q, psi21, n)
    gpu = amd.TableAVX2.from_prime_and_root(q, psi, n)
    assert gpu.policy == "mont"
    base, avx = orc.Table.from_prime_and_root(q, psi, n), orc.TableAVX2.from_prime_and_root(q, psi, n)
    a = orc.splitmix_residues(q, n, 3)
    w1, w2, g = a.copy(), a.copy().view(np.int64), a.copy().view(np.int64)
    base.forward_inplace(w1)
    avx.forward_inplace(w2)
    gpu.forward_inplace(g)
    assert np.array_equal(g.view(np.uint64), w1) and np.array_equal(g, w2)
    base.backward_inplace(w1)
    gpu.backward_inplace(g)
    assert np.array_equal(g.view(np.uint64), w1) and np.array_equal(g.view(np.uint64), a)


def test_small_and_borderline_moduli(amd):
    """Moduli the reference handles (its k is right) from 97 up to just below 2^62, through whichever
    policy the plan picks; the ones it derives a wrong k for are refused."""
    picked = {}
    for q in (97, 257, 12289, 65537, 786433, 998244353, 4611686018326724609, 2305843009213554689):
        for logn in (2, 4, 6, 8, 10, 12):
            n = 1 << logn
            if (q - 1) % (2 * n):
                continue
            psi = _primes.root_of_unity(q, 2 * n)
            try:
                gpu = amd.Table.from_prime_and_root(q, psi, n)
            except amd.Panic:
                picked[q] = "rejected"
                break
            picked[q] = gpu.policy
            ref = orc.Table.from_prime_and_root(q, psi, n)
            rng = np.random.default_rng(q % 1000 + logn)
            a = rng.integers(0, 2**64, n * 5, dtype=np.uint64)
            a[:n] = rng.integers(0, q, n, dtype=np.uint64)
            w = a.copy()
            for b in range(5):
                ref.forward_inplace(w[b * n:(b + 1) * n])
            gpu.forward_inplace_batch(a, 5)
            assert np.array_equal(a, w), (q, n, "fwd")
            for b in range(5):
                ref.backward_inplace(w[b * n:(b + 1) * n])
            gpu.backward_inplace_batch(a, 5)
            assert np.array_equal(a, w), (q, n, "bwd")
    # utils.rs:148-152 derives a wrong k for all of these but two
    assert picked == {97: "rejected", 257: "rejected", 12289: "rejected", 65537: "rejected", 786433: "rejected",
                      998244353: "shoup61", 4611686018326724609: "rejected", 2305843009213554689: "shoup61_near"}
