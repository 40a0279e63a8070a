"""Negacyclic product (nttx_negacyclic_mul_*): equal, bit for bit, to the composition the
reference's callers write -- forward_inplace(a), forward_inplace(b), coefficient-wise product,
backward_inplace -- computed here with the CPU oracle, and to the schoolbook product in
Z_q[X]/(X^n + 1) computed with Python integers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as orc
from _primes import MID

Q61 = 0x1FFFFFFFFFE00001
PSI61_2_21 = 0x0B4F9CB5C38B40B1
QGL = 0xFFFFFFFF00000001
Q32 = 4293918721
PSI32_2_17 = 2004365341


@pytest.fixture(scope="module")
def amd():
    import ntt_amd
    return ntt_amd


def make(amd, kind, n):
    if kind == "mont61":
        psi = pow(PSI61_2_21, (1 << 20) // n, Q61)
        return amd.Table.from_prime_and_root(Q61, psi, n), orc.Table.from_prime_and_root(Q61, psi, n), Q61
    if kind == "mont32":
        psi = pow(PSI32_2_17, (1 << 16) // n, Q32)
        return amd.Table.from_prime_and_root(Q32, psi, n), orc.Table.from_prime_and_root(Q32, psi, n), Q32
    psi = pow(7, (QGL - 1) // (2 * n), QGL)
    if kind == "montwide":   # the Goldilocks prime through the Montgomery table (ntt.rs:60-62)
        return amd.Table.from_prime_and_root(QGL, psi, n), orc.Table.from_prime_and_root(QGL, psi, n), QGL
    if kind == "goldilocks":
        return amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n), QGL
    q, root, _ = MID[1]   # strict Montgomery kernels (2^61 <= q < 2^63)
    psi = pow(root, (1 << 20) // n, q)
    return amd.Table.from_prime_and_root(q, psi, n), orc.Table.from_prime_and_root(q, psi, n), q


def composition(ref, q, a, b, n):
    """forward(a), forward(b), coefficient-wise product mod q, backward -- per polynomial."""
    out = a.copy()
    for i in range(a.size // n):
        fa, fb = a[i * n:(i + 1) * n].copy(), b[i * n:(i + 1) * n].copy()
        ref.forward_inplace(fa)
        ref.forward_inplace(fb)
        prod = (fa.astype(object) * fb.astype(object) % q).astype(np.uint64)
        ref.backward_inplace(prod)
        out[i * n:(i + 1) * n] = prod
    return out


def schoolbook(a, b, q):
    n = len(a)
    c = [0] * n
    for i, x in enumerate(a):
        x = int(x)
        for j, y in enumerate(b):
            k = i + j
            if k < n:
                c[k] = (c[k] + x * int(y)) % q
            else:
                c[k - n] = (c[k - n] - x * int(y)) % q
    return np.array(c, dtype=np.uint64)


@pytest.mark.parametrize("kind", ["mont61", "mont32", "montwide", "goldilocks"])
@pytest.mark.parametrize("n", [16, 256])
def test_schoolbook(amd, kind, n):
    gpu, _, q = make(amd, kind, n)
    a = orc.splitmix_residues(q, n, 11 + n)
    b = orc.splitmix_residues(q, n, 12 + n)
    want = schoolbook(a, b, q)
    got = a.copy()
    gpu.negacyclic_mul_inplace(got, b)
    assert np.array_equal(got, want)
    # X^(n-1) * X = -1
    x1, xn1 = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    x1[1], xn1[n - 1] = 1, 1
    gpu.negacyclic_mul_inplace(xn1, x1)
    assert int(xn1[0]) == q - 1 and not xn1[1:].any()


@pytest.mark.parametrize("kind", ["mont61", "montwide", "goldilocks"])
@pytest.mark.parametrize("logn,batch", [(8, 3), (10, 5), (12, 17), (16, 1), (16, 40), (18, 3), (20, 2)])
def test_matches_composition_host_and_device(amd, kind, logn, batch):
    import torch
    n = 1 << logn
    gpu, ref, q = make(amd, kind, n)
    a = orc.splitmix_residues(q, n * batch, 100 + logn)
    b = orc.splitmix_residues(q, n * batch, 200 + logn)
    want = composition(ref, q, a, b, n)
    got, b_before = a.copy(), b.copy()
    gpu.negacyclic_mul_inplace(got, b)
    assert np.array_equal(got, want), "host path"
    assert np.array_equal(b, b_before), "the host entry point must leave b alone"
    da = torch.from_numpy(a.view(np.int64).copy()).cuda()
    db = torch.from_numpy(b.view(np.int64).copy()).cuda()
    gpu.negacyclic_mul_inplace(da, db)
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy().view(np.uint64), want), "device path"


@pytest.mark.parametrize("kind", ["mont61", "mont32", "montwide", "goldilocks", "montstrict"])
@pytest.mark.parametrize("batch", [16, 35])
def test_fused_middle(amd, kind, batch):
    """n = 2^16 with 16 or more pairs: the last forward pass of both operands, the product and the
    first backward pass run in one launch on groups of 16 pairs; the pairs left over take the
    separate launches.  Product and square, on device memory."""
    import torch
    n = 1 << 16
    gpu, ref, q = make(amd, kind, n)
    a = orc.splitmix_residues(q, n * batch, 400 + batch)
    b = orc.splitmix_residues(q, n * batch, 500 + batch)
    da = torch.from_numpy(a.view(np.int64).copy()).cuda()
    db = torch.from_numpy(b.view(np.int64).copy()).cuda()
    gpu.negacyclic_mul_inplace(da, db)
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy().view(np.uint64), composition(ref, q, a, b, n))
    ds = torch.from_numpy(a.view(np.int64).copy()).cuda()
    gpu.negacyclic_mul_inplace(ds, ds)
    torch.cuda.synchronize()
    assert np.array_equal(ds.cpu().numpy().view(np.uint64), composition(ref, q, a, a, n))


def test_square_and_overlap(amd):
    import torch
    n, batch = 1 << 12, 20
    gpu, ref, q = make(amd, "mont61", n)
    a = orc.splitmix_residues(q, n * batch, 300)
    want = composition(ref, q, a, a, n)
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    gpu.negacyclic_mul_inplace(d, d)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), want)
    # a host batch large enough to be streamed in chunks, squared in place: same bits as the device path
    batch = 600
    a = orc.splitmix_residues(q, n * batch, 301)
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    gpu.negacyclic_mul_inplace(d, d)
    torch.cuda.synchronize()
    h = a.copy()
    gpu.negacyclic_mul_inplace(h, h)
    assert np.array_equal(h, d.cpu().numpy().view(np.uint64))
    assert np.array_equal(h[:n], composition(ref, q, a[:n], a[:n], n))
    big = torch.zeros(3 * n, dtype=torch.int64, device="cuda")
    with pytest.raises(amd.Panic):
        gpu.negacyclic_mul_inplace(big[:2 * n], big[n:])     # partial overlap
    hbig = np.zeros(3 * n, dtype=np.uint64)
    with pytest.raises(amd.Panic):
        gpu.negacyclic_mul_inplace(hbig[:2 * n], hbig[n:])   # the host entry point refuses it too


def test_goldilocks_product_after_forward_emits_q(amd):
    """The forward transform can store q itself (ntt_goldilock.rs:107); the coefficient-wise
    product is mul_reduce, canonical for that operand too."""
    n = 256
    gpu, ref, q = make(amd, "goldilocks", n)
    s = int(ref.powers_psi_bo[1])
    v = 12345
    a = np.zeros(n, dtype=np.uint64)
    a[n // 2] = v
    a[0] = (q - (v * s) % q) % q
    fa = a.copy()
    ref.forward_inplace(fa)
    assert (fa == np.uint64(q)).any()
    b = orc.splitmix_residues(q, n, 301)
    want = composition(ref, q, a, b, n)
    got = a.copy()
    gpu.negacyclic_mul_inplace(got, b)
    assert np.array_equal(got, want)
    assert np.array_equal(got, schoolbook(a, b, q))
