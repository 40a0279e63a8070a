"""Transforms with odd log2 n (plans of the corrected constructor).

The reference cannot build such a table: its bit_reverse (utils.rs:14-17) returns twice the reversal
for an odd bit count and with_precomputes indexes out of bounds (ntt.rs:253), so there is no
reference output to compare with -- parity unpinned.  The checker here is the negacyclic NTT itself,
stage by stage in plain modular arithmetic with the index structure of ntt.rs:129-171 and twiddles
at the true bit reversal; the CPU oracle restates the reference and panics for these sizes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from _primes import MID, WIDE

Q61 = 0x1FFFFFFFFFE00001
PSI61_2_21 = 0x0B4F9CB5C38B40B1   # primitive 2^21-th root of unity
QGL = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def amd():
    import ntt_amd
    return ntt_amd


def bit_reverse(idx, bits):
    out = np.zeros_like(idx)
    for b in range(bits):
        out |= ((idx >> b) & 1) << (bits - 1 - b)
    return out


def tables(psi, q, n):
    """w[bit_reverse(i)] = psi^i as Python integers."""
    logn = n.bit_length() - 1
    pw = np.empty(n, dtype=object)
    x = 1
    for i in range(n):
        pw[i] = x
        x = x * psi % q
    w = np.empty(n, dtype=object)
    w[bit_reverse(np.arange(n), logn)] = pw
    return w


def model_forward(a, w, q):
    n = a.size
    a = a.astype(object) % q
    t, m = n, 1
    while m < n:
        t //= 2
        x = a.reshape(m, 2, t)
        s = w[m:2 * m].reshape(m, 1)
        u, v = x[:, 0, :], x[:, 1, :] * s % q
        a = np.stack([(u + v) % q, (u - v) % q], axis=1).reshape(-1)
        m *= 2
    return a.astype(np.uint64)


def make(amd, kind, n):
    if kind == "goldilocks":
        psi = pow(7, (QGL - 1) // (2 * n), QGL)
        return amd.TableGoldilock.from_root(psi, n, corrected=True), psi, QGL
    if kind == "mont61":
        q, root = Q61, PSI61_2_21
    elif kind == "mont":
        q, root, _ = MID[0]
    else:
        q, root, _ = WIDE[0]
    psi = pow(root, (1 << 20) // n, q)
    return amd.Table.from_prime_and_root(q, psi, n, exact_k=True), psi, q


def describe(amd, table):
    from ntt_amd._lib import lib
    out = []
    for i in range(lib.nttx_pass_count(table._plan)):
        buf = C.create_string_buffer(128)
        assert lib.nttx_pass_describe(table._plan, 0, i, buf, 128) == 0
        out.append(buf.value.decode())
    return out


def test_reference_constructors_still_refuse_odd_sizes(amd):
    with pytest.raises(amd.Panic):
        amd.Table.from_prime_and_root(Q61, pow(PSI61_2_21, 1 << 11, Q61), 1 << 9)
    with pytest.raises(amd.Panic):
        amd.TableGoldilock.from_root(pow(7, (QGL - 1) >> 10, QGL), 1 << 9)
    with pytest.raises(amd.Panic):
        amd.Table.new_simple()                                   # ntt.rs:72-75


@pytest.mark.parametrize("kind", ["mont61", "goldilocks", "mont", "mont_wide"])
@pytest.mark.parametrize("logn,batch", [(3, 7), (5, 3), (7, 40), (9, 5), (11, 33), (13, 2), (15, 21), (17, 3), (19, 2)])
def test_forward_and_backward(amd, kind, logn, batch):
    n = 1 << logn
    gpu, psi, q = make(amd, kind, n)
    w = tables(psi, q, n)
    rng = np.random.default_rng(1000 * logn + batch)
    a = rng.integers(0, q, n * batch, dtype=np.uint64)
    a[:min(n, 64)] = rng.integers(0, 2**64, min(n, 64), dtype=np.uint64)   # any u64 is accepted
    if kind == "goldilocks":
        a[:min(n, 64)] %= np.uint64(q)                                    # TableGoldilock has no input sweep
    check = 1 if logn >= 17 else min(batch, 2)                            # vectors compared with the model
    want = np.concatenate([model_forward(a[b * n:(b + 1) * n], w, q) for b in range(check)])
    got = a.copy()
    gpu.forward_inplace_batch(got, batch)
    assert np.array_equal(got[:check * n], want), describe(amd, gpu)
    # the other vectors of the batch: the same transform, by linearity against the first
    if batch > 2:
        s = (got[2 * n:3 * n].astype(object) + got[:n].astype(object)) % q
        both = ((a[2 * n:3 * n].astype(object) + a[:n].astype(object)) % q).astype(np.uint64)
        t = both.copy()
        gpu.forward_inplace(t)
        assert np.array_equal(t.astype(object) % q, s)
    gpu.backward_inplace_batch(got, batch)
    assert np.array_equal(got, a % np.uint64(q))
    # tables read back: true bit reversal
    if logn <= 13 and kind == "goldilocks":
        assert np.array_equal(gpu.powers_psi_bo.astype(object), w)


def test_schedules(amd):
    """Odd sizes above 2^8: a first pass with an odd stage count, then the even schedule."""
    want = {9: 2, 11: 2, 13: 2, 15: 2, 17: 2, 19: 2, 21: 3}
    for logn, launches in want.items():
        gpu, _, _ = make(amd, "goldilocks", 1 << logn)
        d = describe(amd, gpu)
        assert len(d) == launches, (logn, d)


def test_large_size_by_closed_form(amd):
    """n = 2^21 (tables generated on the device): a delta at position j transforms to
    psi^(j * (2 * bit_reverse(i) + 1)); plus the round trip of a random vector."""
    logn = 21
    n = 1 << logn
    gpu, psi, q = make(amd, "goldilocks", n)
    j = 5
    a = np.zeros(n, dtype=np.uint64)
    a[j] = 1
    gpu.forward_inplace(a)
    rev = bit_reverse(np.arange(n), logn)
    sample = np.random.default_rng(3).integers(0, n, 4096)
    for i in sample:
        assert int(a[i]) == pow(psi, j * (2 * int(rev[i]) + 1), q)
    r = np.random.default_rng(4).integers(0, q, n, dtype=np.uint64)
    x = r.copy()
    gpu.forward_inplace(x)
    gpu.backward_inplace(x)
    assert np.array_equal(x, r)


@pytest.mark.parametrize("kind", ["mont61", "goldilocks"])
@pytest.mark.parametrize("logn,batch", [(5, 3), (9, 4), (15, 20)])
def test_product(amd, kind, logn, batch):
    """Negacyclic product at odd sizes (2^15 with 16 or more pairs takes the fused middle launch)
    against forward, coefficient-wise product, backward done with the verified transforms."""
    import torch
    n = 1 << logn
    gpu, psi, q = make(amd, kind, n)
    rng = np.random.default_rng(77 + logn)
    a = rng.integers(0, q, n * batch, dtype=np.uint64)
    b = rng.integers(0, q, n * batch, dtype=np.uint64)
    fa, fb = a.copy(), b.copy()
    gpu.forward_inplace_batch(fa, batch)
    gpu.forward_inplace_batch(fb, batch)
    want = (fa.astype(object) * fb.astype(object) % q).astype(np.uint64)
    gpu.backward_inplace_batch(want, batch)
    got = a.copy()
    gpu.negacyclic_mul_inplace(got, b)
    assert np.array_equal(got, want)
    da = torch.from_numpy(a.view(np.int64).copy()).cuda()
    db = torch.from_numpy(b.view(np.int64).copy()).cuda()
    gpu.negacyclic_mul_inplace(da, db)
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy().view(np.uint64), want)
    if logn == 5:   # X^(n-1) * X = -1
        x1, xn1 = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        x1[1], xn1[n - 1] = 1, 1
        gpu.negacyclic_mul_inplace(xn1, x1)
        assert int(xn1[0]) == q - 1 and not xn1[1:].any()
