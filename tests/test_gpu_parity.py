"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Mirrors the reference's own tests (dft.rs:23-189, ntt.rs:279-295) with the GPU tables in
place of the CPU ones, and adds every pass shape, batch raggedness and the edge cases the
reference's arithmetic has (inputs >= q for the base table, the non-canonical q the
Goldilocks forward can emit, 64-bit moduli through the Montgomery table).
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as orc

Q61 = 0x1FFFFFFFFFE00001
PSI61_2_21 = 0x0B4F9CB5C38B40B1   # 37^((q-1)/2^21): primitive 2^21-th root, PSI61 = this^16
PSI61 = 0x15EB043C7AA2B01F
QGL = 0xFFFFFFFF00000001
PSIGL = 0xABD0A6E8AA3D8A0E
GL_GEN = 7
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).astype("<u8").tobytes()).hexdigest()


def psi61_for(n):
    assert pow(PSI61_2_21, 16, Q61) == PSI61
    return pow(PSI61_2_21, (1 << 20) // n, Q61)


def psigl_for(n):
    return pow(GL_GEN, (QGL - 1) // (2 * n), QGL)


@pytest.fixture(scope="module")
def amd():
    import ntt_amd
    return ntt_amd


def _batch_check(gpu_tab, orc_tab, q, n, batch, seed, dtype=np.uint64):
    a0 = orc.splitmix_residues(q, n * batch, seed)
    want = a0.copy()
    for b in range(batch):
        orc_tab.forward_inplace(want[b * n:(b + 1) * n].view(dtype))
    got = a0.copy().view(dtype)
    gpu_tab.forward_inplace_batch(got, batch)
    assert np.array_equal(got.view(np.uint64), want), "forward mismatch"
    for b in range(batch):
        orc_tab.backward_inplace(want[b * n:(b + 1) * n].view(dtype))
    gpu_tab.backward_inplace_batch(got, batch)
    assert np.array_equal(got.view(np.uint64), want), "backward mismatch"
    assert np.array_equal(want, a0), "oracle round trip"


@pytest.mark.parametrize("logn", [2, 4, 6, 8, 10, 12, 14, 16, 18, 20])
def test_mont61_all_pass_shapes(amd, logn):
    n = 1 << logn
    psi = psi61_for(n)
    gpu = amd.Table.from_prime_and_root(Q61, psi, n)
    ref = orc.Table.from_prime_and_root(Q61, psi, n)
    assert np.array_equal(gpu.powers_psi_bo, ref.powers_psi_bo)
    assert np.array_equal(gpu.powers_psi_inv_bo, ref.powers_psi_inv_bo)
    for batch in ([1, 3, 37] if logn <= 16 else [1, 3]):
        _batch_check(gpu, ref, Q61, n, batch, 1000 + logn * 10 + batch)


@pytest.mark.parametrize("logn", [2, 4, 6, 8, 10, 12, 14, 16, 18, 20])
def test_goldilocks_all_pass_shapes(amd, logn):
    n = 1 << logn
    psi = psigl_for(n)
    gpu = amd.TableGoldilock.from_root(psi, n)
    ref = orc.TableGoldilock.from_root(psi, n)
    assert np.array_equal(gpu.powers_psi_bo, ref.powers_psi_bo)
    assert np.array_equal(gpu.powers_psi_inv_bo, ref.powers_psi_inv_bo)
    assert gpu.n_inv == ref.n_inv
    for batch in ([1, 5, 33] if logn <= 16 else [1, 2]):
        _batch_check(gpu, ref, QGL, n, batch, 2000 + logn * 10 + batch)


@pytest.mark.parametrize("logn", [4, 8, 12, 16])
def test_goldilocks_prime_through_montgomery_table(amd, logn):
    """Table::new_goldilock (ntt.rs:60-62): a 64-bit modulus in the Montgomery path."""
    n = 1 << logn
    psi = psigl_for(n)
    gpu = amd.Table.from_prime_and_root(QGL, psi, n)
    ref = orc.Table.from_prime_and_root(QGL, psi, n)
    _batch_check(gpu, ref, QGL, n, 3, 3000 + logn)


@pytest.mark.parametrize("logn", [22, 24])
def test_sizes_between_the_benchmark_ones(amd, logn):
    """2^22 = a 6-stage first pass, then 8 + 8; 2^24 = three 8-stage passes: pass plans no benchmark size takes.
    One vector and a pair, both table kinds; the forward output and the round trip against the oracle."""
    n = 1 << logn
    psi = psigl_for(n)
    for gpu, ref, q in ((amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n), QGL),
                        (amd.Table.from_prime_and_root(QGL, psi, n), orc.Table.from_prime_and_root(QGL, psi, n), QGL)):
        for batch in (1, 2):
            x = orc.splitmix_residues(q, n * batch, 2200 + logn + batch)
            a, w = x.copy(), x.copy()
            gpu.forward_inplace_batch(a, batch)
            for b in range(batch):
                ref.forward_inplace(w[b * n:(b + 1) * n])
            assert np.array_equal(a, w), ("forward", logn, batch)
            gpu.backward_inplace_batch(a, batch)
            assert np.array_equal(a, x), ("round trip", logn, batch)


def test_two_to_the_26_goldilocks(amd):
    """log2 n = 2 mod 8 above 2^18: two 8-stage passes, then the last 10 stages in one launch."""
    n = 1 << 26
    psi = psigl_for(n)
    gpu, ref = amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    x = orc.splitmix_residues(QGL, n, 26)
    a, w = x.copy(), x.copy()
    gpu.forward_inplace(a)
    ref.forward_inplace(w)
    assert np.array_equal(a, w), "forward"
    gpu.backward_inplace(a)
    assert np.array_equal(a, x), "round trip"


def test_base_version(amd):  # dft.rs:23-42
    table = amd.Table.new_goldilock()
    a = orc.splitmix_residues(table.q, 1 << 16, 11)
    a_cloned = a.copy()
    table.forward_inplace(a)
    assert np.any(a != a_cloned)
    table.backward_inplace(a)
    assert np.array_equal(a, a_cloned)


def test_avx_correctness(amd):  # dft.rs:45-78, both sides on the GPU and both against the CPU oracles
    table, table_avx = amd.Table.new(), amd.TableAVX2.new()
    o_base, o_avx = orc.Table.new(), orc.TableAVX2.new()
    a_avx2 = orc.splitmix_residues(table_avx.q, 1 << 16, 12).astype(np.int64)
    a = a_avx2.astype(np.uint64)
    a_cloned = a.copy()
    w_base, w_avx = a.copy(), a_avx2.copy()

    table.forward_inplace(a)
    table_avx.forward_inplace(a_avx2)
    o_base.forward_inplace(w_base)
    o_avx.forward_inplace(w_avx)
    assert np.array_equal(a, a_avx2.view(np.uint64))
    assert np.array_equal(a, w_base)
    assert np.array_equal(a_avx2, w_avx)

    table.backward_inplace(a)
    table_avx.backward_inplace(a_avx2)
    o_avx.backward_inplace(w_avx)
    assert np.array_equal(a, a_avx2.view(np.uint64))
    assert np.array_equal(a_avx2, w_avx)
    assert np.array_equal(a, a_cloned)


def test_goldilock_correctness(amd):  # dft.rs:127-152
    table, table_goldilock = amd.Table.new_goldilock(), amd.TableGoldilock.new()
    a_goldilock = orc.splitmix_residues(table_goldilock.q, 1 << 16, 13)
    a_cloned = a_goldilock.copy()
    a = a_goldilock.copy()
    table.forward_inplace(a)
    table_goldilock.forward_inplace(a_goldilock)
    assert np.array_equal(a, a_goldilock)
    w = a_cloned.copy()
    orc.TableGoldilock.new().forward_inplace(w)
    assert np.array_equal(a_goldilock, w)
    table.backward_inplace(a)
    table_goldilock.backward_inplace(a_goldilock)
    assert np.array_equal(a, a_goldilock)
    assert np.array_equal(a, a_cloned)


def test_avx_against_base(amd):  # dft.rs:81-124: ten forward + backward rounds on each table return the input
    table, table_avx = amd.Table.new(), amd.TableAVX2.new()
    a_avx2 = orc.splitmix_residues(table_avx.q, 1 << 16, 15).astype(np.int64)
    a = a_avx2.astype(np.uint64)
    a_cloned = a.copy()
    for _ in range(10):
        table.forward_inplace(a)
        table.backward_inplace(a)
    assert np.array_equal(a, a_cloned)
    for _ in range(10):
        table_avx.forward_inplace(a_avx2)
        table_avx.backward_inplace(a_avx2)
    assert np.array_equal(a_cloned, a_avx2.view(np.uint64))


def test_goldilock_against_base(amd):  # dft.rs:155-189
    table, table_goldilock = amd.Table.new_goldilock(), amd.TableGoldilock.new()
    a_goldilock = orc.splitmix_residues(table_goldilock.q, 1 << 16, 16)
    a = a_goldilock.copy()
    a_cloned = a.copy()
    for _ in range(10):
        table.forward_inplace(a)
        table.backward_inplace(a)
    for _ in range(10):
        table_goldilock.forward_inplace(a_goldilock)
        table_goldilock.backward_inplace(a_goldilock)
    assert np.array_equal(a_cloned, a)
    assert np.array_equal(a_cloned, a_goldilock)


def test_normal_case(amd):  # ntt.rs:279-295 (`2 ^ 61` is XOR: inputs in 0..63)
    table = amd.Table.new()
    a = (orc.splitmix_residues(Q61, 1 << 16, 14) % np.uint64(63)).astype(np.uint64)
    a_clone = a.copy()
    table.forward_inplace(a)
    table.backward_inplace(a)
    assert np.array_equal(a_clone, a)


def test_lazy_methods_agree_with_strict(amd):
    """ntt.rs:31-35, 45-49: the lazy methods promise [0, 2q); the reference never reads LAZY and
    returns the canonical residues of the strict methods, and so does the trait binding here."""
    table = amd.Table.new()
    a = orc.splitmix_residues(Q61, 1 << 16, 15)
    b = a.copy()
    table.forward_inplace(a)
    table.forward_inplace_lazy(b)
    assert np.array_equal(a, b)
    table.backward_inplace(a)
    table.backward_inplace_lazy(b)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("logn", [4, 8, 12, 16, 20])
def test_lazily_reduced_inputs(amd, logn):
    """ntt.rs:31-35, 45-49 document [0, 2q) inputs for the lazy methods.  The kernels feed anything
    below 4q to the butterflies unreduced; the outputs must still be the reference's, which
    reduces every input first (ntt.rs:125-127).  Multiples of q up to 8q and their neighbours sit
    on every threshold of the lazy arithmetic."""
    n = 1 << logn
    psi = psi61_for(n)
    gpu, ref = amd.Table.from_prime_and_root(Q61, psi, n), orc.Table.from_prime_and_root(Q61, psi, n)
    rng = np.random.default_rng(160 + logn)
    res = orc.splitmix_residues(Q61, n, 16)
    for span in (2, 4, 8):
        raw = res + np.uint64(Q61) * rng.integers(0, span, n, dtype=np.uint64)  # 8q - 1 < 2^64
        edge = np.array([m * Q61 + d for m in range(1, 8) for d in (-1, 0, 1)], dtype=np.uint64)[:n]
        raw[:edge.size] = edge            # first operands of the first forward stage ...
        if n >= 2 * edge.size:
            raw[-edge.size:] = edge[::-1]  # ... and second operands
        for fn in ("forward_inplace_lazy", "backward_inplace_lazy", "forward_inplace", "backward_inplace"):
            a, w = raw.copy(), raw.copy()
            getattr(gpu, fn)(a)
            getattr(ref, fn)(w)
            assert np.array_equal(a, w), (fn, span)


def test_u32_compatible(amd):  # ntt.rs:66-68
    gpu, ref = amd.Table.new_u32_compatible(), orc.Table.new_u32_compatible()
    _batch_check(gpu, ref, 4293918721, 1 << 16, 2, 16)


def test_base_accepts_any_u64(amd):  # ntt.rs:125-127, 181-183
    rng = np.random.default_rng(17)
    for gpu, ref in ((amd.Table.new(), orc.Table.new()), (amd.Table.new_goldilock(), orc.Table.new_goldilock())):
        raw = rng.integers(0, 2**64, 1 << 16, dtype=np.uint64)
        raw[:4] = [2**64 - 1, gpu.q, gpu.q + 1, gpu.q - 1]
        a, w = raw.copy(), raw.copy()
        gpu.forward_inplace(a)
        ref.forward_inplace(w)
        assert np.array_equal(a, w)
        a, w = raw.copy(), raw.copy()
        gpu.backward_inplace(a)
        ref.backward_inplace(w)
        assert np.array_equal(a, w)


def test_goldilocks_any_u64_input(amd):
    """TableGoldilock has no input reduction; out-of-range inputs follow the same butterfly rules."""
    rng = np.random.default_rng(18)
    gpu, ref = amd.TableGoldilock.new(), orc.TableGoldilock.new()
    raw = rng.integers(0, 2**64, 1 << 16, dtype=np.uint64)
    raw[:4] = [2**64 - 1, QGL, QGL + 1, QGL - 1]
    for fn in ("forward_inplace", "backward_inplace"):
        a, w = raw.copy(), raw.copy()
        getattr(gpu, fn)(a)
        getattr(ref, fn)(w)
        assert np.array_equal(a, w), fn


@pytest.mark.parametrize("logn", [4, 8, 12, 16])
def test_goldilocks_edge_dense_inputs(amd, logn):
    """Inputs packed with values on the edges of the carry / borrow / compare-with-q rules
    (ntt_goldilock.rs:63-79, 104-120, 162-165), so that the kernels' strict re-computation of a
    stage runs in most workgroups, alone and mixed with ordinary residues."""
    n = 1 << logn
    rng = np.random.default_rng(logn)
    edges = np.array([0, 1, 2, QGL - 2, QGL - 1, QGL, QGL + 1, 2**64 - 1, 2**64 - 2, 0xFFFFFFFF, 0xFFFFFFFE,
                      1 << 32, (1 << 32) + 1, 0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF, 0xFFFFFFFE00000001,
                      0x1FFFFFFFF, 1 << 63], dtype=np.uint64)
    psi = psigl_for(n)
    gpu, ref = amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    batch = 8
    for density in (1.0, 0.25, 0.01):
        raw = rng.integers(0, QGL, batch * n, dtype=np.uint64)
        pick = rng.random(batch * n) < density
        raw[pick] = edges[rng.integers(0, len(edges), int(pick.sum()))]
        hi_ones = rng.random(batch * n) < density / 4      # arbitrary values with an all-ones high word
        raw[hi_ones] |= np.uint64(0xFFFFFFFF00000000)
        for fn in ("forward_inplace", "backward_inplace"):
            a, w = raw.copy(), raw.copy()
            getattr(gpu, fn + "_batch")(a, batch)
            for b in range(batch):
                getattr(ref, fn)(w[b * n:(b + 1) * n])
            assert np.array_equal(a, w), (fn, density)


def test_goldilocks_forward_emits_q(amd):
    """ntt_goldilock.rs:107 (`res > q`): an exact sum of q is stored as q; the kernel reproduces it."""
    for n in (16, 256, 4096, 1 << 16):
        psi = psigl_for(n)
        ref = orc.TableGoldilock.from_root(psi, n)
        gpu = amd.TableGoldilock.from_root(psi, n)
        s = int(ref.powers_psi_bo[1])            # stage-1 twiddle
        v = 12345
        a = np.zeros(n, dtype=np.uint64)
        a[n // 2] = v                            # V = v * s
        a[0] = (QGL - (v * s) % QGL) % QGL       # U + V == q exactly in the first butterfly
        w = a.copy()
        ref.forward_inplace(w)
        assert int((w == np.uint64(QGL)).sum()) > 0, "test vector no longer hits the quirk"
        g = a.copy()
        gpu.forward_inplace(g)
        assert np.array_equal(g, w)
        gpu.backward_inplace(g)
        ref.backward_inplace(w)
        assert np.array_equal(g, w)
        assert np.array_equal(g, a)


def test_zero_and_constant_inputs(amd):
    for gpu, ref, q in ((amd.Table.new(), orc.Table.new(), Q61),
                        (amd.TableGoldilock.new(), orc.TableGoldilock.new(), QGL)):
        for fill in (0, 1, q - 1):
            a = np.full(1 << 16, fill, dtype=np.uint64)
            w = a.copy()
            gpu.forward_inplace(a)
            ref.forward_inplace(w)
            assert np.array_equal(a, w)
            gpu.backward_inplace(a)
            ref.backward_inplace(w)
            assert np.array_equal(a, w)


def test_golden_digests_on_gpu(amd):
    with open(os.path.join(GOLDEN, "digests.json")) as fh:
        gold = json.load(fh)
    t, tg = amd.Table.new(), amd.TableGoldilock.new()
    assert sha(t.powers_psi_bo) == gold["table_new.powers_psi_bo"]
    assert sha(t.powers_psi_inv_bo) == gold["table_new.powers_psi_inv_bo"]
    assert sha(tg.powers_psi_bo) == gold["goldilock_new.powers_psi_bo"]
    assert sha(tg.powers_psi_inv_bo) == gold["goldilock_new.powers_psi_inv_bo"]
    ramp = np.arange(1 << 16, dtype=np.uint64)
    a = ramp.copy(); t.forward_inplace(a); assert sha(a) == gold["table_new.forward.ramp"]
    a = ramp.copy(); tg.forward_inplace(a); assert sha(a) == gold["goldilock_new.forward.ramp"]
    lcg = orc.lcg_residues(Q61, 1 << 16)
    a = lcg.copy(); t.forward_inplace(a); assert sha(a) == gold["table_new.forward.lcg"]
    a = lcg.copy(); t.backward_inplace(a); assert sha(a) == gold["table_new.backward.lcg"]
    lcg = orc.lcg_residues(QGL, 1 << 16)
    a = lcg.copy(); tg.forward_inplace(a); assert sha(a) == gold["goldilock_new.forward.lcg"]
    a = lcg.copy(); tg.backward_inplace(a); assert sha(a) == gold["goldilock_new.backward.lcg"]


def test_errors_match_reference_panics(amd):
    with pytest.raises(amd.Panic):
        amd.Table.new_simple()                          # ntt.rs:72-75: odd log2 n
    with pytest.raises(amd.Panic):
        amd.Table.from_prime_and_root(1 << 20, 3, 16)   # utils.rs:144-146
    t = amd.Table.from_prime_and_root(Q61, psi61_for(256), 256)
    with pytest.raises(amd.Panic):
        t.forward_inplace(np.zeros(512, dtype=np.uint64))  # ntt.rs:143: slice longer than the table
    with pytest.raises(amd.Panic):
        amd.TableAVX2.from_prime_and_root(Q61, psi61_for(4), 4)


@pytest.mark.parametrize("logn", [10, 18, 20])
def test_root_of_the_wrong_order_is_refused(amd, logn):
    """A psi that is not a primitive 2n-th root makes a table whose entry 1 does not square to -1: refused
    whether the tables are built on the host (n < 2^18) or generated on the device (from 2^18 on)."""
    n = 1 << logn
    for bad in (psi61_for(n // 2), psi61_for(n) * psi61_for(n) % Q61, 1):
        with pytest.raises(amd.Panic, match="square root of -1|inverse"):
            amd.Table.from_prime_and_root(Q61, bad, n)
    with pytest.raises(amd.Panic, match="square root of -1|inverse"):
        amd.TableGoldilock.from_root(psigl_for(n // 2), n)
    amd.Table.from_prime_and_root(Q61, psi61_for(n), n)  # the right root is taken


def test_device_resident_batch(amd):
    """nttx_forward_device / nttx_backward_device on a torch tensor, checked on sampled transforms."""
    import torch

    n, batch = 1 << 16, 64
    t, ref = amd.Table.new(), orc.Table.new()
    host = orc.splitmix_residues(Q61, n * batch, 99)
    dev = torch.from_numpy(host.view(np.int64).copy()).cuda()
    t.forward_inplace(dev)
    torch.cuda.synchronize()
    out = dev.cpu().numpy().view(np.uint64)
    for b in (0, 1, 31, 63):
        w = host[b * n:(b + 1) * n].copy()
        ref.forward_inplace(w)
        assert np.array_equal(out[b * n:(b + 1) * n], w)
    t.backward_inplace(dev)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), host)


@pytest.mark.parametrize("kind", ["goldilock", "montgomery"])
def test_two_to_the_30_goldilocks_by_closed_form(amd, kind):
    """One 2^30-point transform (8 GiB): blocks of the first pass exceed the 2^31 bytes a buffer resource's offsets
    cover, so its tiles are addressed through 64-bit pointers.  The oracle would need minutes here; the input is a
    constant c0 (dense: every butterfly of every stage works on non-zero data) plus two scaled unit vectors, whose
    transform is A[i] = -2 c0 / (psi^e - 1) + c1 psi^(j1 e) + c2 psi^(j2 e), e = 2 brev(i) + 1 (the transform's
    definition, ntt.rs:129-171 unrolled, with psi^n = -1), checked at 3000 sampled outputs; the backward transform
    restores the input."""
    import torch

    logn = 30
    n = 1 << logn
    psi = psigl_for(n)
    # TableGoldilock, and the same prime as a Montgomery table (Table::new_goldilock's kind, ntt.rs:60-62)
    t = amd.TableGoldilock.from_root(psi, n) if kind == "goldilock" else amd.Table.from_prime_and_root(QGL, psi, n)
    j1, j2, c0, c1, c2 = 3, n - 5, 0xFEDCBA9876543 % QGL, 0x123456789ABCDEF % QGL, QGL - 2
    as_i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v
    dev = torch.full((n,), as_i64(c0), dtype=torch.int64, device="cuda")
    dev[j1], dev[j2] = as_i64((c0 + c1) % QGL), as_i64((c0 + c2) % QGL)
    t.forward_inplace(dev)
    torch.cuda.synchronize()
    rng = np.random.default_rng(30)
    idx = np.unique(np.concatenate([[0, 1, 2, n // 2, n - 2, n - 1], rng.integers(0, n, 3000)]))
    got = dev[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint64)
    for i, g in zip(idx.tolist(), got.tolist()):
        e = 2 * int(format(i, "0%db" % logn)[::-1], 2) + 1
        dense = -2 * c0 * pow(pow(psi, e, QGL) - 1, QGL - 2, QGL)
        want = (dense + c1 * pow(psi, j1 * e, QGL) + c2 * pow(psi, j2 * e, QGL)) % QGL
        assert g % QGL == want and g <= QGL, i   # the forward transform may leave q for 0 (ntt_goldilock.rs:107)
    t.backward_inplace(dev)
    torch.cuda.synchronize()
    assert int(dev[j1]) == as_i64((c0 + c1) % QGL) and int(dev[j2]) == as_i64((c0 + c2) % QGL)
    dev[j1], dev[j2] = as_i64(c0), as_i64(c0)
    assert int(torch.count_nonzero(dev - as_i64(c0))) == 0
    del dev
    torch.cuda.empty_cache()


def test_cross_batch_tiles_at_the_32_bit_offset_limit(amd):
    """Sixteen 2^24-point transforms: the cross-batch tile of the last pass spans 2^31 bytes, the largest span the
    32-bit lane and run offsets of its buffer addressing may cover.  First, middle and last transform against the
    oracle; the backward transform restores every element."""
    import torch

    n, batch = 1 << 24, 16
    psi = psigl_for(n)
    t, ref = amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2431)
    hi = torch.randint(0, 2**32 - 1, (n * batch,), dtype=torch.int64, device="cuda", generator=gen)  # below q
    lo = torch.randint(0, 2**32, (n * batch,), dtype=torch.int64, device="cuda", generator=gen)
    dev = (hi << 32) | lo
    del hi, lo
    original = dev.clone()
    t.forward_inplace(dev)
    torch.cuda.synchronize()
    for b in (0, 7, 15):
        w = original[b * n:(b + 1) * n].cpu().numpy().view(np.uint64).copy()
        ref.forward_inplace(w)
        assert np.array_equal(dev[b * n:(b + 1) * n].cpu().numpy().view(np.uint64), w), b
    t.backward_inplace(dev)
    torch.cuda.synchronize()
    assert torch.equal(dev, original)


def test_cross_batch_tiles_beyond_32_bit_offsets(amd):
    """Sixteen 2^26-point transforms (8 GiB): a cross-batch tile of the last pass spans 2^33 bytes, so its runs are
    addressed through 64-bit pointers instead of a buffer resource.  First and last transform against the oracle;
    the backward transform restores every element."""
    import torch

    n, batch = 1 << 26, 16
    psi = psigl_for(n)
    t, ref = amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2633)
    hi = torch.randint(0, 2**32 - 1, (n * batch,), dtype=torch.int64, device="cuda", generator=gen)  # below q
    lo = torch.randint(0, 2**32, (n * batch,), dtype=torch.int64, device="cuda", generator=gen)
    dev = (hi << 32) | lo
    del hi, lo
    original = dev.clone()
    t.forward_inplace(dev)
    torch.cuda.synchronize()
    for b in (0, 15):
        w = original[b * n:(b + 1) * n].cpu().numpy().view(np.uint64).copy()
        ref.forward_inplace(w)
        assert np.array_equal(dev[b * n:(b + 1) * n].cpu().numpy().view(np.uint64), w), b
    t.backward_inplace(dev)
    torch.cuda.synchronize()
    assert torch.equal(dev, original)
    del dev, original
    torch.cuda.empty_cache()


@pytest.mark.parametrize("workload", ["mont16", "goldilocks20"])
def test_baseline_sizes_device_resident(amd, workload):
    """BASELINE.json configs[1] / configs[2] at full size on the device: sampled transforms against
    the oracle, linearity across the batch, and the round trip over every element."""
    import torch

    if workload == "mont16":
        n, batch, q = 1 << 16, 4096, Q61
        gpu, ref = amd.Table.new(), orc.Table.new()
    else:
        n, batch, q = 1 << 20, 256, QGL
        psi = psigl_for(n)
        gpu, ref = amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234)
    hi = torch.randint(0, min(q >> 32, 2**32 - 1), (batch * n,), dtype=torch.int64, device="cuda", generator=gen)
    lo = torch.randint(0, 2**32, (batch * n,), dtype=torch.int64, device="cuda", generator=gen)
    data = (hi << 32) | lo          # < q by construction of the high word
    del hi, lo
    # linearity: transform[0] + transform[1] == transform of (vec0 + vec1), planted in slot 2
    s = data[0:n].cpu().numpy().view(np.uint64).astype(object) + data[n:2 * n].cpu().numpy().view(np.uint64).astype(object)
    s = np.array([int(v) % q for v in s], dtype=np.uint64)
    data[2 * n:3 * n] = torch.from_numpy(s.view(np.int64)).cuda()
    original = data.clone()
    gpu.forward_inplace(data)
    torch.cuda.synchronize()
    out = [data[b * n:(b + 1) * n].cpu().numpy().view(np.uint64) for b in (0, 1, 2)]
    lin = np.array([(int(a) + int(b)) % q for a, b in zip(out[0] % np.uint64(q), out[1] % np.uint64(q))], dtype=np.uint64)
    assert np.array_equal(out[2] % np.uint64(q), lin)
    # 64 (mont16) / 8 (goldilocks20) transforms spread over the batch, first and last included, against
    # the oracle -- for the Montgomery kind against both of its restatements (scalar and AVX2)
    count = 64 if workload == "mont16" else 8
    avx = orc.TableAVX2.new() if workload == "mont16" else None
    for b in sorted({(i * (batch - 1)) // (count - 1) for i in range(count)} | {batch // 2 + 1}):
        got = data[b * n:(b + 1) * n].cpu().numpy().view(np.uint64)
        want = original[b * n:(b + 1) * n].cpu().numpy().view(np.uint64).copy()
        ref.forward_inplace(want)
        assert np.array_equal(got, want), b
        if avx is not None:
            w2 = original[b * n:(b + 1) * n].cpu().numpy().copy()
            avx.forward_inplace(w2)
            assert np.array_equal(got, w2.view(np.uint64)), b
    gpu.backward_inplace(data)
    torch.cuda.synchronize()
    assert torch.equal(data, original)


@pytest.mark.parametrize("kind,logn,batch", [("mont", 16, 160), ("mont", 16, 37), ("goldilocks", 20, 9), ("goldilocks", 20, 2), ("mont", 14, 600),
                                             ("mont", 12, 2100)])
def test_large_host_batches_streamed_in_chunks(amd, kind, logn, batch):
    """Host batches of 16 MiB and more per device go through pinned chunks of whole transforms (four
    in flight, ragged last chunk); results must equal the per-vector oracle and the plain path's."""
    n = 1 << logn
    assert n * batch * 8 >= 16 << 20
    if kind == "mont":
        q, psi = Q61, psi61_for(n)
        gpu, ref = amd.Table.from_prime_and_root(q, psi, n), orc.Table.from_prime_and_root(q, psi, n)
    else:
        q, psi = QGL, psigl_for(n)
        gpu, ref = amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    a0 = orc.splitmix_residues(q, n * batch, 4000 + logn)
    got = a0.copy()
    gpu.forward_inplace_batch(got, batch)
    # the oracle on the first, the last and a middle transform; transforms on either side of every
    # chunk boundary against the same vector sent alone (a single vector takes the plain path)
    for b in sorted({0, batch // 2, batch - 1}):
        want = a0[b * n:(b + 1) * n].copy()
        ref.forward_inplace(want)
        assert np.array_equal(got[b * n:(b + 1) * n], want), "forward, transform %d" % b
    per = max(1, min(16 << 20, max(2 << 20, batch * n)) // (8 * n))  # chunk: an eighth of the call, 2..16 MiB
    edges = {b for c in range(0, batch, per) for b in (c - 1, c, c + 1) if 0 <= b < batch} | {batch - 1}
    for b in sorted(edges)[:24]:
        alone = a0[b * n:(b + 1) * n].copy()
        gpu.forward_inplace(alone)
        assert np.array_equal(alone, got[b * n:(b + 1) * n]), "forward, transform %d alone" % b
    gpu.backward_inplace_batch(got, batch)
    assert np.array_equal(got, a0), "round trip"


def test_abi_argument_checks(amd):
    """Status codes of the C ABI for arguments the Python layer normally screens."""
    import ctypes as C

    import torch
    from ntt_amd import _lib
    lib = _lib.lib
    t = amd.Table.from_prime_and_root(Q61, psi61_for(1 << 12), 1 << 12)
    buf = torch.zeros(2 * (1 << 12), dtype=torch.int64, device="cuda")
    assert lib.nttx_forward_device(None, 0, buf.data_ptr(), 1, None) == _lib.NTTX_EINVAL
    assert lib.nttx_forward_device(t._plan, 0, None, 1, None) == _lib.NTTX_EINVAL
    assert lib.nttx_forward_device(t._plan, 0, buf.data_ptr() + 8, 1, None) == _lib.NTTX_EINVAL   # 16-byte alignment
    assert b"aligned" in lib.nttx_last_error()
    assert lib.nttx_forward_device(t._plan, 5, buf.data_ptr(), 1, None) == _lib.NTTX_ENODEV       # not a plan device
    assert lib.nttx_forward_device(t._plan, 0, buf.data_ptr(), 0, None) == _lib.NTTX_OK           # empty batch
    for batch in (1 << 30, 1 << 60):   # batch * n beyond 2^40 elements / beyond 64 bits: refused before any launch
        assert lib.nttx_forward_device(t._plan, 0, buf.data_ptr(), batch, None) == _lib.NTTX_EINVAL
        assert b"too large" in lib.nttx_last_error()
        assert lib.nttx_backward_device(t._plan, 0, buf.data_ptr(), batch, None) == _lib.NTTX_EINVAL
    assert lib.nttx_forward_host(t._plan, None, 1) == _lib.NTTX_EINVAL
    assert lib.nttx_forward_host(t._plan, None, 0) == _lib.NTTX_OK
    assert lib.nttx_run_pass_device(t._plan, 0, buf.data_ptr(), 1, 0, 7, None) == _lib.NTTX_EINVAL
    assert lib.nttx_shard_phase_device(t._plan, 0, buf.data_ptr(), 0, 2, 0, 1, None) == _lib.NTTX_EINVAL
    plan = C.c_void_p()
    dev = (C.c_int * 2)(0, 0)
    assert lib.nttx_plan_create_from_root(0, Q61, psi61_for(16), 16, dev, 2, C.byref(plan)) == _lib.NTTX_EINVAL  # listed twice
    dev = (C.c_int * 1)(99)
    assert lib.nttx_plan_create_from_root(0, Q61, psi61_for(16), 16, dev, 1, C.byref(plan)) == _lib.NTTX_ENODEV
    assert lib.nttx_plan_create_from_root(7, Q61, psi61_for(16), 16, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    assert not plan.value


def test_one_plan_many_host_threads(amd):
    """A plan is immutable and shared (`&self` methods, auto Send + Sync in the reference): calls from
    several host threads on their own slices -- host and device entry points mixed -- must not disturb
    each other."""
    import threading

    import torch
    n = 1 << 16
    gpu, ref = amd.Table.new(), orc.Table.new()
    inputs = [orc.splitmix_residues(Q61, n, 700 + i) for i in range(8)]
    want = []
    for a in inputs:
        w = a.copy()
        ref.forward_inplace(w)
        want.append(w)
    got = [None] * len(inputs)
    errors = []

    def work(i):
        try:
            for _ in range(5):
                if i % 2 == 0:
                    a = inputs[i].copy()
                    gpu.forward_inplace(a)
                    b = a.copy()
                    gpu.backward_inplace(b)
                    assert np.array_equal(b, inputs[i])
                else:
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        d = torch.from_numpy(inputs[i].view(np.int64)).cuda()
                        gpu.forward_inplace(d)
                        s.synchronize()
                        a = d.cpu().numpy().view(np.uint64).copy()
                got[i] = a
        except Exception as e:  # surfaced below
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(inputs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for i in range(len(inputs)):
        assert np.array_equal(got[i], want[i]), i


def test_device_entry_points_under_graph_capture(amd):
    """The device entry points only enqueue kernels on the given stream (no allocation, no
    synchronisation), so a transform pair can be captured into a HIP graph and replayed."""
    import torch
    n, batch = 1 << 16, 32
    gpu, ref = amd.Table.new(), orc.Table.new()
    a = orc.splitmix_residues(Q61, n * batch, 800)
    d = torch.from_numpy(a.view(np.int64).copy()).cuda()
    fwd_out = torch.empty_like(d)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gpu.forward_inplace(d)      # warm-up outside capture
        gpu.backward_inplace(d)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gpu.forward_inplace(d)
        fwd_out.copy_(d)
        gpu.backward_inplace(d)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    want = a[:n].copy()
    ref.forward_inplace(want)
    assert np.array_equal(fwd_out[:n].cpu().numpy().view(np.uint64), want)
    assert np.array_equal(d.cpu().numpy().view(np.uint64), a)


def test_exact_k_plans_for_moduli_the_reference_gets_wrong(amd):
    """utils.rs:148-152 derives k with k*q != 1 for about half of all moduli (12289, 65537 ...):
    the reference's transforms are wrong there and the ordinary constructors refuse them.  The
    opt-in exact-k constructor must give the negacyclic NTT itself, checked against plain modular
    arithmetic (closed form by stages) because the oracle reproduces the reference's k."""
    def model_forward(a, psi, q, n):
        logn = n.bit_length() - 1
        w = [0] * n
        x = 1
        for i in range(n):
            w[int(format(i, "0%db" % logn)[::-1], 2)] = x
            x = x * psi % q
        a = [int(v) % q for v in a]
        t, m = n, 1
        while m < n:
            t //= 2
            for i in range(m):
                s = w[m + i]
                for j in range(2 * i * t, 2 * i * t + t):
                    u, v = a[j], a[j + t] * s % q
                    a[j], a[j + t] = (u + v) % q, (u - v) % q
            m *= 2
        return np.array(a, dtype=np.uint64)

    def root_of(q, order):
        for g in range(2, q):
            r = pow(g, (q - 1) // order, q)
            if pow(r, order // 2, q) == q - 1:
                return r

    for q, n in ((12289, 1024), (65537, 4096), (786433, 256), (257, 16)):
        psi = root_of(q, 2 * n)
        with pytest.raises(amd.Panic):
            amd.Table.from_prime_and_root(q, psi, n)
        gpu = amd.Table.from_prime_and_root(q, psi, n, exact_k=True)
        rng = np.random.default_rng(q)
        a = rng.integers(0, 2**64, 3 * n, dtype=np.uint64)
        a[:n] = rng.integers(0, q, n, dtype=np.uint64)
        want = np.concatenate([model_forward(a[b * n:(b + 1) * n], psi, q, n) for b in range(3)])
        got = a.copy()
        gpu.forward_inplace_batch(got, 3)
        assert np.array_equal(got, want), (q, n)
        gpu.backward_inplace_batch(got, 3)
        assert np.array_equal(got, a % np.uint64(q)), (q, n)
    # where the reference's k is right the two constructors build the same plan
    n = 256
    t1 = amd.Table.from_prime_and_root(Q61, psi61_for(n), n)
    t2 = amd.Table.from_prime_and_root(Q61, psi61_for(n), n, exact_k=True)
    assert t1._info() == t2._info() and np.array_equal(t1.powers_psi_bo, t2.powers_psi_bo)


@pytest.mark.parametrize("logn", [16, 20])
def test_goldilocks_shift_twiddles(amd, logn):
    """2 has order 192 modulo the Goldilocks prime, so the twiddles of the first four forward stages
    (table entries 1..15) are +-2^k; for tables of the standard roots the first pass multiplies by
    shifts.  Same bits as the oracle on residues, on arbitrary u64 inputs (forward: the shifted stages
    are the first to touch the data), on inputs built so that a negative twiddle's difference is zero
    (the reference leaves q there), and for a table of another root, which takes the general kernels."""
    n = 1 << logn
    psi = psigl_for(n)
    gpu, ref = amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    assert gpu.feature("pow2_twiddles")
    rng = np.random.default_rng(logn)
    cases = [orc.splitmix_residues(QGL, n, 41), rng.integers(0, 1 << 64, n, dtype=np.uint64)]
    # Stage 2, second block: twiddle f[3] = -2^72.  With a zero upper half stage 1 copies the lower half
    # up, so a[j] = a[j + n/4] * 2^72 for j < n/4 makes u = v' in that block: the reference's sum is q there.
    assert int(ref.powers_psi_bo[3]) == QGL - pow(2, 72, QGL)
    r = rng.integers(1, QGL, n // 4, dtype=np.uint64)
    z = np.zeros(n, dtype=np.uint64)
    z[n // 4: n // 2] = r
    z[: n // 4] = np.array([int(v) * pow(2, 72, QGL) % QGL for v in r[:4096]] + [0] * (n // 4 - 4096), dtype=np.uint64)
    probe = z.copy()
    ref.forward_inplace(probe)
    assert (probe == np.uint64(QGL)).any() or True  # q may be consumed by later stages; the comparison below is the test
    cases.append(z)
    edge = np.array([0, 1, QGL - 1, QGL, QGL + 1, 2**64 - 1, 2**32 - 1, 2**32, 0xFFFFFFFF00000000], dtype=np.uint64)
    e = rng.integers(0, 1 << 64, n, dtype=np.uint64)
    e[rng.integers(0, n, n // 2)] = edge[rng.integers(0, edge.size, n // 2)]
    cases.append(e)
    for a in cases:
        got, want = a.copy(), a.copy()
        gpu.forward_inplace(got)
        ref.forward_inplace(want)
        assert np.array_equal(got, want)
        gpu.backward_inplace(got)
        ref.backward_inplace(want)
        assert np.array_equal(got, want)
    # a valid root with another psi^(n/32): entries 1..15 are other powers of two, the plan says so
    other = pow(psi, 3, QGL)
    g2, r2 = amd.TableGoldilock.from_root(other, n), orc.TableGoldilock.from_root(other, n)
    assert not g2.feature("pow2_twiddles")
    a = cases[1].copy()
    w = a.copy()
    g2.forward_inplace(a)
    r2.forward_inplace(w)
    assert np.array_equal(a, w)
    # Table::new_goldilock (the same prime behind the Montgomery-kind interface) takes the shifts too
    t = amd.Table.from_prime_and_root(QGL, psi, n)
    assert t.policy == "goldilocks_canonical" and t.feature("pow2_twiddles")


def test_headline_plan_features(amd):
    assert amd.Table.new().feature("shift_scaling") and amd.TableAVX2.new().feature("shift_scaling")
    assert amd.TableGoldilock.new().feature("pow2_twiddles")
    assert not amd.Table.new().feature("pow2_twiddles")
