"""Pins the CPU oracle against every known answer the reference's own tests carry.

Each test names the reference test it restates.  The reference draws its
full-transform inputs from an unseeded RNG and keeps no golden vectors, so the
full transforms are pinned by its cross-implementation / round-trip properties,
by an independent O(N^2) closed form, and by digests recorded in tests/golden.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as orc

Q61 = 0x1FFFFFFFFFE00001
PSI61 = 0x15EB043C7AA2B01F
QGL = 0xFFFFFFFF00000001
PSIGL = 0xABD0A6E8AA3D8A0E
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).astype("<u8").tobytes()).hexdigest()


def s64(x):
    """two's-complement view of a u64 as i64"""
    return x - (1 << 64) if x >= 1 << 63 else x


# ------------------------------------------------------------------ utils.rs
def test_bit_reverse_doctest():  # utils.rs:9-13
    assert orc.bit_reverse(2, 2) == 1


def test_bit_reverse_even_bits_is_true_reversal():
    for bits in (2, 4, 8, 16, 20):
        for n in (0, 1, 2, 3, (1 << bits) - 1, 0x5A5A5 & ((1 << bits) - 1)):
            want = int(format(n, "0%db" % bits)[::-1], 2)
            assert orc.bit_reverse(n, bits) == want


def test_bit_reverse_odd_bits_doubles():  # utils.rs:15: (64 - bits) / 2 truncates
    assert max(orc.bit_reverse(i, 3) for i in range(8)) == 14


def test_egcd_doctest():  # utils.rs:42-50
    a, b = Q61, 2**64
    g, x, y = orc.egcd_i128(a, b)

    def signed(v):
        return v - (1 << 128) if v >= 1 << 127 else v

    g, x, y = signed(g), signed(x), signed(y)
    assert g == a * x + b * y == 1


def test_montgomery_params_generation():  # utils.rs:200-211
    p = orc.MontgomeryTransformer.from_modulus(Q61)
    assert (p.k, p.r_square, p.modulus) == (6917533425689690113, 281474708275264, 2305843009211596801)


def test_montgomery_transformation():  # utils.rs:214-222
    p = orc.MontgomeryTransformer.from_modulus(Q61)
    for val in (1, 16777208):
        assert p.reduce(p.transform(val)) == val


def test_montgomery_even_modulus_panics():  # utils.rs:144-146
    with pytest.raises(orc.OraclePanic):
        orc.MontgomeryTransformer.from_modulus(1 << 20)


def test_montgomery_negative_x_quirk():  # utils.rs:148-152, prime 7681
    p = orc.MontgomeryTransformer.from_modulus(7681)
    assert (p.k * 7681) % 2**64 != 1  # wrong k: lifted by the modulus instead of 2^64
    for q in (Q61, QGL, 4293918721):
        p = orc.MontgomeryTransformer.from_modulus(q)
        assert (p.k * q) % 2**64 == 1


def test_mod_exp():
    assert orc.mod_exp_u64(3, 200, 1000003) == pow(3, 200, 1000003)
    assert orc.mod_exp_u64(PSIGL, 1 << 17, QGL) == 1
    assert orc.mod_exp_u64(PSIGL, 1 << 16, QGL) == QGL - 1
    assert orc.mod_exp_i64(PSI61, 1 << 16, Q61) == Q61 - 1


# -------------------------------------------------------- ntt_goldilock.rs
def test_u128_u64_conversion():  # ntt_goldilock.rs:247-254
    for v in (1, 2**64, 2**64 + 2):
        t = orc.u128_to_u64(v)
        assert orc.u64_to_u128(t[0], t[1]) == v
    assert orc.u128_to_u64(2**64 + 2) == (1, 2)  # (high, low)


def test_mul_reduce_159():  # ntt_goldilock.rs:257-274
    t = orc.TableGoldilock.new()
    data = [1, 0x57EF77BBD0411810]
    for a in data:
        for b in data:
            assert t.mul_reduce(a, b) == (a * b) % QGL


def test_mul_reduce_any_u64():
    t = orc.TableGoldilock.new()
    rng = np.random.default_rng(7)
    vals = [0, 1, QGL - 1, QGL, QGL + 1, 2**64 - 1, 2**32, 2**32 - 1] + [int(x) for x in rng.integers(0, 2**63, 64)]
    for a in vals:
        for b in vals[:12]:
            assert t.mul_reduce(a, b) == (a * b) % QGL


def test_goldilock_n_inv_constant():  # ntt_goldilock.rs:53
    t = orc.TableGoldilock.new()
    assert t.n_inv == 0xFFFEFFFF00010001
    assert (t.n_inv << 16) % QGL == 1


# ------------------------------------------------------------- ntt_avx2.rs
def test_a_sub_b_wide():  # ntt_avx2.rs:826-838
    hi, lo = orc.a_sub_b([0, 0, 2, 2], [0, 1, 2, 1], [0, 0, 2, 0], [0, 2, 1, 2])
    assert hi == [0, -1, 0, 1]
    assert lo == [0, -1, 1, -1]


def test_karatsuba():  # ntt_avx2.rs:841-854
    hi, lo = orc.a_mul_b_karatsuba([-4294967296, 4294967295, -1, -1], [-4294967296, 4294967295, 4294967295, -1])
    assert hi == [-8589934591, 0, 4294967294, -2]
    assert lo == [0, -8589934591, -4294967295, 1]


def test_avx2_montgomery_transformation():  # ntt_avx2.rs:875-888
    t = orc.TableAVX2.new()
    for val in (1, Q61 - 1048321):
        low = t.to_montgomery([val] * 4)
        assert t.montgomery_reduce_64([0] * 4, low) == [val] * 4


def test_reduce_if_negative():  # ntt_avx2.rs:891-905
    t = orc.TableAVX2.new()
    assert t.reduce_if_negative([-1, -2, 3, 0]) == [Q61 - 1, Q61 - 2, 3, 0]


def test_reduce_if_greater_equal_q():  # ntt_avx2.rs:908-923
    t = orc.TableAVX2.new()
    assert t.reduce_if_greater_equal_q([Q61 + 1, Q61, -1, 0]) == [1, 0, -1, 0]


def test_less_than():  # ntt_avx2.rs:926-939
    assert orc.a_lt_b([1, 1, -1, -2], [2, -1, 1, -1]) == [-1, -1, 0, -1]
    assert orc.a_lt_b([2, -1, 1, -1], [1, 1, -1, -2]) == [0, 0, -1, 0]


def _to_mont_r32(a):  # the closure at ntt_avx2.rs:945-946 (u64 wrapping multiply by 2^32)
    return s64(((a * 2**32) % 2**64) % Q61)


def test_mul_reduce_vec():  # ntt_avx2.rs:942-974
    t = orc.TableAVX2.new()
    a = [_to_mont_r32(v) for v in (4119828181, 1439360284, 1073741824, 1527549775)]
    b = [_to_mont_r32(v) for v in (3770581993, 2870057844, 1073741824, 2318024136)]
    got = t.montgomery_reduce_64([0] * 4, t.montgomery_mul_vec(a, b))
    assert got == [151105094258736720, 37308892573532894, 2161727821135872001, 1644454774947165347]


def test_ntt_kernel_4():  # ntt_avx2.rs:977-1030
    t = orc.TableAVX2.new()
    a_j = [_to_mont_r32(v) for v in (2825332433, 3362019074, 2509869327, 1527549775)]
    a_j_t = [_to_mont_r32(v) for v in (4119828181, 1439360284, 2754719283, 357587952)]
    s = [_to_mont_r32(v) for v in (3770581993, 2870057844, 3898594835, 2318024136)]
    x, y = t.cs_butterfly_simd_4(s, a_j, a_j_t)
    assert t.montgomery_reduce_64([0] * 4, x) == [
        1281256975236828982, 824160231704762218, 1601391116647552192, 1613286033865546758]
    assert t.montgomery_reduce_64([0] * 4, y) == [
        979046786719355542, 749542446557696430, 2010948492306909607, 1756774469921775392]


# ----------------------------------------------------- table construction
def test_table_constants():
    t = orc.Table.new()
    f, b = t.powers_psi_bo, t.powers_psi_inv_bo
    assert [int(x) for x in f[1:4]] == [0x0D947C30DCA49281, 0x1D21E810ED55F8C8, 0x02DE17EB128A4739]
    assert [int(x) for x in b[1:4]] == [0x126B83CF233B6D80, 0x1D21E814ED55B8C8, 0x02DE17EF128A0739]
    assert t.n_inv_montgomery == 0x1000000000000
    # f[brev(i)] = psi^i in Montgomery form
    m = orc.MontgomeryTransformer.from_modulus(Q61)
    for i in (0, 1, 2, 77, 65535):
        assert m.reduce(int(f[orc.bit_reverse(i, 16)])) == pow(PSI61, i, Q61)
        assert m.reduce(int(b[orc.bit_reverse(i, 16)])) == pow(PSI61, -i, Q61)


def test_table_goldilock_via_montgomery_constants():
    t = orc.Table.new_goldilock()
    assert (t.k, t.r_square) == (0x100000001, 18446744065119617025)


def test_goldilock_table_constants():
    t = orc.TableGoldilock.new()
    f = t.powers_psi_bo
    assert [int(x) for x in f[1:4]] == [0x1000000000000, 0xFFFFFFFEFF000001, 0xFFFFFEFF00000101]
    for i in (0, 1, 5, 65535):
        assert int(f[orc.bit_reverse(i, 16)]) == pow(PSIGL, i, QGL)
        assert int(t.powers_psi_inv_bo[orc.bit_reverse(i, 16)]) == pow(PSIGL, -i, QGL)


def test_avx2_tables_equal_base_tables():
    assert np.array_equal(orc.Table.new().powers_psi_bo, orc.TableAVX2.new().powers_psi_bo.view(np.uint64))
    assert np.array_equal(orc.Table.new().powers_psi_inv_bo, orc.TableAVX2.new().powers_psi_inv_bo.view(np.uint64))


def test_odd_log_n_panics():  # utils.rs:14-17 + ntt.rs:253
    with pytest.raises(orc.OraclePanic):
        orc.Table.from_prime_and_root(Q61, pow(PSI61, 1 << 13, Q61), 8)
    with pytest.raises(orc.OraclePanic):
        orc.Table.new_simple()  # ntt.rs:72-75


def test_u32_compatible_root():
    t = orc.Table.new_u32_compatible()
    assert t.k == 0x0FE000FF00100001
    a = orc.lcg_residues(4293918721, 1 << 16)
    b = a.copy()
    t.forward_inplace(b)
    t.backward_inplace(b)
    assert np.array_equal(a, b)


# ------------------------------------------- full transforms: properties
@pytest.mark.parametrize("logn", [2, 4, 6, 8])
def test_closed_form_base_and_goldilock(logn):
    n = 1 << logn
    for q, psi_big in ((Q61, PSI61), (QGL, PSIGL)):
        psi = pow(psi_big, (1 << 16) // n, q)
        a = orc.lcg_residues(q, n)
        want = orc.closed_form_forward(q, psi, a)
        t = orc.Table.from_prime_and_root(q, psi, n)
        got = a.copy()
        t.forward_inplace(got)
        assert np.array_equal(got, want)
        t.backward_inplace(got)
        assert np.array_equal(got, a)
    psi = pow(PSIGL, (1 << 16) // n, QGL)
    a = orc.lcg_residues(QGL, n)
    tg = orc.TableGoldilock.from_root(psi, n)
    got = a.copy()
    tg.forward_inplace(got)
    assert np.array_equal(got % np.uint64(QGL), orc.closed_form_forward(QGL, psi, a))
    tg.backward_inplace(got)
    assert np.array_equal(got, a)


@pytest.mark.parametrize("logn", [4, 6, 8])
def test_closed_form_avx2(logn):
    n = 1 << logn
    psi = pow(PSI61, (1 << 16) // n, Q61)
    a = orc.lcg_residues(Q61, n)
    t = orc.TableAVX2.from_prime_and_root(Q61, psi, n)
    got = a.astype(np.int64)
    t.forward_inplace(got)
    assert np.array_equal(got.view(np.uint64), orc.closed_form_forward(Q61, psi, a))
    t.backward_inplace(got)
    assert np.array_equal(got.view(np.uint64), a)


def test_base_version():  # dft.rs:23-42
    t = orc.Table.new_goldilock()
    a = orc.splitmix_residues(QGL, 1 << 16, 1)
    b = a.copy()
    t.forward_inplace(b)
    assert np.any(a != b)
    t.backward_inplace(b)
    assert np.array_equal(a, b)


def test_avx_correctness():  # dft.rs:45-78
    t, tv = orc.Table.new(), orc.TableAVX2.new()
    a0 = orc.splitmix_residues(Q61, 1 << 16, 2)
    a, av = a0.copy(), a0.astype(np.int64)
    t.forward_inplace(a)
    tv.forward_inplace(av)
    assert np.array_equal(a, av.view(np.uint64))
    t.backward_inplace(a)
    tv.backward_inplace(av)
    assert np.array_equal(a, av.view(np.uint64))
    assert np.array_equal(a, a0)


def test_goldilock_correctness():  # dft.rs:127-152
    t, tg = orc.Table.new_goldilock(), orc.TableGoldilock.new()
    a0 = orc.splitmix_residues(QGL, 1 << 16, 3)
    a, ag = a0.copy(), a0.copy()
    t.forward_inplace(a)
    tg.forward_inplace(ag)
    assert np.array_equal(a, ag)
    t.backward_inplace(a)
    tg.backward_inplace(ag)
    assert np.array_equal(a, ag)
    assert np.array_equal(a, a0)


def test_avx_against_base():  # dft.rs:81-124: ten forward + backward rounds on each table return the input
    t, tv = orc.Table.new(), orc.TableAVX2.new()
    a0 = orc.splitmix_residues(Q61, 1 << 16, 5)
    a, av = a0.copy(), a0.astype(np.int64)
    for _ in range(10):
        t.forward_inplace(a)
        t.backward_inplace(a)
    assert np.array_equal(a, a0)
    for _ in range(10):
        tv.forward_inplace(av)
        tv.backward_inplace(av)
    assert np.array_equal(av.view(np.uint64), a0)


def test_goldilock_against_base():  # dft.rs:155-189: the same for the Goldilocks prime on both tables
    t, tg = orc.Table.new_goldilock(), orc.TableGoldilock.new()
    a0 = orc.splitmix_residues(QGL, 1 << 16, 6)
    a, ag = a0.copy(), a0.copy()
    for _ in range(10):
        t.forward_inplace(a)
        t.backward_inplace(a)
    for _ in range(10):
        tg.forward_inplace(ag)
        tg.backward_inplace(ag)
    assert np.array_equal(a, a0)
    assert np.array_equal(ag, a0)  # the reference stops at the base table; the round trip holds here too


def test_normal_case():  # ntt.rs:279-295; `2 ^ 61` there is XOR, i.e. inputs in 0..63
    t = orc.Table.new()
    a0 = (orc.splitmix_residues(Q61, 1 << 16, 4) % np.uint64(63)).astype(np.uint64)
    a = a0.copy()
    t.forward_inplace(a)
    t.backward_inplace(a)
    assert np.array_equal(a, a0)


def test_base_accepts_any_u64():  # ntt.rs:125-127: transform() reduces the input mod q
    t = orc.Table.new()
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 2**64, 1 << 16, dtype=np.uint64)
    a, b = raw.copy(), (raw % np.uint64(Q61)).astype(np.uint64)
    t.forward_inplace(a)
    t.forward_inplace(b)
    assert np.array_equal(a, b)


def test_goldilock_forward_can_emit_q():
    """ntt_goldilock.rs:107: `res > q` leaves an exact sum of q in place, and the forward
    transform has no canonicalising pass; the Montgomery table gives 0 there (ntt.rs:173-175)."""
    n = 16
    psi = pow(PSIGL, 1 << 12, QGL)
    tg = orc.TableGoldilock.from_root(psi, n)
    assert int(tg.powers_psi_bo[1]) == 1 << 48
    a = np.zeros(n, dtype=np.uint64)
    a[0], a[8] = 1, 1 << 48
    g = a.copy()
    tg.forward_inplace(g)
    b = a.copy()
    orc.Table.from_prime_and_root(QGL, psi, n).forward_inplace(b)
    assert int(g[0]) == QGL and int(b[0]) == 0
    assert np.array_equal(g[1:], b[1:])
    tg.backward_inplace(g)
    assert np.array_equal(g, a)


# ----------------------------------------------- digests (tests/golden)
def test_golden_digests():
    with open(os.path.join(GOLDEN, "digests.json")) as fh:
        gold = json.load(fh)
    t, tg = orc.Table.new(), orc.TableGoldilock.new()
    got = {}
    got["table_new.powers_psi_bo"] = sha(t.powers_psi_bo)
    got["table_new.powers_psi_inv_bo"] = sha(t.powers_psi_inv_bo)
    got["goldilock_new.powers_psi_bo"] = sha(tg.powers_psi_bo)
    got["goldilock_new.powers_psi_inv_bo"] = sha(tg.powers_psi_inv_bo)
    ramp = np.arange(1 << 16, dtype=np.uint64)  # benches/ntt.rs:9-11
    a = ramp.copy(); t.forward_inplace(a); got["table_new.forward.ramp"] = sha(a)
    assert [int(x) for x in a[:4]] == [0x0E0BF34790418ECB, 0x10C622D2EC7053BD, 0x0B6246D9F8F7B24E, 0x07442D4D8CE37F93]
    a = ramp.copy(); tg.forward_inplace(a); got["goldilock_new.forward.ramp"] = sha(a)
    assert [int(x) for x in a[:4]] == [0xA4A00EDF5BBE3570, 0xF941837741D3FB1E, 0x3D2BC0B1F93817C7, 0xD7FD95C240B0E210]
    lcg = orc.lcg_residues(Q61, 1 << 16)
    assert [int(x) for x in lcg[:2]] == [0x0CEAEE21BF66BBFF, 0x0A80754D1ABA8D4A]
    a = lcg.copy(); t.forward_inplace(a); got["table_new.forward.lcg"] = sha(a)
    assert [int(x) for x in a[:2]] == [0x0FE64CF761810022, 0x00FF05C7EE93F09C]
    a = lcg.copy(); t.backward_inplace(a); got["table_new.backward.lcg"] = sha(a)
    lcg = orc.lcg_residues(QGL, 1 << 16)
    a = lcg.copy(); tg.forward_inplace(a); got["goldilock_new.forward.lcg"] = sha(a)
    a = lcg.copy(); tg.backward_inplace(a); got["goldilock_new.backward.lcg"] = sha(a)
    av = orc.lcg_residues(Q61, 1 << 16).astype(np.int64)
    orc.TableAVX2.new().forward_inplace(av)
    assert sha(av.view(np.uint64)) == got["table_new.forward.lcg"]
    for k, v in got.items():
        assert gold[k] == v, k


def test_n16_known_answer():
    psi = pow(PSI61, 1 << 12, Q61)
    assert psi == 0x055942A07CE15444
    t = orc.Table.from_prime_and_root(Q61, psi, 16)
    a = np.arange(16, dtype=np.uint64)
    t.forward_inplace(a)
    assert [int(x) for x in a[:4]] == [0x10F35C2B1D012D41, 0x048034698C2FD686, 0x01CA46DBB91693DB, 0x0A64A0AB03DADF01]


def test_cpu_baseline_timer_runs():
    t = orc.Table.new()
    assert orc.bench_pairs(0, t, 1 << 16, 1, 2) > 0
