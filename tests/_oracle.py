"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the ntt_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle.so")

u64 = C.c_uint64
i64 = C.c_int64
p_u64 = C.POINTER(C.c_uint64)
p_i64 = C.POINTER(C.c_int64)
vp = C.c_void_p


def _load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    lib = C.CDLL(_SO)
    sig = {
        "orc_bit_reverse": (u64, [u64, C.c_uint32]),
        "orc_egcd_i128_parts": (None, [i64, u64, i64, u64, p_i64, p_u64]),
        "orc_mod_exp_u64": (u64, [u64, u64, u64]),
        "orc_mod_exp_i64": (i64, [i64, i64, i64]),
        "orc_montgomery_from_modulus": (C.c_int, [u64, vp]),
        "orc_montgomery_reduce": (u64, [vp, u64, u64]),
        "orc_montgomery_transform": (u64, [vp, u64]),
        "orc_montgomery_mul": (u64, [vp, u64, u64]),
        "orc_table_from_prime_and_root": (C.c_int, [u64, u64, u64, C.POINTER(vp)]),
        "orc_table_new": (C.c_int, [C.POINTER(vp)]),
        "orc_table_new_goldilock": (C.c_int, [C.POINTER(vp)]),
        "orc_table_new_u32_compatible": (C.c_int, [C.POINTER(vp)]),
        "orc_table_new_simple": (C.c_int, [C.POINTER(vp)]),
        "orc_table_free": (None, [vp]),
        "orc_table_forward_inplace": (C.c_int, [vp, vp, C.c_size_t]),
        "orc_table_backward_inplace": (C.c_int, [vp, vp, C.c_size_t]),
        "orc_table_powers_psi_bo": (p_u64, [vp]),
        "orc_table_powers_psi_inv_bo": (p_u64, [vp]),
        "orc_table_n_inv_montgomery": (u64, [vp]),
        "orc_table_k": (u64, [vp]),
        "orc_table_r_square": (u64, [vp]),
        "orc_goldilock_new": (C.c_int, [C.POINTER(vp)]),
        "orc_goldilock_from_root": (C.c_int, [u64, u64, C.POINTER(vp)]),
        "orc_goldilock_free": (None, [vp]),
        "orc_goldilock_reduce_159": (u64, [vp, u64, u64, u64]),
        "orc_goldilock_mul_reduce": (u64, [vp, u64, u64]),
        "orc_goldilock_forward_inplace": (C.c_int, [vp, vp, C.c_size_t]),
        "orc_goldilock_backward_inplace": (C.c_int, [vp, vp, C.c_size_t]),
        "orc_goldilock_powers_psi_bo": (p_u64, [vp]),
        "orc_goldilock_powers_psi_inv_bo": (p_u64, [vp]),
        "orc_goldilock_n_inv": (u64, [vp]),
        "orc_u128_to_u64": (None, [u64, u64, p_u64]),
        "orc_u64_to_u128": (None, [u64, u64, p_u64]),
        "orc_avx2_from_prime_and_root": (C.c_int, [i64, i64, i64, C.POINTER(vp)]),
        "orc_avx2_new": (C.c_int, [C.POINTER(vp)]),
        "orc_avx2_free": (None, [vp]),
        "orc_avx2_forward_inplace": (C.c_int, [vp, vp, C.c_size_t]),
        "orc_avx2_backward_inplace": (C.c_int, [vp, vp, C.c_size_t]),
        "orc_avx2_powers_psi_bo": (p_i64, [vp]),
        "orc_avx2_powers_psi_inv_bo": (p_i64, [vp]),
        "orc_avx2_a_mul_b_karatsuba": (None, [p_i64, p_i64, p_i64, p_i64]),
        "orc_avx2_a_lt_b": (None, [p_i64, p_i64, p_i64]),
        "orc_avx2_a_sub_b": (None, [p_i64] * 6),
        "orc_avx2_a_lt_b_wide": (None, [p_i64] * 5),
        "orc_avx2_montgomery_reduce_64": (None, [vp, p_i64, p_i64, p_i64]),
        "orc_avx2_montgomery_mul_vec": (None, [vp, p_i64, p_i64, p_i64]),
        "orc_avx2_to_montgomery": (None, [vp, p_i64, p_i64]),
        "orc_avx2_reduce_if_greater_equal_q": (None, [vp, p_i64]),
        "orc_avx2_reduce_if_negative": (None, [vp, p_i64]),
        "orc_avx2_cs_butterfly_simd_4": (None, [vp, p_i64, p_i64, p_i64]),
        "orc_avx2_gs_butterfly_simd_4": (None, [vp, p_i64, p_i64, p_i64]),
        "orc_closed_form_forward": (None, [u64, u64, vp, vp, C.c_size_t]),
        "orc_bench_pairs": (C.c_double, [C.c_int, vp, C.c_size_t, C.c_int, C.c_int]),
        "orc_last_panic": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class OraclePanic(RuntimeError):
    """The reference would panic here."""


def _check(rc):
    if rc != 0:
        raise OraclePanic(lib.orc_last_panic().decode() or "oracle error")


class MontgomeryTransformer(C.Structure):
    """utils.rs:126-135"""
    _fields_ = [("k", u64), ("r_square", u64), ("modulus", u64)]

    @classmethod
    def from_modulus(cls, modulus):
        m = cls()
        _check(lib.orc_montgomery_from_modulus(modulus, C.byref(m)))
        return m

    def reduce(self, a):
        return lib.orc_montgomery_reduce(C.byref(self), (a >> 64) & (2**64 - 1), a & (2**64 - 1))

    def transform(self, a):
        return lib.orc_montgomery_transform(C.byref(self), a)

    def mul(self, a, b):
        return lib.orc_montgomery_mul(C.byref(self), a, b)


def _arr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class _TableBase:
    _free = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and self._free is not None:
            self._free(h)
            self._h = None


class Table(_TableBase):
    """Oracle for ntt.rs `Table<u64>`; transforms operate in place on a numpy uint64 array."""
    _free = staticmethod(lib.orc_table_free)

    def __init__(self, handle, n):
        self._h = handle
        self.n = n

    @classmethod
    def _make(cls, fn, *args):
        h = vp()
        _check(fn(*args, C.byref(h)))
        t = cls(h, 0)
        return t

    @classmethod
    def from_prime_and_root(cls, modulus, root, n):
        t = cls._make(lib.orc_table_from_prime_and_root, modulus, root, n)
        t.n, t.q, t.psi = n, modulus, root
        return t

    @classmethod
    def new(cls):
        return cls.from_prime_and_root(0x1FFFFFFFFFE00001, 0x15EB043C7AA2B01F, 1 << 16)

    @classmethod
    def new_goldilock(cls):
        return cls.from_prime_and_root(0xFFFFFFFF00000001, 0xABD0A6E8AA3D8A0E, 1 << 16)

    @classmethod
    def new_u32_compatible(cls):
        return cls.from_prime_and_root(4293918721, 2004365341, 1 << 16)

    @classmethod
    def new_simple(cls):
        return cls.from_prime_and_root(7681, 1925, 1 << 3)

    @property
    def powers_psi_bo(self):
        return np.ctypeslib.as_array(lib.orc_table_powers_psi_bo(self._h), (self.n,)).copy()

    @property
    def powers_psi_inv_bo(self):
        return np.ctypeslib.as_array(lib.orc_table_powers_psi_inv_bo(self._h), (self.n,)).copy()

    @property
    def n_inv_montgomery(self):
        return lib.orc_table_n_inv_montgomery(self._h)

    @property
    def k(self):
        return lib.orc_table_k(self._h)

    @property
    def r_square(self):
        return lib.orc_table_r_square(self._h)

    def forward_inplace(self, a):
        assert a.dtype == np.uint64 and a.flags.c_contiguous
        _check(lib.orc_table_forward_inplace(self._h, a.ctypes.data, a.size))

    def backward_inplace(self, a):
        assert a.dtype == np.uint64 and a.flags.c_contiguous
        _check(lib.orc_table_backward_inplace(self._h, a.ctypes.data, a.size))

    forward_inplace_lazy = forward_inplace    # ntt.rs:34-36: LAZY is never read
    backward_inplace_lazy = backward_inplace  # ntt.rs:48-50


class TableGoldilock(_TableBase):
    """Oracle for ntt_goldilock.rs `TableGoldilock<u64>`."""
    _free = staticmethod(lib.orc_goldilock_free)
    q = 0xFFFFFFFF00000001

    def __init__(self, handle, psi, n):
        self._h, self.psi, self.n = handle, psi, n

    @classmethod
    def new(cls):
        h = vp()
        _check(lib.orc_goldilock_new(C.byref(h)))
        return cls(h, 0xABD0A6E8AA3D8A0E, 1 << 16)

    @classmethod
    def from_root(cls, psi, n):
        h = vp()
        _check(lib.orc_goldilock_from_root(psi, n, C.byref(h)))
        return cls(h, psi, n)

    @property
    def n_inv(self):
        return lib.orc_goldilock_n_inv(self._h)

    @property
    def powers_psi_bo(self):
        return np.ctypeslib.as_array(lib.orc_goldilock_powers_psi_bo(self._h), (self.n,)).copy()

    @property
    def powers_psi_inv_bo(self):
        return np.ctypeslib.as_array(lib.orc_goldilock_powers_psi_inv_bo(self._h), (self.n,)).copy()

    def reduce_159(self, low, middle, high):
        return lib.orc_goldilock_reduce_159(self._h, low, middle, high)

    def mul_reduce(self, a, b):
        return lib.orc_goldilock_mul_reduce(self._h, a, b)

    def forward_inplace(self, a):
        assert a.dtype == np.uint64 and a.flags.c_contiguous
        _check(lib.orc_goldilock_forward_inplace(self._h, a.ctypes.data, a.size))

    def backward_inplace(self, a):
        assert a.dtype == np.uint64 and a.flags.c_contiguous
        _check(lib.orc_goldilock_backward_inplace(self._h, a.ctypes.data, a.size))

    forward_inplace_lazy = forward_inplace
    backward_inplace_lazy = backward_inplace


class TableAVX2(_TableBase):
    """Oracle for ntt_avx2.rs `TableAVX2`; arrays are numpy int64."""
    _free = staticmethod(lib.orc_avx2_free)

    def __init__(self, handle, q, psi, n):
        self._h, self.q, self.psi, self.n = handle, q, psi, n

    @classmethod
    def from_prime_and_root(cls, modulus, root, n):
        h = vp()
        if lib.orc_avx2_from_prime_and_root(modulus, root, n, C.byref(h)) != 0:
            raise OraclePanic("TableAVX2 construction panicked")
        return cls(h, modulus, root, n)

    @classmethod
    def new(cls):
        return cls.from_prime_and_root(0x1FFFFFFFFFE00001, 0x15EB043C7AA2B01F, 1 << 16)

    @property
    def powers_psi_bo(self):
        return np.ctypeslib.as_array(lib.orc_avx2_powers_psi_bo(self._h), (self.n,)).copy()

    @property
    def powers_psi_inv_bo(self):
        return np.ctypeslib.as_array(lib.orc_avx2_powers_psi_inv_bo(self._h), (self.n,)).copy()

    def forward_inplace(self, a):
        assert a.dtype == np.int64 and a.flags.c_contiguous
        if lib.orc_avx2_forward_inplace(self._h, a.ctypes.data, a.size) != 0:
            raise OraclePanic("TableAVX2 forward out of domain")

    def backward_inplace(self, a):
        assert a.dtype == np.int64 and a.flags.c_contiguous
        if lib.orc_avx2_backward_inplace(self._h, a.ctypes.data, a.size) != 0:
            raise OraclePanic("TableAVX2 backward out of domain")

    forward_inplace_lazy = forward_inplace
    backward_inplace_lazy = backward_inplace

    # 4-lane primitives for the KATs
    def _v(self, x):
        return (i64 * 4)(*[int(v) for v in x])

    def montgomery_reduce_64(self, th, tl):
        out = (i64 * 4)()
        lib.orc_avx2_montgomery_reduce_64(self._h, self._v(th), self._v(tl), out)
        return list(out)

    def montgomery_mul_vec(self, a, b):
        out = (i64 * 4)()
        lib.orc_avx2_montgomery_mul_vec(self._h, self._v(a), self._v(b), out)
        return list(out)

    def to_montgomery(self, a):
        out = (i64 * 4)()
        lib.orc_avx2_to_montgomery(self._h, self._v(a), out)
        return list(out)

    def reduce_if_greater_equal_q(self, a):
        v = self._v(a)
        lib.orc_avx2_reduce_if_greater_equal_q(self._h, v)
        return list(v)

    def reduce_if_negative(self, a):
        v = self._v(a)
        lib.orc_avx2_reduce_if_negative(self._h, v)
        return list(v)

    def cs_butterfly_simd_4(self, s, a_j, a_j_t):
        x, y = self._v(a_j), self._v(a_j_t)
        lib.orc_avx2_cs_butterfly_simd_4(self._h, self._v(s), x, y)
        return list(x), list(y)

    def gs_butterfly_simd_4(self, s, a_j, a_j_t):
        x, y = self._v(a_j), self._v(a_j_t)
        lib.orc_avx2_gs_butterfly_simd_4(self._h, self._v(s), x, y)
        return list(x), list(y)


def _v4(x):
    return (i64 * 4)(*[int(v) for v in x])


def a_mul_b_karatsuba(a, b):
    hi, lo = (i64 * 4)(), (i64 * 4)()
    lib.orc_avx2_a_mul_b_karatsuba(_v4(a), _v4(b), hi, lo)
    return list(hi), list(lo)


def a_lt_b(a, b):
    out = (i64 * 4)()
    lib.orc_avx2_a_lt_b(_v4(a), _v4(b), out)
    return list(out)


def a_sub_b(ah, al, bh, bl):
    hi, lo = (i64 * 4)(), (i64 * 4)()
    lib.orc_avx2_a_sub_b(_v4(ah), _v4(al), _v4(bh), _v4(bl), hi, lo)
    return list(hi), list(lo)


def a_lt_b_wide(ah, al, bh, bl):
    out = (i64 * 4)()
    lib.orc_avx2_a_lt_b_wide(_v4(ah), _v4(al), _v4(bh), _v4(bl), out)
    return list(out)


def bit_reverse(n, bits):
    return lib.orc_bit_reverse(n, bits)


def egcd_i128(a, m):
    def split(x):
        x &= (1 << 128) - 1
        hi = x >> 64
        if hi >= 1 << 63:
            hi -= 1 << 64
        return hi, x & (2**64 - 1)

    oh, ol = (i64 * 3)(), (u64 * 3)()
    lib.orc_egcd_i128_parts(*split(a), *split(m), oh, ol)
    return tuple((int(oh[i]) << 64) | int(ol[i]) for i in range(3))


def mod_exp_u64(b, e, m):
    return lib.orc_mod_exp_u64(b, e, m)


def mod_exp_i64(b, e, m):
    return lib.orc_mod_exp_i64(b, e, m)


def closed_form_forward(q, psi, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib.orc_closed_form_forward(q, psi, a.ctypes.data, out.ctypes.data, a.size)
    return out


def u128_to_u64(a):
    out = (u64 * 2)()
    lib.orc_u128_to_u64(a >> 64, a & (2**64 - 1), out)
    return int(out[0]), int(out[1])


def u64_to_u128(first, second):
    out = (u64 * 2)()
    lib.orc_u64_to_u128(first, second, out)
    return (int(out[0]) << 64) | int(out[1])


def bench_pairs(kind, table, n, pairs, threads):
    return lib.orc_bench_pairs(kind, table._h, n, pairs, threads)


# ---- deterministic inputs shared by the oracle tests and the GPU parity tests ----
MASK64 = (1 << 64) - 1


def lcg_residues(q, n, x0=0x9E3779B97F4A7C15):
    """x <- x*6364136223846793005 + 1442695040888963407 mod 2^64; a[j] = x mod q after each update."""
    out = np.empty(n, dtype=np.uint64)
    x = x0
    for j in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) & MASK64
        out[j] = x % q
    return out


def splitmix_residues(q, n, seed):
    """SplitMix64 stream reduced mod q (vectorised); seed is the generator's initial state."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(q)).astype(np.uint64)
