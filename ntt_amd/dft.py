"""Mirror of the reference's `ntt::dft` module: the `DFT` trait (src/dft.rs:6-11) and the three
table types, with every transform executed by libnttx.so on the GPU.

A slice argument is either a C-contiguous numpy array in host memory (length n, like the
reference's `&mut [O]`) or a contiguous torch tensor resident on one of the plan's GPUs holding
`batch * n` elements (transformed as `batch` independent vectors on the tensor's current stream).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib


class Panic(RuntimeError):
    """Raised where the reference panics (or where its result would be wrong)."""


def _check(rc):
    if rc != _lib.NTTX_OK:
        raise Panic(_lib.last_error() or "nttx error %d" % rc)


class DFT:
    """src/dft.rs:6-11"""

    def forward_inplace(self, x):
        raise NotImplementedError

    def forward_inplace_lazy(self, x):
        raise NotImplementedError

    def backward_inplace(self, x):
        raise NotImplementedError

    def backward_inplace_lazy(self, x):
        raise NotImplementedError


class _PlanTable(DFT):
    _kind = _lib.NTTX_MONT64
    _np_dtype = np.uint64

    def __init__(self, q, psi, n, devices=None, exact_k=False):
        self.q, self.psi, self._n = q, psi, n
        self._plan = C.c_void_p()
        # exact_k: the corrected setup, not the reference's behaviour -- k = q^-1 mod 2^64 for every odd
        # modulus (the reference derives a wrong k for about half of them, utils.rs:148-152, and those
        # are rejected otherwise), and odd log2(n) with the true bit reversal (the reference's
        # constructors panic there: utils.rs:14-17, ntt.rs:253)
        create = lib.nttx_plan_create_from_root_exact_k if exact_k else lib.nttx_plan_create_from_root
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            rc = create(self._kind, q, psi, n, arr, len(devices), C.byref(self._plan))
        else:
            rc = create(self._kind, q, psi, n, None, 0, C.byref(self._plan))
        _check(rc)
        self._devices = list(devices) if devices else [0]

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan:
            lib.nttx_plan_destroy(plan)
            self._plan = None

    # -- read-back of the tables (pub fields of Table, ntt.rs:17-20)
    def _tables(self):
        f = np.empty(self._n, dtype=np.uint64)
        b = np.empty(self._n, dtype=np.uint64)
        _check(lib.nttx_plan_tables(self._plan, f.ctypes.data, b.ctypes.data))
        return f, b

    @property
    def powers_psi_bo(self):
        return self._tables()[0].view(self._np_dtype)

    @property
    def powers_psi_inv_bo(self):
        return self._tables()[1].view(self._np_dtype)

    def _info(self):
        kind, nd = C.c_int(), C.c_int()
        q, k, n, n_inv = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib.nttx_plan_info(self._plan, C.byref(kind), C.byref(q), C.byref(k), C.byref(n), C.byref(n_inv),
                                  C.byref(nd)))
        return dict(kind=kind.value, q=q.value, k=k.value, n=n.value, n_inv=n_inv.value, n_devices=nd.value)

    @property
    def policy(self):
        """Name of the arithmetic the plan's kernels use (nttx_plan_policy)."""
        return lib.nttx_plan_policy(self._plan).decode()

    def feature(self, name):
        """Whether the plan's kernels take the named shortcut (nttx_plan_feature)."""
        return lib.nttx_plan_feature(self._plan, name.encode()) == 1

    def _run(self, x, forward, range_2q=False):
        if range_2q:
            host_fn = lib.nttx_forward_host_lazy if forward else lib.nttx_backward_host_lazy
            dev_fn = lib.nttx_forward_device_lazy if forward else lib.nttx_backward_device_lazy
        else:
            host_fn = lib.nttx_forward_host if forward else lib.nttx_backward_host
            dev_fn = lib.nttx_forward_device if forward else lib.nttx_backward_device
        if isinstance(x, np.ndarray):
            if x.dtype != self._np_dtype or not x.flags.c_contiguous or not x.flags.writeable:
                raise TypeError("expected a writable C-contiguous %s array" % np.dtype(self._np_dtype).name)
            if x.size != self._n:
                # ntt.rs:123: the reference sizes the transform by the slice; only len == n
                # round-trips (ntt.rs:231-235) and len > n indexes out of bounds (ntt.rs:143).
                raise Panic("slice length %d != table size %d" % (x.size, self._n))
            _check(host_fn(self._plan, x.ctypes.data, 1))
            return
        import torch  # device memory plumbing only

        if isinstance(x, torch.Tensor):
            if not x.is_cuda or not x.is_contiguous() or x.element_size() != 8 or x.dtype not in (
                    torch.int64, torch.uint64):
                raise TypeError("expected a contiguous 64-bit integer tensor on a GPU")
            if x.numel() % self._n != 0:
                raise Panic("tensor holds %d elements, not a multiple of n = %d" % (x.numel(), self._n))
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _check(dev_fn(self._plan, x.device.index, x.data_ptr(), x.numel() // self._n, stream))
            return
        raise TypeError("unsupported slice type %r" % type(x))

    def forward_inplace(self, x):
        self._run(x, True)

    def backward_inplace(self, x):
        self._run(x, False)

    # ntt.rs:31-35, 45-49 document "elements in [0, 2q)" for the lazy methods, but the reference never
    # reads its LAZY parameter (ntt.rs:122, 178; ntt_avx2.rs:108, 281; ntt_goldilock.rs:81, 128): they
    # return exactly what the strict methods return.  So do these, bit for bit -- a caller ported from
    # the reference may compare, serialise or REDC-multiply those outputs as canonical residues.
    def forward_inplace_lazy(self, x):
        self._run(x, True)

    def backward_inplace_lazy(self, x):
        self._run(x, False)

    def forward_inplace_core(self, x, lazy=False):  # `lazy` is accepted and ignored, as in the reference
        self._run(x, True)

    def backward_inplace_core(self, x, lazy=False):
        self._run(x, False)

    # Explicit opt-in, no counterpart in the reference: outputs in [0, 2q), congruent to the canonical
    # ones (nttx_*_lazy).  For q < 2^61 the last conditional subtraction is skipped; other plans
    # return the canonical residues.  Such outputs are valid inputs of every transform here.
    def forward_inplace_range_2q(self, x):
        self._run(x, True, range_2q=True)

    def backward_inplace_range_2q(self, x):
        self._run(x, False, range_2q=True)

    # batched host path (no counterpart in the reference, which loops over slices)
    def forward_inplace_batch(self, x, batch):
        self._run_batch(x, batch, True)

    def backward_inplace_batch(self, x, batch):
        self._run_batch(x, batch, False)

    # a <- a * b in Z_q[X]/(X^n + 1) for every pair of polynomials in the two batches; the composition
    # forward(a), forward(b), coefficient-wise product, backward(a), fused (include/nttx.h)
    def negacyclic_mul_inplace(self, a, b):
        if isinstance(a, np.ndarray):
            for x in (a, b):
                if not isinstance(x, np.ndarray) or x.dtype != self._np_dtype or not x.flags.c_contiguous:
                    raise TypeError("expected C-contiguous %s arrays" % np.dtype(self._np_dtype).name)
            if not a.flags.writeable or a.size != b.size or a.size % self._n != 0:
                raise Panic("operands must hold the same whole number of polynomials of n = %d" % self._n)
            _check(lib.nttx_negacyclic_mul_host(self._plan, a.ctypes.data, b.ctypes.data, a.size // self._n))
            return
        import torch

        for x in (a, b):
            if not isinstance(x, torch.Tensor) or not x.is_cuda or not x.is_contiguous() or x.element_size() != 8:
                raise TypeError("expected contiguous 64-bit integer tensors on a GPU")
        if a.device != b.device or a.numel() != b.numel() or a.numel() % self._n != 0:
            raise Panic("operands must hold the same whole number of polynomials of n = %d" % self._n)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _check(lib.nttx_negacyclic_mul_device(self._plan, a.device.index, a.data_ptr(), b.data_ptr(),
                                              a.numel() // self._n, stream))

    # ONE transform split over several shards (a vector too large for one GPU); same result as
    # forward_inplace / backward_inplace on the whole slice.  n_shards = 0: one per plan device.
    def forward_inplace_distributed(self, x, n_shards=0):
        self._run_distributed(x, n_shards, True)

    def backward_inplace_distributed(self, x, n_shards=0):
        self._run_distributed(x, n_shards, False)

    def _run_distributed(self, x, n_shards, forward):
        if not isinstance(x, np.ndarray) or x.dtype != self._np_dtype or not x.flags.c_contiguous \
                or not x.flags.writeable:
            raise TypeError("expected a writable C-contiguous %s array" % np.dtype(self._np_dtype).name)
        if x.size != self._n:
            raise Panic("slice length %d != table size %d" % (x.size, self._n))
        fn = lib.nttx_forward_distributed if forward else lib.nttx_backward_distributed
        _check(fn(self._plan, x.ctypes.data, n_shards))

    def shard_geometry(self, n_shards):
        """(local_elems, rows, cols, rows_per_peer) of the split described in include/nttx.h."""
        v = [C.c_uint64() for _ in range(4)]
        _check(lib.nttx_shard_geometry(self._plan, n_shards, *[C.byref(t) for t in v]))
        return tuple(t.value for t in v)

    def _run_batch(self, x, batch, forward):
        if x.dtype != self._np_dtype or not x.flags.c_contiguous or x.size != batch * self._n:
            raise Panic("expected batch * n contiguous elements")
        fn = lib.nttx_forward_host if forward else lib.nttx_backward_host
        _check(fn(self._plan, x.ctypes.data, batch))


class Table(_PlanTable):
    """`Table<u64>` (src/dft/ntt.rs): scalar Montgomery path."""

    def __init__(self, modulus, root, n, devices=None, exact_k=False):
        super().__init__(modulus, root, n, devices, exact_k)

    @classmethod
    def from_prime_and_root(cls, modulus, root, n, devices=None, exact_k=False):  # ntt.rs:90
        return cls(modulus, root, n, devices, exact_k)

    @classmethod
    def new(cls, devices=None):  # ntt.rs:55-57, 264-268
        return cls(0x1FFFFFFFFFE00001, 0x15EB043C7AA2B01F, 1 << 16, devices)

    default = new

    @classmethod
    def new_goldilock(cls, devices=None):  # ntt.rs:60-62
        return cls(0xFFFFFFFF00000001, 0xABD0A6E8AA3D8A0E, 1 << 16, devices)

    @classmethod
    def new_u32_compatible(cls, devices=None):  # ntt.rs:66-68
        return cls(4293918721, 2004365341, 1 << 16, devices)

    @classmethod
    def new_simple(cls, devices=None):  # ntt.rs:72-75: panics in the reference too (odd log2 n)
        return cls(7681, 1925, 1 << 3, devices)


class TableAVX2(_PlanTable):
    """`TableAVX2` (src/dft/ntt_avx2.rs): i64 slices, primes <= 61 bit; same outputs as `Table`."""
    _np_dtype = np.int64

    def __init__(self, modulus, root, n, devices=None):
        if not (0 < modulus < 1 << 62):
            raise Panic("TableAVX2 compares residues as signed 64-bit values: modulus must be below 2^62")
        if n < 16:
            raise Panic("TableAVX2 needs len >= 16 (ntt_avx2.rs:264-268 falls back to a wrong scalar path)")
        super().__init__(modulus, root, n, devices)

    @classmethod
    def from_prime_and_root(cls, modulus, root, n, devices=None):  # ntt_avx2.rs:74
        return cls(modulus, root, n, devices)

    @classmethod
    def new(cls, devices=None):  # ntt_avx2.rs:57-59
        return cls(0x1FFFFFFFFFE00001, 0x15EB043C7AA2B01F, 1 << 16, devices)


class TableGoldilock(_PlanTable):
    """`TableGoldilock<u64>` (src/dft/ntt_goldilock.rs): plain residues, shift-and-add reduction."""
    _kind = _lib.NTTX_GOLDILOCKS
    Q = 0xFFFFFFFF00000001

    def __init__(self, psi, n, devices=None, corrected=False):
        super().__init__(self.Q, psi, n, devices, corrected)

    @classmethod
    def new(cls, devices=None):  # ntt_goldilock.rs:48-58
        return cls(0xABD0A6E8AA3D8A0E, 1 << 16, devices)

    @classmethod
    def from_root(cls, psi, n, devices=None, corrected=False):
        """Same table for another (psi, n); the reference fixes both (ntt_goldilock.rs:50-52).
        corrected: accept odd log2(n) (true bit reversal; with_precomputes, ntt_goldilock.rs:192-214,
        panics there)."""
        return cls(psi, n, devices, corrected)

    @property
    def n_inv(self):
        return self._info()["n_inv"]

    def reduce_159(self, low, middle, high):  # ntt_goldilock.rs:63-79
        q = self.Q
        low_2 = (low - high) & utils_mask
        if low < high:
            low_2 = (low_2 + q) & utils_mask
        product = (middle << 32) & utils_mask
        product = (product - (product >> 32)) & utils_mask
        res = (low_2 + product) & utils_mask
        if res < product or res >= q:
            res = (res - q) & utils_mask
        return res


utils_mask = (1 << 64) - 1


def u128_to_u64(a):
    """ntt_goldilock.rs:218-227: returns (high, low)."""
    return (a >> 64) & utils_mask, a & utils_mask


def u64_to_u128(first, second):
    """ntt_goldilock.rs:231-238"""
    return (first << 64) | second
