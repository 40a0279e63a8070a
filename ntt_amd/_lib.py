"""ctypes binding of libnttx.so (include/nttx.h).  No fallback: a missing library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NTTX_LIB_PATH: another build of the same library (kernel experiments); never a different implementation
LIB_PATH = os.environ.get("NTTX_LIB_PATH") or os.path.join(_HERE, "libnttx.so")

NTTX_OK, NTTX_EINVAL, NTTX_ENODEV, NTTX_ENOMEM, NTTX_EHIP = range(5)
NTTX_MONT64, NTTX_GOLDILOCKS = 0, 1

# every symbol include/nttx.h declares: (restype, argtypes)
_u64, _vp, _sz = C.c_uint64, C.c_void_p, C.c_size_t
_pu64 = C.POINTER(C.c_uint64)
_pint = C.POINTER(C.c_int)
SYMBOLS = {
    "nttx_plan_create": (C.c_int, [C.c_int, _u64, _u64, _u64, _u64, _vp, _vp, _pint, C.c_int, C.POINTER(_vp)]),
    "nttx_plan_create_from_root": (C.c_int, [C.c_int, _u64, _u64, _u64, _pint, C.c_int, C.POINTER(_vp)]),
    "nttx_plan_create_from_root_exact_k": (C.c_int, [C.c_int, _u64, _u64, _u64, _pint, C.c_int, C.POINTER(_vp)]),
    "nttx_plan_destroy": (None, [_vp]),
    "nttx_plan_info": (C.c_int, [_vp, _pint, _pu64, _pu64, _pu64, _pu64, _pint]),
    "nttx_plan_tables": (C.c_int, [_vp, _vp, _vp]),
    "nttx_forward_host": (C.c_int, [_vp, _vp, _sz]),
    "nttx_backward_host": (C.c_int, [_vp, _vp, _sz]),
    "nttx_forward_device": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp]),
    "nttx_backward_device": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp]),
    "nttx_transform_host_noalloc": (C.c_int, [_vp, _vp, _sz, C.c_int]),
    "nttx_plan_policy": (C.c_char_p, [_vp]),
    "nttx_forward_host_lazy": (C.c_int, [_vp, _vp, _sz]),
    "nttx_backward_host_lazy": (C.c_int, [_vp, _vp, _sz]),
    "nttx_forward_device_lazy": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp]),
    "nttx_backward_device_lazy": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp]),
    "nttx_negacyclic_mul_host": (C.c_int, [_vp, _vp, _vp, _sz]),
    "nttx_negacyclic_mul_device": (C.c_int, [_vp, C.c_int, _vp, _vp, _sz, _vp]),
    "nttx_pass_count": (C.c_int, [_vp]),
    "nttx_pass_describe": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, _sz]),
    "nttx_run_pass_device": (C.c_int, [_vp, C.c_int, _vp, _sz, C.c_int, C.c_int, _vp]),
    "nttx_plan_feature": (C.c_int, [_vp, C.c_char_p]),
    "nttx_clock_probe_device": (C.c_int, [C.c_int, _vp, C.c_int, _vp]),
    "nttx_forward_distributed": (C.c_int, [_vp, _vp, C.c_int]),
    "nttx_backward_distributed": (C.c_int, [_vp, _vp, C.c_int]),
    "nttx_shard_geometry": (C.c_int, [_vp, C.c_int, _pu64, _pu64, _pu64, _pu64]),
    "nttx_shard_phase_device": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "nttx_device_count": (C.c_int, []),
    "nttx_last_error": (C.c_char_p, []),
    "nttx_version": (C.c_char_p, []),
}


def _load():
    # PyTorch bundles its own HIP runtime and asks for it by the unversioned file name, which the
    # dynamic loader does not match against an already loaded /opt/rocm copy; two HIP runtimes in
    # one process leave the second without devices.  Loading torch first makes libnttx.so bind to
    # torch's copy (same SONAME), so tensors and plans share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C ntt_amd/csrc` (or __graft_entry__.build()); "
            "ntt_amd has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    return lib.nttx_last_error().decode()
