"""MI355X engine behind the `DFT` interface of robertkiel/ntt.

`ntt_amd.dft` mirrors the reference's `ntt::dft` module (trait `DFT`, `Table`, `TableAVX2`,
`TableGoldilock`), `ntt_amd.utils` its `ntt::utils` module.  All transforms run in
libnttx.so (hand-written gfx950 kernels behind the C ABI of include/nttx.h); there is no
CPU fallback: importing this package fails if the library has not been built.
"""
from . import _lib  # noqa: F401  (loads libnttx.so or raises)
from . import avx2, dft, distributed, utils  # noqa: F401
from .dft import DFT, Panic, Table, TableAVX2, TableGoldilock  # noqa: F401

__all__ = ["avx2", "dft", "utils", "DFT", "Panic", "Table", "TableAVX2", "TableGoldilock"]
