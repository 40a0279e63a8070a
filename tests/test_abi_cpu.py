"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/nttx.h
declares, and rejects bad arguments without touching a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ntt_amd import _lib

    header = open(os.path.join(ROOT, "include", "nttx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(nttx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(_lib.lib, name)


def test_argument_validation_needs_no_gpu():
    from ntt_amd import _lib

    lib = _lib.lib
    plan = C.c_void_p()
    tab = (C.c_uint64 * 16)(*([1] * 16))
    q61, k61 = 0x1FFFFFFFFFE00001, 6917533425689690113
    # odd log2 n (utils.rs:14-17 makes the reference's constructors panic)
    assert lib.nttx_plan_create_from_root(0, q61, 3, 8, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    assert "even" in _lib.last_error()
    assert lib.nttx_plan_create_from_root(1, 0xFFFFFFFF00000001, 3, 32, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    assert "even" in _lib.last_error()
    # even modulus (utils.rs:144-146)
    assert lib.nttx_plan_create(0, 1 << 20, 1, 16, 1, tab, tab, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    # k that is not q^-1 mod 2^64 (utils.rs:148-152 for e.g. 7681)
    assert lib.nttx_plan_create(0, 7681, 12345, 16, 1, tab, tab, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    # wrong modulus for the Goldilocks kind
    assert lib.nttx_plan_create(1, q61, 0, 16, 1, tab, tab, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    # n not a power of two, NULL tables
    assert lib.nttx_plan_create(0, q61, k61, 12, 1, tab, tab, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    assert lib.nttx_plan_create(0, q61, k61, 16, 1, None, tab, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    assert lib.nttx_plan_create_from_root(0, q61, 3, 8, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    assert lib.nttx_plan_create_from_root(7, q61, 3, 16, None, 0, C.byref(plan)) == _lib.NTTX_EINVAL
    assert plan.value is None
    # tables that are not a table of this modulus: every check runs before a device is touched
    import numpy as np
    import _oracle as orc
    ref = orc.Table.from_prime_and_root(q61, pow(0x0B4F9CB5C38B40B1, (1 << 20) // 16, q61), 16)
    f, b, n_inv = ref.powers_psi_bo.copy(), ref.powers_psi_inv_bo.copy(), ref.n_inv_montgomery

    def create(fw, bw, ninv):
        return lib.nttx_plan_create(0, q61, k61, 16, ninv, fw.ctypes.data, bw.ctypes.data, None, 0, C.byref(plan))

    bad = f.copy(); bad[0] = 1                                   # plain 1 instead of 2^64 mod q
    assert create(bad, b, n_inv) == _lib.NTTX_EINVAL and "entry 0" in _lib.last_error()
    bad = f.copy(); bad[1] = f[2]
    assert create(bad, b, n_inv) == _lib.NTTX_EINVAL and "entries 1" in _lib.last_error()
    assert create(f, f, n_inv) == _lib.NTTX_EINVAL               # forward table passed twice
    assert create(f, b, (n_inv + 1) % q61) == _lib.NTTX_EINVAL and "n_inv" in _lib.last_error()
    bad = f.copy(); bad[5] = np.uint64(q61)
    assert create(bad, b, n_inv) == _lib.NTTX_EINVAL and "residue" in _lib.last_error()
    assert plan.value is None
    assert lib.nttx_forward_host(None, None, 1) == _lib.NTTX_EINVAL
    assert lib.nttx_version().startswith(b"nttx")


def test_python_utils_mirror_matches_oracle():
    import _oracle as orc
    from ntt_amd import utils

    q61 = 0x1FFFFFFFFFE00001
    assert utils.bit_reverse(2, 2) == 1
    for bits in (3, 4, 16):
        for n in range(0, 1 << min(bits, 6)):
            assert utils.bit_reverse(n, bits) == orc.bit_reverse(n, bits)
    g, x, y = utils.egcd_i128(q61, 1 << 64)
    assert g == q61 * x + (1 << 64) * y == 1
    m = utils.MontgomeryTransformer.from_modulus(q61)
    assert (m.k, m.r_square) == (6917533425689690113, 281474708275264)
    o = orc.MontgomeryTransformer.from_modulus(q61)
    for v in (1, 16777208, q61 - 1):
        assert m.transform(v) == o.transform(v)
        assert m.reduce(m.transform(v)) == v
    assert utils.MontgomeryTransformer.from_modulus(7681).k == orc.MontgomeryTransformer.from_modulus(7681).k
    assert utils.mod_exp_u64(3, 200, 1000003) == pow(3, 200, 1000003)
    r = utils.find_nth_unity_root_u64(1 << 17, q61)
    assert pow(r, 1 << 17, q61) == 1 and pow(r, 1 << 16, q61) == q61 - 1


def test_header_is_plain_c_and_links(tmp_path):
    """include/nttx.h compiles as C99 with -Wall -Werror and a C client links against libnttx.so
    referencing every declared function (a Rust/C/Go FFI sees exactly this surface)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    header = open(os.path.join(ROOT, "include", "nttx.h")).read()
    names = sorted(set(re.findall(r"\b(nttx_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S))))
    src = tmp_path / "client.c"
    src.write_text('#include "nttx.h"\n#include <stdio.h>\nint main(void) {\n'
                   '    const void *fns[] = {%s};\n'
                   '    nttx_plan *plan = NULL;\n'
                   '    int rc = nttx_plan_create_from_root(NTTX_MONT64, 7, 3, 8, NULL, 0, &plan);\n'
                   '    printf("%%d %%s %%d\\n", rc, nttx_last_error(), (int)(sizeof fns / sizeof fns[0]));\n'
                   '    return rc == NTTX_EINVAL && plan == NULL ? 0 : 1;\n}\n'
                   % ", ".join("(const void *)%s" % n for n in names))
    exe = tmp_path / "client"
    libdir = os.path.join(ROOT, "ntt_amd")
    inc = os.path.join(ROOT, "include")
    only = tmp_path / "header_only.c"
    only.write_text('#include "nttx.h"\nint nttx_header_ok;\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(only),
                    "-o", str(tmp_path / "header_only.o")], check=True)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lnttx",
                    "-Wl,-rpath," + libdir], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.split()[0] == "1" and res.stdout.strip().endswith(str(len(names)))
