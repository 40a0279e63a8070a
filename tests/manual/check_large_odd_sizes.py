"""Manual check on a GPU box: odd log2 n above the sizes the test suite covers (2^23, 2^25, 2^27,
Goldilocks): the transform of a delta against its closed form, and a round trip."""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import ntt_amd
from test_gpu_odd_sizes import bit_reverse, describe
QGL = 0xFFFFFFFF00000001
for logn in (23, 25, 27):
    n = 1 << logn
    psi = pow(7, (QGL - 1) // (2 * n), QGL)
    t = ntt_amd.TableGoldilock.from_root(psi, n, corrected=True)
    print(logn, describe(ntt_amd, t), flush=True)
    j = 3
    a = np.zeros(n, dtype=np.uint64); a[j] = 1
    t.forward_inplace(a)
    rev = bit_reverse(np.arange(n), logn)
    for i in np.random.default_rng(logn).integers(0, n, 500):
        assert int(a[i]) == pow(psi, j * (2 * int(rev[i]) + 1), QGL), (logn, i)
    r = np.random.default_rng(1).integers(0, QGL, n, dtype=np.uint64)
    x = r.copy(); t.forward_inplace(x); t.backward_inplace(x)
    assert np.array_equal(x, r)
    print("ok", logn, flush=True)
