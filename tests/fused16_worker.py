"""Child process of tests/test_gpu_fused16.py (NTTX_FUSED16=1 is set by the parent): the one-launch
n = 2^16 transform against the oracle for every arithmetic policy, ragged batch sizes included."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import torch

import _oracle as orc
import _primes
import ntt_amd

assert os.environ.get("NTTX_FUSED16") == "1"
Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
Q32, R32 = 4293918721, 2004365341
QGL = 0xFFFFFFFF00000001
n = 1 << 16
psigl = pow(7, (QGL - 1) // (2 * n), QGL)
qw, pw, _ = _primes.WIDE[0]
qm, pm, _ = _primes.MID[0]
cases = [
    ("shoup61_near", ntt_amd.Table.new(), orc.Table.new(), Q61),
    ("shoup61", ntt_amd.Table.new_u32_compatible(), orc.Table.new_u32_compatible(), Q32),
    ("goldilocks", ntt_amd.TableGoldilock.from_root(psigl, n), orc.TableGoldilock.from_root(psigl, n), QGL),
    ("goldilocks_canonical", ntt_amd.Table.new_goldilock(), orc.Table.new_goldilock(), QGL),
    ("mont_wide", ntt_amd.Table.from_prime_and_root(qw, _primes.psi_for(qw, pw, n), n),
     orc.Table.from_prime_and_root(qw, _primes.psi_for(qw, pw, n), n), qw),
    ("mont", ntt_amd.Table.from_prime_and_root(qm, _primes.psi_for(qm, pm, n), n),
     orc.Table.from_prime_and_root(qm, _primes.psi_for(qm, pm, n), n), qm),
]
for policy, gpu, ref, q in cases:
    assert gpu.policy == policy
    for batch in (64, 77, 200) if policy == "shoup61_near" else (65,):
        rng = np.random.default_rng(batch)
        a = orc.splitmix_residues(q, n * batch, 1000 + batch)
        a[n:2 * n] = rng.integers(0, 1 << 64, n, dtype=np.uint64)        # one vector of arbitrary u64
        d = torch.from_numpy(a.view(np.int64).copy()).cuda()
        gpu.forward_inplace(d)
        torch.cuda.synchronize()
        got = d.cpu().numpy().view(np.uint64)
        canon = a.copy()
        for b in sorted({0, 1, 17, batch // 2, batch - 2, batch - 1}):
            w = a[b * n:(b + 1) * n].copy()
            ref.forward_inplace(w)
            assert np.array_equal(got[b * n:(b + 1) * n], w), (policy, batch, b, "forward")
        gpu.backward_inplace(d)
        torch.cuda.synchronize()
        w1 = a[n:2 * n].copy()
        ref.forward_inplace(w1)
        ref.backward_inplace(w1)                                           # the reference reduces arbitrary input
        canon[n:2 * n] = w1
        assert np.array_equal(d.cpu().numpy().view(np.uint64), canon), (policy, batch, "round trip")
        if policy.startswith("shoup61"):
            gpu.forward_inplace_lazy(d)
            torch.cuda.synchronize()
            lz = d.cpu().numpy().view(np.uint64)
            assert int(lz.max()) < 2 * q and np.array_equal(lz[:n] % np.uint64(q), got[:n]), (policy, "lazy")
print("fused16 ok")
