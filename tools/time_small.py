"""Device-resident batches of small transforms (2 GiB per batch): ms per forward + backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ntt_amd

Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
QGL = 0xFFFFFFFF00000001
for logn in (8, 10, 12, 14):
    n = 1 << logn
    batch = (1 << 28) >> logn
    for name, t, q in (("mont61", ntt_amd.Table.from_prime_and_root(Q61, pow(R61, (1 << 20) // n, Q61), n), Q61),
                       ("goldilocks", ntt_amd.TableGoldilock.from_root(pow(7, (QGL - 1) // (2 * n), QGL), n), 2**63 - 1)):
        a = torch.randint(0, q, (batch * n,), dtype=torch.int64, device="cuda")
        orig = a.clone()
        for _ in range(3):
            t.forward_inplace(a); t.backward_inplace(a)
        torch.cuda.synchronize()
        assert torch.equal(a, orig)
        t0 = time.perf_counter()
        for _ in range(10):
            t.forward_inplace(a); t.backward_inplace(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%-10s n=2^%-2d batch %7d: %.3f ms per fwd+inv of the batch, %.2f M pairs/s" % (name, logn, batch, dt * 1e3, batch / dt / 1e6),
              flush=True)
