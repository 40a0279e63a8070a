"""Negacyclic product of resident batches: fused call versus the three transforms it contains."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ntt_amd

n, batch, steps = 1 << 16, 4096, 10
for name, t, q in (("mont61", ntt_amd.Table.new(), 0x1FFFFFFFFFE00001),
                   ("goldilocks", ntt_amd.TableGoldilock.new(), 0xFFFFFFFF00000001)):
    a = torch.randint(0, min(q, 2**63 - 1), (batch * n,), dtype=torch.int64, device="cuda")
    b = torch.randint(0, min(q, 2**63 - 1), (batch * n,), dtype=torch.int64, device="cuda")

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def three():
        t.forward_inplace(a); t.forward_inplace(b); t.backward_inplace(a)

    t3 = timed(three)
    tm = timed(lambda: t.negacyclic_mul_inplace(a, b))
    ts = timed(lambda: t.negacyclic_mul_inplace(a, a))
    print("%-10s batch %d x 2^16: product %.3f ms (%.0f products/s), square %.3f ms; forward, forward, backward alone %.3f ms"
          % (name, batch, tm * 1e3, batch / tm, ts * 1e3, t3 * 1e3))
