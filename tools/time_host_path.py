"""PCIe-inclusive rate of the host-slice entry points (pageable numpy memory in, same memory out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ntt_amd

n = 1 << 16
q = 0x1FFFFFFFFFE00001
t = ntt_amd.Table.new()
rng = np.random.default_rng(1)
for batch in (1, 16, 32, 64, 256, 1024, 4096):
    a = rng.integers(0, q, batch * n, dtype=np.uint64)
    t.forward_inplace_batch(a, batch)
    t.backward_inplace_batch(a, batch)
    reps = 200 if batch == 1 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        t.forward_inplace_batch(a, batch)
        t.backward_inplace_batch(a, batch)
    dt = (time.perf_counter() - t0) / reps
    print("batch %5d: %.2f ms per fwd+inv of the batch -> %.0f pairs/s, %.1f GB/s over PCIe each way"
          % (batch, dt * 1e3, batch / dt, 2 * batch * n * 8 / dt / 1e9))

# negacyclic products from and to pageable memory (a <- a * b; b is only read)
for batch in (64, 1024):
    a = rng.integers(0, q, batch * n, dtype=np.uint64)
    b = rng.integers(0, q, batch * n, dtype=np.uint64)
    t.negacyclic_mul_inplace(a, b)
    t0 = time.perf_counter()
    for _ in range(3):
        t.negacyclic_mul_inplace(a, b)
    dt = (time.perf_counter() - t0) / 3
    print("batch %5d: %.2f ms per batch of products -> %.0f products/s" % (batch, dt * 1e3, batch / dt))
