import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, ntt_amd, _oracle as orc
def root(q, order):
    for g in range(2, q):
        r = pow(g, (q - 1) // order, q)
        if pow(r, order // 2, q) == q - 1: return r
for q in (97, 257, 12289, 65537, 786433, 998244353, 4611686018326724609, 2305843009213554689):
    for logn in (2, 4, 6, 8, 10, 12):
        n = 1 << logn
        if (q - 1) % (2 * n): continue
        psi = root(q, 2 * n)
        try:
            gpu = ntt_amd.Table.from_prime_and_root(q, psi, n)
        except ntt_amd.Panic as e:
            print(q, n, "rejected:", str(e)[:60]); break
        ref = orc.Table.from_prime_and_root(q, psi, n)
        rng = np.random.default_rng(q % 1000 + logn)
        a = rng.integers(0, 2**64, n * 5, dtype=np.uint64); a[:n] = rng.integers(0, q, n, dtype=np.uint64)
        w = a.copy()
        for b in range(5): ref.forward_inplace(w[b*n:(b+1)*n])
        gpu.forward_inplace_batch(a, 5)
        assert np.array_equal(a, w), (q, n, "fwd")
        for b in range(5): ref.backward_inplace(w[b*n:(b+1)*n])
        gpu.backward_inplace_batch(a, 5)
        assert np.array_equal(a, w), (q, n, "bwd")
    else:
        print(q, "ok")
