import sys, time, os
sys.path.insert(0, os.getcwd())
import ntt_amd
QGL = 0xFFFFFFFF00000001
for logn in (16, 20, 24, 28):
    n = 1 << logn
    psi = pow(7, (QGL - 1) // (2 * n), QGL)
    t0 = time.perf_counter()
    t = ntt_amd.TableGoldilock.from_root(psi, n)
    print("plan creation, Goldilocks 2^%d: %.3f s" % (logn, time.perf_counter() - t0), flush=True)
    del t
Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
for logn in (16, 20):
    n = 1 << logn
    t0 = time.perf_counter()
    t = ntt_amd.Table.from_prime_and_root(Q61, pow(R61, (1 << 20) // n, Q61), n)
    print("plan creation, 61-bit 2^%d: %.3f s" % (logn, time.perf_counter() - t0), flush=True)
