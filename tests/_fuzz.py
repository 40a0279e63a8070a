"""Randomised parity sweep (test helper): random table kinds, sizes, batches and inputs -- residues,
arbitrary u64, lazily reduced and edge values -- through the host entry points, the device entry points
(batches resident on the GPU, on a random stream), the [0, 2q) opt-in, the split transform and the
negacyclic product, each compared with the CPU oracle.  tests/test_gpu_fuzz.py runs a
bounded, seeded slice; tests/manual/fuzz_parity.py runs it for as long as asked."""
import time

import numpy as np

import _oracle as orc
import _primes
import ntt_amd

Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
QGL = 0xFFFFFFFF00000001
Q32, R32 = 4293918721, 2004365341
KINDS = ["mont61", "mont32", "montwide", "goldilocks", "mont62", "mont63", "mont64"]
_STRICT = {"mont62": _primes.MID[1], "mont63": _primes.MID[0], "mont64": _primes.WIDE[0]}


def run(budget, seed, progress=False):
    """Runs random cases for `budget` seconds; returns the number of cases and the kinds seen."""
    rng = np.random.default_rng(seed)
    tables = {}

    def get(kind, logn):
        key = (kind, logn)
        if key not in tables:
            n = 1 << logn
            if kind == "mont61":
                psi = pow(R61, (1 << 20) // n, Q61)
                tables[key] = (ntt_amd.Table.from_prime_and_root(Q61, psi, n), orc.Table.from_prime_and_root(Q61, psi, n), Q61)
            elif kind == "mont32":
                psi = pow(R32, (1 << 16) // n, Q32)
                tables[key] = (ntt_amd.Table.from_prime_and_root(Q32, psi, n), orc.Table.from_prime_and_root(Q32, psi, n), Q32)
            elif kind in _STRICT:
                q, psi21, policy = _STRICT[kind]
                psi = _primes.psi_for(q, psi21, n)
                tables[key] = (ntt_amd.Table.from_prime_and_root(q, psi, n), orc.Table.from_prime_and_root(q, psi, n), q)
                assert tables[key][0].policy == policy
            else:
                psi = pow(7, (QGL - 1) // (2 * n), QGL)
                if kind == "montwide":
                    tables[key] = (ntt_amd.Table.from_prime_and_root(QGL, psi, n), orc.Table.from_prime_and_root(QGL, psi, n), QGL)
                else:
                    tables[key] = (ntt_amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n), QGL)
        return tables[key]


    def inputs(q, count, style):
        if style == 0:
            return rng.integers(0, q, count, dtype=np.uint64)
        if style == 1:
            return rng.integers(0, 2**64, count, dtype=np.uint64)
        if style == 3:     # lazily reduced: residue + a small multiple of q (wraps past 2^64 for 64-bit q)
            a = rng.integers(0, q, count, dtype=np.uint64) + np.uint64(q) * rng.integers(0, 8, count, dtype=np.uint64)
            near = rng.random(count) < 0.05
            a[near] = (np.uint64(q) * rng.integers(1, 9, int(near.sum()), dtype=np.uint64)
                       + rng.integers(0, 3, int(near.sum()), dtype=np.uint64) - np.uint64(1))
            return a
        a = rng.integers(0, q, count, dtype=np.uint64)
        edges = np.array([0, 1, q - 1, q, (q + 1) % 2**64, 2**64 - 1, 2**32 - 1, 2**32, q - 2, 4 * q % 2**64, 2**63],
                         dtype=np.uint64)
        pick = rng.random(count) < 0.3
        a[pick] = edges[rng.integers(0, len(edges), int(pick.sum()))]
        return a


    t_end = time.time() + budget
    cases = 0
    seen = set()
    last = time.time()
    while time.time() < t_end:
        kind = KINDS[rng.integers(0, len(KINDS))]
        logn = int(rng.choice([2, 4, 6, 8, 10, 12, 14, 16] if kind == "mont32" else [2, 4, 6, 8, 10, 12, 14, 16, 18, 20]))
        n = 1 << logn
        batch = int(rng.integers(1, 70 if logn <= 14 else (24 if logn <= 16 else 3)))
        gpu, ref, q = get(kind, logn)
        mode = int(rng.integers(0, 6))
        style = int(rng.integers(0, 4))
        a = inputs(q, n * batch, style)
        tag = (kind, logn, batch, mode, style)
        if mode == 0:      # forward
            w = a.copy()
            for b in range(batch):
                ref.forward_inplace(w[b * n:(b + 1) * n])
            gpu.forward_inplace_batch(a, batch)
            assert np.array_equal(a, w), tag
        elif mode == 1:    # backward
            w = a.copy()
            for b in range(batch):
                ref.backward_inplace(w[b * n:(b + 1) * n])
            gpu.backward_inplace_batch(a, batch)
            assert np.array_equal(a, w), tag
        elif mode == 4:    # device-resident batch on a stream of its own: forward, then backward, both against the oracle
            import torch
            w = a.copy()
            for b in range(batch):
                ref.forward_inplace(w[b * n:(b + 1) * n])
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d = torch.from_numpy(a.view(np.int64)).cuda()
                gpu.forward_inplace(d)
                f = d.cpu().numpy().view(np.uint64).copy()
                gpu.backward_inplace(d)
                r = d.cpu().numpy().view(np.uint64).copy()
            assert np.array_equal(f, w), tag
            for b in range(batch):
                ref.backward_inplace(w[b * n:(b + 1) * n])
            assert np.array_equal(r, w), tag
        elif mode == 5:    # outputs in [0, 2q) on request: below 2q (mod 2^64 for 64-bit q) and congruent to the oracle's
            fwd = bool(rng.integers(0, 2))
            x, w = a[:n].copy(), a[:n].copy()
            (ref.forward_inplace if fwd else ref.backward_inplace)(w)
            (gpu.forward_inplace_range_2q if fwd else gpu.backward_inplace_range_2q)(x)
            if 2 * q < 2**64:
                assert int(x.max()) < 2 * q, tag
                assert np.array_equal(x % np.uint64(q), w % np.uint64(q)), tag
            else:          # policies without lazy outputs return the strict ones
                assert np.array_equal(x, w), tag
        elif mode == 2 and logn >= 16:   # split transform of one vector
            f = logn % 8 or 8
            shards = int(rng.choice([s for s in (1, 2, 4, 8, 16) if s <= (1 << f) and logn >= f + s.bit_length() - 1 + 4]))
            x, w = a[:n].copy(), a[:n].copy()
            fwd = bool(rng.integers(0, 2))
            (ref.forward_inplace if fwd else ref.backward_inplace)(w)
            (gpu.forward_inplace_distributed if fwd else gpu.backward_inplace_distributed)(x, shards)
            assert np.array_equal(x, w), tag + (shards, fwd)
        else:              # negacyclic product of residues
            a = inputs(q, n * batch, 0)
            b2 = inputs(q, n * batch, 0)
            want = a.copy()
            for b in range(batch):
                fa, fb = a[b * n:(b + 1) * n].copy(), b2[b * n:(b + 1) * n].copy()
                ref.forward_inplace(fa)
                ref.forward_inplace(fb)
                prod = np.array([int(x) * int(y) % q for x, y in zip(fa, fb)], dtype=np.uint64) if n <= 4096 else None
                if prod is None:   # big products with Python ints are slow: object arrays
                    prod = (fa.astype(object) * fb.astype(object) % q).astype(np.uint64)
                ref.backward_inplace(prod)
                want[b * n:(b + 1) * n] = prod
            gpu.negacyclic_mul_inplace(a, b2)
            assert np.array_equal(a, want), tag
        cases += 1
        seen.add(kind)
        if progress and time.time() - last > 20:
            last = time.time()
            print("fuzz: %d cases so far, last %s" % (cases, tag), flush=True)
    return cases, seen
