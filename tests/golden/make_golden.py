#!/usr/bin/env python3
"""Regenerates tests/golden/digests.json from an independent pure-Python model of the reference.

The reference keeps no full-transform golden vectors (its tests draw inputs from an unseeded RNG),
and no Rust toolchain exists here to run it, so the fixtures are SHA-256 digests of what a second,
independent restatement computes: this file, written from the reference's semantics with Python
integers, sharing no code with oracle/ (C).  Agreement of the two on every digest is what pins
full transforms.  Inputs: N = 2^16, the default constructors (ntt.rs:55-57,
ntt_goldilock.rs:48-58), a ramp 0..N-1 (benches/ntt.rs:9-11) and an LCG stream reduced mod q.

    python tests/golden/make_golden.py            # rewrite digests.json
    python tests/golden/make_golden.py --check    # compare with the committed file
"""
import hashlib
import json
import os
import struct
import sys

N = 1 << 16
LOGN = 16
Q61, PSI61 = 0x1FFFFFFFFFE00001, 0x15EB043C7AA2B01F
QGL, PSIGL = 0xFFFFFFFF00000001, 0xABD0A6E8AA3D8A0E
M64 = (1 << 64) - 1


def sha(values):
    return hashlib.sha256(struct.pack("<%dQ" % len(values), *values)).hexdigest()


def brev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2)


def lcg(q, n, x=0x9E3779B97F4A7C15):
    out = []
    for _ in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) & M64
        out.append(x % q)
    return out


def power_table(psi, q):
    """t[brev(i)] = psi^i mod q (ntt.rs:239-261 stores these times 2^64; ntt_goldilock.rs:192-214 as is)."""
    t = [0] * N
    x = 1
    for i in range(N):
        t[brev(i, LOGN)] = x
        x = x * psi % q
    return t


# ---- Table<u64>: both sweeps of ntt.rs:125-127 / 173-175 cancel, leaving plain residues ----------
def mont_forward(a, w, q):
    a = [x % q for x in a]
    t, m = N, 1
    while m < N:
        t //= 2
        for i in range(m):
            s = w[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], a[j + t] * s % q
                a[j], a[j + t] = (u + v) % q, (u - v) % q
        m *= 2
    return a


def mont_backward(a, winv, q):
    a = [x % q for x in a]
    t, m = 1, N
    while m > 1:
        h, j1 = m // 2, 0
        for i in range(h):
            s = winv[h + i]
            for j in range(j1, j1 + t):
                u, v = a[j], a[j + t]
                a[j], a[j + t] = (u + v) % q, (u - v) * s % q
            j1 += 2 * t
        t *= 2
        m //= 2
    n_inv = pow(N, q - 2, q)
    return [x * n_inv % q for x in a]


# ---- TableGoldilock: wrapping u64 arithmetic with the reference's compare rules -------------------
def gl_add(u, v):  # ntt_goldilock.rs:104-112
    s = u + v
    carry, s = s >> 64, s & M64
    return (s - QGL) & M64 if (carry or s > QGL) else s


def gl_forward(a, w):
    a = list(a)
    t, m = N, 1
    while m < N:
        t //= 2
        for i in range(m):
            s = w[m + i]
            for j in range(2 * i * t, 2 * i * t + t):
                u, v = a[j], a[j + t] * s % QGL          # mul_reduce is canonical (ntt_goldilock.rs:63-79)
                d = (u - v) & M64
                if u < v:
                    d = (d + QGL) & M64
                a[j], a[j + t] = gl_add(u, v), d % QGL    # :115-120
        m *= 2
    return a


def gl_backward(a, winv):
    a = list(a)
    t, m = 1, N
    while m > 1:
        h, j1 = m // 2, 0
        for i in range(h):
            s = winv[h + i]
            for j in range(j1, j1 + t):
                u, v = a[j], a[j + t]
                d = (u - v) & M64
                if u < v:
                    d = (d + QGL) & M64                    # :162-165, no further reduction
                a[j], a[j + t] = gl_add(u, v), d * s % QGL
            j1 += 2 * t
        t *= 2
        m //= 2
    n_inv = 0xFFFEFFFF00010001                              # hard-coded at ntt_goldilock.rs:53
    assert n_inv == pow(N, QGL - 2, QGL)
    return [x * n_inv % QGL for x in a]                    # :177-179


def main():
    out = {"_comment": "SHA-256 over little-endian u64 bytes, N=2^16, default constructors. Produced by "
                       "tests/golden/make_golden.py (independent pure-Python model of the reference) and "
                       "reproduced by the C oracle and by the GPU path."}
    R = 1 << 64
    w61, w61inv = power_table(PSI61, Q61), power_table(pow(PSI61, Q61 - 2, Q61), Q61)
    out["table_new.powers_psi_bo"] = sha([x * R % Q61 for x in w61])
    out["table_new.powers_psi_inv_bo"] = sha([x * R % Q61 for x in w61inv])
    ramp = list(range(N))
    out["table_new.forward.ramp"] = sha(mont_forward(ramp, w61, Q61))
    l61 = lcg(Q61, N)
    out["table_new.forward.lcg"] = sha(mont_forward(l61, w61, Q61))
    out["table_new.backward.lcg"] = sha(mont_backward(l61, w61inv, Q61))
    wgl, wglinv = power_table(PSIGL, QGL), power_table(pow(PSIGL, QGL - 2, QGL), QGL)
    out["goldilock_new.powers_psi_bo"] = sha(wgl)
    out["goldilock_new.powers_psi_inv_bo"] = sha(wglinv)
    out["goldilock_new.forward.ramp"] = sha(gl_forward(ramp, wgl))
    lgl = lcg(QGL, N)
    out["goldilock_new.forward.lcg"] = sha(gl_forward(lgl, wgl))
    out["goldilock_new.backward.lcg"] = sha(gl_backward(lgl, wglinv))

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "digests.json")
    if "--check" in sys.argv:
        with open(path) as fh:
            have = json.load(fh)
        bad = [k for k in out if k != "_comment" and have.get(k) != out[k]]
        if bad:
            raise SystemExit("digest mismatch: %s" % ", ".join(bad))
        print("all %d digests reproduced" % (len(out) - 1))
    else:
        with open(path, "w") as fh:
            json.dump(out, fh, indent=2)
            fh.write("\n")
        print("wrote", path)


if __name__ == "__main__":
    main()
