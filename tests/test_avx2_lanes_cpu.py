"""Host mirrors of the reference's AVX2 lane helpers (ntt_avx2.rs:704-813): the reference's own
known answers (ntt_avx2.rs:826-854, 926-939) and agreement with the oracle's restatement, which runs
the real intrinsics, on random and edge lanes.  The C++ mirror is compiled and run here too."""
import os
import random
import subprocess

import _oracle as orc
from ntt_amd import avx2

HERE = os.path.dirname(os.path.abspath(__file__))
EDGE = [0, 1, -1, -2, 2**63 - 1, -2**63, 2**32, 2**32 - 1, -2**32, 4294967295, -4294967296]


def lanes(rng):
    return [rng.choice(EDGE) if rng.random() < 0.4 else rng.randrange(-2**63, 2**63) for _ in range(4)]


def test_reference_known_answers():
    hi, lo = avx2.a_sub_b([0, 0, 2, 2], [0, 1, 2, 1], [0, 0, 2, 0], [0, 2, 1, 2])          # ntt_avx2.rs:826-838
    assert (hi, lo) == ([0, -1, 0, 1], [0, -1, 1, -1])
    hi, lo = avx2.a_mul_b_karatsuba([-4294967296, 4294967295, -1, -1], [-4294967296, 4294967295, 4294967295, -1])
    assert (hi, lo) == ([-8589934591, 0, 4294967294, -2], [0, -8589934591, -4294967295, 1])  # :841-854
    assert avx2.a_lt_b([1, 1, -1, -2], [2, -1, 1, -1]) == [-1, -1, 0, -1]                  # :926-939
    assert avx2.a_lt_b([2, -1, 1, -1], [1, 1, -1, -2]) == [0, 0, -1, 0]
    assert avx2.extract_m256i_to_i64([1, -1, 2**63, 2**64 - 1]) == [1, -1, -2**63, -1]
    assert avx2.extract_m256i_to_i64([1, 2, 3, 4]) == [1, 2, 3, 4]                          # :856-862
    assert avx2.extract_m256d_to_f64([1.0, 2.0, 3.0, 4.0]) == [1.0, 2.0, 3.0, 4.0]          # :865-872


def test_python_mirror_matches_oracle_intrinsics():
    rng = random.Random(7)
    for _ in range(2000):
        a, b, c, d = lanes(rng), lanes(rng), lanes(rng), lanes(rng)
        assert avx2.a_lt_b(a, b) == orc.a_lt_b(a, b)
        assert avx2.a_mul_b_karatsuba(a, b) == tuple(orc.a_mul_b_karatsuba(a, b))
        assert avx2.a_sub_b(a, b, c, d) == tuple(orc.a_sub_b(a, b, c, d))
        assert avx2.a_lt_b_wide(a, b, c, d) == orc.a_lt_b_wide(a, b, c, d)
        assert avx2.a_lt_b_wide(a, b, a, d) == orc.a_lt_b_wide(a, b, a, d)   # equal high words


def test_cpp_mirror(tmp_path):
    rng = random.Random(11)
    cases = [(lanes(rng), lanes(rng), lanes(rng), lanes(rng)) for _ in range(60)]
    cases.append(([0, 0, 2, 2], [0, 1, 2, 1], [0, 0, 2, 0], [0, 2, 1, 2]))
    src = tmp_path / "t.cpp"
    body = []
    for a, b, c, d in cases:
        hi, lo = avx2.a_mul_b_karatsuba(a, b)
        sh, sl = avx2.a_sub_b(a, b, c, d)
        fmt = lambda v: "{" + ", ".join("(int64_t)%dull" % (x & (2**64 - 1)) for x in v) + "}"
        body.append("check(%s, %s, %s, %s, %s, %s, %s, %s, %s, %s);" % tuple(
            fmt(v) for v in (a, b, c, d, avx2.a_lt_b(a, b), hi, lo, sh, sl, avx2.a_lt_b_wide(a, b, c, d))))
    src.write_text("""
#include <cstdio>
#include "avx2_lanes.hpp"
using namespace ntt::dft::avx2;
static int bad = 0;
static void check(Lanes a, Lanes b, Lanes c, Lanes d, Lanes lt, Lanes mh, Lanes ml, Lanes sh, Lanes sl, Lanes ltw)
{
    auto m = a_mul_b_karatsuba(a, b);
    auto s = a_sub_b(a, b, c, d);
    if (a_lt_b(a, b) != lt || m.first != mh || m.second != ml || s.first != sh || s.second != sl ||
        a_lt_b_wide(a, b, c, d) != ltw || extract_m256i_to_i64(a) != a)
        bad++;
}
int main()
{
%s
    std::printf("%%d\\n", bad);
    return bad != 0;
}
""" % "\n".join("    " + line for line in body))
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-I", os.path.join(os.path.dirname(HERE), "ntt_amd", "host"),
                           str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr
