"""Randomised parity sweep on a GPU box for as long as asked (tests/_fuzz.py holds the cases; the
test suite runs a bounded slice of it).  Usage: python tests/manual/fuzz_parity.py [seconds] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _fuzz  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cases, seen = _fuzz.run(budget, int(sys.argv[2]) if len(sys.argv) > 2 else 1, progress=True)
print("fuzz: %d random cases passed in %.0f s (%s)" % (cases, budget, ", ".join(sorted(seen))))
