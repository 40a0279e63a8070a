"""The committed digests are what tests/golden/make_golden.py -- an independent pure-Python model of
the reference, sharing no code with the C oracle -- computes; test_oracle.py and the GPU parity
tests check the oracle and the kernels against the same file."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_python_model_reproduces_committed_digests():
    res = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_golden.py"), "--check"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
