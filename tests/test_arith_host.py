"""Builds and runs the host-side unit test of ntt_amd/csrc/arith.hpp (the device arithmetic
compiled for the CPU): Shoup/Montgomery/Goldilocks products against unsigned __int128 arithmetic,
lazy-range bounds, and whole radix-2 transforms in the reference's loop order."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_arithmetic_policies_on_host():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "test_arith"])
    out = subprocess.run([os.path.join(HERE, "cpp", "test_arith")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all arithmetic tests passed" in out.stdout
