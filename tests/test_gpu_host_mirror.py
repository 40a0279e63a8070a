"""Runs the C++ host mirror's restatement of the reference's dft.rs tests on the GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_cpp_host_mirror_passes_reference_tests():
    exe = os.path.join(HERE, "cpp", "test_dft")
    assert os.path.exists(exe), "build it first: make -C tests/cpp (done by __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all host-mirror tests passed" in out.stdout
