"""The opt-in one-launch transform of n = 2^16 (fused16_kernel, NTTX_FUSED16=1): both 8-stage passes in
one launch, tiles handed from pass to pass through the XCD's L2 under a ticket protocol
(tests/test_fused16_protocol.py models the protocol).  The switch is read once per process, so the
parity check runs in a child."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_fused16_matches_oracle():
    env = dict(os.environ, NTTX_FUSED16="1")
    out = subprocess.run([sys.executable, os.path.join(HERE, "fused16_worker.py")], env=env, capture_output=True,
                         text=True, timeout=280)
    assert out.returncode == 0 and "fused16 ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
