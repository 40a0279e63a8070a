"""Code paths the default runs do not take, selected by environment switches the library reads
once per process -- so each variant runs the pass-shape parity tests in a child process:

  NTTX_NO_BUFFER_ADDRESSING  64-bit global addressing, the path of blocks above 2^28 elements
  NTTX_NO_CROSS_BATCH        plain contiguous tiles for every batch (otherwise only batch mod 16)
  NTTX_NO_HOST_STAGING       host batches of any size copied straight from the caller's memory
  NTTX_NO_FUSED_SMALL        n = 2^10 and 2^12 as two launches (otherwise only split transforms and
                             the profiling entry point run them so)
  NTTX_NO_POW2_TWIDDLES      Goldilocks plans of the standard roots multiply by their early twiddles
                             instead of shifting (otherwise only plans of other roots do)
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SELECT = "all_pass_shapes or montgomery_table or device_resident_batch or large_host or any_u64 or lazily"


@pytest.mark.parametrize("switches", [("NTTX_NO_BUFFER_ADDRESSING",), ("NTTX_NO_CROSS_BATCH", "NTTX_NO_HOST_STAGING", "NTTX_NO_FUSED_SMALL", "NTTX_NO_POW2_TWIDDLES")])
def test_parity_on_alternative_paths(switches):
    env = dict(os.environ)
    for s in switches:
        env[s] = "1"
    cmd = [sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_gpu_polymul.py"),
           "-x", "-q", "-p", "no:cacheprovider", "-k", SELECT + " or composition or fused_middle"]
    out = subprocess.run(cmd, env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
