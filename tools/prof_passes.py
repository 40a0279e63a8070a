"""Runs every pass kernel of the headline plan a few times on a resident batch (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ntt_amd
from ntt_amd._lib import lib

kind = sys.argv[1] if len(sys.argv) > 1 else "mont"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batch = int(sys.argv[3]) if len(sys.argv) > 3 else (1 << 28) >> logn
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n = 1 << logn
if kind == "mont":
    q = 0x1FFFFFFFFFE00001
    t = ntt_amd.Table.from_prime_and_root(q, pow(0x0B4F9CB5C38B40B1, (1 << 20) // n, q), n)
else:
    q = 0xFFFFFFFF00000001
    t = ntt_amd.TableGoldilock.from_root(pow(7, (q - 1) // (2 * n), q), n)
data = torch.randint(0, min(q, 2**63 - 1), (batch * n,), dtype=torch.int64, device="cuda")
original = data.clone()
s = torch.cuda.current_stream()
for d in (0, 1):
    for p in range(lib.nttx_pass_count(t._plan)):
        for r in range(reps):
            if p == 0:
                data.copy_(original)  # first passes screen their input: give them residues
            assert lib.nttx_run_pass_device(t._plan, 0, data.data_ptr(), batch, d, p, s.cuda_stream) == 0
torch.cuda.synchronize()
