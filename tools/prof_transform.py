"""Runs whole forward and backward transforms of a resident batch a few times (for rocprofv3).
Usage: prof_transform.py <mont|goldilocks> <logn> [batch] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ntt_amd

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
for r in range(reps):
    t.forward_inplace(data)
    t.backward_inplace(data)
torch.cuda.synchronize()
